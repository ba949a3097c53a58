#!/bin/bash
# tools/pmc_passes.sh OUTDIR [bench args...] -- rocprofv3 PMC passes over bench.py, one pass per run
# (counter-slot limits: 8 SQ, 4 TCC, 2 GRBM per pass).  Run on the GPU box from the repo root.
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame-leg --no-single-process-leg $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
}
BENCH_ARGS="${BENCH_ARGS:-}"
pass mix   SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32
pass time  SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT
pass misc  SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 profiles/summarize_pmc.py render_kernel "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"

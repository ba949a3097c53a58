#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/pmc_ab.sh OUTDIR LIB... -- the 'mix' and 'time' PMC passes of tools/pmc_passes.sh for each build of
# libluma_hip.so named (LUMA_HIP_LIB), summaries side by side.  Run on the GPU box from the repo root.
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  export LUMA_HIP_LIB=$PWD/$lib
  for p in "mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" \
           "time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    set -- $p; name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame-leg ${BENCH_ARGS:-} > "$OUT/$tag.$name.log" 2>&1 || { echo "pass $tag/$name failed"; tail -3 "$OUT/$tag.$name.log"; exit 1; }
  done
  echo "== $tag"; python3 profiles/summarize_pmc.py render_kernel "$OUT/$tag" | tee "$OUT/$tag.summary.txt"
done

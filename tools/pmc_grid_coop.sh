#!/bin/bash
# tools/pmc_grid_coop.sh OUTDIR -- PMC passes of one of config #5's eight bands under LUMA_FLAG_GRID, for the
# lane-per-ray walk and (LUMA_GRID_COOP=1) the wave-cooperative walk.  Run on the GPU box from the repo root.
set -u
OUT=$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for coop in 0 1; do
  export LUMA_GRID_COOP=$coop
  tag=coop$coop
  for p in "mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32" \
           "time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    set -- $p; name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload config5 --grid --emulate-world 8 --emulate-rank 3 > "$OUT/$tag.$name.log" 2>&1 || { echo "pass $tag/$name failed"; tail -3 "$OUT/$tag.$name.log"; exit 1; }
  done
  echo "== $tag"; python3 profiles/summarize_pmc.py render_kernel "$OUT/$tag" | tee "$OUT/$tag.summary.txt"
done

#!/bin/bash
# tools/pmc_quick.sh OUTDIR -- one PMC pass (instruction and lane counters) over bench.py; BENCH_ARGS as in pmc_passes.sh
set -u
OUT=$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/q" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/q.log" 2>&1 || { echo "pass failed"; tail -3 "$OUT/q.log"; exit 1; }
python3 profiles/summarize_pmc.py render_kernel "$OUT" 

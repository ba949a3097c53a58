#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/profile_grid.sh OUTDIR [OLD_LIB] -- evidence for LUMA_FLAG_GRID, on the GPU box from the repo root: where the
# time of one of config #5's eight bands goes (tools/grid_split_probe.py), kernel time by scene size
# (tools/grid_coop_sweep.py), rocprofv3 --kernel-trace --stats and the PMC passes of that band under bench.py --grid.
# OLD_LIB: an earlier build of libluma_hip.so to run beside the current one (LUMA_HIP_LIB).
set -u
OUT=$1; OLD=${2:-}; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in $OLD raytracer_amd/libluma_hip.so; do
  tag=$(basename "$lib" .so)
  export LUMA_HIP_LIB=$PWD/$lib
  timeout -k 10 200 python3 tools/grid_split_probe.py > "$OUT/grid_split_probe.$tag.txt" 2>&1 || { echo "probe $tag failed"; exit 1; }
  timeout -k 10 300 python3 tools/grid_coop_sweep.py > "$OUT/grid_coop_sweep.$tag.txt" 2>&1 || { echo "sweep $tag failed"; exit 1; }
done
unset LUMA_HIP_LIB
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --no-cpu-baseline --no-frame-leg --workload config5 --emulate-world 8 --grid --steps 5 --warmup 1 > "$OUT/kt.json" 2> "$OUT/kt.err" || { echo "kernel-trace failed"; tail -3 "$OUT/kt.err"; exit 1; }
f=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/config5_band_of_8_grid_kernel_stats.csv"
BENCH_ARGS="--workload config5 --emulate-world 8 --grid" bash tools/pmc_ab.sh "$OUT/pmc" $OLD raytracer_amd/libluma_hip.so > "$OUT/config5_band8_grid_pmc_summary.txt" 2>&1 || { tail -5 "$OUT/config5_band8_grid_pmc_summary.txt"; exit 1; }
echo done

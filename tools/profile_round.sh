#!/bin/bash
# tools/profile_round.sh OUTDIR -- the evidence set of a round, on the GPU box from the repo root:
# rocprofv3 --kernel-trace --stats of the bench workloads, then the PMC passes (tools/pmc_passes.sh)
# for config #3, config #2 and one of config #5's eight bands.  Summaries land in OUTDIR; copy what is
# to be judged into profiles/rNN/.
set -u
OUT=$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {   # name, bench args...
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -- python3 bench.py --no-cpu-baseline --no-frame-leg --no-single-process-leg "$@" > "$OUT/kt_$name.json" 2> "$OUT/kt_$name.err" || { echo "kernel-trace $name failed"; tail -3 "$OUT/kt_$name.err"; exit 1; }
  f=$(find "$OUT/kt_$name" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
  echo "== $name"; head -4 "$OUT/${name}_kernel_stats.csv"
}
stats config3 --steps 100 --warmup 10 || exit 1
for w in 2 4 8; do stats config3_band_of_$w --steps 100 --warmup 10 --emulate-world $w || exit 1; done
stats config2 --workload config2 --steps 50 --warmup 5 || exit 1
stats config5_band_of_8 --workload config5 --emulate-world 8 --steps 3 --warmup 1 || exit 1
stats config5_band_of_8_grid --workload config5 --emulate-world 8 --grid --steps 5 --warmup 1 || exit 1
bash tools/pmc_passes.sh "$OUT/pmc_config3" > "$OUT/config3_pmc_summary.txt" 2>&1 || { tail -5 "$OUT/config3_pmc_summary.txt"; exit 1; }
BENCH_ARGS="--workload config2" bash tools/pmc_passes.sh "$OUT/pmc_config2" > "$OUT/config2_pmc_summary.txt" 2>&1 || exit 1
BENCH_ARGS="--workload config5 --emulate-world 8" bash tools/pmc_passes.sh "$OUT/pmc_config5" > "$OUT/config5_band8_pmc_summary.txt" 2>&1 || exit 1
echo done

#!/bin/bash
# tools/all_configs.sh -- kernel ms, Mrays/s and roofline fraction of every BASELINE GPU config, whole frame on one GPU and
# (configs #3-#5) one of eight bands; config #5 also with LUMA_FLAG_GRID.  Run on the GPU box from the repo root.
pp='import json,sys; d=json.loads(sys.stdin.read()); print("%-8s %-14s kernel %9.3f ms  %9.0f Mrays/s  frac %.3f" % (sys.argv[1], sys.argv[2], d["kernel_ms"], d["value"], d["roofline"]["frac"]))'
run() { name=$1; what=$2; shift 2; timeout -k 10 300 python bench.py --workload $name "$@" --no-cpu-baseline --no-frame-leg 2>/dev/null | python -c "$pp" $name "$what"; }
run config2 whole --steps 200 --warmup 10
run config3 whole --steps 100 --warmup 10
run config3 band-of-8 --steps 100 --warmup 10 --emulate-world 8
run config4 whole --steps 10 --warmup 2
run config4 band-of-8 --steps 20 --warmup 2 --emulate-world 8
run config5 band-of-8 --steps 3 --warmup 1 --emulate-world 8
run config5 band-of-8-grid --steps 10 --warmup 2 --emulate-world 8 --grid
run config5 whole --steps 1 --warmup 1
run config5 whole-grid --steps 3 --warmup 1 --grid

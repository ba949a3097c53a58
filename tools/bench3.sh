#!/bin/bash
# tools/bench3.sh -- GPU parity tests, then the three bench workloads in short form.
pp='import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:44], "ms=%.4f"%d["ms_per_step"], "Mrays/s=%.0f"%d["value"], "frac=%.4f"%d["roofline"]["frac"])'
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
for w in config3 config2; do timeout -k 10 100 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pp"; done
timeout -k 10 300 python bench.py --workload config5 --emulate-world 8 --steps 3 --warmup 1 2>/dev/null | python -c "$pp"

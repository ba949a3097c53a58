#!/bin/bash
# tools/wf_pmc2.sh OUTDIR WF... -- latency / queue-level PMC passes over bench.py for each LUMA_WF value
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wf in "$@"; do
  export LUMA_WF=$wf
  for p in "lat SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "lvl SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
    set -- $p; name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/wf$wf/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame-leg ${BENCH_ARGS:-} > "$OUT/wf$wf.$name.log" 2>&1 || { echo "pass wf$wf/$name failed"; tail -3 "$OUT/wf$wf.$name.log"; exit 1; }
  done
  echo "== LUMA_WF=$wf"; python3 profiles/summarize_pmc.py render_kernel "$OUT/wf$wf" | tee "$OUT/wf$wf.summary.txt"
done

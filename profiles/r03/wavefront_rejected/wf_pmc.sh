#!/bin/bash
# tools/wf_pmc.sh OUTDIR WF... -- two PMC passes (instruction mix; time and lanes) over bench.py for each LUMA_WF value
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wf in "$@"; do
  export LUMA_WF=$wf
  for p in "mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    set -- $p; name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/wf$wf/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame-leg ${BENCH_ARGS:-} > "$OUT/wf$wf.$name.log" 2>&1 || { echo "pass wf$wf/$name failed"; tail -3 "$OUT/wf$wf.$name.log"; exit 1; }
  done
  echo "== LUMA_WF=$wf"; python3 profiles/summarize_pmc.py render_kernel "$OUT/wf$wf" | tee "$OUT/wf$wf.summary.txt"
done

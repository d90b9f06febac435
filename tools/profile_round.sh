#!/bin/bash
# Round profiles on the GPU box: rocprofv3 kernel statistics + last-step summaries of the bench workloads, and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) of tools/pmc_run.py.   bash tools/profile_round.sh <tag> [workloads...]
# Results land in gpurun_out/<tag>/ (copy what is to be kept into profiles/).
set -u
TAG=${1:-r02}; shift || true
WLS=${*:-"sdxl_lora aura_lora lumina_qlora sdxl_full"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$wl" -- python3 "$ROOT/bench.py" --single --workload "$wl" --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/$wl.log" 2>&1
  echo "$wl rc=$?"
  python3 "$ROOT/tools/trace_summary.py" "$OUT/$wl" 45 > "$OUT/${wl}_last_step_summary.txt" 2>&1
  cp "$(find "$OUT/$wl" -name '*_kernel_stats.csv' | head -1)" "$OUT/${wl}_kernel_stats.csv"
  find "$OUT/$wl" -name '*_kernel_trace.csv' -delete   # hundreds of MB: not pulled back
  head -12 "$OUT/${wl}_last_step_summary.txt"
done
cd "$ROOT"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/tools/pmc_run.py" > "$OUT/pmc_f.log" 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/tools/pmc_run.py" > "$OUT/pmc_w.log" 2>&1 )
python3 tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_hbm_traffic.csv" && cat "$OUT/pmc_hbm_traffic.csv"
find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*_kernel_trace.csv' -delete
du -sh "$OUT"

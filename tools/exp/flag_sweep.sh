#!/bin/bash
# A/B of library builds on one box: bash tools/exp/flag_sweep.sh <tag> <name=path> ...   (SDXL-LoRA and AuraFlow-LoRA steps,
# alternating, twice; results in gpurun_out/<tag>/)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for round in 1 2; do
  for spec in "$@"; do
    v=${spec%%=*}; lib=$PWD/${spec#*=}
    for wl in sdxl_lora aura_lora; do
      VFT_LIB_PATH=$lib timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload $wl --steps 8 --warmup 2 > gpurun_out/$TAG/${v}_${wl}_$round.json 2> gpurun_out/$TAG/${v}_${wl}_$round.err || echo "FAIL $v $wl"
      python - <<PY
import json
l=[x for x in open("gpurun_out/$TAG/${v}_${wl}_$round.json") if x.startswith("{")]
d=json.loads(l[-1]); print("$v $wl $round", d["value"], d["ms_per_step"], d["roofline"]["families_ms_frac"].get("linear_gemm"), d["roofline"]["families_ms_frac"].get("conv3x3"), flush=True)
PY
    done
  done
done

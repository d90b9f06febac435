#!/bin/bash
# A/B of library builds (VFT_LIB_PATH) on one box: SDXL-LoRA and AuraFlow-LoRA step, alternating; results in gpurun_out/r5n/
mkdir -p gpurun_out/r5n
for round in 1 2; do
  for v in hip maxilp trackers bias0 nopost; do
    if [ $v = hip ]; then lib=$PWD/vision_ft_amd/libvft_hip.so; else lib=$PWD/tools/exp/libvft_$v.so; fi
    for wl in sdxl_lora aura_lora; do
      VFT_LIB_PATH=$lib timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload $wl --steps 8 --warmup 2 > gpurun_out/r5n/${v}_${wl}_$round.json 2> gpurun_out/r5n/${v}_${wl}_$round.err || echo "FAIL $v $wl"
      python - <<PY
import json
l=[x for x in open("gpurun_out/r5n/${v}_${wl}_$round.json") if x.startswith("{")]
d=json.loads(l[-1]); print("$v $wl $round", d["value"], d["ms_per_step"], flush=True)
PY
    done
  done
done

#!/bin/bash
# Produce vision_ft_amd/tuning_tables/gfx950_gemm.csv on an MI355X: every bench workload once with
# library-GEMM tuning on (vision_ft_amd/tuning.py).  Run from the repo root; ~5 minutes.
set -e
for w in sdxl_lora sdxl_full aura_lora lumina_lora lumina_qlora sdxl_buckets; do
    python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --tune-gemms | cut -c1-200
done
wc -l vision_ft_amd/tuning_tables/gfx950_gemm.csv

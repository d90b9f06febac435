#!/bin/bash
# two PMC passes over tools/exp/pmc_attn64.py (one shape, variants in VFT_PMC_VARIANTS); tables into gpurun_out/$1/
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/exp/pmc_attn64.py 24,10,4096,64 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/exp/pmc_attn64.py 24,10,4096,64 > $OUT/p2.log 2>&1
cd $ROOT
python3 tools/exp/pmc_table.py $OUT/p1 attn_bwd1 > $OUT/table.txt 2>&1; python3 tools/exp/pmc_table.py $OUT/p2 attn_bwd1 >> $OUT/table.txt 2>&1
find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*_kernel_trace.csv' -delete
cat $OUT/table.txt

#!/bin/bash
mkdir -p gpurun_out/r03
args="--int8 --ctx 4096"
bash tools/pmc.sh attn_i8_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" python tools/attn_bench.py $args --iters 4 > /dev/null 2>&1
bash tools/pmc.sh attn_i8_b "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" python tools/attn_bench.py $args --iters 4 > /dev/null 2>&1
bash tools/pmc.sh attn_i8_c "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" python tools/attn_bench.py $args --iters 4 > /dev/null 2>&1
for s in a b c; do python tools/pmc_sum.py gpurun_out/pmc_attn_i8_$s paged_attn; done > gpurun_out/r03/attn_pmc_int8_after.txt 2>&1
cat gpurun_out/r03/attn_pmc_int8_after.txt

#!/bin/bash
# instruction mix of the INT8-KV attention kernel (b=64, ctx 4096) vs fp16 (ctx 2048): separate PMC passes
mkdir -p gpurun_out/r03
for mode in int8 fp16; do
  if [ $mode = int8 ]; then args="--int8 --ctx 4096"; else args="--ctx 2048"; fi
  bash tools/pmc.sh attn_${mode}_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" python tools/attn_bench.py $args --iters 4 > /dev/null 2>&1
  bash tools/pmc.sh attn_${mode}_b "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" python tools/attn_bench.py $args --iters 4 > /dev/null 2>&1
  bash tools/pmc.sh attn_${mode}_c "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16" python tools/attn_bench.py $args --iters 4 > /dev/null 2>&1
  for s in a b c; do python tools/pmc_sum.py gpurun_out/pmc_attn_${mode}_$s paged_attn; done
done > gpurun_out/r03/attn_pmc_int8_vs_fp16.txt 2>&1
cat gpurun_out/r03/attn_pmc_int8_vs_fp16.txt

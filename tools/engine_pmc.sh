#!/bin/bash
# SQ counters of the ENGINE's own launches (b = 64 headline step): two rocprofv3 --pmc passes (--kernel-trace only) over `bench.py --no-graph`, summarised per
# kernel by tools/engine_pmc.py (BENCH_EXTRA: extra bench.py arguments, e.g. --debug-set switches of the tuning build; PMC_TAG: suffix of the output file): wave cycles split into waiting / issue-stalled / active, VALU and MFMA instruction counts, MFMA pipe busy against the
# kernel's duration (the "MFMA utilisation against chip peak" of north_star).  usage (GPU box): bash tools/engine_pmc.sh -> gpurun_out/$ROUND/pmc_engine_sq.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND:-r05}; mkdir -p $O
export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd $R && timeout 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_sq_$i -o run -- \
      python bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-sweep $BENCH_EXTRA > $O/pmc_sq_$i.log 2>&1 )
done
python $R/tools/engine_pmc.py $O/pmc_sq_1 $O/pmc_sq_2 > $O/pmc_engine_sq$PMC_TAG.txt
rm -rf $O/pmc_sq_1 $O/pmc_sq_2
cat $O/pmc_engine_sq$PMC_TAG.txt

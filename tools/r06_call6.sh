#!/bin/bash
cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rope or prefill or generate" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -12 )
for b in 16 17 32 33 48 49; do bash tools/trace_bench.sh b$b --batch $b --steps 8 --warmup 2 --no-cpu-baseline --no-sweep; echo "== b=$b"; head -10 $O/kernel_stats_b$b.txt | cut -c1-150; done

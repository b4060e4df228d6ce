#!/bin/bash
# two chunks of weights ahead in the wide GEMM at <= 32 rows (A/B through the tuning build's switch), the in-launch prefetch of the fused
# all-reduce (2 ranks on one GPU), then the default bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fused_linear.py tests/test_gpu_parity.py -x -q -m gpu -k "wide or deferred or image or img or silu" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_allreduce.py -x -q -m gpu -k "kernels-2 or engine-2" 2>&1 | tail -4
timeout 600 python tools/wide_img_time.py --ms 8,16,32 --tuning --dbg 0,16,2,3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/wide_ring2.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_ring2.json 2> gpurun_out/r04/bench_ring2.log; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_ring2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('batch_sweep') or d.get('sweep'))
PY

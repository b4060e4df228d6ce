cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
( python -m pytest tests/test_gpu_fused_linear.py tests/test_gpu_baseline_shapes.py tests/test_gpu_full_depth.py tests/test_gpu_bf16.py tests/test_gpu_allreduce.py tests/test_gpu_tp_engine.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -8 )
python bench.py --no-cpu-baseline --steps 64 2>/dev/null | tail -1 > $O/bench_jobs.json; python -c "
import json; d=json.load(open('$O/bench_jobs.json')); print(d['ms_per_step'], d['ms_per_step_repeats'], d.get('sweep'))"

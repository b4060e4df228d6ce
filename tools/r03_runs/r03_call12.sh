#!/bin/bash
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -x -q -m gpu -k "rmsnorm or engine or partial" > gpurun_out/r03/pytest_norm.txt 2>&1; echo "rc=$?" >> gpurun_out/r03/pytest_norm.txt
tail -3 gpurun_out/r03/pytest_norm.txt
python bench.py --no-sweep --no-cpu-baseline --steps 64 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['frac'], d['step_roofline']['eager_kernel_ms_per_step'])"

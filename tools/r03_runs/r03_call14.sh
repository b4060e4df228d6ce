#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_full2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/pytest_full2.txt
tail -4 gpurun_out/r03/pytest_full2.txt
python bench.py --no-sweep --no-cpu-baseline --steps 64 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['step_roofline']['eager_kernel_ms_per_step'])"
python tools/attn_bench.py 2>&1 | tail -1; python tools/attn_bench.py --ctx 4096 2>&1 | tail -1; python tools/attn_bench.py --ctx 4096 --int8 2>&1 | tail -1

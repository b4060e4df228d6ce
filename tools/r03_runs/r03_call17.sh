#!/bin/bash
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -k "attention or engine_greedy or int8_kv" 2>&1 | tail -3
{ python tools/attn_bench.py; python tools/attn_bench.py --ctx 4096; python tools/attn_bench.py --ctx 4096 --int8; python tools/attn_bench.py --ctx 1024 --int8; python tools/attn_bench.py --batch 16; python tools/attn_bench.py --page 64; } 2>&1 | grep -v amdgpu

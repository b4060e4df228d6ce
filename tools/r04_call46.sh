#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_native_ops.py -x -q -m gpu -k "rope" -s 2>&1 | grep -E "passed|failed|error|Error|assert|scaled RoPE" | tail -12 )

#!/bin/bash
# down_proj at a few rows: weight ring of 3 / 5 chunks (5 = the ten-chunk slice in two rounds) against 2
timeout 400 python tools/fullk_time.py --sweep 5=0,3,5 --ms 1,8 2>&1 | grep -v amdgpu.ids

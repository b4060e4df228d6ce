#!/bin/bash
# timing-only variants of the full-K launches (tuning build, key 5): which of the fixed costs is worth removing
timeout 500 python tools/fullk_time.py --sweep 5=0,1,2,4,8,16,6,31 --ms 1,8 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# (1) cold-window burst reads: what one launch can get out of HBM for 68 / 34 / 8 MB, by access pattern
# (2) full-K launches, K slices blocked (4=1) vs interleaved (4=2), same library (norm partial sums first in both)
# (3) stamps of the new source
timeout 200 tools/probe/burst_read 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/fullk_time.py --set 4=1 --ms 1,8,16 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/fullk_time.py --set 4=2 --ms 1,8,16 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/fullk_stamps.py --ms 1 --set 4=1 2>&1 | grep -v amdgpu.ids

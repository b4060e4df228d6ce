#!/bin/bash
# per-wave stamps of the full-K launches at 1 and 8 rows
timeout 300 python tools/fullk_stamps.py --ms 1,8 2>&1 | grep -v amdgpu.ids

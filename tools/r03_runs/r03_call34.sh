#!/bin/bash
timeout 300 python tools/fullk_stamps.py --ms 1 2>&1 | grep -v amdgpu.ids

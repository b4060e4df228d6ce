#!/bin/bash
for fr in 8 16; do echo "== fuse_rows $fr"; for b in 8 9 12 16; do MI355_FUSE_ROWS=$fr python bench.py --no-sweep --no-cpu-baseline --batch $b --steps 48 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b', d['config']['batch'], d['ms_per_step'], d['step_roofline']['eager_kernel_ms_per_step'])"; done; done

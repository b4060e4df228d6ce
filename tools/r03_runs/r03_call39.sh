#!/bin/bash
# final state of the round: HBM traffic of the engine's launches + the default bench line, then the whole GPU suite (4 workers: the
# time is the CPU oracle's)
bash tools/r03_collect_b.sh 2>&1 | grep -v amdgpu.ids | tail -14
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -6

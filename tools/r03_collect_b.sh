#!/bin/bash
# part B: HBM traffic of the engine's own launches (separate FETCH_SIZE / WRITE_SIZE passes), then the default bench line
export ROUND=r03
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
bash $R/tools/engine_traffic.sh | tail -12
cd $R
cp $O/traffic.json $R/profiles/r03_traffic.json 2>/dev/null
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']); print([ (s['batch'], s['ms_per_step']) for s in d['sweep']])"

#!/bin/bash
# round-3 closing measurements: bench lines of every configuration (fresh process each)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final_r3
rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; exit 1; }; python3 -c "
import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value'],1), d['unit'], 'ms/step', round(d['ms_per_step'],5), 'frac', d.get('roofline',{}).get('frac'))"; }
run r03_bench_c2_default
run r03_bench_c2_steps20 --gpus 1 --steps 20 --warmup 5
run r03_bench_c3 --config C3
run r03_bench_cship --config Cship
run r03_bench_c2_impl4_fast --pyramid-impl 4 --no-cpu
run r03_bench_c4_streams16 --streams 16 --no-cpu
run r03_bench_c4_streams64 --streams 64 --no-cpu
run r03_bench_c5 --config C5
run r03_bench_c2_2ranks_shared --gpus 2 --share-gpu --steps 64 --warmup 16 --no-cpu

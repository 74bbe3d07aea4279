#!/bin/bash
# A/B: coarse levels of BATCHED builds in one launch (GKLT_CHAIN=2) vs one launch per level (1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_chain
rm -rf $O; mkdir -p $O
GKLT_CHAIN=2 timeout -k 10 300 python -u -m pytest tests/test_gpu_lookahead.py tests/test_gpu_multistream.py tests/test_gpu_track.py -x -q --timeout 120 --timeout-method thread > $O/tests_chain2.log 2>&1 || { tail -20 $O/tests_chain2.log; exit 1; }
tail -1 $O/tests_chain2.log
for rep in 1 2; do
for CFG in C2 Cship C3; do
  for K in 1 2; do
    GKLT_CHAIN=$K timeout -k 10 200 python3 bench.py --config $CFG --no-cpu --main-only --steps 2048 --warmup 128 > $O/${CFG}_$K.json 2> $O/${CFG}_$K.err || exit 1
    python3 -c "
import json; d=json.load(open('$O/${CFG}_$K.json')); print('$CFG chain $K: %.1f M, kernels %s' % (d['value']/1e6, {k: (round(v['avg_us'],1), v['launches']) for k,v in d['kernels'].items()}))" | tee -a $O/summary.txt
  done
done
done
for K in 1 2; do
  GKLT_CHAIN=$K timeout -k 10 200 python3 bench.py --streams 16 --no-cpu --steps 512 --warmup 64 > $O/ms16_$K.json 2> $O/ms16_$K.err || exit 1
  python3 -c "
import json; d=json.load(open('$O/ms16_$K.json')); print('16 streams chain $K: %.1f M' % (d['value']/1e6))" | tee -a $O/summary.txt
done

#!/bin/bash
# A/B: build stream and tracker stream on disjoint CU sets (GKLT_CU_SPLIT) vs sharing the chip
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_cumask
rm -rf $O; mkdir -p $O
for CFG in Cship C2; do
  for K in 0 96 128 160; do
    GKLT_CU_SPLIT=$K timeout -k 10 200 python3 bench.py --config $CFG --no-cpu --main-only --steps 1024 --warmup 64 > $O/${CFG}_$K.json 2> $O/${CFG}_$K.err || exit 1
    python3 -c "
import json; d=json.load(open('$O/${CFG}_$K.json')); print('$CFG split $K: %.1f M, kernels %s' % (d['value']/1e6, {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}))" | tee -a $O/summary.txt
  done
done

#!/bin/bash
# tracker wavefront cap (gklt_ctx_set_track_waves) against the batch period, one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/sweep_waves
rm -rf $O; mkdir -p $O
for CFG in ${CFGS:-C2}; do
for W in ${WAVES:-0 2048 3072 4096 6144}; do
  timeout -k 10 300 python3 bench.py --config $CFG --track-waves $W --no-cpu --no-pcie --no-shim --steps 1024 --warmup 64 > $O/${CFG}_$W.json 2> $O/${CFG}_$W.err || exit 1
  python3 -c "
import json; d=json.load(open('$O/${CFG}_$W.json')); a=d['roofline'].get('level0_alone') or {}
print('$CFG waves $W: %.1f M, kernels %s' % (d['value']/1e6, {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}))" | tee -a $O/summary.txt
done
done

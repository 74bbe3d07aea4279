#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/sweep_waves_ms
rm -rf $O; mkdir -p $O
for W in ${WAVES:-4096 6144 8192}; do
  timeout -k 10 300 python3 bench.py --streams 16 --track-waves $W --no-cpu --steps 256 --warmup 32 > $O/ms_$W.json 2> $O/ms_$W.err || exit 1
  python3 -c "
import json; d=json.load(open('$O/ms_$W.json'))
print('16 streams waves $W: %.1f M' % (d['value']/1e6))" | tee -a $O/summary.txt
done

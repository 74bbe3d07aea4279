#!/bin/bash
# A/B of two builds of the library on one box: bench lines of the default library and of $1
# (GKLT_LIB override, tools/bin/...), configs in $2 (default "C2")
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ALT=$1; CFGS=${2:-C2}
O=gpurun_out/ab_lib
rm -rf $O; mkdir -p $O
for CFG in $CFGS; do
  for L in base alt base alt; do
    if [ $L = alt ]; then export GKLT_LIB=$GRAFT_REPO_ROOT/$ALT; else unset GKLT_LIB; fi
    timeout -k 10 300 python3 bench.py --config $CFG --no-cpu --no-pcie --no-shim --steps 1024 --warmup 64 > $O/${CFG}_$L.json 2> $O/${CFG}_$L.err || exit 1
    python3 -c "
import json; d=json.load(open('$O/${CFG}_$L.json')); a=d['roofline'].get('level0_alone') or {}
print('$CFG $L: %.1f M, kernels %s, level0 alone %.1f us (%.3f of the HBM roof)' % (d['value']/1e6, {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, a.get('avg_launch_us', 0), a.get('frac', 0)))" | tee -a $O/summary.txt
  done
done

#!/bin/bash
# exact (impl 0) vs contracted + polyphase (impl 4) level kernels: bench lines and VALU counts
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/ab_impl4
rm -rf $O; mkdir -p $O
for CFG in C2 Cship C3; do
  for I in 0 4; do
    timeout -k 10 300 python3 bench.py --config $CFG --pyramid-impl $I --no-cpu --no-pcie --no-shim --steps 1024 --warmup 64 > $O/${CFG}_$I.json 2> $O/${CFG}_$I.err || exit 1
    python3 -c "
import json; d=json.load(open('$O/${CFG}_$I.json')); a=d['roofline'].get('level0_alone') or {}
print('$CFG impl $I: %.1f M, kernels %s, level0 alone %.1f us (%.3f of the HBM roof)' % (d['value']/1e6, {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, a.get('avg_launch_us', 0), a.get('frac', 0)))" | tee -a $O/summary.txt
  done
done
for I in 0 4; do
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace -d $O/p$I -o p -- python3 bench.py --config C2 --pyramid-impl $I --no-cpu --main-only --steps 64 --warmup 16 > $O/p$I.out 2>&1 || exit 1
  echo "== C2 impl $I" >> $O/summary.txt
  python3 tools/pmc_summary.py $O/p$I | grep "unsigned char" >> $O/summary.txt
done
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
cat $O/summary.txt

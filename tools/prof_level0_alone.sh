#!/bin/bash
# rocprofv3 kernel statistics of the level kernels with nothing beside them (empty feature list, one
# build stream): tools/exp_alone.py, 32 batches of 16 frames
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GKLT_COARSE_SPLIT=0
export CFG=${CFG:-C2}
O=gpurun_out/prof_l0alone_$CFG
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o t -- python3 tools/exp_alone.py > $O/out.txt 2> $O/trace.err || exit 1
find $O -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*.db' -delete
cat $O/out.txt; head -5 $O/kernel_stats.csv | cut -c1-120,300-420

#!/bin/bash
# HIP API time breakdown (rocprofv3 --hip-trace --stats) of the frame-at-a-time host loop (tools/bin/pipe_host)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_host
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --hip-trace --stats --output-format csv -d $O -o h -- $R/tools/bin/pipe_host > $O/out.txt 2>&1
cd $R
tail -4 $O/out.txt
find $O -name "*hip_api_stats.csv" -exec cat {} \; | head -20
find $O -name "*trace.csv" -delete; find $O -name "*.db" -delete

#!/bin/bash
# kernel timeline of the default bench (lookahead 16) for gap analysis; $1 = extra bench flags
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_tl
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o t -- python3 bench.py --no-cpu --main-only --steps 512 --warmup 64 $1 > $O/bench.json 2> $O/trace.err
python3 tools/trace_summary.py $O/trace 60 > $O/timeline.txt 2>&1
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*.db' -delete

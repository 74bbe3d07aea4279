set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_ms
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o t -- python3 tools/exp_multistream.py ${1:-4} > $O/exp.out 2> $O/trace.err
python3 tools/trace_summary.py $O/trace 40 > $O/kernel_by_grid.txt 2>&1 || true
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*.db' -delete
cat $O/exp.out

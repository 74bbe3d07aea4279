set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CFG=${CFG:-C2}
O=gpurun_out/prof_final_$CFG$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 240 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --config $CFG $EXTRA --no-cpu --main-only --steps 2048 --warmup 128 > $O/bench_under_rocprof.json 2> $O/trace.err
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/p1 -o p -- python3 bench.py --config $CFG $EXTRA --no-cpu --main-only --steps 64 --warmup 16 > $O/p1.out 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $O/p2 -o p -- python3 bench.py --config $CFG $EXTRA --no-cpu --main-only --steps 64 --warmup 16 > $O/p2.out 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/p3 -o p -- python3 bench.py --config $CFG $EXTRA --no-cpu --main-only --steps 64 --warmup 16 > $O/p3.out 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace -d $O/p4 -o p -- python3 bench.py --config $CFG $EXTRA --no-cpu --main-only --steps 64 --warmup 16 > $O/p4.out 2>&1
python3 tools/pmc_summary.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/pmc_by_kernel.txt
python3 tools/trace_summary.py $O/trace > $O/kernel_by_grid.txt 2>&1 || true
find $O -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
tail -3 $O/bench_under_rocprof.json | cut -c1-400

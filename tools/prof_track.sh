# PMC passes over the tracker alone (tools/exp_track.py), one mapping per run: IMPLS=<n> CFG=<config>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_track_${CFG}_${IMPLS}
rm -rf $O; mkdir -p $O
timeout -s KILL 100 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace -d $O/p1 -o p -- python3 tools/exp_track.py $CFG > $O/p1.out 2>&1 &&
timeout -s KILL 100 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace -d $O/p2 -o p -- python3 tools/exp_track.py $CFG > $O/p2.out 2>&1 &&
python3 tools/pmc_summary.py $O/p1 $O/p2 > $O/pmc_by_kernel.txt
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
cat $O/pmc_by_kernel.txt; tail -2 $O/p1.out $O/p2.out

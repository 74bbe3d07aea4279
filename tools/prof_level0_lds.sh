#!/bin/bash
# what bounds the level-0 kernel inside a CU: LDS activity / bank conflicts / VALU activity per launch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_l0lds$TAG
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/p1 -o p -- python3 bench.py --config ${CFG:-C2} --no-cpu --main-only --steps 64 --warmup 16 > $O/p1.out 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace -d $O/p2 -o p -- python3 bench.py --config ${CFG:-C2} --no-cpu --main-only --steps 64 --warmup 16 > $O/p2.out 2>&1 || exit 1
python3 tools/pmc_summary.py $O/p1 $O/p2 > $O/pmc_by_kernel.txt
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
cat $O/pmc_by_kernel.txt

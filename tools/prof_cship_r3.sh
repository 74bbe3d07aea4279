#!/bin/bash
# Why does k_level<u8> take twice as long beside the Cship tracker launch?  Memory-path counters of the
# level-0 kernel ALONE (tools/exp_alone.py, empty feature list) and OVERLAPPED with the tracker
# (bench.py --config Cship --main-only), one rocprofv3 --pmc pass per counter group and run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_cship_r3
rm -rf $O; mkdir -p $O
G1="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G2="TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
G3="TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
G4="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  CFG=Cship timeout -s KILL 100 rocprofv3 --output-format csv --pmc $G --kernel-trace -d $O/alone$i -o p -- python3 tools/exp_alone.py > $O/alone$i.out 2>&1 || exit 1
  timeout -s KILL 100 rocprofv3 --output-format csv --pmc $G --kernel-trace -d $O/over$i -o p -- python3 bench.py --config Cship --no-cpu --main-only --steps 64 --warmup 16 > $O/over$i.out 2>&1 || exit 1
done
echo "== alone (level kernels only, empty feature list)" > $O/pmc_cship.txt
python3 tools/pmc_summary.py $O/alone1 $O/alone2 $O/alone3 $O/alone4 >> $O/pmc_cship.txt
echo "== overlapped with the tracker launch (bench.py --config Cship --main-only)" >> $O/pmc_cship.txt
python3 tools/pmc_summary.py $O/over1 $O/over2 $O/over3 $O/over4 >> $O/pmc_cship.txt
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
cat $O/pmc_cship.txt

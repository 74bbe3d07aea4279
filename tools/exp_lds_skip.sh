#!/bin/bash
# LDS counters of the level kernels with one class of LDS access removed per build (tools/bin/libgklt_exp<K>.so,
# outputs wrong, only the counters matter): which accesses the bank-conflict cycles belong to
# (the variant builds were made from a scratch copy of gklt_pyramid.hip in which the reads / writes of one
# phase of level_tile were replaced by register constants / made conditional on an impossible value, one
# class per build: second row pass reads 1, its writes 2, gradient column reads 4, pyramid column reads 8,
# centre pixel reads 16; they are not kept in the tree.  KS lists the variants present; 0 = the library)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/exp_lds_skip
rm -rf $O; mkdir -p $O
for K in 0 $KS; do
  if [ $K = 0 ]; then unset GKLT_LIB; else export GKLT_LIB=$GRAFT_REPO_ROOT/tools/bin/libgklt_exp$K.so; fi
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/p$K -o p -- python3 bench.py --config C2 --no-cpu --main-only --steps 64 --warmup 16 > $O/p$K.out 2>&1 || exit 1
  echo "== skip $K" >> $O/summary.txt
  python3 tools/pmc_summary.py $O/p$K | grep -E "k_level<(unsigned char, 2, 5, 3, tr   grid 4177920|float, 0, 5, 3, true, 256>   grid 1044480)" | sed -E 's/SQ_(WAVES|BUSY_CYCLES|WAIT_INST_LDS|ACTIVE_INST_VALU)=[^ ]+ *//g' >> $O/summary.txt
done
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
cat $O/summary.txt

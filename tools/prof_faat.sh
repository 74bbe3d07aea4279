#!/bin/bash
# kernel timeline of frame-at-a-time operation (gklt_pipeline_step), steady state: last 40 kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_faat
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o t -- python3 bench.py --no-cpu --main-only --lookahead 0 --steps 96 --warmup 16 $1 > $O/bench.json 2> $O/trace.err
python3 - <<'PY' > $O/timeline.txt
import csv, glob
f = glob.glob("gpurun_out/prof_faat/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed leg: kernels 400..460 are well inside it
rows = rows[300:345]
base = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].replace("void gklt::(anonymous namespace)::", "").split("(")[0][:44]
    print("%10.1f %8.1f  q%s %-44s %sx%sx%s" % ((int(r["Start_Timestamp"]) - base) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
PY
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*.db' -delete
cat $O/timeline.txt

#!/bin/bash
# suppression rounds of the C5 selection for several iteration limits per launch: launches per
# selection and their summed / first / median duration (rocprofv3 kernel trace of tools/exp_c5.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_c5_iters
rm -rf $O; mkdir -p $O
for it in "$@"; do
  timeout -k 10 120 rocprofv3 --output-format csv --kernel-trace -d $O/t$it -o t -- python3 tools/exp_c5.py uniform $it > $O/out_$it.txt 2> $O/err_$it.txt || exit 1
  python3 - $it <<'PY' >> $O/summary.txt
import csv, glob, sys, statistics
it = sys.argv[1]
rows = []
for f in glob.glob("gpurun_out/prof_c5_iters/t%s/**/*kernel_trace.csv" % it, recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sel, cur = [], None
for s, e, n in rows:
    if "k_score" in n:
        cur = []
        sel.append(cur)
    elif "k_tops_round" in n and cur is not None:
        cur.append((e - s) / 1e3)
sel = [c for c in sel if c][len(sel) // 2:]
print("iters %s: selections %d, launches %.1f, sum %.1f us, first %.1f, median launch %.1f" % (
    it, len(sel), statistics.mean(len(c) for c in sel), statistics.mean(sum(c) for c in sel),
    statistics.mean(c[0] for c in sel), statistics.median(x for c in sel for x in c)))
print("   one selection:", " ".join("%.0f" % x for x in sel[-1]))
PY
  rm -rf $O/t$it
  tail -3 $O/out_$it.txt >> $O/summary.txt
done
cat $O/summary.txt

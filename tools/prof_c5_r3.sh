#!/bin/bash
# kernel + copy timeline of one steady-state selection of `bench.py --config C5`, and the bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_c5_r3
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $O/trace -o t -- python3 bench.py --config C5 --no-cpu --steps 24 --warmup 8 > $O/bench.json 2> $O/trace.err || exit 1
python3 - <<'PY' > $O/timeline.txt
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_c5_r3/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("gklt::(anonymous namespace)::", "").split("(")[0][:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for f in glob.glob("gpurun_out/prof_c5_r3/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_score")]
if not idx:
    print(sorted(set(r[2] for r in rows)))
print("rows", len(rows), "selections", len(idx))
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
base = rows[i0][0]
prev_end = base
for s, e, n in rows[i0:i1 + 1]:
    print("%9.1f  +%7.1f  gap %6.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = max(prev_end, e)
PY
find $O -name '*.db' -delete
rm -rf $O/trace
cat $O/timeline.txt

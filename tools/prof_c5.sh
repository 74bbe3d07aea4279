# kernel trace of tools/exp_c5.py <variant> <iters>; prints the launches of the last selection in order
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=${1:-plain}; I=${2:-1}
O=gpurun_out/prof_c5_$V
rm -rf $O; mkdir -p $O
timeout -k 10 240 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o t -- python3 tools/exp_c5.py $V $I > $O/out.txt 2> $O/err.txt
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("gklt::(anonymous namespace)::", "")
    return n.split("(")[0][:40]
# last selection = from the last k_score to the next k_track
idx = [i for i, r in enumerate(rows) if "k_score" in r["Kernel_Name"]]
i0 = idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
with open(O + "/last_selection.txt", "w") as out:
    for r in rows[i0 - 3:idx[-1]]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.write("%9.1f us  +%7.1f us  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"])))
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
cat $O/out.txt

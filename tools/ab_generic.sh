# A/B: runs bench.py with each argument string in ARGS (separated by |), prints the headline numbers
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VARIANTS <<< "$ARGS"
for rep in 1 2; do
i=0
for v in "${VARIANTS[@]}"; do
  i=$((i+1))
  timeout -k 10 150 python bench.py $v --no-pcie --no-cpu > gpurun_out/abg_$i.log 2>&1 || { tail -5 gpurun_out/abg_$i.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/abg_$i.log").read().strip().split("\n")[-1])
k=d.get("kernels",{})
print("%-60s %.1f M  faat %.1f M  " % ("$v", d["value"]/1e6, d["frame_at_a_time"]["value"]/1e6) + "  ".join("%s %.0f" % (n, k[n]["avg_us"]) for n in k))
PY
done
done

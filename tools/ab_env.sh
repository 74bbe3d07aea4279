# A/B over environment settings: ENVS="A=1|B=2 C=3|" (| separated, empty = baseline), ARGS = bench flags
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VARIANTS <<< "$ENVS|"
for rep in 1 2; do
i=0
for v in "${VARIANTS[@]}"; do
  i=$((i+1))
  env $v timeout -k 10 150 python bench.py $ARGS --no-pcie --no-cpu > gpurun_out/abe_$i.log 2>&1 || { tail -5 gpurun_out/abe_$i.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/abe_$i.log").read().strip().split("\n")[-1])
k=d.get("kernels",{})
r=d.get("roofline",{})
print("%-44s %.1f M  faat %.1f M  " % ("$v", d["value"]/1e6, d["frame_at_a_time"]["value"]/1e6) + "  ".join("%s %.0f" % (n, k[n]["avg_us"]) for n in k) + "  frac %.3f" % r.get("frac", 0))
PY
done
done

# A/B of tracker knobs on the Cship lookahead bench: prints value, frame_at_a_time, kernel times
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for h in 0 1 2; do
  GKLT_TRACK_HALO=$h timeout -k 10 120 python bench.py --config ${CFG:-Cship} --steps 128 --warmup 32 --no-pcie --no-cpu > gpurun_out/ab_$h.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$h.log").read().strip().split("\n")[-1])
k=d["kernels"]
print("halo $h: %.1f M  faat %.1f M  track %.0f us  level0 %.0f us" % (d["value"]/1e6, d["frame_at_a_time"]["value"]/1e6, k["track"]["avg_us"], k["level0"]["avg_us"]))
PY
done
done

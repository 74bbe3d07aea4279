#!/usr/bin/env python3
"""The shim drop-in loop alone (bench.py's shim_drop_in block), with the per-part breakdown."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GKLT_DRV_BREAKDOWN"] = "1"
os.environ["GKLT_SHIM_BREAKDOWN"] = "1"
from gslam_amd import klt, synth
import bench
klt.lib()
frames = synth.make_stream(1080, 1920, 16)
print(json.dumps(bench.shim_drop_in(frames, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 128), indent=1))

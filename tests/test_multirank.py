"""world_size-2 gloo test of bench.py's multi-GPU harness logic (no GPU): stream sharding by rank
and the max-time / sum-count reduction.  The data path itself has no collective."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    r = dist.get_rank()
    # each rank "measured" a different time and count
    dt, tracked, iters = bench.reduce_over_ranks(dist, 1.0 + r, 1000 * (r + 1), 10 * (r + 1))
    seeds = bench.rank_seed(r)
    out = dict(rank=r, dt=dt, tracked=tracked, iters=iters, seed=seeds,
               streams=bench.shard_streams(7, dist.get_world_size(), r))
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_reduction_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    res = sorted((json.loads(o[0].split("RESULT ")[1].splitlines()[0]) for o in outs), key=lambda d: d["rank"])
    for d in res:
        assert d["dt"] == 2.0            # max over ranks
        assert d["tracked"] == 3000.0    # summed over ranks
        assert d["iters"] == 30.0
    assert res[0]["seed"] != res[1]["seed"]   # independent streams per rank
    # config 4: stream s -> rank s mod world; every stream on exactly one rank
    assert res[0]["streams"] == [0, 2, 4, 6] and res[1]["streams"] == [1, 3, 5]

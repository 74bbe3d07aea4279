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


def _bench(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=e,
                          capture_output=True, text=True, timeout=120)


def test_bench_launcher_starts_one_rank_per_gpu():
    """`bench.py --gpus N` outside torch.distributed.run starts N ranks itself (no HIP call in the
    launcher): every rank sees its own RANK / LOCAL_RANK -> device and the config-4 partition."""
    import json
    r = _bench("--gpus", "4", "--streams", "64", "--dry-launch")
    assert r.returncode == 0, r.stderr
    ranks = [json.loads(l) for l in r.stdout.splitlines()]
    assert [d["rank"] for d in ranks] == [0, 1, 2, 3]
    assert all(d["world"] == 4 and d["device"] == d["local_rank"] == d["rank"] for d in ranks)
    assert sorted(s for d in ranks for s in d["streams"]) == list(range(64))
    assert all(len(d["streams"]) == 16 for d in ranks)
    # rehearsal on a 1-GPU box: every rank on device 0
    r = _bench("--gpus", "2", "--share-gpu", "--dry-launch")
    assert [json.loads(l)["device"] for l in r.stdout.splitlines()] == [0, 0]


def test_bench_rejects_gpus_that_disagree_with_world_size():
    r = _bench("--gpus", "8", "--dry-launch", env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    # bad arguments fail in the launcher, before any rank starts
    r = _bench("--gpus", "2", "--dry-launch", "--config", "nope")
    assert r.returncode != 0


def test_bench_launcher_does_not_hang_when_a_rank_dies():
    """A rank that exits before the first barrier leaves the others waiting: the launcher kills them
    after the grace period and reports failure; a job that outlives --launch-timeout is killed too."""
    import time
    t0 = time.monotonic()
    r = _bench("--gpus", "3", "--dry-launch", "--fail-rank", "1", "--launch-grace", "1")
    assert r.returncode != 0 and time.monotonic() - t0 < 60
    assert "rank 1 exited with code 3" in r.stderr and "killing rank 0 (another rank failed)" in r.stderr
    t0 = time.monotonic()
    r = _bench("--gpus", "2", "--dry-launch", "--fail-rank", "7", "--launch-timeout", "2")   # nobody fails, all wait
    assert r.returncode != 0 and time.monotonic() - t0 < 60
    assert "deadline of 2 s passed" in r.stderr

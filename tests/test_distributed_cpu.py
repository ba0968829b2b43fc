"""CPU suite, part 4: the N > 1 path -- shard arithmetic and the optional observation gather -- with two
gloo ranks on CPU tensors (the kernels themselves need no collective)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["T1D_ROOT"])
import torch, torch.distributed as dist
from simglucose_amd.distributed import shard_range, gather_observations, local_actions
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for n_total in (10, 7, 1024):
    lo, hi = shard_range(n_total, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float64) * 1.5 + 0.25     # stands for this shard's CGM slice
    full = gather_observations(local, n_total)
    assert full.shape == (n_total,) and torch.equal(full, torch.arange(n_total, dtype=torch.float64) * 1.5 + 0.25), (rank, n_total)
    acts = local_actions(full * 2, n_total)
    assert torch.equal(acts, local * 2)
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                 # bench.py's max-over-ranks timing
assert float(t) == world
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_shard_range_partitions():
    from simglucose_amd.distributed import shard_range
    for n, w in ((1 << 20, 8), (10, 3), (7, 8), (61440, 4)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, T1D_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


BENCH_WORKER = r'''
import io, json, os, sys, contextlib
sys.path.insert(0, os.environ["T1D_ROOT"])
import torch, torch.distributed as dist
import bench
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
plans = {}
for scaling in ("strong", "weak"):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--envs", "1048577", "--scaling", scaling, "--plan-only"])     # an odd total: shards differ by one
    plans[scaling] = json.loads(buf.getvalue())
# every rank learns every rank's plan: sizes add up, offsets are the running sum, nothing overlaps
for scaling in ("strong", "weak"):
    mine = torch.tensor([plans[scaling]["n_local"], plans[scaling]["env_offset"], plans[scaling]["n_global"]], dtype=torch.int64)
    allp = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    sizes = [int(p[0]) for p in allp]; offs = [int(p[1]) for p in allp]; glob = {int(p[2]) for p in allp}
    assert len(glob) == 1 and sum(sizes) == glob.pop(), (scaling, sizes)
    assert offs == [sum(sizes[:r]) for r in range(world)], (scaling, offs)
    if scaling == "strong":
        assert sum(sizes) == 1048577 and max(sizes) - min(sizes) <= 1
    else:
        assert sizes == [1048577] * world
# the timing reduction of bench.py: max over ranks
t = torch.tensor([0.5 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == world - 0.5
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_bench_rank_arithmetic_under_two_gloo_ranks(tmp_path):
    """bench.py's own shard plan (--scaling strong: 1 Mi envs in all; weak: 1 Mi per rank) evaluated by two ranks
    launched the way the driver launches them (RANK / WORLD_SIZE / MASTER_* in the environment), no GPU."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, T1D_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_bench_plan_and_refill_count():
    import bench
    assert bench.plan_shard(1 << 20, "strong", 3, 8) == (131072, 3 * 131072, 1 << 20)
    assert bench.plan_shard(1 << 20, "weak", 3, 8) == (1 << 20, 3 << 20, 8 << 20)
    assert [bench.plan_shard(10, "strong", r, 4)[:2] for r in range(4)] == [(3, 0), (3, 3), (2, 6), (2, 8)]
    # 1-minute sensor: blocks of 150 samples, reset consumed samples 0 and 1 -> the step taking minute 149 opens block 1
    assert bench.refills_in(0, 148, 1, 150) == 0 and bench.refills_in(0, 149, 1, 150) == 1
    assert bench.refills_in(800, 1000, 1, 150) == 7

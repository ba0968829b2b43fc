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


def test_bench_shards_env_offsets_like_global_ids():
    """Philox streams are indexed by GLOBAL env id: rank r of W with env_offset = r * n sees the streams
    [r n, (r+1) n) -- the host-side arithmetic bench.py relies on."""
    from simglucose_amd.distributed import shard_range
    n, w = 1 << 20, 8
    offs = [shard_range(n * w, r, w)[0] for r in range(w)]
    assert offs == [r * n for r in range(w)]

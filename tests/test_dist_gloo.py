"""CPU-only, world size 2 over gloo: the multi-GPU path's sharding and its single exchange step
(all_gather of [B,8] f32 velocity records + MAX-over-ranks timing), as bench.py uses them over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from __graft_entry__ import load_package
    load_package()
    from of_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 13
    lo, hi = sharding.shard_range(total, rank, world)
    # every rank "solves" its own pairs: record k of the global batch is a pure function of k
    B = 7
    local = torch.tensor([[1000 * rank + k + 0.125 * j for j in range(8)] for k in range(B)], dtype=torch.float32)
    allrec = sharding.gather_records(dist, local)
    assert allrec.shape == (world * B, 8)
    for r in range(world):
        assert torch.equal(allrec[r * B:(r + 1) * B, 0], torch.arange(B, dtype=torch.float32) + 1000 * r)
    t = sharding.max_over_ranks(dist, 0.5 + rank)
    assert t == 0.5 + (world - 1)
    owned = torch.zeros(total, dtype=torch.int32); owned[lo:hi] = 1
    dist.all_reduce(owned)
    assert bool((owned == 1).all()), owned          # every unit owned exactly once
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_world2_gloo_gather_and_shard(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_shard_range_properties():
    from __graft_entry__ import load_package
    load_package()
    from of_amd.sharding import shard_range
    for total in (0, 1, 7, 8, 1024, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1

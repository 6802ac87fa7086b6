"""CPU-only, world size 2: the multi-GPU path's sharding, its bootstrap and its exchange semantics.

The product's transport is RCCL through libofk.so (ofk_comm_*, no torch).  Without GPUs it cannot run, so two things are
checked here instead:
  * test_world2_gloo_gather_and_shard — gloo (torch.distributed) stands in for the transport: shard_range ownership, the
    rank-major all-gather layout of the [B, 8] records and MAX-over-ranks timing, i.e. the semantics ofk_comm_gather_records /
    ofk_comm_allreduce_f64 implement on the GPUs;
  * test_world2_bootstrap_with_fake_collective — sharding.exchange_unique_id + env_ranks + combine_moments with two plain
    processes and a file-based fake all-reduce: the torch-free bootstrap path bench.py and simulation.sweep_flow_errors take;
    a stale unique-id file from "a crashed earlier run" lies at the rendezvous path and must be ignored; two more exchanges of
    the same processes on the default path must not meet each other's files (generation counter).
  * test_world2_comm_sequence_with_fake_context — sharding.Comm's full call sequence (two communicators, two slots, gather per
    slice, fetch) against a fake context whose "all-gather" goes through files and whose reassembly is libofk.so's own host
    function ofk_comm_reorder_records: the [world][B][8] rank-major result for world = 2, S = 2, uneven slices.
  * test_reorder_records_matches_numpy_for_uneven_cuts — that host function against numpy for worlds of 1..8, 1..4 slices.
"""
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
    assert sharding.env_ranks() == (rank, world, rank)
    total = 13
    lo, hi = sharding.shard_range(total, rank, world)
    # every rank "solves" its own pairs: record k of the global batch is a pure function of k
    B = 7
    local = torch.tensor([[1000 * rank + k + 0.125 * j for j in range(8)] for k in range(B)], dtype=torch.float32)
    allrec = torch.empty((world * B, 8), dtype=torch.float32)
    dist.all_gather_into_tensor(allrec, local)                 # what ncclAllGather does with the ranks' send buffers
    assert allrec.shape == (world * B, 8)
    for r in range(world):
        assert torch.equal(allrec[r * B:(r + 1) * B, 0], torch.arange(B, dtype=torch.float32) + 1000 * r)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                     # Comm.max
    assert float(t) == 0.5 + (world - 1)
    owned = torch.zeros(total, dtype=torch.int32); owned[lo:hi] = 1
    dist.all_reduce(owned)
    assert bool((owned == 1).all()), owned          # every unit owned exactly once
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT

BOOT = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, %r)
    from __graft_entry__ import load_package
    load_package()
    from of_amd import sharding
    assert "torch" not in sys.modules
    rank, world, local = sharding.env_ranks()
    path = os.environ["OFK_TEST_RDV"]
    made = []
    def make_id():
        made.append(1)
        return bytes((7 * i + 3) %% 256 for i in range(128))
    uid = sharding.exchange_unique_id(make_id, rank, world, path=path, timeout=60)
    assert uid == bytes((7 * i + 3) %% 256 for i in range(128)) and len(made) == (1 if rank == 0 else 0)
    # two more exchanges on the default path (MASTER_ADDR / MASTER_PORT key): consecutive generations, distinct files, distinct ids
    g0 = sharding.rendezvous_path()
    u1 = sharding.exchange_unique_id(lambda: bytes([1]) * 128, rank, world, timeout=60)
    g1 = sharding.rendezvous_path()
    u2 = sharding.exchange_unique_id(lambda: bytes([2]) * 128, rank, world, timeout=60)
    assert g0 != g1 and g0.endswith("_g0") and g1.endswith("_g1") and os.environ["MASTER_PORT"] in g0
    assert u1 == bytes([1]) * 128 and u2 == bytes([2]) * 128

    # a fake all-reduce(sum) over files: each rank publishes its vector, waits for the others', adds them up in rank order
    def allreduce(v, tag=[0]):
        tag[0] += 1
        np.save(f"{path}.red{tag[0]}.{rank}.tmp.npy", v); os.replace(f"{path}.red{tag[0]}.{rank}.tmp.npy", f"{path}.red{tag[0]}.{rank}.npy")
        parts = []
        for r in range(world):
            f = f"{path}.red{tag[0]}.{r}.npy"
            t0 = time.time()
            while not os.path.exists(f):
                assert time.time() - t0 < 60
                time.sleep(0.002)
            parts.append(np.load(f))
        return np.sum(parts, axis=0)

    # Monte-Carlo statistics over sharded trials == statistics over all trials (simulation.py:199-200: np.mean / np.std)
    rng = np.random.default_rng(11)
    v_all = 1.0 + 0.02 * rng.standard_normal((101, 3))
    lo, hi = sharding.shard_range(len(v_all), rank, world)
    mine = v_all[lo:hi]
    mean, std, n = sharding.combine_moments(mine.sum(0), (mine * mine).sum(0), len(mine), allreduce)
    assert n == 101
    np.testing.assert_allclose(mean, v_all.mean(0), rtol=1e-13)
    np.testing.assert_allclose(std, v_all.std(0), rtol=1e-9)
    print("rank", rank, "ok")
""") % ROOT


def _run_world(script_text, tmp_path, extra_env, expect_rc=0):
    script = tmp_path / "worker.py"
    script.write_text(script_text)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", **extra_env)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == expect_rc and (expect_rc != 0 or f"rank {r} ok" in o), o
    return outs


def test_world2_gloo_gather_and_shard(tmp_path):
    _run_world(WORKER, tmp_path, {})


def test_world2_bootstrap_with_fake_collective(tmp_path):
    import time
    stale = tmp_path / "rdv"
    stale.write_bytes(bytes(128))                               # a crashed earlier run left a well-formed id behind ...
    old = time.time() - 3600
    os.utime(stale, (old, old))                                 # ... an hour ago
    (tmp_path / "rdv.ack1").write_bytes(b"1")
    _run_world(BOOT, tmp_path, {"OFK_TEST_RDV": str(stale)})
    assert not stale.exists() and not (tmp_path / "rdv.ack1").exists()      # rank 0 cleaned the rendezvous files up


def test_world2_bootstrap_ignores_a_fresh_file_of_a_dead_publisher(tmp_path, pkg):
    """The long-lived-parent case: the launcher (a shell, a slurm step daemon, pytest) is OLDER than the file a crashed earlier run
    left under the same MASTER_ADDR:PORT, so the modification-time rule accepts it.  The launch token does not: the file names a
    rank-0 process that no longer exists, readers skip it and rank 0 of this launch replaces it."""
    from of_amd import sharding
    dead = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(30)"])
    tok = sharding._token(dead.pid)
    assert sharding._token_alive(tok) is True
    dead.kill(); dead.wait()
    assert sharding._token_alive(tok) is False and sharding._token_alive(b"x" * 24) is False
    stale = tmp_path / "rdv"
    stale.write_bytes(bytes([7]) * 128 + tok)                   # well-formed, brand new, published by nobody alive
    _run_world(BOOT, tmp_path, {"OFK_TEST_RDV": str(stale)})
    assert not stale.exists()


COMM = open(os.path.join(ROOT, "tests", "comm_sequence_worker.py")).read()


def test_world2_comm_sequence_with_fake_context(tmp_path):
    _run_world(COMM, tmp_path, {"OFK_TEST_RDV": str(tmp_path / "rdv"), "OFK_TEST_ROOT": ROOT})


def test_world2_communicator_count_is_agreed_before_any_is_created(tmp_path):
    """Rank 1 can afford one communicator, rank 0 asks for two: the count is settled by an all-reduce over communicator 0 BEFORE a
    second ncclCommInitRank is attempted anywhere, so both ranks end on one communicator, nobody waits inside a collective call the
    other never enters, and the gathers (one slice) deliver every rank's records."""
    _run_world(COMM, tmp_path, {"OFK_TEST_RDV": str(tmp_path / "rdv"), "OFK_TEST_ROOT": ROOT, "OFK_TEST_SCENARIO": "afford1"})


def test_world2_failed_communicator_behind_the_agreement_is_fatal_on_both_ranks(tmp_path):
    """A rank-local failure of the second ncclCommInitRank AFTER the ranks agreed on two: the failing rank raises and exits non-zero,
    its peer is released from the collective call by the time-out with an error - no rank carries on with a different communicator
    count, none hangs."""
    outs = _run_world(COMM, tmp_path, {"OFK_TEST_RDV": str(tmp_path / "rdv"), "OFK_TEST_ROOT": ROOT, "OFK_TEST_SCENARIO": "addfail"}, expect_rc=7)
    assert "simulated local failure" in outs[1] and "peers never entered" in outs[0]


def test_reorder_records_matches_numpy_for_uneven_cuts():
    import numpy as np
    from __graft_entry__ import load_package
    load_package()
    from of_amd import ofk
    rng = np.random.default_rng(3)
    for world, batch, slices in ((3, 7, 2), (3, 10, 3), (2, 256, 2), (8, 5, 4), (1, 9, 2), (4, 6, 1)):
        per_rank = rng.standard_normal((world, batch, 8)).astype(np.float32)
        recv = []
        for s in range(slices):
            b0, b1 = batch * s // slices, batch * (s + 1) // slices
            for r in range(world):
                recv.append(per_rank[r, b0:b1].ravel())
        out = ofk.comm_reorder_records(np.concatenate(recv), world, batch, slices)
        assert np.array_equal(out, per_rank), (world, batch, slices)


def test_rendezvous_path_is_per_launch():
    from __graft_entry__ import load_package
    load_package()
    from of_amd.sharding import rendezvous_path, env_ranks
    a = rendezvous_path({"MASTER_PORT": "29500", "MASTER_ADDR": "127.0.0.1"}, "/tmp"); b = rendezvous_path({"MASTER_PORT": "29501"}, "/tmp")
    assert a != b and "127-0-0-1_29500" in a and str(os.getppid()) not in os.path.basename(a)      # ranks need not share a parent
    assert f"ppid{os.getppid()}" in rendezvous_path({}, "/tmp")                                       # last resort: no rendezvous address exported
    assert rendezvous_path({"MASTER_PORT": "1"}, "/tmp", generation=0) != rendezvous_path({"MASTER_PORT": "1"}, "/tmp", generation=1)
    assert rendezvous_path({"MASTER_PORT": "1", "TORCHELASTIC_RUN_ID": "a"}, "/tmp") != rendezvous_path({"MASTER_PORT": "1", "TORCHELASTIC_RUN_ID": "b"}, "/tmp")
    assert env_ranks({}) == (0, 1, 0) and env_ranks({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"}) == (3, 8, 3)


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

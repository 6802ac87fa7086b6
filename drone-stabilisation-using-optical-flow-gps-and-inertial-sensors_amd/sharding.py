"""sharding.py — how frame pairs and Monte-Carlo trials are spread over the GPUs of a node, and the one exchange step.

Frame pairs (and Monte-Carlo trials) are independent, so the data path needs no collective: rank r of W owns the contiguous
slice [r*B/W, (r+1)*B/W) of a global batch (or simply its own B pairs under weak scaling).  The only exchange is an all-gather
of the per-pair velocity records ([B_local, 8] float32: vx, vy, vz, residual, n_used, s_min, rank, corners) — and, for the
Monte-Carlo sweep, an all-reduce of (sum v, sum v^2, count) per sigma step.

Transport: RCCL over xGMI through libofk.so (ofk_comm_*, which dlopens librccl.so) — numpy + ctypes only, NO PyTorch in the
process (BASELINE.json north_star): the collectives are queued on the library's own HIP streams, so a gather is
stream-ordered behind the step that produced its records and nothing waits on the host.  One process per GPU, launched by any
launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run does; it is only the process launcher here).

Bootstrap: rank 0 draws the 128-byte RCCL unique id and publishes it through a file the launcher's ranks share (all ranks of a
run sit on one node: the bench contract); the file name carries the launcher's pid and MASTER_PORT so concurrent runs do not
meet.  `exchange_unique_id` is transport-agnostic (it moves 128 bytes) and is tested on the CPU with two processes.
"""
import os
import tempfile
import time

import numpy as np


def shard_range(total, rank, world):
    """Contiguous slice of `total` units owned by `rank`; remainders go to the low ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_ranks(environ=None):
    """(rank, world, local_rank) from the launcher's environment; (0, 1, 0) for a plain single-process run."""
    e = os.environ if environ is None else environ
    if "RANK" not in e or "WORLD_SIZE" not in e:
        return 0, 1, 0
    return int(e["RANK"]), int(e["WORLD_SIZE"]), int(e.get("LOCAL_RANK", e["RANK"]))


def rendezvous_path(environ=None, directory=None):
    """File through which rank 0 hands the unique id to the other ranks of THIS launch: keyed by the launcher (parent) pid, the
    rendezvous port and the elastic run id, under /dev/shm when present (memory-backed, node-local)."""
    e = os.environ if environ is None else environ
    d = directory or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir())
    key = f"{e.get('MASTER_PORT', '0')}_{e.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
    return os.path.join(d, f"ofk_rccl_uid_{key}")


def exchange_unique_id(make_id, rank, world, path=None, timeout=120.0, nbytes=128):
    """Rank 0 calls make_id() -> bytes and publishes them (write + atomic rename); the other ranks poll the file.  Returns the
    id on every rank.  Rank 0 removes the file once every rank has acknowledged (one small file per rank)."""
    path = path or rendezvous_path()
    if world == 1:
        return bytes(make_id())
    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != nbytes:
            raise ValueError(f"unique id has {len(uid)} bytes, expected {nbytes}")
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        deadline = time.monotonic() + timeout
        pending = set(range(1, world))
        while pending and time.monotonic() < deadline:
            pending = {r for r in pending if not os.path.exists(f"{path}.ack{r}")}
            if pending:
                time.sleep(0.005)
        for r in range(1, world):
            try:
                os.remove(f"{path}.ack{r}")
            except OSError:
                pass
        try:
            os.remove(path)
        except OSError:
            pass
        if pending:
            raise TimeoutError(f"ranks {sorted(pending)} never picked up the unique id at {path}")
        return uid
    deadline = time.monotonic() + timeout
    while time.monotonic() < deadline:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == nbytes:
                with open(f"{path}.ack{rank}", "wb") as f:
                    f.write(b"1")
                return uid
        except OSError:
            pass
        time.sleep(0.005)
    raise TimeoutError(f"rank {rank}: no unique id at {path} after {timeout} s")


class Comm:
    """The exchange step of one rank.  `ctx` is the rank's ofk.Context (its device = LOCAL_RANK).

        rank, world, local = env_ranks()
        comm = Comm(pipe.ctx, rank, world, n_comms=2)   # RCCL bootstrap (file rendezvous) + ncclCommInitRank per slice
        pipe.run_async(); comm.gather_async(B, slot=k % 2)     # queued behind the step, returns at once
        records = comm.fetch(B, slot=k % 2)         # [world, B, 8] float32, when somebody needs them on the host
        comm.barrier(); t = comm.max(seconds)
    """

    def __init__(self, ctx, rank, world, path=None, n_comms=1):
        try:
            from . import ofk
        except ImportError:
            import ofk
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        # one communicator per free-running slice (ofk_set_streams): each slice then gathers its own records on its own stream
        uid = exchange_unique_id(lambda: ofk.comm_unique_id(n_comms), self.rank, self.world, path=path, nbytes=128 * int(n_comms))
        ctx.comm_init(uid, self.rank, self.world)

    def gather_async(self, batch, slot=0):
        self.ctx.comm_gather_records(batch, slot)

    def fetch(self, batch, slot=0):
        return self.ctx.comm_fetch_records(batch, slot)

    def allreduce(self, values, op="sum"):
        return self.ctx.comm_allreduce(values, op)

    def barrier(self):
        self.ctx.comm_allreduce([0.0], "sum")

    def max(self, value):
        return float(self.ctx.comm_allreduce([float(value)], "max")[0])

    def close(self):
        self.ctx.comm_destroy()


def combine_moments(local_sum, local_sumsq, local_count, allreduce=None):
    """Mean and (population) standard deviation per component from per-rank (sum v, sum v^2, count), all-reduced when an
    `allreduce(values) -> summed values` is given: the statistic the Monte-Carlo sweeps keep per sigma step
    (simulation.py:183-202: np.mean / np.std over the trials)."""
    s = np.concatenate([np.ravel(local_sum), np.ravel(local_sumsq), [float(local_count)]]).astype(np.float64)
    if allreduce is not None:
        s = np.asarray(allreduce(s), np.float64)
    k = (s.size - 1) // 2
    n = s[-1]
    mean = s[:k] / n
    var = np.maximum(s[k:2 * k] / n - mean * mean, 0.0)
    return mean, np.sqrt(var), int(round(n))

"""sharding.py — how frame pairs and Monte-Carlo trials are spread over the GPUs of a node, and the one exchange step.

Frame pairs (and Monte-Carlo trials) are independent, so the data path needs no collective: rank r of W owns the contiguous
slice [r*B/W, (r+1)*B/W) of a global batch (or simply its own B pairs under weak scaling).  The only exchange is an all-gather
of the per-pair velocity records ([B_local, 8] float32: vx, vy, vz, residual, n_used, s_min, rank, corners) — and, for the
Monte-Carlo sweep, an all-reduce of (sum v, sum v^2, count) per sigma step.

Transport: RCCL over xGMI through libofk.so (ofk_comm_*, which dlopens librccl.so) — numpy + ctypes only, NO PyTorch in the
process (BASELINE.json north_star): the collectives are queued on the library's own HIP streams, so a gather is
stream-ordered behind the step that produced its records and nothing waits on the host.  One process per GPU, launched by any
launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run does; it is only the process launcher here).

Bootstrap: rank 0 draws the 128-byte RCCL unique ids and publishes them through a file the launcher's ranks share (all ranks of a
run sit on one node: the bench contract); the file name carries MASTER_ADDR, MASTER_PORT, the elastic run id and the number of the
exchange inside the launch, so concurrent runs and consecutive exchanges do not meet, and readers ignore files older than the launch.  `exchange_unique_id` is transport-agnostic (it moves 128 bytes) and is tested on the CPU with two processes.
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


_exchange_generation = 0          # exchanges this process has taken part in: every rank of a launch makes them in the same order


def _proc_start_time(pid):
    """Start time of process `pid` in seconds since the epoch (Linux /proc), None if it cannot be read."""
    try:
        with open(f"/proc/{pid}/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])            # field 22: starttime in clock ticks since boot
        with open("/proc/stat") as f:
            btime = next(int(line.split()[1]) for line in f if line.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return None


def launch_epoch():
    """A time no file of THIS launch can be older than: the start of the launcher (the ranks' common parent under
    torch.distributed.run / mpirun) when it can be read, else of this process.  A unique-id file left behind by a crashed earlier
    run whose MASTER_PORT recurs is older than that and is ignored by the readers."""
    t_self = _proc_start_time(os.getpid())
    t_parent = _proc_start_time(os.getppid()) if os.getppid() > 1 else None
    ts = [t for t in (t_self, t_parent) if t is not None]
    return (min(ts) - 1.0) if ts else 0.0


def rendezvous_path(environ=None, directory=None, generation=None):
    """File through which rank 0 hands the unique ids to the other ranks of THIS launch, under /dev/shm when present
    (memory-backed, node-local).  Keyed by what the launcher gives every rank alike — MASTER_ADDR : MASTER_PORT and the elastic
    run id — so the ranks need not share a parent process; the launcher's pid is only the last resort when no rendezvous address
    is exported.  `generation` numbers the exchanges of a launch (a process that builds a second Comm must not meet the first
    one's file)."""
    e = os.environ if environ is None else environ
    d = directory or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir())
    if "MASTER_PORT" in e:
        addr = "".join(ch if ch.isalnum() else "-" for ch in e.get("MASTER_ADDR", "local"))
        key = f"{addr}_{e['MASTER_PORT']}_{e.get('TORCHELASTIC_RUN_ID', 'none')}"
    else:
        key = f"ppid{os.getppid()}"
    g = _exchange_generation if generation is None else generation
    return os.path.join(d, f"ofk_rccl_uid_{key}_g{g}")


_TOKEN = b"OFKTOKEN"


def _token(pid=None):
    """Launch token rank 0 appends to the published ids: its pid and that process's start time.  A reader accepts a file only while
    that very process is alive - whatever a crashed earlier run with the same MASTER_ADDR:PORT left behind fails the check even when the
    launcher (an interactive shell, a slurm step daemon, pytest) is older than the leftover and the modification-time rule cannot tell."""
    pid = os.getpid() if pid is None else pid
    t = _proc_start_time(pid)
    return _TOKEN + int(pid).to_bytes(8, "little") + int(round((t or 0.0) * 100)).to_bytes(8, "little")


def _token_alive(tok):
    """True / False for a token whose writer is / is not the live process it names; None when /proc cannot tell."""
    if len(tok) != 24 or tok[:8] != _TOKEN:
        return False
    pid, t100 = int.from_bytes(tok[8:16], "little"), int.from_bytes(tok[16:24], "little")
    t = _proc_start_time(pid)
    if t is None:
        # not visible.  Sibling ranks of one launcher share its PID namespace: if this process can see its own launcher, a live rank 0
        # would be visible too, so the publisher is gone; otherwise (no /proc, or every rank in a namespace of its own) /proc cannot tell
        # and the modification-time rule decides alone
        ppid = os.getppid()
        return False if (ppid > 1 and _proc_start_time(ppid) is not None) else None
    return abs(int(round(t * 100)) - t100) <= 1


def exchange_unique_id(make_id, rank, world, path=None, timeout=120.0, nbytes=128, not_before=None):
    """Rank 0 calls make_id() -> bytes and publishes them followed by its launch token (stale file removed, write, atomic rename); the
    other ranks poll the file, ignore one that is older than the launch (`not_before`, default launch_epoch()) or whose token does not
    name a live process, and return the ids.  nbytes = None accepts any positive multiple of 128 bytes (rank 0 decides how many ids).
    Rank 0 removes the file once every rank has acknowledged (one small file per rank).  Every call of a process uses the next
    generation's file name unless `path` is given."""
    global _exchange_generation
    if path is None:
        path = rendezvous_path()
        _exchange_generation += 1
    if world == 1:
        return bytes(make_id())
    if rank == 0:
        for stale in [path] + [f"{path}.ack{r}" for r in range(1, world)]:        # leftovers of a crashed run with the same key
            try:
                os.remove(stale)
            except OSError:
                pass
        uid = bytes(make_id())
        if (nbytes is not None and len(uid) != nbytes) or not uid or len(uid) % 128:
            raise ValueError(f"unique id has {len(uid)} bytes, expected {nbytes or 'a multiple of 128'}")
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid + _token())
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
            raise TimeoutError(f"rank 0: ranks {sorted(pending)} of {world} never picked up the unique id at {path} within {timeout} s "
                               f"(do all ranks see MASTER_ADDR/MASTER_PORT = {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')} and this directory?)")
        return uid
    if not_before is None:
        not_before = launch_epoch()
    deadline = time.monotonic() + timeout
    stale_seen = False
    while time.monotonic() < deadline:
        try:
            if os.stat(path).st_mtime < not_before:
                stale_seen = True                               # a crashed earlier run's file: rank 0 of this launch replaces it
            else:
                with open(path, "rb") as f:
                    blob = f.read()
                uid, tok = blob[:-24], blob[-24:]
                if uid and len(uid) % 128 == 0 and (nbytes is None or len(uid) == nbytes) and _token_alive(tok) is not False:
                    with open(f"{path}.ack{rank}", "wb") as f:
                        f.write(b"1")
                    return uid
                stale_seen = True                               # well-formed or not: not published by a live rank 0 of this launch
        except OSError:
            pass
        time.sleep(0.005)
    raise TimeoutError(f"rank {rank} of {world}: no unique id at {path} after {timeout} s" +
                       (" (only a stale file - older than this launch, or written by a process that no longer exists - was there)" if stale_seen else "") +
                       f"; rank 0 publishes it there - MASTER_ADDR/MASTER_PORT here: {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}")


class Comm:
    """The exchange step of one rank.  `ctx` is the rank's ofk.Context (its device = LOCAL_RANK).

        rank, world, local = env_ranks()
        comm = Comm(pipe.ctx, rank, world, n_comms=2)   # RCCL bootstrap (file rendezvous) + ncclCommInitRank per slice
        pipe.run_async(); comm.gather_async(B, slot=k % 2)     # queued behind the step, returns at once
        records = comm.fetch(B, slot=k % 2)         # [world, B, 8] float32, when somebody needs them on the host
        comm.barrier(); t = comm.max(seconds)
    """

    def __init__(self, ctx, rank, world, path=None, n_comms=1, make_id=None):
        try:
            from . import ofk
        except ImportError:
            import ofk
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        # one communicator per free-running slice (ofk_set_streams): each slice then gathers its own records on its own stream
        # (make_id: a stand-in for ncclGetUniqueId where no RCCL exists - the CPU tests drive this class against a fake context)
        uid = exchange_unique_id(make_id or (lambda: ofk.comm_unique_id(n_comms)), self.rank, self.world, path=path, nbytes=None)
        have = len(uid) // 128                                   # ids rank 0 drew: the same number on every rank
        ctx.comm_init(uid[:128], self.rank, self.world)          # communicator 0: every rank needs it, a failure here raises
        # ncclCommInitRank is collective, so the number of per-slice communicators is decided BEFORE any of them is created: the
        # smallest count any rank wants (its n_comms, at most the ids there are), one all-reduce over communicator 0.  Every rank
        # then creates exactly that many; an error behind the agreement raises (fatal for the job: the peers are inside the call).
        agreed = 1
        if have > 1:
            agreed = max(1, int(ctx.comm_allreduce([float(min(int(n_comms), have))], "min")[0]))
        for k in range(1, agreed):
            ctx.comm_add(uid[128 * k:128 * (k + 1)])
        self.n_comms = ctx.comm_count()
        if self.n_comms != agreed:
            raise RuntimeError(f"rank {self.rank}: holds {self.n_comms} communicators, the ranks agreed on {agreed}")

    def pending(self, slot=0):
        """Bitmask of the slices whose gather of `slot` is still travelling (non-blocking; for watchdogs)."""
        return self.ctx.comm_pending(slot)

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

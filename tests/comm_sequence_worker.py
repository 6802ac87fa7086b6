"""Worker of tests/test_dist_gloo.py::test_world2_comm_sequence_with_fake_context (one process per rank, no torch, no GPU)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["OFK_TEST_ROOT"])
from __graft_entry__ import load_package
load_package()
from of_amd import sharding, ofk
assert "torch" not in sys.modules
rank, world, local = sharding.env_ranks()
root = os.environ["OFK_TEST_RDV"]
B, S = 7, 2                                                  # an uneven cut: slices of 3 and 4 pairs
# scenario "sym": both ranks want S communicators; "afford1": rank 1 wants one only -> both must end on ONE communicator, no rank
# creates a second; "addfail": the ranks agree on two, rank 1's second ncclCommInitRank fails locally -> that is fatal on rank 1, and
# rank 0 - inside the collective call - is released by its own time-out with an error (nobody hangs, nobody carries on alone)
SCEN = os.environ.get("OFK_TEST_SCENARIO", "sym")
WANT = 1 if (SCEN == "afford1" and rank == 1) else S


def records(r, k):                                           # step k's [B, 8] f32 records of rank r
    return (1000.0 * r + 10.0 * k + np.arange(B)[:, None] + 0.125 * np.arange(8)[None, :]).astype(np.float32)


class FakeCtx:
    """The calls sharding.Comm makes on ofk.Context: files are the wire, libofk.so's host function does the reassembly."""

    def comm_init(self, uid, rank, world):
        assert len(uid) == 128                               # communicator 0 only: the others follow the agreement
        self.rank, self.world, self.latest, self.nsteps, self.ncomm, self.nred = rank, world, {}, 0, 1, 0

    def comm_add(self, uid):
        assert len(uid) == 128
        if SCEN == "afford1":
            raise AssertionError("a second communicator although rank 1 wanted one")
        if SCEN == "addfail" and self.rank == 1:
            raise ofk.OfkError(ofk.E_HIP, "ncclCommInitRank(communicator 1): simulated local failure")
        # the collective part of ncclCommInitRank: every rank has to arrive
        open(f"{root}.add{self.ncomm}.r{self.rank}", "wb").close()
        t0 = time.time()
        while not all(os.path.exists(f"{root}.add{self.ncomm}.r{r}") for r in range(self.world)):
            if time.time() - t0 > 5:
                raise TimeoutError(f"rank {self.rank}: peers never entered ncclCommInitRank of communicator {self.ncomm}")
            time.sleep(0.002)
        self.ncomm += 1

    def comm_count(self):
        return self.ncomm

    def comm_destroy(self):
        pass

    def comm_gather_records(self, batch, slot):
        k = self.nsteps
        self.nsteps += 1
        self.latest[slot] = k
        rec = records(self.rank, k)
        S = self.ncomm
        for s in range(S):                                   # every slice "all-gathers" its own pairs on its own communicator
            b0, b1 = batch * s // S, batch * (s + 1) // S
            tmp = f"{root}.g{k}.s{s}.r{self.rank}.tmp.npy"
            np.save(tmp, rec[b0:b1]); os.replace(tmp, f"{root}.g{k}.s{s}.r{self.rank}.npy")

    def comm_fetch_records(self, batch, slot):
        k = self.latest[slot]
        recv = []
        S = self.ncomm
        for s in range(S):                                   # receive buffer: slice after slice, [world][pairs of the slice][8]
            for r in range(self.world):
                f = f"{root}.g{k}.s{s}.r{r}.npy"
                t0 = time.time()
                while not os.path.exists(f):
                    assert time.time() - t0 < 60
                    time.sleep(0.002)
                recv.append(np.load(f).ravel())
        return ofk.comm_reorder_records(np.concatenate(recv), self.world, batch, S)

    def comm_pending(self, slot):
        return 0

    def comm_allreduce(self, values, op="sum"):              # over files, like the gathers: every rank publishes, all combine
        k = self.nred
        self.nred += 1
        v = np.asarray(values, np.float64)
        tmp = f"{root}.red{k}.r{self.rank}.tmp.npy"
        np.save(tmp, v); os.replace(tmp, f"{root}.red{k}.r{self.rank}.npy")
        parts = []
        for r in range(self.world):
            f = f"{root}.red{k}.r{r}.npy"
            t0 = time.time()
            while not os.path.exists(f):
                assert time.time() - t0 < 60
                time.sleep(0.002)
            parts.append(np.load(f))
        return {"sum": np.sum, "max": np.max, "min": np.min}[op](np.stack(parts), axis=0)


ctx = FakeCtx()
try:
    comm = sharding.Comm(ctx, rank, world, path=root, n_comms=WANT, make_id=lambda: bytes(128 * S))
except (ofk.OfkError, TimeoutError) as e:
    assert SCEN == "addfail", e
    print("rank", rank, "fatal:", e)
    sys.exit(7)
assert SCEN != "addfail"
assert comm.n_comms == (1 if SCEN == "afford1" else S) and comm.pending(0) == 0
for k in range(4):                                           # bench.py's loop: step k gathers into slot k % 2
    comm.gather_async(B, k % 2)
for slot, k in ((0, 2), (1, 3)):                             # the latest gather of each slot
    got = comm.fetch(B, slot)
    assert got.shape == (world, B, 8)
    for r in range(world):
        assert np.array_equal(got[r], records(r, k)), (slot, r)
comm.close()
print("rank", rank, "ok")

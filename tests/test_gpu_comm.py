"""GPU: the exchange step through libofk.so's RCCL binding (ofk_comm_*, of_amd/sharding.Comm) with a world of ONE rank - all that
one GPU can host (the driver runs N = 2, 4, 8 at round end): bootstrap, ncclCommInitRank, the stream-ordered gather of a step's
records, the all-reduce used for barriers / timing / Monte-Carlo statistics.  No torch in this path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_world1_gather_is_stream_ordered_and_carries_the_records(pkg, ofk, tmp_path):
    from of_amd import sharding, synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    B = 6
    pairs = [synth.render_pair(240, 320, 40 + b, v=(0.004, -0.003 + 0.001 * b, 0.002), omega=(0.003, -0.002, 0.004)) for b in range(B)]
    prev = np.stack([p["prev"] for p in pairs]); nxt = np.stack([p["next"] for p in pairs])
    sensors = np.concatenate([ofk.make_sensors(1, d=p["d"], normal=p["n"], omega=p["omega"], scaling=p["scaling"], cx=p["cx"], cy=p["cy"]) for p in pairs])
    cfg = PipelineConfig(max_corners=80, quality=0.05, min_distance=8, block_size=7, max_level=2)
    pipe = FlowPipeline(320, 240, B, cfg, streams=2)
    pipe.upload(prev, nxt, sensors)
    comm = sharding.Comm(pipe.ctx, 0, 1, path=str(tmp_path / "rdv"), n_comms=2)
    assert comm.world == 1 and comm.rank == 0 and comm.n_comms == 2   # communicator 0 + one ofk_comm_add behind the agreement
    for k in range(4):                                           # queue steps and gathers back to back, no host wait in between
        pipe.run_async()
        comm.gather_async(B, k % 2)
    g = comm.fetch(B, 1)                                         # the last step's gather
    rec = pipe.ctx.pairs_download(points=False)["records"]
    mine = np.stack([rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 11], rec[:, 7], rec[:, 4], rec[:, 12]], 1).astype(np.float32)
    assert g.shape == (1, B, 8) and np.array_equal(g[0], mine) and np.all(g[0, :, 6] == 3)
    assert np.array_equal(comm.fetch(B, 0)[0], mine)             # the same frames every step: the other slot holds the same records
    np.testing.assert_array_equal(comm.allreduce([1.5, -2.0, 7.0]), [1.5, -2.0, 7.0])
    assert comm.max(3.25) == 3.25
    comm.barrier()
    pipe.ctx.set_streams(1)                                      # one slice: the gather runs behind the context's stream on communicator 0
    pipe.run_async(); comm.gather_async(B, 0)
    assert np.array_equal(comm.fetch(B, 0)[0], mine)
    with pytest.raises(ofk.OfkError):
        pipe.ctx.comm_init(b"\0" * 128, 0, 1)                   # one communicator set per context
    comm.close()
    # the C entry point with several ids does the same internally: communicator 0, agreement over it, then exactly that many
    pipe.ctx.comm_init(ofk.comm_unique_id(3), 0, 1)
    assert pipe.ctx.comm_count() == 3
    pipe.ctx.comm_destroy()
    pipe.close()


def test_monte_carlo_sweep_sharded_statistics_equal_the_single_rank_ones(pkg, ofk, golden, tmp_path):
    from of_amd import sharding, simulation as sim
    pts = golden["g1_points"]
    sig = dict(ang_vel_sig=0.00071, translation_sig=0.005, height_sig=0.01, normal_sig=0.00065)
    args = (pts, np.array([1.0, 1, 1]), np.array([1.0, 1, 1]), 1.0, np.array([0, 0, 1.0]), np.array([0.02, 0, 0.205]), sig)
    ref = sim.sweep_flow_errors(*args, k=3, trials=64, generator=np.random.default_rng(5))
    ctx = ofk.default_context()
    comm = sharding.Comm(ctx, 0, 1, path=str(tmp_path / "rdv2"))
    got = sim.sweep_flow_errors(*args, k=3, trials=64, generator=np.random.default_rng(5), comm=comm)
    comm.close()
    np.testing.assert_allclose(got[:9], ref[:9], rtol=1e-12)     # means
    np.testing.assert_allclose(got[9:], ref[9:], rtol=1e-7)      # standard deviations (moment form)

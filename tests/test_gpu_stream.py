"""GPU: persistent tracks of a video stream (feature lifecycle) — the loop of
velocity_measurment_node:92-177 with its commented-out blocks restored — against the same loop written with the
oracle's functions.  Tracks, counts and velocities must agree frame by frame (tracks bit-exact)."""
import numpy as np
import pytest

from oracle import image_oracle as io, estimation_oracle as eo

pytestmark = pytest.mark.gpu


from stream_oracle import oracle_stream  # noqa: E402  (tests/stream_oracle.py: the node's loop written with the oracle's functions)


@pytest.mark.parametrize("min_feat,motion", [(90, (0.02, 0.008, 0.0)), (70, (0.012, 0.004, 0.0)), (10, (0.004, -0.003, 0.002))])
def test_stream_tracks_match_oracle_loop(pkg, ofk, min_feat, motion):
    from of_amd import synth
    from of_amd.pipeline import FlowStream, PipelineConfig
    h, w, nf = 240, 320, 7
    cfg = PipelineConfig(max_corners=90, quality=0.04, min_distance=9, block_size=7, win=15, max_level=2, max_count=20, eps=0.03)
    seqs = [synth.render_sequence(h, w, 500 + b, nf, v=motion, omega=(0.0, 0.0, 0.004 * b), d=1.0) for b in range(2)]
    frames = np.stack([s[0] for s in seqs])                      # [B, nf, h, w, 3]
    infos = [s[1] for s in seqs]
    sensors = np.concatenate([ofk.make_sensors(1, d=i["d"], normal=i["n"], omega=i["omega"], scaling=i["scaling"], cx=i["cx"], cy=i["cy"]) for i in infos])
    fs = FlowStream(w, h, batch=2, cfg=cfg, min_features=min_feat, mask_radius=12)
    tracks, counts = fs.begin(frames[:, 0])
    refs = [oracle_stream(frames[b], cfg, sensors[b], min_feat, 12) for b in range(2)]
    for b in range(2):
        assert counts[b] == len(refs[b][0]) and np.array_equal(tracks[b, :counts[b]], refs[b][0])
    redetected = False
    for t in range(1, nf):
        rec, tracks, counts = fs.step(frames[:, t], sensors)
        for b in range(2):
            v, tr, n_old, n_tracked = refs[b][1][t - 1]
            assert counts[b] == len(tr), (t, b, counts[b], len(tr))
            assert np.array_equal(tracks[b, :counts[b]].view(np.uint32), tr.astype(np.float32).view(np.uint32)), (t, b)
            assert rec[b, 12] == n_old and rec[b, 13] == n_tracked
            if v is not None:
                np.testing.assert_allclose(rec[b, :3], v, rtol=1e-9, atol=1e-13)
            redetected = redetected or len(tr) > n_tracked
    if min_feat == 90:
        assert redetected                                      # min_features == maxCorners: every lost track is replaced
    fs.close()


def test_stream_velocity_is_physical(pkg, ofk):
    from of_amd import synth
    from of_amd.pipeline import FlowStream, PipelineConfig
    h, w = 480, 640
    frames, info = synth.render_sequence(h, w, 9, 5, v=(0.004, -0.003, 0.002), omega=(0.002, 0.001, -0.003), d=1.0)
    cfg = PipelineConfig(max_corners=150, quality=0.02, min_distance=10, block_size=7)
    sensors = ofk.make_sensors(1, d=1.0, normal=info["n"], omega=info["omega"], scaling=info["scaling"], cx=info["cx"], cy=info["cy"])
    fs = FlowStream(w, h, 1, cfg, min_features=20, mask_radius=30)
    fs.begin(frames[0][None])
    for t in range(1, 5):
        rec, tracks, counts = fs.step(frames[t][None], sensors)
        assert rec[0, 4] == 3 and np.linalg.norm(rec[0, :3] - info["v"]) < 0.05 * np.linalg.norm(info["v"])     # measured: 0.9-2.8 %
    fs.close()


def test_stream_from_jpeg_frames_equals_stream_from_decoded_frames(pkg, ofk):
    """The node's real input (node:112, 215): CompressedImage payloads.  begin_jpeg / step_jpeg decode on the device into the
    stream's frame buffer; tracks, counts and records must equal those of the same loop fed with the frames the JPEG oracle
    decodes (two cameras in one context, restart markers on one of them)."""
    Image = pytest.importorskip("PIL.Image")
    import io as _io
    from oracle import jpeg_oracle as jo
    from of_amd import synth
    from of_amd.pipeline import FlowStream, PipelineConfig
    h, w, T = 480, 640, 5
    seqs = [synth.render_sequence(h, w, 21 + k, T, v=(0.004, -0.003 + 0.001 * k, 0.002), omega=(0.002, 0.001, -0.003), d=1.0) for k in range(2)]
    info = seqs[0][1]

    def enc(img, k):
        buf = _io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=90, **({"restart_marker_rows": 2} if k else {}))
        return buf.getvalue()

    streams = [[enc(seqs[k][0][t], k) for k in range(2)] for t in range(T)]
    decoded = [np.stack([jo.decode(s) for s in streams[t]]) for t in range(T)]
    cfg = PipelineConfig(max_corners=120, quality=0.02, min_distance=10, block_size=7)
    sensors = ofk.make_sensors(2, d=1.0, normal=info["n"], omega=info["omega"], scaling=info["scaling"], cx=info["cx"], cy=info["cy"])
    a = FlowStream(w, h, 2, cfg, min_features=100, mask_radius=20)
    b = FlowStream(w, h, 2, cfg, min_features=100, mask_radius=20)
    ta, ca = a.begin_jpeg(streams[0]); tb, cb = b.begin(decoded[0])
    assert np.array_equal(ca, cb) and np.array_equal(ta, tb) and int(ca.min()) > 30
    for t in range(1, T):
        ra, ta, ca = a.step_jpeg(streams[t], sensors)
        rb, tb, cb = b.step(decoded[t], sensors)
        assert np.array_equal(ca, cb) and np.array_equal(ta, tb) and np.array_equal(ra, rb), t
    with pytest.raises(ofk.OfkError):
        a.step_jpeg([enc(seqs[0][0][0][:240], 0)] * 2, sensors)   # another frame size
    a.close(); b.close()

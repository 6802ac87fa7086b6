"""GPU parity on the one real camera frame the reference holds (tests/golden/reference_frame.npz = flight_experiments/pic2.txt.npy,
240 x 320 BGR).  Every other parity image is band-limited noise on which 99.99 % of the pixels clear a 1 % quality level; this frame
has large flat areas (65 % of its pixels above 1 %, 39 pixels above the reference's 0.7) and drives the threshold, empty-strip and
few-candidate branches of the response kernels and the selection with the reference's own parameter sets."""
import os

import numpy as np
import pytest

from oracle import image_oracle as io, estimation_oracle as eo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def frame():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_frame.npz"))["frame_bgr"]


# (maxCorners, qualityLevel, minDistance, blockSize): node:97-102, evaluate_exp.py:38-43, of_module.py:15-18, BASELINE configs[1]
PARAM_SETS = [(100, 0.7, 10, 12), (20, 0.7, 10, 7), (50, 0.3, 20, 32), (500, 0.01, 10, 7)]


@pytest.mark.parametrize("mc,q,md,bs", PARAM_SETS)
def test_good_features_on_the_real_frame(gpu_ctx, frame, mc, q, md, bs):
    """goodFeaturesToTrack with the reference's parameter sets: gray bit-exact, corner list identical (block 12 = one column per lane,
    block 7 = the pair kernel, block 32 = the LDS-tile kernel)."""
    g = gpu_ctx.gray_bgr8(frame[None])[0]
    assert np.array_equal(g, io.gray_bgr8(frame))
    ref = io.good_features(g, mc, q, md, bs)
    got = gpu_ctx.good_features(g, mc, q, md, bs)
    assert np.array_equal(got, ref), (len(got), len(ref))
    assert 5 <= len(ref) <= mc                                   # 11 / 8 / 30 / 338 corners: far below the budgets at the reference's quality levels


@pytest.mark.parametrize("tile", [(1, 1), (3, 3), (4, 6)])
def test_good_features_on_tiled_real_frames(gpu_ctx, frame, tile):
    """The same on the frame tiled to 720 x 960 and 960 x 1920 (mirrored tiles, so that the seams are not corners): interior strips and
    chunks of the pair kernel see flat regions whose every response is below the threshold - strips and rows without a single key."""
    ty, tx = tile
    rows = [np.concatenate([frame[:, ::(1 if (i + j) % 2 == 0 else -1)][::(1 if i % 2 == 0 else -1)] for j in range(tx)], 1) for i in range(ty)]
    big = np.ascontiguousarray(np.concatenate(rows, 0))
    g = io.gray_bgr8(big)
    for mc, q, md, bs in ((100, 0.7, 10, 12), (20, 0.7, 10, 7), (500, 0.05, 10, 7), (300, 0.3, 10, 5), (200, 0.5, 8, 3)):
        ref = io.good_features(g, mc, q, md, bs)
        got = gpu_ctx.good_features(g, mc, q, md, bs)
        assert np.array_equal(got, ref), (tile, mc, q, md, bs, len(got), len(ref))


def test_masked_redetection_on_the_real_frame(gpu_ctx, frame):
    """node:157-173: when tracks are lost, corners are re-detected under a mask that blanks a disc around every surviving track.
    The mask removes the strongest responses, so the image maximum under the mask - and with it the threshold - changes."""
    g = io.gray_bgr8(frame)
    first = io.good_features(g, 100, 0.7, 10, 12).reshape(-1, 2)
    assert len(first) >= 8
    keep = first[::2]                                            # half of the tracks survived
    yy, xx = np.mgrid[0:g.shape[0], 0:g.shape[1]]
    mask = np.full(g.shape, 255, np.uint8)
    for x, y in keep:
        mask[(xx - x) ** 2 + (yy - y) ** 2 <= 10 ** 2] = 0       # cv2.circle(mask, (x, y), 10, 0, -1)
    for mc, q, md, bs in ((100 - len(keep), 0.7, 10, 12), (20, 0.7, 10, 7), (50, 0.3, 20, 32)):
        ref = io.good_features(g, mc, q, md, bs, mask)
        got = gpu_ctx.good_features(g, mc, q, md, bs, mask)
        assert np.array_equal(got, ref), (mc, q, md, bs, len(got), len(ref))
        d2 = ((ref.reshape(-1, 1, 2) - keep.reshape(1, -1, 2)) ** 2).sum(-1)
        assert len(ref) >= 1 and d2.min() > 10 ** 2               # nothing inside a blanked disc


@pytest.mark.parametrize("preset", ["node", "evaluate", "baseline"])
def test_lk_and_velocity_on_the_warped_real_frame(pkg, ofk, frame, preset):
    """The frame tiled 3 x 3 (720 x 960, mirrored tiles) and seen again through synth's plane-induced homography: the whole resident
    pipeline against the oracle chain - corners identical, LK positions / status / error bit-exact, velocity 1e-9."""
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    rows = [np.concatenate([frame[:, ::(1 if (i + j) % 2 == 0 else -1)][::(1 if i % 2 == 0 else -1)] for j in range(3)], 1) for i in range(3)]
    prev = np.ascontiguousarray(np.concatenate(rows, 0))
    h, w = prev.shape[:2]
    v, om, d, n = (0.004, -0.003, 0.002), (0.003, -0.002, 0.004), 1.0, (0.0, 0.0, 1.0)
    scaling, cx, cy = 1.0 / w, w / 2.0, h / 2.0
    nxt = synth.warp_frame(prev, synth.pixel_homography(v, om, d, n, scaling, cx, cy))
    cfg = {"node": PipelineConfig.node(), "evaluate": PipelineConfig.evaluate_exp(), "baseline": PipelineConfig.baseline_1080p()}[preset]
    sensors = ofk.make_sensors(1, d=d, normal=n, omega=om, scaling=scaling, cx=cx, cy=cy)
    pipe = FlowPipeline(w, h, 1, cfg)
    pipe.upload(prev[None], nxt[None], sensors)
    out = pipe.run()
    g0, g1 = io.gray_bgr8(prev), io.gray_bgr8(nxt)
    pts = io.good_features(g0, cfg.max_corners, cfg.quality, cfg.min_distance, cfg.block_size)
    cnt = int(out["counts"][0])
    assert cnt == len(pts) and cnt >= 5                          # 6 corners with the node's parameters: real footage at quality 0.7
    assert np.array_equal(out["prev_pts"][0, :cnt], pts.reshape(-1, 2))
    rn, rs, re = io.lk_pyr(g0, g1, pts, cfg.win, cfg.max_level, cfg.max_count, cfg.eps, cfg.min_eig_thr)
    assert np.array_equal(out["status"][0, :cnt], rs.ravel())
    assert np.array_equal(out["next_pts"][0, :cnt].view(np.uint32), rn.reshape(-1, 2).view(np.uint32))
    assert np.array_equal(out["err"][0, :cnt].view(np.uint32), re.ravel().view(np.uint32))
    ok = rs.ravel() == 1
    assert ok.sum() >= 4
    new = rn.reshape(-1, 2).astype(np.float64); old = pts.reshape(-1, 2).astype(np.float64)
    vref = eo.solve_lgs_node((new[ok] - [cx, cy]) * scaling, (new[ok] - old[ok]) * scaling, d, np.asarray(n), np.asarray(om))[0]
    np.testing.assert_allclose(out["records"][0, :3], vref, rtol=1e-9, atol=1e-13)
    pipe.close()

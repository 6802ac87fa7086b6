"""GPU parity, image stages: every HIP kernel is compared BIT-EXACTLY with oracle/image_oracle.c through the
C ABI (ctypes).  The oracle's image semantics are parity-unpinned w.r.t. OpenCV (absent); see its header."""
import numpy as np
import pytest

from oracle import image_oracle as io

pytestmark = pytest.mark.gpu


def rand_u8(shape, seed):
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def textured(h, w, seed, pkg):
    from of_amd import synth
    return np.clip(np.rint(synth.make_texture(h, w, seed)), 0, 255).astype(np.uint8)


SHAPES = [(240, 320), (33, 47), (135, 241), (480, 640), (1080, 1920)]


@pytest.mark.parametrize("shape", SHAPES)
def test_gray(gpu_ctx, shape):
    bgr = rand_u8((2,) + shape + (3,), 1)
    got = gpu_ctx.gray_bgr8(bgr)
    for b in range(2):
        assert np.array_equal(got[b], io.gray_bgr8(bgr[b]))
    # KATs: white stays white, pure channels hit the coefficient table
    kat = np.zeros((1, 16, 16, 3), np.uint8); kat[0, 0] = 255; kat[0, 1, :, 0] = 255; kat[0, 2, :, 1] = 255; kat[0, 3, :, 2] = 255
    g = gpu_ctx.gray_bgr8(kat)[0]
    assert g[0, 0] == 255 and g[1, 0] == 29 and g[2, 0] == 150 and g[3, 0] == 76 and g[4, 0] == 0


# widths that are multiples of 8 run the streaming kernel (strips of 496 source columns): one strip, exactly one strip,
# the right edge in the strip's last lane / first lanes of the next strip, odd heights, the smallest size it accepts
PYR_STREAM_SHAPES = [(4, 16), (5, 24), (37, 496), (64, 504), (35, 512), (21, 992), (67, 1000), (135, 960), (540, 960), (3, 16)]


@pytest.mark.parametrize("shape", SHAPES + [(16, 17), (2, 2), (3, 300)] + PYR_STREAM_SHAPES)
def test_pyr_down(gpu_ctx, shape):
    src = rand_u8((2,) + shape, 2)
    got = gpu_ctx.pyr_down(src)
    for b in range(2):
        assert np.array_equal(got[b], io.pyr_down(src[b]))
    assert np.all(gpu_ctx.pyr_down(np.full(shape, 77, np.uint8)) == 77)


# fused three-level pass (widths % 16 == 0, heights % 8 == 0, >= 64 x 32): one strip, two strips (1920), three and more strips,
# chunk boundaries at several heights; other shapes take the level-by-level kernels through the same entry point
PYR3_SHAPES = [(32, 64), (40, 976), (48, 992), (56, 1936), (1080, 1920), (64, 2000), (264, 4000), (544, 1024), (136, 80), (30, 64), (32, 72)]


@pytest.mark.parametrize("shape", PYR3_SHAPES)
def test_pyramid_levels(ofk, shape):
    h, w = shape
    levels = 4 if min(h, w) >= 64 else 3
    src = rand_u8((3, h, w), 11)
    src[1] = 255; src[2, ::2] = 0                              # saturated and striped images: rounding and mirrored borders
    ctx = ofk.Context(0, w, h, 3, 16, levels)
    got = ctx.pyramid(src, levels)
    assert len(got) == levels
    for b in range(3):
        ref = src[b]
        for l in range(levels):
            ref = io.pyr_down(ref)
            assert np.array_equal(got[l][b], ref), (b, l)
    ctx.close()


@pytest.mark.parametrize("shape", SHAPES + [(2, 2), (5, 130)])
def test_scharr(gpu_ctx, shape):
    src = rand_u8((2,) + shape, 3)
    got = gpu_ctx.scharr(src)
    for b in range(2):
        assert np.array_equal(got[b], io.scharr(src[b]))


@pytest.mark.parametrize("block", [3, 7, 12, 32])
@pytest.mark.parametrize("shape", [(240, 320), (97, 131), (480, 640)])
def test_mineig_bit_exact(gpu_ctx, pkg, shape, block):
    imgs = np.stack([rand_u8(shape, 4), textured(shape[0], shape[1], 5, pkg)])
    got = gpu_ctx.mineig(imgs, block)
    for b in range(2):
        ref = io.mineig(imgs[b], block)
        assert np.array_equal(got[b].view(np.uint32), ref.view(np.uint32)), f"max diff {np.abs(got[b] - ref).max()}"


def test_mineig_1080p_and_invariance(gpu_ctx, pkg):
    img = textured(1080, 1920, 6, pkg)
    got = gpu_ctx.mineig(img, 7)
    assert np.array_equal(got.view(np.uint32), io.mineig(img, 7).view(np.uint32))
    # response is invariant under intensity inversion (gradients flip sign, products do not)
    assert np.array_equal(gpu_ctx.mineig(255 - img, 7).view(np.uint32), got.view(np.uint32))
    assert np.all(gpu_ctx.mineig(np.full((64, 64), 9, np.uint8), 7) == 0)


def corners_equal(gpu_ctx, img, mc, q, md, bs, mask=None):
    got = gpu_ctx.good_features(img, mc, q, md, bs, mask=mask)
    ref = io.good_features(img, mc, q, md, bs, mask=mask)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(got, ref)
    return got


PARAMS = [(50, 0.3, 20, 32), (100, 0.7, 10, 12), (20, 0.7, 10, 7), (500, 0.01, 10, 7), (2000, 0.001, 3, 3), (300, 0.05, 0, 5)]


@pytest.mark.parametrize("mc,q,md,bs", PARAMS)
def test_good_features_param_sets(gpu_ctx, pkg, mc, q, md, bs):
    for seed, shape in [(7, (480, 640)), (8, (240, 320))]:
        pts = corners_equal(gpu_ctx, textured(shape[0], shape[1], seed, pkg), mc, q, md, bs)
        assert len(pts) > 0


def test_good_features_1080p_500(gpu_ctx, pkg):
    pts = corners_equal(gpu_ctx, textured(1080, 1920, 9, pkg), 500, 0.01, 10, 7)
    assert len(pts) == 500


def test_good_features_many_candidates_chunked(gpu_ctx):
    """Noise image at a tiny quality level: tens of thousands of candidates, several 4096-key selection rounds."""
    img = rand_u8((480, 640), 10)
    eig = io.mineig(img, 3)
    _, ncand = io.select_corners(eig, 2000, 1e-4, 12.0)
    assert ncand > 3 * 4096
    pts = corners_equal(gpu_ctx, img, 2000, 1e-4, 12.0, 3)
    assert len(pts) > 1000


def test_good_features_ties_mask_and_empty(gpu_ctx):
    # checkerboard: massive exact ties -> exercises the (value desc, index asc) order
    yy, xx = np.mgrid[0:240, 0:320]
    chk = (((yy // 16) + (xx // 16)) % 2 * 200 + 20).astype(np.uint8)
    pts = corners_equal(gpu_ctx, chk, 400, 0.2, 5, 5)
    assert len(pts) > 50
    # single bright square: its 4 corners, symmetric responses
    sq = np.zeros((120, 160), np.uint8); sq[40:80, 50:110] = 255
    pts = corners_equal(gpu_ctx, sq, 10, 0.5, 10, 3)
    assert len(pts) == 4
    # constant image -> no corners
    assert len(gpu_ctx.good_features(np.full((64, 64), 128, np.uint8), 10, 0.1, 5, 3)) == 0
    # mask excludes the left half (also from the max)
    img = rand_u8((240, 320), 11)
    mask = np.ones((240, 320), np.uint8); mask[:, :160] = 0
    pts = corners_equal(gpu_ctx, img, 100, 0.1, 8, 7, mask=mask)
    assert len(pts) and np.all(pts[:, 0, 0] >= 160)


def test_select_corners_from_map_batch(gpu_ctx):
    eig = np.stack([io.mineig(rand_u8((200, 300), s), 5) for s in (12, 13, 14)])
    pts, cnt = gpu_ctx.select_corners(eig, 64, 0.05, 9.0)
    for b in range(3):
        ref, _ = io.select_corners(eig[b], 64, 0.05, 9.0)
        assert cnt[b] == len(ref) and np.array_equal(pts[b, :cnt[b]], ref.reshape(-1, 2))


def lk_equal(gpu_ctx, prev, nxt, pts, **kw):
    gn, gs, ge = gpu_ctx.lk_pyr(prev, nxt, pts, **kw)
    rn, rs, re = io.lk_pyr(prev, nxt, pts, **kw)
    assert np.array_equal(gs, rs), f"status differs at {np.nonzero(gs != rs)[0][:8]}"
    assert np.array_equal(gn.view(np.uint32), rn.view(np.uint32)), f"max |d| {np.abs(gn - rn).max()}"
    assert np.array_equal(ge.view(np.uint32), re.view(np.uint32))
    return gn, gs, ge


@pytest.fixture(scope="module")
def pair480(pkg):
    from of_amd import synth
    p = synth.render_pair(480, 640, 21, v=(0.004, -0.003, 0.002), omega=(0.004, -0.002, 0.006), d=1.0)
    return io.gray_bgr8(p["prev"]), io.gray_bgr8(p["next"]), p


@pytest.mark.parametrize("kw", [dict(win=15, max_level=3, max_count=20, eps=0.03), dict(win=15, max_level=3, max_count=10, eps=0.5),
                                dict(win=15, max_level=0, max_count=30, eps=0.01), dict(win=21, max_level=2, max_count=20, eps=0.03),
                                dict(win=31, max_level=5, max_count=20, eps=0.03), dict(win=7, max_level=4, max_count=5, eps=0.0),
                                dict(win=15, max_level=3, max_count=0, eps=0.03)])
def test_lk_bit_exact(gpu_ctx, pair480, kw):
    g0, g1, p = pair480
    pts = io.good_features(g0, 300, 0.01, 7, 7)
    gn, gs, ge = lk_equal(gpu_ctx, g0, g1, pts, **kw)
    if kw["max_count"] >= 10 and kw["max_level"] >= 2:
        from of_amd import synth
        ok = gs.ravel() == 1
        assert ok.mean() > 0.9
        flow = (gn - pts).reshape(-1, 2)[ok]
        truth = synth.true_flow_px(p["H"], pts.reshape(-1, 2)[ok])
        assert np.median(np.abs(flow - truth)) < 0.1          # tracks the rendered motion


def test_lk_edge_cases(gpu_ctx, pair480):
    g0, g1, _ = pair480
    h, w = g0.shape
    # points on and outside the border, plus interior ones
    pts = np.array([[0, 0], [w - 1, h - 1], [3.5, 2.25], [w - 2.5, 10.75], [-30, 50], [w + 40, 20], [100, -25], [w / 2, h / 2],
                    [7, 7], [w - 8, h - 8], [6.99, 100.01], [-8, -8], [w + 7, h + 7]], np.float32).reshape(-1, 1, 2)
    lk_equal(gpu_ctx, g0, g1, pts, win=15, max_level=3, max_count=20, eps=0.03)
    # identical frames: zero flow, status 1, err 0 on textured points
    tp = io.good_features(g0, 50, 0.05, 10, 7)
    gn, gs, ge = lk_equal(gpu_ctx, g0, g0, tp, win=15, max_level=3, max_count=20, eps=0.03)
    assert np.all(gs == 1) and np.array_equal(gn, tp) and np.all(ge == 0)
    # flat image: rejected by the min-eigenvalue test
    flat = np.full((h, w), 100, np.uint8)
    gn, gs, ge = lk_equal(gpu_ctx, flat, flat, tp, win=15, max_level=3, max_count=20, eps=0.03)
    assert np.all(gs == 0)


def test_lk_batch_ragged_and_translation(gpu_ctx, pkg):
    from of_amd import synth
    imgs = [textured(300, 400, 30 + i, pkg) for i in range(3)]
    prev = np.stack(imgs)
    nxt = np.stack([np.roll(im, (2 + i, -3 - i), axis=(0, 1)) for i, im in enumerate(imgs)])
    counts = np.array([40, 0, 17], np.int32)
    pts = np.zeros((3, 40, 2), np.float32)
    for b in range(3):
        pts[b, :counts[b]] = io.good_features(prev[b], 40, 0.05, 12, 7).reshape(-1, 2)[:counts[b]]
    gn, gs, ge = gpu_ctx.lk_pyr(prev, nxt, pts, counts, win=15, max_level=2, max_count=20, eps=0.03)
    for b in range(3):
        n = counts[b]
        rn, rs, re = io.lk_pyr(prev[b], nxt[b], pts[b, :n], win=15, max_level=2, max_count=20, eps=0.03)
        assert np.array_equal(gs[b, :n], rs.ravel()) and np.array_equal(gn[b, :n], rn.reshape(-1, 2)) and np.array_equal(ge[b, :n], re.ravel())
        if n:
            inner = (pts[b, :n, 0] > 30) & (pts[b, :n, 0] < 370) & (pts[b, :n, 1] > 30) & (pts[b, :n, 1] < 270) & (rs.ravel() == 1)
            d = (rn.reshape(-1, 2) - pts[b, :n])[inner]
            assert np.allclose(d, [-3 - b, 2 + b], atol=0.02)      # pure integer translation is recovered


def test_lk_1080p_500_points(gpu_ctx, pkg):
    from of_amd import synth
    p = synth.render_pair(1080, 1920, 41, v=(0.002, -0.0015, 0.001), omega=(0.002, -0.001, 0.003), d=1.0)
    g0, g1 = io.gray_bgr8(p["prev"]), io.gray_bgr8(p["next"])
    pts = io.good_features(g0, 500, 0.01, 10, 7)
    gn, gs, ge = lk_equal(gpu_ctx, g0, g1, pts, win=15, max_level=3, max_count=20, eps=0.03)
    assert (gs == 1).mean() > 0.95


def test_lk_quad_kernel_borders_and_dead_rows(gpu_ctx, pkg):
    """k_lk15q (four points per wave): windows and staged regions over every border of a 1080p pair — mirrored columns are built
    from the mirrored dwords, rows are reflected per lane — with point counts that leave rows of the last wave idle, and a next
    frame shifted far enough that tracks leave the image and the staged region has to follow the window."""
    g0 = textured(1080, 1920, 77, pkg)
    g1 = np.roll(g0, (5, -12), axis=(0, 1))
    h, w = g0.shape
    rng = np.random.default_rng(5)
    edge = []
    for x in (-14.5, -3.25, 0, 0.5, 3.75, 7, 8.5, 12.25, 16, 23.5, 24, 27.75, 31.5):
        for y in (-14, 0.25, 7.5, 19, 400.5):
            edge += [[x, y], [w - 1 - x, y], [x, h - 1 - y], [w - 1 - x, h - 1 - y], [y, x], [w - 1 - y, h - 1 - x]]
    pts = np.array(edge + rng.uniform([-10, -10], [w + 10, h + 10], (203, 2)).tolist(), np.float32).reshape(-1, 1, 2)
    for n in (len(pts), 1, 2, 3, 5):
        lk_equal(gpu_ctx, g0, g1, pts[:n], win=15, max_level=3, max_count=30, eps=0.01)
    lk_equal(gpu_ctx, g0, np.roll(g0, (-40, 60), axis=(0, 1)), pts, win=15, max_level=1, max_count=30, eps=0.01)   # long walks at a fine level


def test_lk_quad_kernel_full_contrast(gpu_ctx):
    """Window sums beyond int32: binary noise puts A11, A22 above the Cauchy-Schwarz limit of the 32-bit tree (and a step edge
    pattern puts them above 2^31), so the mismatch sums and A12 go through the split reduction."""
    rng = np.random.default_rng(9)
    noise = (rng.integers(0, 2, (256, 320)) * 255).astype(np.uint8)
    stripes = np.zeros((256, 320), np.uint8); stripes[:, (np.arange(320) // 2) % 2 == 1] = 255; stripes[::9] ^= 255   # |Ix| = 4080 nearly everywhere
    checker = (((np.add.outer(np.arange(256), np.arange(320)) // 2) & 1) * 255).astype(np.uint8)                      # diagonal: A12 ~ A11 > 2^31
    pts = rng.uniform([5, 5], [315, 250], (64, 2)).astype(np.float32).reshape(-1, 1, 2)
    for img in (noise, stripes, checker):
        for shift in ((0, 0), (1, -1), (0, 2)):
            gn, gs, ge = lk_equal(gpu_ctx, img, np.roll(img, shift, axis=(0, 1)), pts, win=15, max_level=2, max_count=20, eps=0.03)
    mixed = noise.copy(); mixed[:, 160:] = (rng.integers(0, 256, (256, 160)) // 4 + 96).astype(np.uint8)      # both paths inside one wave
    lk_equal(gpu_ctx, mixed, np.roll(mixed, (1, 1), axis=(0, 1)), pts, win=15, max_level=2, max_count=20, eps=0.03)


def test_lk_quad_kernel_smallest_levels(gpu_ctx, pkg):
    """The four-point kernel at the smallest geometry it accepts (coarsest level 40 x 32: one reflection has to cover staged rows 23
    and staged columns 27 past the border, and the 36-byte strip of a row nearly spans the level), points on and beyond every border."""
    g0 = textured(256, 320, 91, pkg)
    g1 = np.roll(g0, (3, -4), axis=(0, 1))
    rng = np.random.default_rng(17)
    h, w = g0.shape
    grid = np.array([[x, y] for x in (-14, -6.5, 0, 3.25, 9, 40.5, w / 2, w - 41, w - 10.5, w - 4, w - 1, w + 5.5, w + 13)
                     for y in (-13, -2.5, 0, 7.75, 31, h / 2, h - 33, h - 8.25, h - 1, h + 6, h + 12.5)], np.float32)
    pts = np.concatenate([grid, rng.uniform([-14, -14], [w + 14, h + 14], (97, 2)).astype(np.float32)]).reshape(-1, 1, 2)
    lk_equal(gpu_ctx, g0, g1, pts, win=15, max_level=3, max_count=30, eps=0.01)
    lk_equal(gpu_ctx, g0, np.roll(g0, (-9, 11), axis=(0, 1)), pts, win=15, max_level=3, max_count=30, eps=0.01)
    # one level narrower than the kernel accepts: the one-wave-per-point kernel takes over, same answers required
    g2 = textured(256, 312, 92, pkg)
    lk_equal(gpu_ctx, g2, np.roll(g2, (2, 2), axis=(0, 1)), pts[:64], win=15, max_level=3, max_count=30, eps=0.01)


@pytest.mark.parametrize("bs", [7, 5, 3])
def test_good_features_plateau_rows_fill_the_key_buffer(gpu_ctx, bs):
    """A texture whose period equals the box size makes every box sum - and so the response - identical over a whole patch: every
    pixel of the patch ties with its neighbours and passes the 3x3 test, i.e. a strip of the response kernel produces one key per
    column for several rows in a row (more would run into the segment capacity, which assumes one key per four pixels of a strip chunk).
    The wave's key buffer holds two such rows (k_mineig_pair: NBUF = 64 + 2 SW) and has to
    spill in front of every second row; corners, their order (ties: higher index first) and the count must still be the oracle's."""
    rng = np.random.default_rng(3 + bs)
    h, w = 216, 640
    img = rng.integers(100, 112, (h, w)).astype(np.uint8)                     # faint background: the plateau holds the strongest responses
    tile = rng.integers(0, 256, (bs, bs)).astype(np.uint8)
    ph, pw = 6 + 2 * (bs // 2 + 1) + 1, 300                                   # patch: spans three 116-column strips; ~6 rows of plateau
    img[40:40 + ph, 100:100 + pw] = np.tile(tile, (ph // bs + 1, pw // bs + 1))[:ph, :pw]
    for mc, q, md in ((300, 0.001, 0.0), (500, 0.01, 3.0)):
        corners_equal(gpu_ctx, img, mc, q, md, bs)
    eig = io.mineig(img, bs)
    m = bs // 2 + 1
    inner = eig[40 + m:40 + ph - m, 100 + m:100 + pw - m]
    assert inner.shape[0] >= 5 and np.ptp(inner) == 0 and inner[0, 0] > 0.2 * eig.max()    # one plateau, among the strongest responses


def test_good_features_on_nearly_flat_images_with_isolated_dots(gpu_ctx):
    """The response kernels take lambda_min's square root from v_rsq_f32 + one Newton step and clamp its argument away from zero
    (k_corners.hip sqrt_rn_normal).  The argument IS zero wherever Sxx == Syy and Sxy == 0 - flat regions, but also every window that
    holds one isolated dot (its Sobel pattern is symmetric: sum dx^2 == sum dy^2, sum dx dy == 0), where a + c is as small as it gets
    (amplitude 1: 24 kd ~ 2e-7).  There (a + c) - root must still round to a + c, the plateaus of equal responses around each dot must
    come out as the oracle's, and on an image of nothing but such dots they ARE the maxima the quality level is taken from."""
    h, w = 216, 464
    rng = np.random.default_rng(11)
    for amp, base in ((1, 0), (1, 254), (2, 17), (3, 128), (255, 0)):
        img = np.full((h, w), base, np.uint8)
        ys = rng.integers(12, h - 12, 40); xs = rng.integers(12, w - 12, 40)
        img[ys, xs] = np.clip(base + amp, 0, 255) if base + amp <= 255 else base - amp
        for bs in (3, 5, 7, 12):
            assert np.array_equal(gpu_ctx.mineig(img, bs).view(np.uint32), io.mineig(img, bs).view(np.uint32)), (amp, base, bs)
            for mc, q, md in ((400, 0.001, 0.0), (60, 0.5, 5.0), (400, 0.999, 1.0)):
                corners_equal(gpu_ctx, img, mc, q, md, bs)
    # dots beside real texture: the threshold comes from the texture, the dots' plateaus fall below it or not as in the oracle
    img = np.zeros((h, w), np.uint8)
    img[:, : w // 2] = rng.integers(0, 256, (h, w // 2))
    img[rng.integers(12, h - 12, 30), rng.integers(w // 2 + 12, w - 12, 30)] = 1
    for bs in (3, 7, 12):
        for q in (1e-6, 1e-3):
            corners_equal(gpu_ctx, img, 1000, q, 0.0, bs)


def test_tuning_knobs_do_not_change_results(gpu_ctx, pkg, ofk):
    """ofk_set_tuning (include/ofk.h) moves strip lengths and picks between kernels that compute the same thing: corners, pyramid
    levels and decoded-and-tracked records are bit-identical under every knob (what the header promises)."""
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    p = synth.render_pair(272, 448, 5)                           # width % 16 == 0, height % 8 == 0: the three-level pyramid pass applies
    g = io.gray_bgr8(p["prev"])
    base_pts = gpu_ctx.good_features(g, 120, 0.02, 6, 7)
    base_pyr = gpu_ctx.pyramid(g, 3)
    cfg = PipelineConfig(max_corners=80, quality=0.03, min_distance=7, max_level=3)
    sensors = ofk.make_sensors(2, scaling=p["scaling"], cx=p["cx"], cy=p["cy"])
    pipe = FlowPipeline(448, 272, 2, cfg)
    pipe.upload(np.stack([p["prev"]] * 2), np.stack([p["next"]] * 2), sensors)
    base_out = pipe.run()
    assert len(base_pts) > 40
    try:
        for knob, values in (("no_pair", (1,)), ("eig_rows", (8, 33, 100)), ("no_pyr3", (1,)), ("pyr3_chunks", (1, 3)), ("pyr_rows", (4, 9)),
                             ("gray_px", (16, 32, 64))):
            for v in values:
                ofk.set_tuning(knob, v)
                assert np.array_equal(gpu_ctx.good_features(g, 120, 0.02, 6, 7), base_pts), (knob, v)
                for a, b in zip(gpu_ctx.pyramid(g, 3), base_pyr):
                    assert np.array_equal(a, b), (knob, v)
                out = pipe.run()
                for key in ("counts", "prev_pts", "next_pts", "status", "records"):
                    assert np.array_equal(out[key], base_out[key]), (knob, v, key)
            ofk.set_tuning(knob, 0)
    finally:
        for knob in ("no_pair", "eig_rows", "no_pyr3", "pyr3_chunks", "pyr_rows", "jpeg_chunk", "jpeg_sub", "gray_px"):
            ofk.set_tuning(knob, 0)
    pipe.close()

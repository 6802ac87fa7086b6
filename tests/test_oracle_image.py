"""CPU-only: analytic known-answer tests and an independent numpy cross-check of oracle/image_oracle.c
(the image-stage oracle is parity-unpinned w.r.t. OpenCV; these pin it to SURVEY.md Appendix B)."""
import numpy as np
import pytest

from oracle import image_oracle as io


def r101(i, n):
    i = np.asarray(i)
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def np_conv_sep(img, kx, ky):
    """Separable correlation with REFLECT_101, integer."""
    h, w = img.shape
    rx, ry = len(kx) // 2, len(ky) // 2
    a = img.astype(np.int64)
    t = sum(kx[i] * a[:, r101(np.arange(w) + i - rx, w)] for i in range(len(kx)))
    return sum(ky[j] * t[r101(np.arange(h) + j - ry, h), :] for j in range(len(ky)))


@pytest.fixture(scope="module")
def img():
    return np.random.default_rng(0).integers(0, 256, (61, 83), dtype=np.uint8)


def test_gray_formula():
    bgr = np.random.default_rng(1).integers(0, 256, (17, 19, 3), dtype=np.uint8)
    b, g, r = [bgr[..., i].astype(np.int64) for i in range(3)]
    assert np.array_equal(io.gray_bgr8(bgr), ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8))
    assert io.gray_bgr8(np.full((2, 2, 3), 255, np.uint8))[0, 0] == 255


def test_pyr_down_vs_numpy(img):
    full = np_conv_sep(img, [1, 4, 6, 4, 1], [1, 4, 6, 4, 1])
    assert np.array_equal(io.pyr_down(img), ((full[::2, ::2] + 128) >> 8).astype(np.uint8))
    assert io.pyr_down(img).shape == (31, 42)
    assert io.lk_levels(1080, 1920, 15, 3) == 3 and io.lk_levels(60, 60, 15, 5) == 1 and io.lk_levels(1080, 1920, 15, 8) == 6


def test_scharr_vs_numpy(img):
    d = io.scharr(img)
    assert np.array_equal(d[..., 0], np_conv_sep(img, [-1, 0, 1], [3, 10, 3]))
    assert np.array_equal(d[..., 1], np_conv_sep(img, [3, 10, 3], [-1, 0, 1]))


@pytest.mark.parametrize("bs", [3, 6, 7])
def test_mineig_vs_numpy(img, bs):
    dx = np_conv_sep(img, [-1, 0, 1], [1, 2, 1]); dy = np_conv_sep(img, [1, 2, 1], [-1, 0, 1])
    h, w = img.shape
    an = bs // 2

    def box(p):
        t = sum(p[:, r101(np.arange(w) - an + i, w)] for i in range(bs))
        return sum(t[r101(np.arange(h) - an + j, h), :] for j in range(bs))

    sxx, sxy, syy = box(dx * dx), box(dx * dy), box(dy * dy)
    scale = 1.0 / (4.0 * bs * 255.0)
    kd, ko = np.float32(0.5 * scale * scale), np.float32(scale * scale)
    a = sxx.astype(np.float32) * kd; b = sxy.astype(np.float32) * ko; c = syy.astype(np.float32) * kd
    amc = a - c
    ref = (a + c) - np.sqrt(amc * amc + b * b, dtype=np.float32)
    assert np.array_equal(io.mineig(img, bs).view(np.uint32), ref.astype(np.float32).view(np.uint32))


def test_corner_kats():
    assert len(io.good_features(np.full((40, 40), 7, np.uint8), 10, 0.1, 3, 3)) == 0          # constant image
    sq = np.zeros((120, 160), np.uint8); sq[40:80, 50:110] = 255
    pts = io.good_features(sq, 10, 0.5, 10, 3).reshape(-1, 2)
    assert len(pts) == 4
    for cx, cy in [(50, 40), (109, 40), (50, 79), (109, 79)]:
        assert np.min(np.hypot(pts[:, 0] - cx, pts[:, 1] - cy)) <= 1.5
    # translation equivariance away from the border
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (140, 180), dtype=np.uint8)
    a = io.good_features(base[10:110, 10:150], 30, 0.2, 8, 5).reshape(-1, 2)
    eig_a = io.mineig(base[10:110, 10:150], 5); eig_b = io.mineig(base[15:115, 17:157], 5)
    assert np.array_equal(eig_a[15:80, 17:120], eig_b[10:75, 10:113])
    assert len(a) > 5
    # min-distance respected, ordering by response
    eig = io.mineig(base, 5)
    p, _ = io.select_corners(eig, 50, 0.05, 9.0)
    p = p.reshape(-1, 2)
    dist = np.hypot(p[:, None, 0] - p[None, :, 0], p[:, None, 1] - p[None, :, 1]) + 1e9 * np.eye(len(p))
    assert dist.min() >= 9.0
    vals = eig[p[:, 1].astype(int), p[:, 0].astype(int)]
    assert np.all(np.diff(vals) <= 0)
    # mask
    m = np.ones_like(base); m[:, :90] = 0
    pm = io.good_features(base, 20, 0.1, 5, 5, mask=m).reshape(-1, 2)
    assert len(pm) and np.all(pm[:, 0] >= 90)


def test_lk_kats():
    rng = np.random.default_rng(5)
    t = rng.standard_normal((200, 260))
    k = np.exp(-0.5 * (np.arange(-6, 7) / 2.0) ** 2); k /= k.sum()
    t = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, t); t = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, t)
    img = np.clip(127 + 60 * t / t.std(), 0, 255).astype(np.uint8)
    pts = io.good_features(img, 40, 0.05, 12, 7)
    # identical frames: zero displacement, status 1, err 0
    n, s, e = io.lk_pyr(img, img, pts)
    assert np.all(s == 1) and np.array_equal(n, pts) and np.all(e == 0)
    # integer translation of a smooth texture
    sh = np.roll(img, (3, -2), axis=(0, 1))
    n, s, e = io.lk_pyr(img, sh, pts)
    p2 = pts.reshape(-1, 2)
    inner = (p2[:, 0] > 30) & (p2[:, 0] < 230) & (p2[:, 1] > 30) & (p2[:, 1] < 170) & (s.ravel() == 1)
    assert inner.sum() > 10 and np.allclose((n.reshape(-1, 2) - p2)[inner], [-2, 3], atol=0.02)
    # window leaves the padded image -> status 0; flat patch -> rejected
    far = np.array([[-40.0, 10.0], [400.0, 50.0]], np.float32)
    n, s, e = io.lk_pyr(img, sh, far)
    assert np.all(s == 0) and np.all(e == 0)
    flat = np.full_like(img, 90)
    n, s, e = io.lk_pyr(flat, flat, pts)
    assert np.all(s == 0)

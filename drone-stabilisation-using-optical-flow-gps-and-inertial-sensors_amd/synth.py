"""synth.py — synthetic frame-pair renderer (measurement harness; numpy only).

The reference never synthesises images (numerical_simulation/simulation.py generates point flows,
its videos are missing blobs), so every benchmark/test frame comes from here: a seeded, band-limited
texture on the ground plane, viewed before and after the plane-induced homography
    p2 ~ (I + [w]x + v n^T / d) p1
whose first-order displacement is exactly the reference's flow model (node:25-29; simulation.py:7-12).
v and w are per-frame quantities (velocity*dt, rate*dt), p in normalised centred coordinates
x = (px - c) * scaling (node:229-235).
"""
import numpy as np


def _blur(a, sigma):
    r = max(1, int(3 * sigma + 0.5))
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2); k /= k.sum()
    for axis in (0, 1):
        pad = [(r, r) if ax == axis else (0, 0) for ax in range(2)]
        ap = np.pad(a, pad, mode="reflect")
        out = np.zeros_like(a)
        for i, kv in enumerate(k):
            sl = [slice(None)] * 2
            sl[axis] = slice(i, i + a.shape[axis])
            out += kv * ap[tuple(sl)]
        a = out
    return a


def make_texture(h, w, seed, sigma=1.8):
    """Band-limited noise + soft-edged random rectangles (plenty of Shi-Tomasi corners), float32 in [8, 247]."""
    rng = np.random.default_rng(seed)
    t = rng.standard_normal((h, w)).astype(np.float32)
    t = _blur(t, sigma)
    t /= max(1e-6, float(t.std()))
    blocks = np.zeros((h, w), np.float32)
    nrect = max(8, (h * w) // 6000)
    ys = rng.integers(0, h, nrect); xs = rng.integers(0, w, nrect)
    hs = rng.integers(6, 40, nrect); ws = rng.integers(6, 40, nrect)
    amp = rng.uniform(-2.0, 2.0, nrect).astype(np.float32)
    for y, x, hh, ww_, a in zip(ys, xs, hs, ws, amp):
        blocks[y:y + hh, x:x + ww_] += a
    blocks = _blur(blocks, 1.0)
    t = 0.7 * t + 0.6 * blocks
    t = 127.5 + 45.0 * t
    return np.clip(t, 8, 247).astype(np.float32)


def pixel_homography(v, omega, d, n, scaling, cx, cy):
    v = np.asarray(v, np.float64); om = np.asarray(omega, np.float64); n = np.asarray(n, np.float64)
    W = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    Hn = np.eye(3) + W + np.outer(v, n) / d
    K = np.array([[1 / scaling, 0, cx], [0, 1 / scaling, cy], [0, 0, 1.0]])
    return K @ Hn @ np.linalg.inv(K)


def _bilinear(img, xs, ys):
    h, w = img.shape
    xs = np.clip(xs, 0, w - 1.001); ys = np.clip(ys, 0, h - 1.001)
    x0 = np.floor(xs).astype(np.int32); y0 = np.floor(ys).astype(np.int32)
    a = (xs - x0).astype(np.float32); b = (ys - y0).astype(np.float32)
    i00 = img[y0, x0]; i01 = img[y0, x0 + 1]; i10 = img[y0 + 1, x0]; i11 = img[y0 + 1, x0 + 1]
    return (i00 * (1 - a) + i01 * a) * (1 - b) + (i10 * (1 - a) + i11 * a) * b


def render_pair(h, w, seed, v=(0.003, -0.002, 0.001), omega=(0.002, -0.001, 0.003), d=1.0, n=(0, 0, 1), scaling=None,
                margin=48):
    """Returns dict(prev, next: [h,w,3] uint8 BGR; H: 3x3 pixel homography prev->next; scaling, cx, cy)."""
    scaling = scaling or 1.0 / max(h, w)
    cx, cy = w / 2.0, h / 2.0
    T = make_texture(h + 2 * margin, w + 2 * margin, seed)
    T2 = make_texture(h + 2 * margin, w + 2 * margin, seed + 7919, sigma=3.0)
    H = pixel_homography(v, omega, d, n, scaling, cx, cy)
    Hi = np.linalg.inv(H)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
    sx = (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den + margin
    sy = (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den + margin

    def colour(base, tint):
        out = np.empty(base.shape + (3,), np.uint8)
        for ch, g in enumerate((0.10, -0.06, 0.08)):
            out[..., ch] = np.clip(np.rint(base + g * (tint - 127.5)), 0, 255).astype(np.uint8)
        return out

    prev = colour(T[margin:margin + h, margin:margin + w], T2[margin:margin + h, margin:margin + w])
    nxt = colour(_bilinear(T, sx, sy), _bilinear(T2, sx, sy))
    return dict(prev=prev, next=nxt, H=H, scaling=scaling, cx=cx, cy=cy, v=np.asarray(v, np.float64),
                omega=np.asarray(omega, np.float64), d=float(d), n=np.asarray(n, np.float64))


def warp_frame(bgr, H):
    """A BGR uint8 frame seen through the pixel homography H (prev -> next), bilinear, edge pixels replicated: the `next` frame of a
    pair whose `prev` is a given picture (a real camera frame instead of make_texture)."""
    h, w = bgr.shape[:2]
    Hi = np.linalg.inv(H)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
    sx = (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den
    sy = (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den
    out = np.empty_like(bgr)
    for ch in range(bgr.shape[2]):
        out[..., ch] = np.clip(np.rint(_bilinear(bgr[..., ch].astype(np.float32), sx, sy)), 0, 255).astype(np.uint8)
    return out


def render_sequence(h, w, seed, n_frames, v=(0.003, -0.002, 0.001), omega=(0.002, -0.001, 0.003), d=1.0, n=(0, 0, 1),
                    scaling=None, margin=96):
    """`n_frames` BGR frames of one stream under constant per-frame motion: frame k shows the texture through H^k.
    Returns (frames [n,h,w,3] uint8, info dict as render_pair)."""
    scaling = scaling or 1.0 / max(h, w)
    cx, cy = w / 2.0, h / 2.0
    T = make_texture(h + 2 * margin, w + 2 * margin, seed)
    T2 = make_texture(h + 2 * margin, w + 2 * margin, seed + 7919, sigma=3.0)
    H = pixel_homography(v, omega, d, n, scaling, cx, cy)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    frames = np.empty((n_frames, h, w, 3), np.uint8)
    Hk = np.eye(3)
    for k in range(n_frames):
        Hi = np.linalg.inv(Hk)
        den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
        sx = (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den + margin
        sy = (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den + margin
        base, tint = _bilinear(T, sx, sy), _bilinear(T2, sx, sy)
        for ch, g in enumerate((0.10, -0.06, 0.08)):
            frames[k, ..., ch] = np.clip(np.rint(base + g * (tint - 127.5)), 0, 255).astype(np.uint8)
        Hk = H @ Hk
    return frames, dict(H=H, scaling=scaling, cx=cx, cy=cy, v=np.asarray(v, np.float64), omega=np.asarray(omega, np.float64),
                        d=float(d), n=np.asarray(n, np.float64))


def true_flow_px(H, pts_xy):
    """Exact displacement (pixels) of prev-frame points under the pair's homography."""
    p = np.concatenate([np.asarray(pts_xy, np.float64).reshape(-1, 2), np.ones((len(pts_xy), 1))], 1) @ H.T
    return p[:, :2] / p[:, 2:3] - np.asarray(pts_xy, np.float64).reshape(-1, 2)


def make_batch(batch, h, w, seed, distinct=4, **kw):
    """`batch` frame pairs: `distinct` rendered pairs, the rest cyclic shifts of them (different pixels,
    same statistics) so that large benchmark batches build in seconds."""
    base = [render_pair(h, w, seed + i, **kw) for i in range(min(distinct, batch))]
    prev = np.empty((batch, h, w, 3), np.uint8); nxt = np.empty((batch, h, w, 3), np.uint8)
    for b in range(batch):
        src = base[b % len(base)]
        k = b // len(base)
        if k == 0:
            prev[b] = src["prev"]; nxt[b] = src["next"]
        else:
            sh = (17 * k) % h, (29 * k) % w
            prev[b] = np.roll(src["prev"], sh, axis=(0, 1)); nxt[b] = np.roll(src["next"], sh, axis=(0, 1))
    return prev, nxt, base

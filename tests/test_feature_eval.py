"""Per-feature estimators around the path: calc_height, dynamic_immobile (with convert_to_of) and eval_ft of
of_library.py, batched on the device.  PARITY UNPINNED: the reference's three functions carry undefined names and cannot be
executed, so the oracle restates the formulas they spell out; a hand-computed case pins the oracle, the device
is compared with the oracle."""
import numpy as np
import pytest

from oracle import estimation_oracle as eo


def make_set(rng, n, img=(320, 240)):
    pos = np.stack([rng.uniform(20, img[0] - 20, n), rng.uniform(20, img[1] - 20, n)], 1)
    flow = rng.normal(0, 3.0, (n, 2)) + np.array([2.5, -1.5])
    return pos, pos - flow, rng.uniform(0.01, 0.5, n), rng.uniform(0.01, 0.5, n)


def test_oracle_known_answer():
    # one feature right of the centre, pure x motion: height = f v_x / u_x on the x axis, (f v_y - y v_z) / u_y on the y axis
    pos = np.array([[200.0, 140.0], [100.0, 100.0]]); old = np.array([[196.0, 138.0], [101.0, 99.0]])
    h, he, imm, score, order = eo.feature_eval(pos, [0.1, 0.1], old, [0.1, 0.1], [2.0, 1.0, 0.0], [0.0, 0.0, 0.0], 100.0, -1.0, (320, 240), [1, 0, 0, 0])
    np.testing.assert_allclose(h, [0.5 * (200.0 / 4 + 100.0 / 2), 0.5 * (200.0 / -1 + 100.0 / 1)])
    # variance: only the flow-error terms survive (no velocity error, v_z = 0): ((f v_x) e / u_x^2)^2 + ((f v_y) e / u_y^2)^2
    np.testing.assert_allclose(he, [(200 * 0.1 / 16) ** 2 + (100 * 0.1 / 4) ** 2, (200 * 0.1 / 1) ** 2 + (100 * 0.1 / 1) ** 2])
    assert list(order) == [0, 1] and score[0] == 0.0 and score[1] == 1.0       # weight on height only: the highest feature scores 0


def _g12():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_association.npz"))
    return [(g[f"g12_{k}_d"], g[f"g12_{k}_sorted"], g[f"g12_{k}_diff"]) for k in range(int(g["g12_n"]))]


def test_oracle_d_split_matches_reference_lines():
    """node:250-251 exec'd by line range (tests/golden/make_golden_association.py, G12): pinned."""
    for d, srt, dif in _g12():
        s, g, n = eo.d_split(d, 0.3)
        assert np.array_equal(s, srt) and np.array_equal(g, dif) and n == int(np.sum(dif >= 0.3))


@pytest.mark.gpu
def test_device_d_split_matches_reference_lines(pkg, ofk, gpu_ctx):
    sets = _g12()
    n = max(len(d) for d, _, _ in sets)
    d = np.zeros((len(sets), n)); counts = np.array([len(x[0]) for x in sets], np.int32)
    for k, (x, _, _) in enumerate(sets):
        d[k, :len(x)] = x
    srt, dif, ns = gpu_ctx.d_split(d, 0.3, counts=counts)
    for k, (x, s, g) in enumerate(sets):
        c = len(x)
        assert np.array_equal(srt[k, :c], s) and np.array_equal(dif[k, :max(c - 1, 0)], g) and ns[k] == int(np.sum(g >= 0.3))
    s1, g1, n1 = gpu_ctx.d_split(sets[3][0], 0.05)                            # single set
    assert np.array_equal(s1, sets[3][1]) and np.array_equal(g1, sets[3][2]) and n1 == int(np.sum(sets[3][2] >= 0.05))
    with pytest.raises(ofk.OfkError):
        gpu_ctx.d_split(np.zeros((1, 5000)), 0.1)


@pytest.mark.gpu
def test_device_feature_eval_matches_oracle(pkg, ofk, gpu_ctx):
    rng = np.random.default_rng(21)
    B, n = 4, 300
    counts = np.array([300, 257, 1, 64], np.int32)
    pos = np.zeros((B, n, 2)); old = np.zeros((B, n, 2)); pe = np.zeros((B, n)); oe = np.zeros((B, n))
    for b in range(B):
        pos[b], old[b], pe[b], oe[b] = make_set(rng, n)
    old[0, 5] = [-1.0, 17.0]; old[0, 9, 1] = -1.0                               # dummy coordinates
    old[1, 3, 0] = pos[1, 3, 0]                                                # zero x flow: inf height, NaN score -> sorted last
    pe[3, :] = 0.25                                                            # constant error: its normalisation is all zeros
    vel = rng.normal(0, 1.0, (B, 3)) + [3.0, -2.0, 0.2]; vel_err = np.abs(rng.normal(0, 0.05, (B, 3)))
    w = [0.4, 0.3, 0.2, 0.1]
    out = gpu_ctx.feature_eval(pos, pe, old, oe, vel, vel_err, 320.0, -1.0, (320, 240), w, counts=counts)
    for b in range(B):
        c = int(counts[b])
        h, he, imm, score, order = eo.feature_eval(pos[b, :c], pe[b, :c], old[b, :c], oe[b, :c], vel[b], vel_err[b], 320.0, -1.0, (320, 240), w)
        np.testing.assert_allclose(out["height"][b, :c], h, rtol=1e-14, atol=0, equal_nan=True)
        np.testing.assert_allclose(out["height_err"][b, :c], he, rtol=1e-13, atol=0, equal_nan=True)
        assert np.array_equal(out["immobile"][b, :c].astype(bool), imm)
        np.testing.assert_allclose(out["score"][b, :c], score, rtol=1e-12, atol=1e-15, equal_nan=True)
        assert np.array_equal(out["order"][b, :c], order)
        assert np.all(out["order"][b, c:] == -1) and np.all(out["height"][b, c:] == 0)
    assert not out["immobile"][0, 5] and not out["immobile"][0, 9]
    assert out["order"][1, counts[1] - 1] == 3                                 # the NaN score comes last
    assert out["bad_height"]                                                   # random flows give negative heights somewhere
    # a set with positive heights only: flag stays clear; single-set call
    p1 = np.array([[200.0, 140.0], [120.0, 90.0], [250.0, 200.0]]); o1 = p1 - [[4.0, 2.0], [3.0, 1.5], [5.0, 2.5]]
    s1 = gpu_ctx.feature_eval(p1, 0.1, o1, 0.1, [2.0, 1.0, 0.0], [0.01, 0.01, 0.01], 100.0, -1.0, (320, 240), [1, 1, 1, 1])
    assert not s1["bad_height"] and s1["height"].shape == (3,) and np.all(s1["height"] > 0)

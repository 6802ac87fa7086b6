"""CPU-only: host-side logic of the facade that needs no kernel (of_library helpers, presets, sensor records)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def of(pkg):
    import of_amd.of_library as m
    return m


def test_pix_trans(of, golden):
    for i, o in zip(golden["g7_in"], golden["g7_out"]):
        assert of.pix_trans(tuple(int(v) for v in i)) == tuple(o)
    assert of.pix_trans((320, 240)) == (160.0, 120.0) and of.pix_trans((321, 241)) == (161.0, 121.0)


def test_static_immobile(of, golden):
    g = golden
    assert np.array_equal(of.static_immobile(g["g7_newpos"], g["g7_oldpos"], 3.0, 1.5, -1.0), g["g7_static"])


def test_value_errors(of):
    with pytest.raises(ValueError):
        of.convert_to_of(np.zeros((2, 3)), np.zeros((2, 3)), [1, 1], [0.1, 0.1], np.array([0.0, 1, 1]), 0.1, 1.0, (320, 240))
    with pytest.raises(ValueError):
        of.initialize_ft([], {}, {}, 0, 5, None, None, 1.0, -1, (320, 240), [1, 1, 1, 1])
    with pytest.raises(ValueError):
        of.initialize_ft([], {}, {}, 5, 0, None, None, 1.0, -1, (320, 240), [1, 1, 1, 1])


def test_mask_helpers(of):
    m = np.ones((60, 80), np.uint8)
    of.circles(np.array([[40.0, 20.0]]), m, 5)
    assert m[20, 40] == 0 and m[20, 45] == 0 and m[20, 46] == 1 and m[26, 40] == 1 and m.sum() == 60 * 80 - 81
    m = np.ones((60, 80), np.uint8)
    tri = np.array([[[10.0, 10.0]], [[50.0, 10.0]], [[30.0, 40.0]]], np.float32)
    of.convexhull([tri], m, 3)
    assert m[15, 30] == 0 and m[39, 30] == 0 and m[45, 30] == 1 and m[15, 5] == 1
    m = np.ones((60, 80), np.uint8)
    of.boundingboxes([tri, np.array([[[70.0, 50.0]]], np.float32)], m, 3)
    assert m[20, 30] == 0 and m[12, 12] == 0 and m[50, 70] == 0 and m[50, 60] == 1


def test_eval_ft_and_calc_height(of):
    rng = np.random.default_rng(0)
    n = 9
    of_flow = rng.uniform(0.5, 2.0, (n, 2)); vel = np.tile([1.0, 0.8, 0.1], (n, 1)); pos = rng.uniform(0, 100, (n, 2))
    hgt, herr = of.calc_height(of_flow, 0.01 * np.ones((n, 2)), vel, 0.05 * np.ones((n, 3)), 100.0, pos, 0.1 * np.ones((n, 2)))
    np.testing.assert_allclose(hgt, 0.5 * ((100 * 1.0 - pos[:, 0] * 0.1) / of_flow[:, 0] + (100 * 0.8 - pos[:, 1] * 0.1) / of_flow[:, 1]))
    assert np.all(herr > 0)
    h2, he2, p2, pe2 = of.eval_ft([1, 0, 0, 0], hgt, herr, pos, 0.1 * np.ones(n), (100, 100))
    assert np.all(np.diff(h2) <= 1e-12) and sorted(h2.tolist()) == sorted(hgt.tolist())     # weight on (1 - height_norm): tallest first


def test_clusters(of):
    pts = np.array([[0, 0], [1, 1], [0, 1], [50, 50], [51, 50], [50, 51]], np.float32).reshape(-1, 1, 2)
    cl = of.kmeancluster(pts, 2)
    assert sorted(len(c) for c in cl) == [3, 3]
    clusters, cloud = of.distancecluster(np.zeros((0, 2)), [[0, 0], [1, 1], [30, 30], [2, 1]], 3.0, [])
    assert sorted(sorted(c) for c in clusters) == [[0, 1, 3], [2]] and cloud.shape == (4, 2)


def test_presets_and_sensors(pkg, ofk):
    from of_amd.pipeline import PipelineConfig
    n = PipelineConfig.node(); m = PipelineConfig.of_module(); e = PipelineConfig.evaluate_exp()
    assert (n.max_corners, n.quality, n.min_distance, n.block_size, n.max_count, n.eps) == (100, 0.7, 10, 12, 20, 0.03)
    assert (m.max_corners, m.quality, m.min_distance, m.block_size, m.max_count, m.eps) == (50, 0.3, 20, 32, 10, 0.5)
    assert (e.max_corners, e.block_size) == (20, 7)
    p = n.to_params()
    assert p.max_corners == 100 and p.block_size == 12 and p.win == 15 and abs(p.eps - 0.03) < 1e-15
    s = ofk.make_sensors(3, d=0.75, omega=(.1, .2, .3), scaling=0.01, cx=160, cy=120)
    assert s.shape == (3, ofk.SENSOR_DOUBLES) and np.all(s[:, 0] == 0.75) and np.allclose(s[0, 7:16], np.eye(3).ravel())
    assert np.allclose(s[1, 16:19], [0, 0, 0.1]) and s[2, 19] == 0.01 and s[2, 20] == 160


def test_read_yaml_imu_safe_loader(of, tmp_path):
    y = tmp_path / "imu.yaml"
    y.write_text("""
- !!python/object/new:sensor_msgs.msg._Imu.Imu
  state:
  - !!python/object/new:std_msgs.msg._Header.Header
    state: [1, {secs: 5, nsecs: 250000000}, base]
  header: {stamp: {secs: 5, nsecs: 250000000}}
  orientation: {x: 0.0, y: 0.0, z: 0.0, w: 1.0}
  orientation_covariance: [0, 0, 0, 0, 0, 0, 0, 0, 0]
  linear_acceleration: {x: 0.1, y: 0.2, z: 9.8}
  linear_acceleration_covariance: [0, 0, 0, 0, 0, 0, 0, 0, 0]
  angular_velocity: {x: 0.01, y: 0.02, z: 0.03}
  angular_velocity_covariance: [0, 0, 0, 0, 0, 0, 0, 0, 0]
""".replace("- !!python/object/new:sensor_msgs.msg._Imu.Imu\n  state:\n  - !!python/object/new:std_msgs.msg._Header.Header\n    state: [1, {secs: 5, nsecs: 250000000}, base]\n", "- \n"))
    st = of.read_yaml_imu(str(y))
    assert len(st) == 1 and st[0][0] == 5 + 250.0 and st[0][3] == [0.1, 0.2, 9.8] and st[0][1] == [0.0, 0.0, 0.0, 1.0]


def test_failed_context_growth_keeps_the_old_default_context(pkg, monkeypatch):
    """ofk.default_context builds the bigger context FIRST: when that fails (a corrupt JPEG header asking for 65535 x 65535, out of
    device memory) the old default context must stay installed, open and usable (advisor finding of round 1, fixed in ofk.py)."""
    import of_amd.ofk as ofk

    class FakeContext:
        fail = False
        closed = []

        def __init__(self, device, w, h, batch, pts, lvl):
            if FakeContext.fail:
                raise ofk.OfkError(ofk.E_HIP, "hipMalloc failed")
            self.max_w, self.max_h, self.max_pts, self.max_level = w, h, pts, lvl

        def close(self):
            FakeContext.closed.append(self)

    monkeypatch.setattr(ofk, "Context", FakeContext)
    monkeypatch.setattr(ofk, "_default", None)
    first = ofk.default_context(640, 480, 64, 3)
    assert ofk.default_context(320, 240) is first                # fits: no rebuild
    FakeContext.fail = True
    with pytest.raises(ofk.OfkError):
        ofk.default_context(65535, 65535)
    assert ofk._default is first and first not in FakeContext.closed
    assert ofk.default_context(640, 480) is first                # still served by the old context
    FakeContext.fail = False
    bigger = ofk.default_context(4096, 2160)
    assert bigger is not first and FakeContext.closed == [first] and bigger.max_w >= 4096 and bigger.max_pts >= first.max_pts


def test_tuning_knobs_validate(pkg):
    import of_amd.ofk as ofk
    assert ofk.get_tuning("eig_rows") == 0
    ofk.set_tuning("eig_rows", 128)
    assert ofk.get_tuning("eig_rows") == 128
    ofk.set_tuning("eig_rows", 0)
    for knob, bad in (("eig_rows", 4), ("jpeg_chunk", 100), ("no_pair", 2), ("nonsense", 1)):
        with pytest.raises(ofk.OfkError):
            ofk.set_tuning(knob, bad)
    assert ofk.get_tuning("eig_rows") == 0 and ofk.get_tuning("jpeg_chunk") == 0

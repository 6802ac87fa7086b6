"""CPU-only: the committed measurement summaries are self-consistent (profiles/r02_traffic_pmc.json was half the
truth for every stage because a substring match doubled its step count, and bench.py passed it on).

  * the calibrated case: k_gray_bgr8's counter bytes == 2 * 4 * P * batch within 1 %;
  * no stage moves less than 0.8 x its algorithmic bytes (a cold kernel cannot);
  * the FETCH_SIZE factor in use is the calibrated one (tools/fetch_calib.hip: 2.0 for every access pattern of the pipeline);
  * bench.py refuses a traffic number that violates the second rule."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r04"


def load(name):
    p = os.path.join(ROOT, "profiles", f"{TAG}_{name}.json")
    if not os.path.exists(p):
        pytest.skip(f"{p} not committed yet")
    return json.load(open(p))


def test_traffic_summary_is_self_consistent():
    t = load("traffic_pmc")
    B, P = t["batch"], 1920 * 1080
    gray = t["stages"]["gray"]["hbm_bytes_per_step"]
    assert abs(gray / (2 * 4 * P * B) - 1) < 0.01
    assert t["kernels"]["k_gray_bgr8"]["launches_per_step"] == 2 and t["kernels"]["k_mineig_pair"]["launches_per_step"] == 1
    # compulsory bytes per stage and pair (SURVEY.md §8(d) with the fused kernels' real inputs/outputs)
    floor = {"gray": 2 * 4 * P, "pyr": 2 * (P + P // 4 + P // 16 + P // 64), "eig": P, "lk": 500 * 4 * (17 * 17 + 22 * 22) // 2}
    for s, per_pair in floor.items():
        assert t["stages"][s]["hbm_bytes_per_step"] >= 0.8 * per_pair * B, s


@pytest.mark.parametrize("cfg,P,corners,levels", [("c2", 640 * 480, 500, 3), ("c4", 3840 * 2160, 2000, 5)])
def test_traffic_summaries_of_the_other_configurations(cfg, P, corners, levels):
    """bench.py --config c2 / c4 take roofline.traffic from their own PMC passes (profiles/<tag>c2_*, <tag>c4_*): the same checks."""
    p = os.path.join(ROOT, "profiles", f"{TAG}{cfg}_traffic_pmc.json")
    if not os.path.exists(p):
        pytest.skip(f"{p} not committed yet")
    t = json.load(open(p))
    B = t["batch"]
    assert abs(t["stages"]["gray"]["hbm_bytes_per_step"] / (2 * 4 * P * B) - 1) < 0.01
    pyr = 2 * sum(P // 4 ** l for l in range(levels + 1))
    for s, per_pair in {"gray": 2 * 4 * P, "pyr": pyr, "eig": P}.items():
        assert t["stages"][s]["hbm_bytes_per_step"] >= 0.8 * per_pair * B, s
    v = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}{cfg}_valu_pmc.json")))
    assert v["batch"] == B and v["stages"]["lk"]["SQ_INSTS_VALU_per_launch"] > 1000 * corners * B / 4      # > 1 k wave-instructions per point


def test_committed_bench_lines_follow_the_contract():
    """The bench lines kept under profiles/ (configs[1], [2], [4]) carry the contract's keys, a roofline and - at N = 1 - a CPU baseline."""
    for name, tag in (("r04_bench_b512", ""), ("r04c2_bench", "c2"), ("r04c4_bench", "c4")):
        p = os.path.join(ROOT, "profiles", name + ".json")
        if not os.path.exists(p):
            pytest.skip(f"{p} not committed yet")
        d = json.load(open(p))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (name, k)
        assert d["vs_baseline"] is None and d["scaling"] == "weak" and "workload" in d["config"] and "BASELINE configs" in d["config"]["workload"]
        assert abs(d["value"] - d["n_gpus"] * d["config"]["pairs_per_gpu_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
        assert d["roofline"]["avg_ms"] <= d["ms_per_step"]
        ns = d["north_star_kernel"]
        assert ns["frac_fused_minimum"] <= ns["frac"] and (ns["frac_traffic"] is None or ns["frac_fused_minimum"] <= ns["frac_traffic"] * 1.05)


def test_fetch_factor_is_calibrated():
    c = load("fetch_calibration")
    for k, p in c["patterns"].items():
        assert abs(p["factor128"] - 2.0) < 0.1, (k, p["factor128"])          # the L2 asks for whole 128-B lines, tallied at 64 B
    t = load("traffic_pmc")
    for k, v in t["kernels"].items():
        assert v["fetch_factor"] == c["fetch_factor_of_kernel"].get(k, 2.0)


def test_bench_refuses_impossible_traffic(monkeypatch, capsys):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "pmc_traffic", lambda stage, pairs: 300)
    assert bench.checked_traffic("eig", 256, 625) is None and "refused" in capsys.readouterr().err
    assert bench.checked_traffic("eig", 256, 300) == 300


def test_bench_algorithmic_bytes_follow_survey_8d():
    """bench.py's per-stage byte counts are SURVEY.md 8(d)'s formulas: at 1080p / 3 levels / 500 points G_gray = 16.59 MB, G_pyr = 6.80 MB,
    G_lk ~ 1.6 MB per pair; the fused minimum of the pyramid pass is 2 (P0 + .. + P3) = 5.51 MB; every configuration the bench runs has
    a workload string that names its BASELINE entry."""
    import importlib.util
    from types import SimpleNamespace
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ab = bench.algorithmic_bytes(SimpleNamespace(max_level=3, win=15), 500, 46000, 1080, 1920)
    assert abs(ab["gray"] / 1e6 - 16.59) < 0.01 and abs(ab["pyr"] / 1e6 - 6.80) < 0.01 and abs(ab["pyr_fused_minimum"] / 1e6 - 5.51) < 0.01
    assert 1.5e6 < ab["lk"] < 1.7e6 and ab["eig"] == 1080 * 1920 + 8 * 46000
    for name, c in bench.CONFIGS.items():
        assert f"BASELINE configs[{name[1]}]" in c["workload"] and c["batch"] >= 128 and c["streams"] in (1, 2)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1920" in base["configs"][1] and "1024" in base["configs"][2] and "3840" in base["configs"][4]


def test_decoder_profile_and_the_bench_line_agree():
    """profiles/<tag>_jpeg_decoder.json (SQ counter passes over the decoder alone) is what bench.py reports as ingest_inclusive.decoder:
    the per-kernel times add up to the total, the committed bench line of configs[1] carries exactly these numbers, its double-buffered
    ingest figure names its batch count and a steady-state rate that is not below the ramp-inclusive one."""
    dec = load("jpeg_decoder")
    ks = {k: v["ms_per_512_frames"] for k, v in dec["kernels"].items() if k.startswith("k_jpeg")}
    assert abs(sum(ks.values()) - dec["decoder_ms_per_512_frames"]) < 2e-3 and dec["frames_per_decode"] == 512
    assert {"k_jpeg_sync", "k_jpeg_write", "k_jpeg_idct", "k_jpeg_dc"} <= set(ks) and "k_jpeg_color" not in ks and "k_jpeg_zero_upper" not in ks
    p = os.path.join(ROOT, "profiles", f"{TAG}_bench_b512.json")
    if not os.path.exists(p):
        pytest.skip("bench line not committed yet")
    ing = json.load(open(p))["ingest_inclusive"]
    assert ing["decoder"]["ms_per_512_frames"] == dec["decoder_ms_per_512_frames"] and ing["decoder"]["kernels_ms_per_512_frames"] == ks
    db = ing["jpeg_double_buffered"]
    assert db["batches"] >= 20 and db["steady_state"] >= db["value"] > ing["jpeg_decode_on_device"]["value"] > ing["raw_bgr_upload"]["value"]


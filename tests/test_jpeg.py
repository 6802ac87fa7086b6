"""Compressed-image ingest (reference: cv_bridge.compressed_imgmsg_to_cv2 = cv::imdecode = libjpeg,
velocity_measurment_node.py:112).  The golden fixture holds JPEG streams with the pixels libjpeg-turbo's default decompressor
returned for them (tests/golden/make_golden_jpeg.py); the oracle restatement and the device decoder must both reproduce them bit
for bit.  Larger streams are encoded with Pillow when it is importable (an encoder is not part of this repo) and checked against
the oracle."""
import io
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as jo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def names(gold):
    return [str(n) for n in gold["names"]]


# ------------------------------------------------------------------------------------------------ CPU: oracle, host parser
def test_oracle_reproduces_libjpeg_pixels(gold):
    for n in names(gold):
        got = jo.decode(gold[f"jpg_{n}"].tobytes())
        assert np.array_equal(got, gold[f"bgr_{n}"]), n


def test_oracle_refuses_progressive(gold):
    with pytest.raises(ValueError):
        jo.decode(gold["jpg_progressive"].tobytes())


def test_host_header_parser_matches_oracle(pkg, ofk, gold):
    """ofk_jpeg_info is host-only: it runs without a GPU."""
    for n in names(gold):
        data = gold[f"jpg_{n}"].tobytes()
        i = jo.info(data)
        assert ofk.jpeg_info(data) == (i["h"], i["w"], i["ncomp"])
        assert gold[f"bgr_{n}"].shape == (i["h"], i["w"], 3)
    for bad in (b"", b"\xff\xd8", b"not a jpeg at all", gold["jpg_progressive"].tobytes(), gold["jpg_c444"].tobytes()[:300]):
        with pytest.raises(ofk.OfkError):
            ofk.jpeg_info(bad)


def _scan_data(data):
    """(offset of the entropy-coded segment, restart interval) of a baseline stream: walk the marker segments up to SOS."""
    i, ri = 2, 0
    while True:
        assert data[i] == 0xFF
        m, L = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        if m == 0xDD:
            ri = (data[i + 4] << 8) | data[i + 5]
        if m == 0xDA:
            return i + 2 + L, ri
        i += 2 + L


def _destuff_reference(data):
    """What the decoders read: FF00 -> FF, RSTn (streams with a restart interval) taken out and the offsets behind them listed, the data
    ends at any other marker (T.81 B.1.1.5, F.1.2.3; libjpeg jdhuff.c fill_bit_buffer / process_restart)."""
    e0, ri = _scan_data(data)
    out, rst, i = bytearray(), [], e0
    while i < len(data):
        b = data[i]
        if b != 0xFF:
            out.append(b); i += 1
        elif i + 1 >= len(data):
            out.append(0xFF); break
        elif data[i + 1] == 0:
            out.append(0xFF); i += 2
        elif ri and 0xD0 <= data[i + 1] <= 0xD7:
            rst.append(len(out)); i += 2
        else:
            break
    return bytes(out), rst


def test_host_destuffing_matches_a_python_restatement(pkg, ofk, gold):
    """ofk_jpeg_destuff is the routine the staging of the compressed ingest runs (k_jpeg.hip jdestuff), host-only: every golden stream
    (4:4:4 / 4:2:2 / 4:2:0 / gray, odd sizes, restart intervals, quality 100 = dense stuffing) against a byte-wise restatement, plus
    streams with a foreign marker or surplus RSTn markers spliced into the entropy data."""
    seen_ff, seen_rst = 0, 0
    for n in names(gold):
        data = gold[f"jpg_{n}"].tobytes()
        want, want_rst = _destuff_reference(data)
        got, got_rst = ofk.jpeg_destuff(data)
        assert got == want and got_rst == want_rst, n
        e0, _ = _scan_data(data)
        seen_ff += data[e0:].count(b"\xff\x00"); seen_rst += len(want_rst)
        assert len(got) == len(data) - e0 - 2 - data[e0:-2].count(b"\xff\x00") - 2 * len(want_rst)      # nothing else is dropped (EOI ends the stream)
        # a foreign marker in the middle ends the data there; surplus restart markers are listed, not decoded
        mid = e0 + (len(data) - e0) // 2
        while data[mid - 1] == 0xFF or data[mid] == 0xFF:
            mid += 1
        cut = data[:mid] + b"\xff\xe1" + data[mid:]
        assert ofk.jpeg_destuff(cut) == _destuff_reference(cut) and len(ofk.jpeg_destuff(cut)[0]) < len(got)
        if want_rst:
            more = data[:-2] + b"\xff\xd3\xff\xd4" + data[-2:]
            assert ofk.jpeg_destuff(more) == _destuff_reference(more) and len(ofk.jpeg_destuff(more)[1]) == len(want_rst) + 2
    assert seen_ff > 50 and seen_rst > 10          # the fixtures exercise both
    with pytest.raises(ofk.OfkError):
        ofk.jpeg_destuff(gold["jpg_progressive"].tobytes())


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_device_decoder_reproduces_libjpeg_pixels(gpu_ctx, gold):
    for n in names(gold):
        data = gold[f"jpg_{n}"].tobytes()
        out = gpu_ctx.jpeg_decode([data])
        assert out.shape == (1,) + gold[f"bgr_{n}"].shape
        assert np.array_equal(out[0], gold[f"bgr_{n}"]), n
    # a batch: the same stream three times and twice more after a different one of the same geometry (tables differ per image)
    a, b = gold["jpg_c420_ros"].tobytes(), gold["jpg_c420_ros"].tobytes()
    out = gpu_ctx.jpeg_decode([a, b, a])
    for k in range(3):
        assert np.array_equal(out[k], gold["bgr_c420_ros"])


@pytest.mark.gpu
def test_device_decoder_errors(gpu_ctx, ofk, gold):
    with pytest.raises(ofk.OfkError):
        gpu_ctx.jpeg_decode([gold["jpg_progressive"].tobytes()])
    with pytest.raises(ofk.OfkError, match="differs"):
        gpu_ctx.jpeg_decode([gold["jpg_c420_ros"].tobytes(), gold["jpg_c420_odd"].tobytes()])
    data = gold["jpg_c420_ros"].tobytes()
    with pytest.raises(ofk.OfkError, match="truncated|corrupt"):
        gpu_ctx.jpeg_decode([data[: len(data) // 2] + b"\xff\xd9"])
    # still usable afterwards
    assert np.array_equal(gpu_ctx.jpeg_decode([data])[0], gold["bgr_c420_ros"])


@pytest.mark.gpu
def test_imdecode_facade(pkg, gold):
    from of_amd import cv2_hip as cv2
    img = cv2.imdecode(gold["jpg_c420_ros"], cv2.IMREAD_COLOR)
    assert np.array_equal(img, gold["bgr_c420_ros"])
    g = cv2.imdecode(gold["jpg_gray"], cv2.IMREAD_UNCHANGED)
    assert g.ndim == 2 and np.array_equal(g, gold["bgr_gray"][:, :, 0])
    assert cv2.imdecode(np.frombuffer(b"garbage", np.uint8), cv2.IMREAD_COLOR) is None


def _encode(img, quality, subsampling=None, **extra):
    Image = pytest.importorskip("PIL.Image")
    buf = io.BytesIO()
    kw = {} if subsampling is None else {"subsampling": subsampling}
    kw.update(extra)
    Image.fromarray(img).save(buf, "JPEG", quality=quality, **kw)
    return buf.getvalue()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,ss,q", [(1080, 1920, 2, 80), (1080, 1920, 2, 97), (721, 1283, 1, 60), (333, 517, 0, 90), (1080, 1920, None, 80)])
def test_device_decoder_vs_oracle_large(pkg, ofk, h, w, ss, q):
    """Thousands of decoder threads per image (1080p at quality 80 is ~1500 chunks): the self-synchronisation must end in the
    oracle's coefficients whatever the chunk boundaries hit."""
    from of_amd import synth
    frames = [synth.render_pair(h, w, 700 + k)["prev"] for k in range(3)]
    if ss is None:
        frames = [np.ascontiguousarray(f[:, :, 1]) for f in frames]
    streams = [_encode(f, q, ss) for f in frames]
    ctx = ofk.Context(0, w, h, 3, 64, 1)
    out = ctx.jpeg_decode(streams)
    for k in range(3):
        assert np.array_equal(out[k], jo.decode(streams[k])), k
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,ss,q,opts", [(240, 320, 0, 100, {}), (480, 640, 2, 98, {}), (360, 488, 1, 100, {"optimize": True}),
                                           (480, 640, 2, 100, {"restart_marker_blocks": 3}), (1080, 1920, 2, 85, {"optimize": True})])
def test_device_decoder_high_entropy(pkg, ofk, h, w, ss, q, opts):
    """White noise at the top qualities: every zigzag position of almost every block is non-zero (the write pass's direct path behind
    position 32 carries half of the coefficients), FF bytes and stuffed zeros are as frequent as they get (the byte-wise refill),
    codes run past the 9-bit look-ahead all the time (the long-code search), and `optimize` swaps the standard Huffman tables
    for ones fitted to the image."""
    rng = np.random.default_rng(h * 7 + w + q)
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(2)]
    from of_amd import synth
    frames.append(np.ascontiguousarray(synth.render_pair(h, w, 760)["prev"]))
    streams = [_encode(f, q, ss, **opts) for f in frames]
    ctx = ofk.Context(0, w, h, 3, 64, 1)
    out = ctx.jpeg_decode(streams)
    for k in range(3):
        assert np.array_equal(out[k], jo.decode(streams[k])), k
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,ss,opts", [(1080, 1920, 2, {"restart_marker_rows": 1}), (720, 1280, 2, {"restart_marker_blocks": 7}),
                                         (480, 640, 1, {"restart_marker_blocks": 1}), (1080, 1920, None, {"restart_marker_rows": 4})])
def test_device_decoder_restart_intervals(pkg, ofk, h, w, ss, opts):
    """DRI / RSTn streams: every decoder thread that runs into a marker is in step from there on; DC prediction restarts per
    interval (segmented scan).  Mixed with a stream without markers in one batch (the interval is a per-image property)."""
    from of_amd import synth
    frames = [synth.render_pair(h, w, 800 + k)["prev"] for k in range(3)]
    if ss is None:
        frames = [np.ascontiguousarray(f[:, :, 1]) for f in frames]
    streams = [_encode(frames[0], 80, ss, **opts), _encode(frames[1], 80, ss), _encode(frames[2], 92, ss, **opts)]
    ctx = ofk.Context(0, w, h, 3, 64, 1)
    out = ctx.jpeg_decode(streams)
    for k in range(3):
        assert np.array_equal(out[k], jo.decode(streams[k])), k
    ctx.close()


@pytest.mark.gpu
def test_pairs_upload_jpeg_feeds_the_pipeline(pkg, ofk):
    """Compressed ingest -> resident pairs -> the whole path: identical to uploading the frames the oracle decodes."""
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    B, h, w = 3, 480, 640
    pairs = [synth.render_pair(h, w, 40 + b) for b in range(B)]
    sp = [_encode(p["prev"], 90, 2) for p in pairs]
    sn = [_encode(p["next"], 90, 2) for p in pairs]
    sensors = ofk.make_sensors(B, scaling=pairs[0]["scaling"], cx=pairs[0]["cx"], cy=pairs[0]["cy"])
    cfg = PipelineConfig(max_corners=150, quality=0.03, min_distance=7)
    pipe = FlowPipeline(w, h, B, cfg)
    pipe.upload(np.stack([jo.decode(s) for s in sp]), np.stack([jo.decode(s) for s in sn]), sensors)
    ref = pipe.run()
    pipe.ctx.pairs_upload_jpeg(sp, sn)
    out = pipe.run()
    for k in ("counts", "prev_pts", "next_pts", "status", "records"):
        assert np.array_equal(out[k], ref[k]), k
    assert int(out["counts"].min()) > 20
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("streams_per_gpu", [1, 2])
def test_double_buffered_ingest_matches_the_one_shot_path(pkg, ofk, streams_per_gpu):
    """FlowPipeline.run_jpeg_batches (a helper thread stages batch k + 1 - marker parse, pinned staging, asynchronous H2D on the copy
    stream - while the main thread decodes batch k and queues its pairs run) returns, batch for batch, what upload_jpeg + run return;
    four batches through the two staging slots, the last one smaller than the others, with one and two free-running slices."""
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    h, w = 480, 640
    sizes = [3, 3, 3, 2]
    pairs = [[synth.render_pair(h, w, 60 + 10 * k + b) for b in range(n)] for k, n in enumerate(sizes)]
    batches = [([_encode(p["prev"], 88, 2) for p in ps], [_encode(p["next"], 88, 2) for p in ps]) for ps in pairs]
    p0 = pairs[0][0]
    cfg = PipelineConfig(max_corners=120, quality=0.03, min_distance=7)
    pipe = FlowPipeline(w, h, 3, cfg, streams=streams_per_gpu)
    sens = lambda k: ofk.make_sensors(sizes[k], scaling=p0["scaling"], cx=p0["cx"], cy=p0["cy"], omega=(0.001 * k, 0, 0))
    want = []
    for k, (sp, sn) in enumerate(batches):
        pipe.upload_jpeg(sp, sn, sens(k))
        want.append(pipe.run())
    got = []
    n = pipe.run_jpeg_batches(batches, sens, on_step=lambda k: got.append(pipe.ctx.pairs_download()))
    assert n == 4 and len(got) == 4
    for k in range(4):
        for key in ("counts", "prev_pts", "next_pts", "status", "records"):
            assert np.array_equal(got[k][key], want[k][key]), (k, key)
        assert got[k]["records"].shape[0] == sizes[k] and int(got[k]["counts"].min()) > 20
    # a slot is decoded once; staging an odd number of streams is not a batch of pairs
    with pytest.raises(ofk.OfkError):
        pipe.ctx.pairs_upload_staged(0, (6, h, w))
    st = pipe.ctx.jpeg_stage(1, batches[0][0])
    with pytest.raises(ofk.OfkError):
        pipe.ctx.pairs_upload_staged(1, st)
    pipe.close()


@pytest.mark.gpu
def test_staged_ingest_refuses_more_pairs_than_the_context_holds(pkg, ofk):
    """ofk_jpeg_stage accepts any count; ofk_pairs_upload_staged must refuse 2 * (max_batch + 1) streams - and an odd count - BEFORE a
    decoder kernel is queued: the colour pass would write past the resident frame sets.  The resident pairs of an earlier upload
    stay untouched (records of a re-run are the same bits) and the slot is spent."""
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    h, w, B = 240, 320, 2
    pairs = [synth.render_pair(h, w, 170 + b) for b in range(B + 1)]
    sp = [_encode(p["prev"], 90, 2) for p in pairs]; sn = [_encode(p["next"], 90, 2) for p in pairs]
    sensors = ofk.make_sensors(B, scaling=pairs[0]["scaling"], cx=pairs[0]["cx"], cy=pairs[0]["cy"])
    cfg = PipelineConfig(max_corners=60, quality=0.03, min_distance=7, max_level=2)
    for streams in (1, 2):                                                       # gray-direct path and the BGR path
        pipe = FlowPipeline(w, h, B, cfg, streams=streams)
        pipe.upload_jpeg(sp[:B], sn[:B], sensors)
        ref = pipe.run()
        ref_pyr = pipe.ctx.resident_pyramid(0, B - 1, h, w, 2)
        for bad in (sp + sn, sp[:B] + sn[:B - 1]):                               # B + 1 pairs; an odd count
            st = pipe.ctx.jpeg_stage(1, bad)
            with pytest.raises(ofk.OfkError) as e:
                pipe.ctx.pairs_upload_staged(1, st)
            assert e.value.code == ofk.E_INVALID
            with pytest.raises(ofk.OfkError):                                    # the slot is spent
                pipe.ctx.pairs_upload_staged(1, st)
        pipe.upload_jpeg(sp[:B], sn[:B], sensors)
        out = pipe.run()
        for k in ("counts", "prev_pts", "next_pts", "status", "records"):
            assert np.array_equal(out[k], ref[k]), k
        assert all(np.array_equal(a, b) for a, b in zip(pipe.ctx.resident_pyramid(0, B - 1, h, w, 2), ref_pyr))
        pipe.close()


@pytest.mark.gpu
def test_staging_errors_have_their_own_message(pkg, ofk, gold):
    """ofk_jpeg_stage may run on a helper thread beside the owner: its errors go to the SLOT's message (ofk_jpeg_stage_error), the
    context's message (ofk_last_error) stays the owner's."""
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    good = gold["jpg_c420_ros"].tobytes()
    pipe = FlowPipeline(640, 480, 2, PipelineConfig(max_corners=20))
    with pytest.raises(ofk.OfkError) as e:
        pipe.ctx.pairs_run(pipe._params)                         # the owner's error: nothing resident
    owner_msg = pipe.ctx._L.ofk_last_error(pipe.ctx._h)
    assert b"no resident frame pairs" in owner_msg
    with pytest.raises(ofk.OfkError) as e:
        pipe.ctx.jpeg_stage(1, [good, gold["jpg_c420_odd"].tobytes(), good, good])           # stream 1 has another geometry
    assert "stream 1" in str(e.value)
    assert pipe.ctx._L.ofk_last_error(pipe.ctx._h) == owner_msg  # untouched by the staging failure
    st = pipe.ctx.jpeg_stage(1, [good] * 4)                      # a success clears the slot's message
    assert pipe.ctx._L.ofk_jpeg_stage_error(pipe.ctx._h, 1) == b""
    pipe.close()


@pytest.mark.gpu
def test_gray_direct_ingest_state_machine(pkg, ofk):
    """The compressed ingest writes the gray frames straight into one pyramid set and no BGR frame (ofk_pairs_upload_staged on the
    default schedule): a second run without a new upload, a run after the schedule changed to two slices, odd frame sizes (the byte
    store path of the colour kernel), the resident gray level itself, and a raw upload afterwards must all behave as if the decoded
    frames had been uploaded."""
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    for (h, w) in ((480, 640), (250, 332)):
        B = 3
        pairs = [synth.render_pair(h, w, 140 + b) for b in range(B)]
        sp = [_encode(p["prev"], 90, 2) for p in pairs]
        sn = [_encode(p["next"], 90, 2) for p in pairs]
        dp, dn = np.stack([jo.decode(s) for s in sp]), np.stack([jo.decode(s) for s in sn])
        sensors = ofk.make_sensors(B, scaling=pairs[0]["scaling"], cx=pairs[0]["cx"], cy=pairs[0]["cy"])
        cfg = PipelineConfig(max_corners=100, quality=0.03, min_distance=7, max_level=2)
        pipe = FlowPipeline(w, h, B, cfg)
        pipe.upload(dp, dn, sensors)
        ref = pipe.run()
        ref_pyr = pipe.ctx.resident_pyramid(0, 1, h, w, 2)
        keys = ("counts", "prev_pts", "next_pts", "status", "records")
        same = lambda out, what: [np.testing.assert_array_equal(out[k], ref[k], err_msg=f"{what}: {k} ({h}x{w})") for k in keys]
        pipe.ctx.pairs_upload_jpeg(sp, sn)
        same(pipe.run(), "first run")
        got_pyr = pipe.ctx.resident_pyramid(0, 1, h, w, 2)                       # level 0 = the gray frame the decoder wrote
        assert all(np.array_equal(a, b) for a, b in zip(got_pyr, ref_pyr))
        same(pipe.run(), "second run on the same upload")
        same(pipe.run(), "third run")
        pipe.ctx.pairs_upload_jpeg(sp, sn)
        pipe.ctx.set_streams(2)
        same(pipe.run(), "two slices after a one-slice ingest")
        pipe.ctx.pairs_upload_jpeg(sp, sn)                                      # two slices: the BGR way
        same(pipe.run(), "ingest under two slices")
        pipe.ctx.set_streams(1)
        pipe.ctx.pairs_upload_jpeg(sp[::-1], sn[::-1])                          # other frames in between
        pipe.run()
        pipe.upload(dp, dn, sensors)
        same(pipe.run(), "raw upload after a gray-direct ingest")
        pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ss,h,w", [(0, 250, 332), (1, 480, 640), (1, 123, 201), (None, 250, 332), (2, 123, 201)])
def test_gray_direct_ingest_other_samplings(pkg, ofk, ss, h, w):
    """The fused luma-IDCT / colour kernel writes gray for every sampling the decoder accepts - 4:4:4 (tile of 8 rows x 256 pixels),
    4:2:2, 4:2:0 (16 x 128), gray streams (no chroma pass at all) - and for widths that are no multiple of the tile or of eight
    pixels: the resident gray level and the pipeline's results equal those of the decoded frames uploaded raw."""
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    B = 2
    pairs = [synth.render_pair(h, w, 240 + b) for b in range(B)]
    as_stream = (lambda img: _encode(np.ascontiguousarray(img[:, :, 1]), 85)) if ss is None else (lambda img: _encode(img, 85, ss))
    sp, sn = [as_stream(p["prev"]) for p in pairs], [as_stream(p["next"]) for p in pairs]
    dp, dn = np.stack([jo.decode(s) for s in sp]), np.stack([jo.decode(s) for s in sn])
    sensors = ofk.make_sensors(B, scaling=pairs[0]["scaling"], cx=pairs[0]["cx"], cy=pairs[0]["cy"])
    pipe = FlowPipeline(w, h, B, PipelineConfig(max_corners=60, quality=0.03, min_distance=7, max_level=2))
    try:
        pipe.upload(dp, dn, sensors)
        ref = pipe.run()
        ref_pyr = pipe.ctx.resident_pyramid(0, 1, h, w, 2)
        pipe.ctx.pairs_upload_jpeg(sp, sn)
        out = pipe.run()
        for k in ("counts", "prev_pts", "next_pts", "status", "records"):
            assert np.array_equal(out[k], ref[k]), k
        assert all(np.array_equal(a, b) for a, b in zip(pipe.ctx.resident_pyramid(0, 1, h, w, 2), ref_pyr))
    finally:
        pipe.close()


@pytest.mark.gpu
def test_decoder_chunk_size_does_not_change_pixels(pkg, ofk, gold):
    """ofk_set_tuning("jpeg_chunk"): 64 ... 1024 entropy bytes per decoder thread - other chunk boundaries, other synchronisation
    histories, the same pixels.  "jpeg_sub": fewer second-level Huffman tables than the streams' tables need - the codes left over
    take the canonical search (k_jpeg.hip jslow), the same pixels."""
    from of_amd import synth
    streams = [_encode(synth.render_pair(480, 640, 300 + k)["prev"], q, 2, **o) for k, (q, o) in enumerate(((85, {}), (97, {"optimize": True}), (60, {"restart_marker_rows": 2})))]
    want = [jo.decode(s) for s in streams]
    ctx = ofk.Context(0, 640, 480, 3, 64, 1)
    try:
        for chunk in (64, 128, 256, 512, 1024, 0):
            ofk.set_tuning("jpeg_chunk", chunk)
            out = ctx.jpeg_decode(streams)
            for k in range(3):
                assert np.array_equal(out[k], want[k]), (chunk, k)
        for sub in (1, 3, 7, 13):
            ofk.set_tuning("jpeg_sub", sub)
            out = ctx.jpeg_decode(streams)
            for k in range(3):
                assert np.array_equal(out[k], want[k]), ("jpeg_sub", sub, k)
    finally:
        ofk.set_tuning("jpeg_chunk", 0)
        ofk.set_tuning("jpeg_sub", 0)
        ctx.close()


@pytest.mark.gpu
def test_markers_inside_the_entropy_data(pkg, ofk):
    """The host takes byte stuffing and restart markers out of the entropy segment while it stages it (k_jpeg.hip jdestuff): a marker
    that is neither ends the data there (libjpeg stops reading at it too), and more RSTn markers than the frame has restart
    intervals are refused before anything is queued."""
    from of_amd import synth
    img = synth.render_pair(240, 320, 77)["prev"]
    plain = _encode(img, 80, 2)
    rst = _encode(img, 80, 2, restart_marker_rows=1)
    ctx = ofk.Context(0, 320, 240, 2, 64, 1)
    try:
        assert np.array_equal(ctx.jpeg_decode([rst, plain])[0], jo.decode(rst))
        sos = plain.index(b"\xff\xda")
        mid = sos + (len(plain) - sos) // 2
        while plain[mid - 1] == 0xFF or plain[mid] == 0xFF:      # not inside an FF00 pair
            mid += 1
        with pytest.raises(ofk.OfkError, match="truncated|corrupt"):
            ctx.jpeg_decode([plain[:mid] + b"\xff\xe0" + plain[mid:]])
        extra = rst[:-2] + b"\xff\xd0" * 40 + rst[-2:]            # 15 intervals, 14 markers + 40
        with pytest.raises(ofk.OfkError, match="more restart markers"):
            ctx.jpeg_decode([extra])
        assert np.array_equal(ctx.jpeg_decode([plain])[0], jo.decode(plain))       # still usable
    finally:
        ctx.close()

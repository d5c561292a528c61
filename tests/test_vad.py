"""src/vad.rs: column classification of mel images (vad_boundaries / vad_on / VoiceActivityDetector).

The reference's own test (test_speech_detection, src/vad.rs:620-668) is a known-answer test over twelve TGA
fixtures: seven without speech must not trigger vad_on, five with speech must.  The fixtures are copied to
tests/golden/vad/; the oracle, the kernels' per-thread functions on the host, and the HIP kernels all have to
reproduce it, and the masks of the latter two must be identical to the oracle's (decisions, not floats)."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF_SETTINGS = dict(min_energy=1.0, min_y=10, min_x=10, min_mel=0)      # src/vad.rs:624-629


def fixtures(oracle):
    for kind, want in (("blank", False), ("speech", True)):
        for f in sorted(glob.glob(os.path.join(GOLDEN, "vad", kind, "*.tga"))):
            with open(f, "rb") as fh:
                img = oracle.parse_tga_8bit(fh.read()).reshape(80, -1)          # load_tga_8bit + to_array2
            yield os.path.basename(f), img, want


@pytest.fixture(scope="module")
def vemu():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", d, "-s"])
    L = C.CDLL(os.path.join(d, "libmelspec_emu.so"))
    L.emu_vad_boundaries.restype = C.c_longlong
    L.emu_vad_boundaries.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]

    def run(img, min_energy=0.98, min_y=11, min_x=5, min_mel=2):
        x = np.ascontiguousarray(img, np.float32)
        raw = np.zeros(max(1, x.shape[1]), np.uint8); sm = np.zeros_like(raw)
        n = L.emu_vad_boundaries(x.ctypes.data, x.shape[0], x.shape[1], min_mel, min_y, float(min_energy), raw.ctypes.data, sm.ctypes.data)
        return raw[:n].astype(bool), sm[:n].astype(bool)
    return run


def test_oracle_reproduces_the_reference_known_answers(oracle):
    seen = 0
    for name, img, want in fixtures(oracle):
        raw, sm = oracle.vad_boundaries(img, **REF_SETTINGS)
        assert raw.shape == sm.shape == (img.shape[1] - 2,)
        assert oracle.vad_on(sm, 10) == want, name
        seen += 1
    assert seen == 12


def test_oracle_edge_cases(oracle):
    z = np.zeros((80, 50), np.float32)
    raw, sm = oracle.vad_boundaries(z)
    assert not raw.any() and not sm.any() and not oracle.vad_on(sm, 1)
    assert oracle.vad_boundaries(np.zeros((2, 50), np.float32))[0].shape == (0,)       # height < 3
    assert oracle.vad_boundaries(np.zeros((80, 2), np.float32))[0].shape == (0,)       # width < 3
    raw, sm = oracle.vad_boundaries(z, min_y=0)                                         # min_y == 0: everything active
    assert raw.all() and sm.all() and oracle.vad_on(sm, 48) and not oracle.vad_on(sm, 49)
    # vad_on tests its counter from the second intersected column on (src/vad.rs:241-250)
    assert oracle.vad_on(np.array([0, 1, 0, 0, 1, 0], bool), 1) and not oracle.vad_on(np.array([0, 1, 0, 0, 0, 0], bool), 1)
    assert not oracle.vad_on(np.array([0, 1, 0, 0, 1, 0], bool), 2) and oracle.vad_on(np.array([0, 1, 1, 0, 1, 0], bool), 2)
    # min_mel beyond the image: no rows to count
    step = np.zeros((80, 20), np.float32); step[:, 10:] = 5.0
    assert oracle.vad_boundaries(step, min_energy=1.0, min_y=3, min_mel=0)[0].any()
    assert not oracle.vad_boundaries(step, min_energy=1.0, min_y=3, min_mel=200)[0].any()


def test_emulated_kernels_match_oracle_masks(vemu, oracle):
    for name, img, want in fixtures(oracle):
        for kw in (REF_SETTINGS, dict(), dict(min_energy=0.5, min_y=3, min_x=5, min_mel=7)):
            r0, s0 = oracle.vad_boundaries(img, **kw)
            r1, s1 = vemu(img, **kw)
            assert np.array_equal(r0, r1) and np.array_equal(s0, s1), (name, kw)
    rng = np.random.default_rng(5)
    for shape in ((3, 3), (3, 9), (80, 3), (128, 700), (5, 1000)):
        img = rng.standard_normal(shape).astype(np.float32)
        for kw in (dict(min_energy=2.0, min_y=1, min_mel=0), dict(min_energy=1.0, min_y=2, min_mel=1)):
            r0, s0 = oracle.vad_boundaries(img, **kw)
            r1, s1 = vemu(img, **kw)
            assert np.array_equal(r0, r1) and np.array_equal(s0, s1), (shape, kw)


# ---- GPU ---------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_reproduces_the_reference_known_answers(gpu, oracle):
    st = gpu.DetectionSettings(**REF_SETTINGS)
    for name, img, want in fixtures(oracle):
        e = gpu.vad_boundaries([img], st)
        raw, sm = oracle.vad_boundaries(img, **REF_SETTINGS)
        assert np.array_equal(e.raw, raw) and np.array_equal(e.smoothed, sm), name
        assert gpu.vad_on(e, st.min_x) == want, name
        assert e.longest_run == max([0] + [len(r) for r in "".join("1" if b else "0" for b in sm).split("0")])
        assert sorted(e.intersected() + e.non_intersected()) == list(range(img.shape[1] - 2))


@pytest.mark.gpu
def test_gpu_masks_on_random_images_and_edges(gpu, oracle):
    rng = np.random.default_rng(5)
    for shape in ((3, 3), (3, 9), (80, 3), (128, 700), (5, 1000), (80, 3000)):
        img = rng.standard_normal(shape).astype(np.float32)
        for kw in (dict(min_energy=2.0, min_y=1, min_mel=0), dict(min_energy=1.0, min_y=2, min_mel=1), dict(min_y=0)):
            e = gpu.vad_boundaries([img[:, :shape[1] // 2], img[:, shape[1] // 2:]], gpu.DetectionSettings(**kw))   # frames are concatenated
            raw, sm = oracle.vad_boundaries(img, **{"min_energy": 0.98, "min_y": 11, "min_mel": 2, **kw})
            assert np.array_equal(e.raw, raw) and np.array_equal(e.smoothed, sm), (shape, kw)
            for n in (1, 2, 5, 40):
                assert gpu.vad_on(e, n) == oracle.vad_on(sm, n)
    assert gpu.vad_boundaries([np.zeros((2, 50), np.float32)], gpu.DetectionSettings()).smoothed.shape == (0,)
    assert gpu.vad_boundaries([], gpu.DetectionSettings()).smoothed.shape == (0,)


@pytest.mark.gpu
def test_gpu_mel_to_vad_and_streaming_detector(gpu, oracle, jfk):
    """PCM -> fused mel kernel -> VAD masks: speech in jfk_f32le.wav is found, a silent clip is not; the
    frame-by-frame detector agrees with vad_boundaries on its own windows (src/vad.rs:172-186)."""
    m = gpu.HipMelSpectrogram(400, 160, 16000.0, 80)
    mel = m.compute_mel_spectrogram(jfk)
    img = np.ascontiguousarray(mel.T)
    st = gpu.DetectionSettings(**REF_SETTINGS)
    e = gpu.vad_boundaries([img], st)
    want = oracle.vad_boundaries(oracle.compute_mel_spectrogram_cpu(jfk, 400, 160, 80).T, **REF_SETTINGS)[1]
    assert (e.smoothed != want).mean() < 0.01 and gpu.vad_on(e, 10)      # f32 mel vs f64 mel may flip a column at a threshold
    quiet = m.compute_mel_spectrogram(oracle.synth_pcm(0, 32000) * np.float32(1e-4))
    assert not gpu.vad_on(gpu.vad_boundaries([np.ascontiguousarray(quiet.T)], st), 10)
    det = gpu.VoiceActivityDetector(gpu.DetectionSettings(min_energy=1.0, min_y=10, min_x=5, min_mel=0))
    decisions = [det.add_activity(img[:, t:t + 1]) for t in range(300, 420)]
    assert all(d is None for d in decisions[:4]) and all(d is not None for d in decisions[4:])
    for k, d in enumerate(decisions[4:], start=4):
        t = 300 + k
        w = oracle.vad_boundaries(img[:, t - 4:t + 1], min_energy=1.0, min_y=10, min_x=5, min_mel=0)[1]
        inter = np.nonzero(w)[0]
        assert d.frame_index == k and d.window_columns == 3 and d.active_columns == len(inter)
        assert d.active == (len(inter) > 0 and inter[0] == 0)
    m.close()


# ---- the detector stage of the streaming bank (melspec_stream_enable_vad): VoiceActivityDetector::add_activity per stream ----------
def _emu_stream_vad():
    L = C.CDLL(os.path.join(os.path.dirname(__file__), "emu", "libmelspec_emu.so"))
    L.emu_stream_vad_push.restype = None
    L.emu_stream_vad_push.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def _mel_like(rng, frames, n_mels):
    """rows in Whisper's output range with speech-like structure: bursts of harmonic ridges over a noisy floor"""
    x = rng.uniform(-1.0, -0.97, (frames, n_mels)).astype(np.float32)
    t = 0
    while t < frames:
        d = int(rng.integers(3, 40))
        if rng.random() < 0.5:
            for m in rng.integers(0, n_mels, 6):
                x[t:t + d, m:m + 2] += np.float32(rng.uniform(0.8, 2.0))
        t += d
    return x


@pytest.mark.parametrize("kw", [dict(min_energy=1.0, min_y=3, min_x=5, min_mel=0), dict(min_energy=0.98, min_y=11, min_x=5, min_mel=2),
                                dict(min_energy=0.7, min_y=2, min_x=12, min_mel=1), dict(min_energy=0.5, min_y=1, min_x=66, min_mel=0),
                                dict(min_energy=1.0, min_y=0, min_x=7, min_mel=0), dict(min_energy=1.0, min_y=3, min_x=2, min_mel=0),
                                dict(min_energy=1.0, min_y=3, min_x=3, min_mel=0), dict(min_energy=1.0, min_y=2, min_x=1, min_mel=0)])
def test_stream_detector_steps_match_the_oracle(oracle, kw):
    """The steps of stream_vad_kernel in their host form (the kernel spreads the Sobel walk over a wave's lanes; same sums), run over pushes of irregular size (1 frame, several, more than the
    history), give for every frame what the reference's detector gives when fed the same rows one at a time."""
    L = _emu_stream_vad()
    rng = np.random.default_rng(11 + kw["min_x"])
    for n_mels in (80, 5, 2):
        rows = _mel_like(rng, 400, n_mels)
        want = oracle.voice_activity_stream(rows, **kw)
        state = np.zeros(2, np.uint64)
        prev = np.zeros((2, n_mels), np.float32)
        got, t = [], 0
        while t < rows.shape[0]:
            f = int(min(rows.shape[0] - t, rng.choice([1, 1, 2, 3, 7, 70, 130])))
            chunk = np.ascontiguousarray(rows[t:t + f])
            acts = np.zeros(f, np.dtype([("valid", "u1"), ("active", "u1"), ("lead", "<u2"), ("n", "<u2"), ("w", "<u2")]))
            L.emu_stream_vad_push(chunk.ctypes.data, f, n_mels, kw["min_mel"], kw["min_y"], kw["min_x"], float(kw["min_energy"]),
                                  state.ctypes.data, prev.ctypes.data, acts.ctypes.data)
            got += [None if not a["valid"] else (bool(a["active"]), t + k, int(a["lead"]), int(a["n"]), int(a["w"])) for k, a in enumerate(acts)]
            t += f
        assert int(state[0]) == rows.shape[0]
        assert got == want, (kw, n_mels, next(i for i, (a, b) in enumerate(zip(got, want)) if a != b))
        if kw["min_x"] >= 3 and n_mels == 80 and kw["min_y"] > 0 and kw["min_x"] < 60:
            assert any(w is not None and w[0] for w in want) and any(w is not None and not w[0] for w in want)      # both answers occur


@pytest.mark.gpu
def test_gpu_stream_bank_detectors(gpu, oracle, jfk):
    """melspec_stream_enable_vad: every push of the bank also feeds its streams' detectors on the device.  Three streams (speech,
    the same speech later, near-silence) pushed in chunks of irregular size: the rows equal the bank's rows without the stage, and
    every frame's record equals what the reference's detector returns when it is fed those rows one by one."""
    kw = dict(min_energy=1.0, min_y=10, min_x=5, min_mel=0)
    m = gpu.HipMelSpectrogram(400, 160, 16000.0, 80)
    bank = gpu.StreamBank(m, 3, 8000)
    plain = gpu.StreamBank(m, 3, 8000)
    bank.enable_vad(gpu.DetectionSettings(**kw))
    src = [jfk[:60000], jfk[40000:120000], oracle.synth_pcm(1, 50000) * np.float32(1e-4)]
    pos, rows, acts = [0, 0, 0], [[], [], []], [[], [], []]
    rng = np.random.default_rng(3)
    while any(p < len(s) for p, s in zip(pos, src)):
        ids = [i for i in range(3) if pos[i] < len(src[i]) and rng.random() < 0.8]
        if not ids:
            continue
        chunks = []
        for i in ids:
            n = int(min(len(src[i]) - pos[i], rng.choice([37, 160, 161, 800, 4000, 8000])))
            chunks.append(src[i][pos[i]:pos[i] + n]); pos[i] += n
        first = [bank.vad_frames(i) for i in ids]
        r, a = bank.push_vad(ids, chunks)
        r0 = plain.push(ids, chunks)
        for i, ri, ai, r0i, f0 in zip(ids, r, a, r0, first):
            assert np.array_equal(ri, r0i) and len(ai) == ri.shape[0]
            assert all(x is None or x.frame_index == f0 + k for k, x in enumerate(ai))
            rows[i].append(ri); acts[i] += ai
    r, a = bank.flush_vad([0, 1, 2])                       # the zero-padded last frame feeds the detector too
    for i in range(3):
        rows[i].append(r[i]); acts[i] += a[i]
    seen_active = False
    for i in range(3):
        x = np.concatenate(rows[i])
        want = oracle.voice_activity_stream(x, **kw)
        got = [None if v is None else (v.active, v.frame_index, v.leading_active_columns, v.active_columns, v.window_columns) for v in acts[i]]
        assert bank.vad_frames(i) == x.shape[0] == len(got)
        assert got == want, (i, next(k for k, (p, q) in enumerate(zip(got, want)) if p != q))
        assert all(v is None or v.confidence == (v.active_columns / v.window_columns if v.window_columns else 0.0) for v in acts[i])
        if i < 2:
            seen_active |= any(w is not None and w[0] for w in want)
        else:
            assert not any(w is not None and w[0] for w in want)      # the silent stream never fires
    assert seen_active
    # reset starts the listed detectors afresh; the stage can be turned off again
    bank.reset([1])
    assert bank.vad_frames(1) == 0 and bank.vad_frames(0) > 0
    r, a = bank.push_vad([1], [jfk[:4000]])
    assert [v for v in a[0][:4]] == [None] * 4 and a[0][4] is not None and a[0][4].frame_index == 4
    with pytest.raises(Exception):
        bank.push_stft([0], [jfk[:800]])                   # the stage needs mel rows
    bank.enable_vad(None)
    with pytest.raises(Exception):
        bank.push_vad([0], [jfk[:800]])
    bank.push([0], [jfk[:800]])
    bank.close(); plain.close(); m.close()


@pytest.mark.gpu
def test_gpu_stream_detectors_device_push_many_streams(gpu, oracle):
    """The device-producer form at scale: 1024 streams, one hop per push written straight into the slots, rows and records to device
    memory; per stream the records equal the reference's detector over the rows the pushes emitted."""
    from mel_spec_amd.stream import ACTIVITY_DTYPE
    kw = dict(min_energy=0.8, min_y=10, min_x=6, min_mel=1)
    n_streams, hop, pushes = 1024, 160, 42
    m = gpu.HipMelSpectrogram(400, hop, 16000.0, 80)
    bank = gpu.StreamBank(m, n_streams, hop)
    bank.enable_vad(gpu.DetectionSettings(**kw))
    ids = np.arange(n_streams, dtype=np.uint32)
    out = gpu.DeviceBuffer(n_streams * 80 * 4)
    acts = gpu.DeviceBuffer(n_streams * 8)
    p0 = bank.input_ptr(0)
    slot = bank.input_ptr(1) - p0
    rows, recs = [], []
    for k in range(pushes):
        gpu.synth_pcm_window(p0, slot // 4, hop, k * hop, n_streams)
        fr = bank.push_device_vad(ids, np.full(n_streams, hop, np.uint32), out.ptr, acts.ptr)
        assert (fr == (1 if k >= 2 else 0)).all()
        if k >= 2:
            rows.append(out.download((n_streams, 80)))
            recs.append(acts.download((n_streams,), ACTIVITY_DTYPE))
    fired = quiet = 0
    for s in list(range(0, n_streams, 37)) + [n_streams - 1]:
        x = np.stack([r[s] for r in rows])
        want = oracle.voice_activity_stream(x, **kw)
        got = [None if not a[s]["valid"] else (bool(a[s]["active"]), k, int(a[s]["leading_active_columns"]), int(a[s]["active_columns"]),
                                               int(a[s]["window_columns"])) for k, a in enumerate(recs)]
        assert got == want, s
        assert bank.vad_frames(s) == len(rows)
        fired += sum(1 for w in want if w is not None and w[0])
        quiet += sum(1 for w in want if w is not None and not w[0])
    assert fired > 0 and quiet > 0
    out.free(); acts.free(); bank.close(); m.close()


def test_reference_stage_test_and_timing_helpers(oracle):
    """test_stage (src/vad.rs:731-758): quantized_mel_golden.tga fed to a detector one column at a time, settings (1.0, 3, 3, 0) -- here
    through the host form of the bank's detector stage against the restated add_activity; plus VadFrameTiming / the duration helpers
    (src/vad.rs:97-117, 579-601) on hand-checked values."""
    import mel_spec_amd as M
    with open(os.path.join(GOLDEN, "quantized_mel_golden.tga"), "rb") as f:
        img = oracle.parse_tga_8bit(f.read()).reshape(80, -1)
    rows = np.ascontiguousarray(img.T)
    kw = dict(min_energy=1.0, min_y=3, min_x=3, min_mel=0)
    want = oracle.voice_activity_stream(rows, **kw)
    L = _emu_stream_vad()
    state, prev = np.zeros(2, np.uint64), np.zeros((2, 80), np.float32)
    got = []
    for t in range(rows.shape[0]):                                      # chunk_size = 1, as the reference's test does
        acts = np.zeros(1, np.dtype([("valid", "u1"), ("active", "u1"), ("lead", "<u2"), ("n", "<u2"), ("w", "<u2")]))
        L.emu_stream_vad_push(rows[t:t + 1].ctypes.data, 1, 80, kw["min_mel"], kw["min_y"], kw["min_x"], kw["min_energy"],
                              state.ctypes.data, prev.ctypes.data, acts.ctypes.data)
        a = acts[0]
        got.append(None if not a["valid"] else (bool(a["active"]), t, int(a["lead"]), int(a["n"]), int(a["w"])))
    assert got == want and got[0] is None and got[1] is None and got[2] is not None and got[2][4] == 1
    assert any(g is not None and g[0] for g in got) and any(g is not None and not g[0] for g in got)
    t = M.VadFrameTiming(400, 160, 16000.0)
    ts = t.timestamps_for_frame(3)                                      # samples 480 / 680 / 880 -> 30 / 42.5 -> 43 / 55 ms
    assert (ts.start_ms, ts.center_ms, ts.end_ms) == (30, 43, 55)
    assert M.VadFrameTiming(512, 160, 16000.0).timestamps_for_frame(0).center_ms == 16
    assert M.n_frames_for_duration(160, 16000.0, 1000) == 100 and M.n_frames_for_duration(160, 16000.0, 1001) == 101
    assert M.duration_ms_for_n_frames(160, 16000.0, 100) == 1000 and M.duration_ms_for_n_frames(160, 16000.0, 3) == 30
    assert M.format_milliseconds(3723004) == "01:02:03.004" and M.format_milliseconds(0) == "00:00:00.000"
    det = M.VoiceActivityDetector.new_with_timing(M.DetectionSettings(**kw), t)
    assert det.timing is t and det.frame_index == 0


@pytest.mark.gpu
def test_gpu_stream_detector_long_pushes(gpu, oracle, jfk):
    """Pushes that emit more frames than the detector kernel keeps in LDS at a time (2048): the history is carried from piece to piece."""
    kw = dict(min_energy=1.0, min_y=8, min_x=9, min_mel=0)
    m = gpu.HipMelSpectrogram(400, 160, 16000.0, 80)
    bank = gpu.StreamBank(m, 2, 400000)
    bank.enable_vad(gpu.DetectionSettings(**kw))
    x = np.tile(jfk, 3)[:520000]
    rows, acts = [], []
    for lo, hi in ((0, 390000), (390000, 390100), (390100, 520000)):
        r, a = bank.push_vad([1], [x[lo:hi]])
        rows.append(r[0]); acts += a[0]
    rows = np.concatenate(rows)
    assert rows.shape[0] > 3000
    want = oracle.voice_activity_stream(rows, **kw)
    got = [None if v is None else (v.active, v.frame_index, v.leading_active_columns, v.active_columns, v.window_columns) for v in acts]
    assert got == want
    assert any(w is not None and w[0] for w in want) and any(w is not None and not w[0] for w in want)
    bank.close(); m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_mels,kw", [(8, dict(min_energy=0.3, min_y=2, min_x=3, min_mel=0)), (80, dict(min_energy=1.0, min_y=0, min_x=7, min_mel=0)),
                                       (80, dict(min_energy=0.6, min_y=3, min_x=66, min_mel=1)), (80, dict(min_energy=0.5, min_y=1, min_x=5, min_mel=500)),
                                       (128, dict(min_energy=0.98, min_y=11, min_x=5, min_mel=2)), (80, dict(min_energy=1.0, min_y=3, min_x=2, min_mel=0)),
                                       (5, dict(min_energy=0.2, min_y=1, min_x=4, min_mel=0))])
def test_gpu_stream_detector_settings_and_banks(gpu, oracle, jfk, n_mels, kw):
    """The detector stage at the edges of its settings (min_y = 0: every column intersected; min_x = 66: the whole 64-bit history;
    min_x = 2: windows too narrow for the stencil; min_mel beyond the image) and on other banks (8 / 128 mels on the fused kernels,
    5 mels on the generic one), against the restated add_activity."""
    m = gpu.HipMelSpectrogram(400, 160, 16000.0, n_mels)
    bank = gpu.StreamBank(m, 1, 16000)
    bank.enable_vad(gpu.DetectionSettings(**kw))
    rows, acts = [], []
    rng = np.random.default_rng(n_mels + kw["min_x"])
    pos = 0
    x = jfk[:120000]
    while pos < len(x):
        n = int(min(len(x) - pos, rng.choice([160, 320, 1000, 7000, 16000])))
        r, a = bank.push_vad([0], [x[pos:pos + n]])
        rows.append(r[0]); acts += a[0]; pos += n
    rows = np.concatenate(rows)
    want = oracle.voice_activity_stream(rows, **kw)
    got = [None if v is None else (v.active, v.frame_index, v.leading_active_columns, v.active_columns, v.window_columns) for v in acts]
    assert got == want, next(k for k, (p, q) in enumerate(zip(got, want)) if p != q)
    bank.close(); m.close()


@pytest.mark.gpu
def test_gpu_stream_detector_argument_errors(gpu, jfk):
    """Error behaviour of the detector stage: settings the 64-bit history cannot serve, negative settings, calls for records while the
    stage is off, a records buffer that is too small or missing -- each a library error, none a crash, and the bank stays usable."""
    import ctypes as C2
    from mel_spec_amd import _lib
    from mel_spec_amd.stream import ACTIVITY_DTYPE
    m = gpu.HipMelSpectrogram(400, 160, 16000.0, 80)
    bank = gpu.StreamBank(m, 2, 4000)
    with pytest.raises(gpu.HipError):
        bank.enable_vad(gpu.DetectionSettings(min_x=67))
    with pytest.raises(gpu.HipError):
        bank.enable_vad(gpu.DetectionSettings(min_y=-1))
    with pytest.raises(gpu.HipError):
        bank.push_vad([0], [jfk[:1600]])                              # the stage is off
    bank.enable_vad(gpu.DetectionSettings(min_x=66))                   # the largest window there is
    L = _lib.lib()
    ids = np.array([0], np.uint32); lens = np.array([1600], np.uint32)
    out = np.zeros((10, 80), np.float32); frames = np.zeros(1, np.uint32); acts = np.zeros(1, ACTIVITY_DTYPE)
    x = np.ascontiguousarray(jfk[:1600])
    rc = L.melspec_stream_push_host_vad(bank._h, ids.ctypes.data_as(C2.POINTER(C2.c_uint32)), x.ctypes.data_as(C2.POINTER(C2.c_float)),
                                        lens.ctypes.data_as(C2.POINTER(C2.c_uint32)), 1, out.ctypes.data_as(C2.POINTER(C2.c_float)), out.size,
                                        frames.ctypes.data_as(C2.POINTER(C2.c_uint32)), acts.ctypes.data_as(C2.c_void_p), 1)
    assert rc == _lib.ERR_CAPACITY                                     # 8 frames come out of 1600 samples at this point, room for one record
    assert bank.vad_frames(0) == 0                                      # nothing was committed
    r, a = bank.push_vad([0], [x])
    assert r[0].shape[0] == len(a[0]) == 8 and all(v is None for v in a[0]) and bank.vad_frames(0) == 8
    d_out = gpu.DeviceBuffer(2 * 30 * 80 * 4)
    with pytest.raises(gpu.HipError):
        bank.push_device_vad([1], [0], d_out.ptr, 0)                   # no frames -> fine; but a NULL records pointer with frames is an error:
        gpu.synth_pcm_window(bank.input_ptr(1), 1, 1600, 0, 1)
        bank.push_device_vad([1], [1600], d_out.ptr, 0)
    d_out.free(); bank.close(); m.close()

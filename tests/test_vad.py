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

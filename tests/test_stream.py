"""Streaming bank (melspec_stream_*): Spectrogram::add + RingBuffer::maybe_mel semantics (src/stft.rs:48-86,
src/rb.rs:60-121) with the overlap-save state on the device.  The CPU part drives the bank's bookkeeping
(csrc/stream_plan.hpp) with host stand-ins for the kernels; the GPU part drives the real thing.  Reference:
the oracle's hop-by-hop streaming loop, which testdata/rust_jfk_golden.npy pins (tests/test_oracle.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

TOL = 1e-4
SR = 16000.0


class EmuBank:
    def __init__(self, hop, n_mels, n_streams, max_chunk, sr=SR):
        d = os.path.join(ROOT, "tests", "emu")
        subprocess.check_call(["make", "-C", d, "-s"])
        L = C.CDLL(os.path.join(d, "libmelspec_emu.so"))
        L.emu_stream_create.restype = C.c_void_p
        L.emu_stream_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint32, C.c_uint32]
        L.emu_stream_destroy.argtypes = [C.c_void_p]
        L.emu_stream_frames_after.restype = C.c_longlong
        L.emu_stream_frames_after.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.emu_stream_push.restype = C.c_longlong
        L.emu_stream_push.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        self.L, self.n_mels = L, n_mels
        self.h = L.emu_stream_create(hop, n_mels, sr, n_streams, max_chunk)

    def frames_after(self, s, n):
        return int(self.L.emu_stream_frames_after(self.h, s, n))

    def _run(self, ids, chunks, flush):
        ids = np.ascontiguousarray(ids, np.uint32)
        xs = [np.ascontiguousarray(c, np.float32).ravel() for c in chunks]
        lens = np.ascontiguousarray([x.shape[0] for x in xs], np.uint32)
        flat = np.concatenate(xs + [np.zeros(1, np.float32)])
        out = np.full((sum((int(n) // 1) for n in lens) // 1 + 2 * len(ids) + 8, self.n_mels), np.nan, np.float32)
        fr = np.zeros(len(ids), np.uint32)
        tot = self.L.emu_stream_push(self.h, ids.ctypes.data, flat.ctypes.data, lens.ctypes.data, len(ids), int(flush), out.ctypes.data, fr.ctypes.data)
        if tot < 0:
            raise ValueError(int(tot))
        res, cur = [], 0
        for f in fr:
            res.append(out[cur:cur + int(f)].copy()); cur += int(f)
        assert cur == tot
        return res

    def push(self, ids, chunks):
        return self._run(ids, chunks, False)

    def flush(self, ids):
        return self._run(ids, [np.zeros(0, np.float32)] * len(ids), True)

    def close(self):
        self.L.emu_stream_destroy(self.h)


def _drive(bank, oracle, jfk, hop, n_mels, n_streams, max_chunk, seed, sr=SR, flush=True, fft=400):
    """Random-sized chunks (0 .. max_chunk, incl. sizes below, at and above a hop) to a random subset of the
    streams per push; every stream's concatenated output must be what the reference's loop emits for its
    concatenated input."""
    rng = np.random.default_rng(seed)
    src = [np.ascontiguousarray(jfk[(7919 * s) % 60000:][:24000] * (1.0 + 0.1 * s)) for s in range(n_streams)]
    pos = [0] * n_streams
    got = [[] for _ in range(n_streams)]
    sizes = [0, 1, hop - 1, hop, hop + 1, 2 * hop, 399, 400, 401, max_chunk]
    live = set(range(n_streams))
    while live:
        ids = [s for s in sorted(live) if rng.random() < 0.7]
        if not ids:
            continue
        chunks = []
        for s in ids:
            n = int(rng.choice(sizes)) if rng.random() < 0.5 else int(rng.integers(0, max_chunk + 1))
            n = min(n, max_chunk, len(src[s]) - pos[s])
            assert bank.frames_after(s, n) >= 0
            chunks.append(src[s][pos[s]:pos[s] + n]); pos[s] += n
        want_counts = [bank.frames_after(s, len(c)) for s, c in zip(ids, chunks)]
        res = bank.push(ids, chunks)
        assert [len(r) for r in res] == want_counts
        for s, r in zip(ids, res):
            got[s].append(r)
            if pos[s] >= len(src[s]):
                live.discard(s)
    tails = bank.flush(list(range(n_streams))) if flush else [np.zeros((0, n_mels), np.float32)] * n_streams
    worst = 0.0
    for s in range(n_streams):
        mine = np.concatenate(got[s] + [tails[s]])
        want = oracle.stream_mel(src[s], fft, hop, n_mels, sr, flush_tail=flush)
        assert mine.shape == want.shape, (s, mine.shape, want.shape)
        worst = max(worst, float(np.abs(mine - want).max()))
    return worst


# ---- CPU: bookkeeping + host stand-ins ----------------------------------------------------------------

@pytest.mark.parametrize("hop,n_mels,max_chunk,seed", [(160, 80, 1000, 1), (160, 80, 160, 2), (128, 40, 517, 3), (320, 100, 2000, 4),
                                                       (400, 80, 900, 5), (2, 80, 50, 6)])
def test_bank_bookkeeping_on_the_host(oracle, jfk, hop, n_mels, max_chunk, seed):
    bank = EmuBank(hop, n_mels, 5 if hop > 2 else 2, max_chunk)
    src = jfk if hop > 2 else jfk[:3000]
    if hop == 2:
        rng = np.random.default_rng(seed)
        x = np.ascontiguousarray(src[1000:1900])
        out = []
        p = 0
        while p < len(x):
            n = int(rng.integers(0, max_chunk + 1)); n = min(n, len(x) - p)
            out.append(bank.push([1], [x[p:p + n]])[0]); p += n
        mine = np.concatenate(out)
        want = oracle.stream_mel(x, 400, hop, n_mels, SR)
        assert mine.shape == want.shape and np.abs(mine - want).max() <= TOL
    else:
        assert _drive(bank, oracle, src, hop, n_mels, 5, max_chunk, seed) <= TOL
    bank.close()


def test_first_frame_alignment_and_golden_pin(oracle, jfk):
    """off = ceil(n_fft/hop)*hop - n_fft: 80 for 400/160; hop-sized pushes emit nothing twice, then
    one frame per push.  The streaming output equals the batch path on samples[80:], which is what the
    reference's quantized_mel_golden.tga holds (tests/test_quant.py)."""
    bank = EmuBank(160, 80, 1, 160)
    counts = [len(bank.push([0], [jfk[i * 160:(i + 1) * 160]])[0]) for i in range(6)]
    assert counts == [0, 0, 1, 1, 1, 1]
    bank.close()
    bank = EmuBank(160, 80, 1, 176000)
    got = bank.push([0], [jfk])[0]
    want = oracle.compute_mel_spectrogram_cpu(jfk[80:], 400, 160, 80)
    assert got.shape == (1098, 80) and np.abs(got - want[:1098]).max() <= TOL
    assert bank.flush([0])[0].shape == (0, 80)           # 176000 is a whole number of hops: nothing pending
    bank.close()


def test_push_errors_on_the_host():
    bank = EmuBank(160, 80, 3, 100)
    z = np.zeros(10, np.float32)
    with pytest.raises(ValueError, match="-1"):
        bank.push([3], [z])                      # id out of range
    with pytest.raises(ValueError, match="-1"):
        bank.push([1, 1], [z, z])                # twice in one push
    with pytest.raises(ValueError, match="-2"):
        bank.push([0], [np.zeros(101, np.float32)])
    bank.close()


# ---- GPU ---------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("fft,hop,n_mels,max_chunk,seed", [(400, 160, 80, 1000, 1), (400, 160, 128, 160, 2), (400, 128, 40, 517, 3),
                                                           (512, 160, 80, 700, 4), (400, 161, 80, 333, 5)])
def test_gpu_bank_random_chunks(gpu, oracle, jfk, fft, hop, n_mels, max_chunk, seed):
    """the fused f32 kernels (n_fft 400, even and odd hop) and the fused f64 kernel (512) behind the same bank"""
    m = gpu.HipMelSpectrogram(fft, hop, SR, n_mels)
    bank = gpu.StreamBank(m, 7, max_chunk)
    assert _drive(bank, oracle, jfk, hop, n_mels, 7, max_chunk, seed, fft=fft) <= TOL
    # a reset stream starts over (zero history, idx = 0); its neighbours keep their state
    n = min(1000, max_chunk)
    bank.reset([2])
    before3 = bank.frames_after(3, n)
    a = bank.push([2, 3], [jfk[:n], jfk[:n]])
    assert len(a[0]) == bank_frames(fft, hop, n) and len(a[1]) == before3
    if len(a[0]):
        want = oracle.stream_mel(jfk[:n], fft, hop, n_mels, SR)
        assert np.abs(a[0] - want).max() <= TOL
    bank.close(); m.close()


def bank_frames(fft, hop, n):
    h = n // hop
    skip = 0
    for j in range(h):
        if (j + 1) * hop < fft:
            skip += 1
    return h - skip


@pytest.mark.gpu
def test_gpu_streaming_reproduces_reference_golden(gpu, jfk):
    """rust_jfk_golden.npy (src/rb.rs:134-179): RingBuffer fed the whole file, 512/160/80, @1e-6 in the reference."""
    want = np.load(os.path.join(GOLDEN, "rust_jfk_golden.npy"))            # (80, 1097)
    m = gpu.HipMelSpectrogram(512, 160, SR, 80)
    rb = gpu.RingBuffer(m, capacity=len(jfk))
    cols = []
    for p in range(0, len(jfk), 4000):                                       # the test's wav reader hands over blocks
        rb.add_frame(jfk[p:p + 4000])
        while True:
            c = rb.maybe_mel()
            if c is None:
                break
            assert c.shape == (80, 1) and c.dtype == np.float64
            cols.append(c)
    got = np.concatenate(cols, axis=1)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-6      # the reference's own gate (src/rb.rs:171-178)
    rb.close(); m.close()


@pytest.mark.gpu
def test_gpu_device_producer_path_and_many_streams(gpu, oracle):
    """4096 live streams, one hop per push, chunks written straight into the slots by a device producer."""
    n_streams, hop = 4096, 160
    m = gpu.HipMelSpectrogram(400, hop, SR, 80)
    bank = gpu.StreamBank(m, n_streams, hop)
    ids = np.arange(n_streams, dtype=np.uint32)
    out = gpu.DeviceBuffer(n_streams * 80 * 4)
    pushes = 6
    got = []
    p0 = bank.input_ptr(0)
    slot = bank.input_ptr(1) - p0
    assert slot > 0 and p0 % 16 == 0 and slot % 16 == 0
    for k in range(pushes):
        # synth clip c, samples [k*hop, (k+1)*hop): generate the whole clip prefix once per push into a scratch and copy the window
        gpu.synth_pcm_window(p0, slot // 4, hop, k * hop, n_streams)
        fr = bank.push_device(ids, np.full(n_streams, hop, np.uint32), out.ptr)
        assert (fr == (1 if k >= 2 else 0)).all()
        if k >= 2:
            got.append(out.download((n_streams, 80)))
    for s in (0, 1, 7, 2049, 4095):
        want = oracle.stream_mel(oracle.synth_pcm(s, pushes * hop), 400, hop, 80, SR)
        mine = np.stack([g[s] for g in got])
        assert mine.shape == want.shape == (4, 80) and np.abs(mine - want).max() <= TOL
    out.free(); bank.close(); m.close()


@pytest.mark.gpu
def test_gpu_spectrogram_and_mel_spectrogram_objects(gpu, oracle, jfk):
    """The reference's per-frame pair (README usage): Spectrogram::add hop by hop (src/stft.rs:48-86) feeding MelSpectrogram::add
    (src/mel.rs:26-32).  Spectra equal compute_all_cpu on the samples from the stream's first frame on; mel columns equal the
    streaming restatement."""
    fft, hop = 400, 160
    sg = gpu.Spectrogram(fft, hop)
    ms = gpu.MelSpectrogram(fft, SR, 80)
    x = jfk[:16000]
    specs, mels = [], []
    for p in range(0, len(x) - hop + 1, hop):
        f = sg.add(x[p:p + hop])
        if f is not None:
            assert f.shape == (fft,) and f.dtype == np.complex128
            specs.append(f)
            col = ms.add(f)
            assert col.shape == (80, 1) and col.dtype == np.float64
            mels.append(col[:, 0])
    first = ((fft + hop - 1) // hop) * hop - fft                       # 80: the first window ends at the third hop
    want = oracle.compute_all_cpu(x[first:], fft, hop)
    assert len(specs) == want.shape[0] == 98
    assert np.abs(np.stack(specs) - want).max() <= 1e-9 * np.abs(want).max()
    want_mel = oracle.stream_mel(x, fft, hop, 80, SR)
    assert np.abs(np.stack(mels) - want_mel).max() <= 2e-6             # f64 from the spectrum to the column (the oracle's rows are f32)
    assert sg.add(x[:37]) is not None                                   # a short block is zero-padded (src/stft.rs:57-60)
    with pytest.raises(AssertionError):
        sg.add(x[:hop + 1])
    sg.close(); ms.close()


@pytest.mark.gpu
def test_gpu_steady_state_pushes_reuse_the_previous_plan(gpu, oracle, jfk):
    """A push that repeats the previous one (same streams, same lengths, same pending counts, every stream past its first window) reuses
    the entry table and the ragged plan that are still on the device.  Runs of such pushes interleaved with everything that must miss --
    another length, a subset of the streams, a pending remainder, a reset, the host and the device form alternating, a flush -- give,
    stream by stream, exactly the frames of the reference's streaming loop."""
    hop = 160
    m = gpu.HipMelSpectrogram(400, hop, SR, 80)
    bank = gpu.StreamBank(m, 3, 4000)
    src = [jfk[:70000], jfk[30000:100000], (oracle.synth_pcm(3, 70000) * np.float32(0.3))]
    pos, got = [0, 0, 0], [[], [], []]

    def push(ids, lens):
        chunks = [src[i][pos[i]:pos[i] + n] for i, n in zip(ids, lens)]
        for i, n in zip(ids, lens):
            pos[i] += n
        for i, r in zip(ids, bank.push(ids, chunks)):
            got[i].append(r)

    for _ in range(8): push([0, 1, 2], [hop] * 3)              # fills, then hits
    push([0, 1, 2], [hop, 2 * hop, hop])                      # another length: miss
    for _ in range(5): push([0, 1, 2], [hop] * 3)              # miss, then hits
    push([0, 2], [hop, hop])                                  # a subset: miss
    for _ in range(4): push([0, 2], [hop, hop])                # hits
    push([0, 1, 2], [37, hop, hop])                           # leaves a remainder in stream 0: miss
    for _ in range(4): push([0, 1, 2], [hop] * 3)              # pending 37 stays: miss once, then hits
    push([0, 1, 2], [hop - 37, hop, hop])                     # remainder gone
    for _ in range(6): push([0, 1, 2], [hop] * 3)
    for i in range(3):
        want = oracle.stream_mel(src[i][:pos[i]], 400, hop, 80, SR)
        mine = np.concatenate(got[i])
        assert mine.shape == want.shape and np.abs(mine - want).max() <= TOL, i
    # a reset invalidates; the stream starts again at its first window
    bank.reset([1])
    pos[1], got[1] = 0, []
    for _ in range(7): push([0, 1, 2], [hop] * 3)
    want = oracle.stream_mel(src[1][:pos[1]], 400, hop, 80, SR)
    assert np.concatenate(got[1]).shape == want.shape and np.abs(np.concatenate(got[1]) - want).max() <= TOL
    # the device form between host pushes, then a flush (never cached)
    from mel_spec_amd import _lib
    import ctypes as C2
    d_out = gpu.DeviceBuffer(3 * 80 * 4)
    for _ in range(3):
        for i in range(3):                                        # the producer writes the next hop straight into the stream's slot
            x = np.ascontiguousarray(src[i][pos[i]:pos[i] + hop])
            assert _lib.lib().melspec_memcpy_h2d(C2.c_void_p(bank.input_ptr(i)), x.ctypes.data_as(C2.c_void_p), x.nbytes) == 0
            pos[i] += hop
        fr = bank.push_device([0, 1, 2], [hop] * 3, d_out.ptr)
        rows = d_out.download((3, 80))
        for i in range(3):
            assert fr[i] == 1
            got[i].append(rows[i:i + 1].copy())
        push([0, 1, 2], [hop] * 3)
    tail = bank.flush([0, 1, 2])
    for i in range(3):
        want = oracle.stream_mel(src[i][:pos[i]], 400, hop, 80, SR)
        mine = np.concatenate(got[i])
        assert mine.shape == want.shape and np.abs(mine - want).max() <= TOL, i
        assert tail[i].shape[0] == 0                              # nothing pending: a flush emits nothing
    d_out.free(); bank.close(); m.close()

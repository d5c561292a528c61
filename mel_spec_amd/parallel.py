"""Per-clip sharding across the GPUs of one node.

The path has no exchange step: every frame (Whisper) / every clip (fbank, because of CMN) is
independent, so ranks take contiguous blocks of clips and no collective
touches the data.  torch.distributed (RCCL) is used by bench.py only for the timing barrier
and the max-over-ranks of the elapsed time.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [lo, hi) block of `n_items` owned by `rank`; sizes differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_by_samples(lengths, world_size: int) -> list[tuple[int, int]]:
    """Contiguous clip ranges balanced by total samples (ragged batches): melspec_shard_by_samples, host logic of the
    library (runs without a GPU)."""
    import ctypes as C
    import numpy as np
    from ._lib import lib
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    bounds = np.zeros(world_size + 1, np.uint32)
    rc = lib().melspec_shard_by_samples(ln.ctypes.data_as(C.POINTER(C.c_uint64)), ln.shape[0], world_size,
                                        bounds.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc:
        raise ValueError("melspec_shard_by_samples failed")
    return [(int(bounds[k]), int(bounds[k + 1])) for k in range(world_size)]


class ShardedMelSpectrogram:
    """HipMelSpectrogram over the GPUs of one node (melspec_sharded_*): one context + stream per device, one host thread per
    device while a call runs, contiguous blocks of clips per device balanced by samples, no collective.  `devices`: list of
    device indices (None = every gfx950 device; listing a device twice gives it two contexts -- how the 1-GPU tests exercise
    the multi-context path)."""

    def __init__(self, fft_size: int, hop_size: int, sampling_rate: float, n_mels: int, devices=None):
        import ctypes as C
        from ._lib import lib
        from .hip import _check
        self._h = None
        h = C.c_void_p()
        if devices is None:
            _check(lib().melspec_sharded_create(C.byref(h), None, 0, fft_size, hop_size, float(sampling_rate), n_mels), construct=True)
        else:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().melspec_sharded_create(C.byref(h), arr, len(devices), fft_size, hop_size, float(sampling_rate), n_mels), construct=True)
        self._h = h
        self.n_mels, self.fft_size, self.hop_size = n_mels, fft_size, hop_size
        self.n_shards = int(lib().melspec_sharded_n_shards(h))
        # MELSPEC_PRECISE: the test mirror's switch for the initial precision mode (see HipMelSpectrogram.__init__), on every shard
        import os
        env = os.environ.get("MELSPEC_PRECISE", "")[:1]
        if env in ("1", "f"):
            for k in range(self.n_shards):
                _check(lib().melspec_set_precision(lib().melspec_sharded_ctx(h, k), 1 if env == "1" else 2))

    def num_frames(self, n: int) -> int:
        return 0 if n < self.fft_size else (n - self.fft_size) // self.hop_size + 1

    def compute_ragged(self, clips) -> list:
        """list of 1-D host arrays -> list of [frames_i, n_mels] arrays; shard k computes clips bounds[k]:bounds[k+1]"""
        import ctypes as C
        import numpy as np
        from ._lib import lib
        from .hip import _check, _f32, _fp
        arrs = [_f32(c).reshape(-1) for c in clips]
        lens = np.array([a.shape[0] for a in arrs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(arrs) else np.zeros(0, np.uint64)
        frames = [self.num_frames(int(n)) for n in lens]
        flat = np.concatenate(arrs) if arrs and int(lens.sum()) else np.zeros(1, np.float32)
        out = np.empty(max(sum(frames) * self.n_mels, 1), np.float32)
        total = C.c_uint64(0)
        u64p = C.POINTER(C.c_uint64)
        _check(lib().melspec_sharded_compute_batch_host(self._h, _fp(flat), offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), len(arrs),
                                                        _fp(out), None, out.size, C.byref(total)))
        assert int(total.value) == sum(frames)
        res, cur = [], 0
        for f in frames:
            res.append(out[cur:cur + f * self.n_mels].reshape(f, self.n_mels))
            cur += f * self.n_mels
        return res

    def compute_uniform_device(self, d_pcm, clip_stride: int, clip_len: int, n_clips, d_out) -> None:
        """melspec_sharded_compute_uniform_device: per-shard device pointers (shard k's clips on device k), asynchronous"""
        import ctypes as C
        import numpy as np
        from ._lib import lib
        from .hip import _check
        n = self.n_shards
        assert len(d_pcm) == len(d_out) == len(n_clips) == n
        pin, pout = (C.c_void_p * n)(*d_pcm), (C.c_void_p * n)(*d_out)
        nc = np.ascontiguousarray(n_clips, dtype=np.uint32)
        _check(lib().melspec_sharded_compute_uniform_device(self._h, pin, clip_stride, clip_len, nc.ctypes.data_as(C.POINTER(C.c_uint32)), pout))

    def compute_ragged_device(self, d_pcm, offsets, lengths, n_clips, d_out, out_offsets=None) -> None:
        """melspec_sharded_compute_ragged_device: the shards' clip tables concatenated, offsets relative to each shard's own buffers"""
        import ctypes as C
        import numpy as np
        from ._lib import lib
        from .hip import _check
        n = self.n_shards
        pin, pout = (C.c_void_p * n)(*d_pcm), (C.c_void_p * n)(*d_out)
        nc = np.ascontiguousarray(n_clips, dtype=np.uint32)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        _check(lib().melspec_sharded_compute_ragged_device(self._h, pin, off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p), nc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                           pout, None if oo is None else oo.ctypes.data_as(u64p)))

    def synchronize(self) -> None:
        from ._lib import lib
        from .hip import _check
        _check(lib().melspec_sharded_synchronize(self._h))

    def close(self) -> None:
        if self._h is not None:
            from ._lib import lib
            lib().melspec_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_peer(dst_device: int, dst_ptr: int, pieces) -> None:
    """melspec_gather_peer: pieces = [(src_device, src_ptr, n_bytes, dst_offset_bytes), ...], copied concurrently (one
    stream per source device, hipMemcpyPeerAsync over xGMI)."""
    import ctypes as C
    from ._lib import lib
    from .hip import _check
    n = len(pieces)
    devs = (C.c_int * n)(*[p[0] for p in pieces])
    srcs = (C.c_void_p * n)(*[p[1] for p in pieces])
    sizes = (C.c_size_t * n)(*[p[2] for p in pieces])
    offs = (C.c_size_t * n)(*[p[3] for p in pieces])
    _check(lib().melspec_gather_peer(dst_device, C.c_void_p(dst_ptr), devs, srcs, sizes, offs, n))


def timed_steps(step, synchronize, steps: int, warmup: int, dist=None, device=None):
    """The bench protocol: `warmup` untimed steps, then exactly `steps` timed steps bracketed by a
    barrier + synchronize on both sides; returns the MAX elapsed seconds over ranks.

    `dist` is torch.distributed (initialised) or None for a single process; `device` is where the
    reduction tensor lives (cuda device for nccl, None/cpu for gloo)."""
    import time
    for _ in range(warmup):
        step()
    synchronize()
    if dist is not None:
        dist.barrier()
    synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed

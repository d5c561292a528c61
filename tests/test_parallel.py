"""Multi-GPU path: per-clip sharding with no data-path collective.  The N>1 logic is exercised
with world_size-2 gloo processes on CPU (the oracle stands in for the kernel as the per-rank
work; what is under test is the partition and the barrier/max-over-ranks timing protocol)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from mel_spec_amd.parallel import shard_by_samples, shard_range
from conftest import ROOT


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1024, 65536, 65537):
        for w in (1, 2, 3, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_shard_by_samples_balances_ragged():
    """melspec_shard_by_samples (host logic of the library, no GPU needed): what ShardedMelSpectrogram splits by."""
    lengths = [480000] * 10 + [16000] * 300 + [160000] * 50
    for w in (1, 2, 4, 8):
        b = shard_by_samples(lengths, w)
        assert len(b) == w and b[0][0] == 0 and b[-1][1] == len(lengths)
        assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
        loads = [sum(lengths[lo:hi]) for lo, hi in b]
        assert max(loads) <= sum(lengths) / w + max(lengths)
    # degenerate inputs: fewer clips than shards, empty clips, nothing at all
    assert shard_by_samples([5, 5], 4)[-1][1] == 2 and sum(hi - lo for lo, hi in shard_by_samples([5, 5], 4)) == 2
    assert shard_by_samples([], 3) == [(0, 0)] * 3
    b = shard_by_samples([0, 0, 7, 0], 2)
    assert b[0][0] == 0 and b[-1][1] == 4 and b[0][1] == b[1][0]
    rng = np.random.default_rng(3)
    for _ in range(20):
        ln = rng.integers(0, 500000, int(rng.integers(1, 400))).tolist()
        w = int(rng.integers(1, 9))
        b = shard_by_samples(ln, w)
        assert b[0][0] == 0 and b[-1][1] == len(ln) and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert max(sum(ln[lo:hi]) for lo, hi in b) <= sum(ln) / w + max(ln)


def test_sharded_object_without_a_gpu_reports_unavailable():
    """The spawn path needs devices; on a box without one construction fails like CudaError::Unavailable (src/cuda.rs:10-25)
    and nothing falls back to the CPU."""
    import mel_spec_amd as M
    if M._lib.lib().melspec_device_count() >= 1:
        pytest.skip("a GPU is visible: covered by the -m gpu test below")
    with pytest.raises(M.HipUnavailable):
        M.ShardedMelSpectrogram(400, 160, 16000.0, 80, devices=[0, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0], None])
def test_sharded_mel_spectrogram_on_the_devices_of_this_box(gpu, oracle, jfk, devices):
    """One context + stream + host thread per listed device; a device listed several times gets several contexts, which is
    how a 1-GPU box runs the multi-shard path (threads, per-shard pipelines, disjoint output ranges)."""
    rng = np.random.default_rng(11)
    clips = [jfk[int(a):int(a) + int(n)] for a, n in zip(rng.integers(0, 60000, 37), rng.integers(0, 90000, 37))]
    clips += [oracle.synth_pcm(3, 3_000_000), np.zeros(399, np.float32), oracle.synth_pcm(5, 400)]      # a clip longer than a pipeline chunk
    sh = gpu.ShardedMelSpectrogram(400, 160, 16000.0, 80, devices=devices)
    assert sh.n_shards == (len(devices) if devices else gpu.device_count())
    got = sh.compute_ragged(clips)
    for g, x in zip(got, clips):
        want = oracle.compute_mel_spectrogram_cpu(x, 400, 160, 80, 16000.0)
        assert g.shape == want.shape and (g.size == 0 or np.abs(g - want).max() <= 1e-4)
    sh.close()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0], None])
def test_sharded_device_resident_calls(gpu, oracle, jfk, devices):
    """melspec_sharded_compute_uniform_device / _ragged_device: every shard's clips already on its device, its frames left there, one
    launch per shard on the shard's own stream; the same bits as one context computing the shard's clips (a device listed several
    times gives the 1-GPU box several shards)."""
    sh = gpu.ShardedMelSpectrogram(400, 160, 16000.0, 80, devices=devices)
    n = sh.n_shards
    single = gpu.HipMelSpectrogram(400, 160, 16000.0, 80)
    single.set_auto_adaptive(False)
    for k in range(n):
        gpu._lib.lib().melspec_set_auto_adaptive(gpu._lib.lib().melspec_sharded_ctx(sh._h, k), 0)      # bits are compared below
    clip_len, fpc = 24000, single.num_frames(24000)
    counts = [5 + 3 * k for k in range(n)]
    pcm, out, want = [], [], []
    for k in range(n):
        x = np.stack([oracle.synth_pcm(100 * k + c, clip_len) if c % 2 else np.resize(jfk[977 * (k + c):], clip_len) for c in range(counts[k])]).astype(np.float32)
        b = gpu.DeviceBuffer(x.nbytes); b.upload(x)
        pcm.append(b); out.append(gpu.DeviceBuffer(counts[k] * fpc * 80 * 4))
        want.append(single.compute_batch(x))
        assert np.abs(want[-1][0] - oracle.compute_mel_spectrogram_cpu(x[0], 400, 160, 80, 16000.0)).max() <= 1e-4
    sh.compute_uniform_device([b.ptr for b in pcm], clip_len, clip_len, counts, [b.ptr for b in out])
    sh.synchronize()
    for k in range(n):
        assert np.array_equal(out[k].download((counts[k], fpc, 80)), want[k])
    # ragged: per shard three clips of different lengths cut from its buffer, packed outputs
    lens = [24000, 399, 5000]
    offs, lns = [], []
    for k in range(n):
        offs += [0, clip_len, 2 * clip_len]; lns += lens
    for b in out:
        b.upload(np.zeros(b.nbytes // 4, np.float32))
    sh.compute_ragged_device([b.ptr for b in pcm], offs, lns, [3] * n, [b.ptr for b in out])
    sh.synchronize()
    f2 = single.num_frames(5000)
    for k in range(n):
        got = out[k].download(((fpc + f2) * 80,))
        assert np.array_equal(got[:fpc * 80].reshape(fpc, 80), want[k][0])
        x2 = pcm[k].download((5000,), offset_bytes=2 * clip_len * 4)
        assert np.abs(got[fpc * 80:].reshape(f2, 80) - oracle.compute_mel_spectrogram_cpu(x2, 400, 160, 80, 16000.0)).max() <= 1e-4
    for b in pcm + out:
        b.free()
    single.close(); sh.close()


def test_eight_rank_rehearsal_of_config5_plans_what_the_full_size_test_samples():
    """`bench.py --gpus 8 --config 5` has never run on eight devices.  Its launch path with eight ranks (gloo, no
    GPU): every rank plans the 8192-clip shard whose boundary clips tests/test_full_size.py compares against the oracle on one GPU,
    and the residency it would allocate -- 15.7 GB of PCM + 7.9 GB of mel + the guard's note list -- fits a 288 GB device 12 times."""
    import json
    from test_full_size import _picks
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "8", "--config", "5"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["shards"] == [[8192 * r, 8192 * (r + 1)] for r in range(8)]
    sampled = set(_picks(65536))
    for lo, hi in line["shards"]:
        assert {lo, hi - 1} <= sampled, (lo, hi)          # first and last clip of every rank's shard are among the clips checked at full size
    fpc = (480000 - 400) // 160 + 1
    for r, res in enumerate(line["residency"]):
        assert res["rank"] == r and res["pcm_bytes"] == 8192 * 480000 * 4 and res["mel_bytes"] == 8192 * fpc * 80 * 4
        assert res["guard_list_bytes"] == (8192 * 500 + 65536) * 8
        assert 23.5 < res["total_GB"] < 23.7 and res["total_GB"] * 12 < line["hbm_per_gpu_GB"]
    assert sum(b - a for a, b in line["shards"]) * fpc == 196476928


@pytest.mark.gpu
def test_gather_peer_names_the_piece_that_failed(gpu):
    """the error paths of melspec_gather_peer, which no 1-GPU run had executed.  A source device that does not
    exist fails in hipSetDevice, a destination that does not exist in hipMemcpyPeerAsync; either way the call returns non-zero after the
    pieces already queued have finished (no stream left behind: the next call works), and melspec_last_error() says which piece, from
    which device to which."""
    from mel_spec_amd._lib import lib
    a = gpu.DeviceBuffer(4096 * 4); b = gpu.DeviceBuffer(4096 * 4); dst = gpu.DeviceBuffer(8192 * 4)
    xa = np.arange(4096, dtype=np.float32)
    a.upload(xa); b.upload(-xa)
    with pytest.raises(Exception) as e1:
        gpu.gather_peer(0, dst.ptr, [(0, a.ptr, 4096 * 4, 0), (99, b.ptr, 4096 * 4, 4096 * 4)])
    msg = lib().melspec_last_error().decode()
    assert "piece 1 of 2" in msg and "source device 99" in msg and "destination device 0" in msg and "hipSetDevice" in msg, msg
    assert str(e1.value)
    assert np.array_equal(dst.download((8192,))[:4096], xa)          # piece 0 was queued before the failure and has landed
    with pytest.raises(Exception):
        gpu.gather_peer(99, dst.ptr, [(0, a.ptr, 4096 * 4, 0)])
    msg = lib().melspec_last_error().decode()
    assert "piece 0 of 1" in msg and "source device 0" in msg and "destination device 99" in msg, msg
    # a device is no peer of itself (hipDeviceCanAccessPeer: 0): the branch without peer access is the one every 1-GPU call takes
    gpu.gather_peer(0, dst.ptr, [(0, b.ptr, 4096 * 4, 0), (0, a.ptr, 4096 * 4, 4096 * 4)])
    got = dst.download((8192,))
    assert np.array_equal(got[:4096], -xa) and np.array_equal(got[4096:], xa)
    for x in (a, b, dst):
        x.free()


@pytest.mark.gpu
def test_gather_peer_consolidates_device_results(gpu):
    """melspec_gather_peer with every piece on device 0 (the only one of this box): the offsets / sizes / streams path."""
    a = gpu.DeviceBuffer(4096 * 4); b = gpu.DeviceBuffer(1000 * 4); dst = gpu.DeviceBuffer(6000 * 4)
    xa, xb = np.arange(4096, dtype=np.float32), -np.arange(1000, dtype=np.float32)
    a.upload(xa); b.upload(xb)
    gpu.gather_peer(0, dst.ptr, [(0, a.ptr, 4096 * 4, 0), (0, b.ptr, 1000 * 4, 5000 * 4)])
    got = dst.download((6000,))
    assert np.array_equal(got[:4096], xa) and np.array_equal(got[5000:], xb)
    for x in (a, b, dst):
        x.free()


WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from mel_spec_amd.parallel import shard_range
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_clips, clip_len = 6, 4000
lo, hi = shard_range(n_clips, rank, world)
dist.barrier(); t0 = time.perf_counter()
outs = [O.compute_mel_spectrogram_cpu(O.synth_pcm(c, clip_len)) for c in range(lo, hi)]
dist.barrier(); dt = time.perf_counter() - t0
t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
# the bench protocol itself (bench.py uses this helper with nccl): rank 1 is slower, MAX must win
from mel_spec_amd.parallel import timed_steps
calls = []
el = timed_steps(lambda: (calls.append(1), time.sleep(0.02 * (rank + 1))), lambda: None, 3, 2, dist, None)
assert len(calls) == 5 and 0.11 <= el < 1.0, (len(calls), el)
frames = torch.tensor([sum(o.shape[0] for o in outs)], dtype=torch.int64); dist.all_reduce(frames)
chk = torch.tensor([float(sum(np.float64(o).sum() for o in outs))], dtype=torch.float64); dist.all_reduce(chk)
if rank == 0:
    print("RESULT", int(frames.item()), repr(float(chk.item())), float(t.item()) >= dt - 1e-9)
dist.destroy_process_group()
'''


def test_two_rank_gloo_run_covers_every_clip_once(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0].split()
    want = [oracle.compute_mel_spectrogram_cpu(oracle.synth_pcm(c, 4000)) for c in range(6)]
    assert int(line[1]) == sum(o.shape[0] for o in want)
    assert abs(float(line[2]) - float(sum(np.float64(o).sum() for o in want))) < 1e-6
    assert line[3] == "True"


@pytest.mark.parametrize("gpus,config", [(2, 2), (2, 5), (1, 5)])
def test_bench_spawns_its_ranks(gpus, config):
    """`python bench.py --gpus N` as the driver runs it, without a launcher: bench.py re-executes itself under
    torch.distributed.run, every rank plans its shard (config 5: shard_range over the 65 536 clips), the barrier / MAX protocol
    runs, rank 0 prints one JSON line with n_gpus == N.  --dry-run keeps it off the GPU (gloo)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--config", str(config), "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == gpus and line["dry_run"] is True
    sh = line["shards"]
    assert len(sh) == gpus
    if config == 5:
        assert line["scaling"] == "strong" and sh[0][0] == 0 and sh[-1][1] == 65536 and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
    else:
        assert line["scaling"] == "weak" and sh == [[1024 * r, 1024 * (r + 1)] for r in range(gpus)]
    assert line["max_over_ranks_s"] >= 0.002 * gpus * 3 * 0.9


def test_plain_bench_line_carries_the_speech_leg_and_config5():
    """What the driver runs: `bench.py` (N = 1) carries the real-input leg next to `value`; `bench.py --gpus N` without --config
    also times north_star's 65 536 x 30 s per-clip split (configs[4]) whose per-rank frames sum to 196 476 928.  --dry-run: the
    plan of both records without a GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}

    def run(*extra):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *extra], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])

    one = run()
    assert one["n_gpus"] == 1 and one["config"] == 2 and "speech" in one and "cfg5" not in one
    # the legs the driver's plain N = 1 run carries next to `value`
    assert one["legs"] == ["cfg3", "cfg3_split", "cfg4", "f64", "mel_major", "w512", "nemo", "nemo_f32", "host_api_single_clip_ms"] and "legs" not in run("--no-legs")
    two = run("--gpus", "2")
    assert two["n_gpus"] == 2 and two["config"] == 2 and two["scaling"] == "weak" and two["shards"] == [[0, 1024], [1024, 2048]]
    c5 = two["cfg5"]
    assert c5["scaling"] == "strong" and c5["shards"][0][0] == 0 and c5["shards"][-1][1] == 65536
    assert sum(c5["per_rank_frames"]) == 196476928
    assert "cfg5" not in run("--gpus", "2", "--config", "2")


def test_a_failing_leg_costs_its_entry_not_the_line():
    """a parity miss or an allocation failure inside one of the extra legs must not cost the bench line its `value`.  The legs
    run against a library stand-in whose every constructor raises: each leg records {"error": ...} and extra_legs returns."""
    import torch
    sys.path.insert(0, ROOT)
    import bench

    class Boom:
        def __getattr__(self, name):
            if name == "synth_pcm_device":
                return lambda *a, **k: None
            if name == "BatchLogMelConfig":
                return lambda **k: None
            def ctor(*a, **k):
                raise RuntimeError(f"{name}: no device in this test")
            return ctor

    legs = bench.extra_legs(Boom(), torch, torch.device("cpu"), None)
    assert set(legs) == set(bench.LEG_NAMES) | {"host_api_single_clip_ms"}
    for name, rec in legs.items():
        assert set(rec) == {"error"} and "no device in this test" in rec["error"], (name, rec)


def test_valu_fields_come_from_the_committed_isa_table():
    """every leg of the line carries the VALU instructions per frame of its kernel and the fraction of the 78.6 TFLOP/s
    f64 vector peak they amount to at the measured rate (profiles/isa_hist.json, written by tools/isa_legs.py from the shipped sources)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    from mel_spec_amd import build as hip_build
    path = os.path.join(ROOT, "profiles", "isa_hist.json")
    if json.load(open(path))["source_hash"] != hip_build.source_hash():          # the sources moved on: the table is rebuilt here, like the library
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_legs.py")], check=True, stdout=subprocess.DEVNULL, timeout=900)
        bench._ISA = None
    table = json.load(open(path))
    assert table["source_hash"] == hip_build.source_hash(), "profiles/isa_hist.json is stale: run tools/isa_legs.py"
    for leg in ("value", "cfg3", "cfg4", "f64", "mel_major", "w512", "nemo", "nemo_f32", "speech", "speech128"):
        v = bench.valu_fields(leg, 2.0e9)
        assert v is not None and "stale" not in v, leg
        assert v["f64_insts_per_frame"] >= 0 and v["other_valu_per_frame"] > 0
        # 128 flops per f64 wave-instruction at 2 G frames/s against 78.6 TFLOP/s
        assert abs(v["frac_of_f64_vector_peak"] - v["f64_insts_per_frame"] * 128 * 2.0e9 / 78.6e12) < 1e-12
    # no kernel the library dispatches reloads a spilled register inside its unit loop (a scratch load's vmcnt(0) also waits for the previous
    # unit's stores: the 64-mel bank's sixteen-wave kernels were 19-29 % slower for six to eight of them, round 6).  The one exception is
    # emitted but never launched: csrc/melspec_runs.hip says why it is still there.
    spilled = {k: v for k, v in table["unit_loop_scratch"].items() if v}
    assert len(table["unit_loop_scratch"]) >= 60
    assert all("whisper400_six_runs_kernelILi9ENS_13LensSixStaticILi40E" in k for k in spilled), spilled
    assert bench.valu_fields("f64", 2.4e9)["f64_insts_per_frame"] == 106.0          # 636 f64 instructions per six-frame unit
    assert bench.valu_fields("value", 3.4e9)["f64_insts_per_frame"] == 0.0
    assert bench.valu_fields("no such leg", 1.0) is None


@pytest.mark.gpu
def test_bench_multi_rank_path_rehearsed_on_one_gpu(gpu):
    """The N > 1 code path of bench.py on real hardware although the box has one GPU: two ranks under torch.distributed.run, both on
    GPU 0, rendezvous over gloo (`--one-device`).  Covers what the dry run cannot: per-rank contexts, the event-timed steps, the
    all_gather of {frames, ms}, closing the first workload and running the config-5 leg (shrunk), one JSON line from rank 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--clips", "64", "--cfg5-clips", "96",
                        "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "rehearsal" in line and line["steps"] == 5
    assert [r["frames_per_step"] for r in line["per_rank"]] == [64 * 998.0] * 2
    assert abs(line["value"] - 2 * 64 * 998 * 5 / (line["ms_per_step"] * 5e-3)) <= 1e-6 * line["value"]
    assert line["parity_max_abs_diff"] <= 1e-4
    c5 = line["config"]["cfg5"]
    assert c5["scaling"] == "strong" and [r["frames_per_step"] for r in c5["per_rank"]] == [48 * 2998.0] * 2 and c5["parity_max_abs_diff"] <= 1e-4
    assert "roofline" in line and "cpu_baseline" not in line            # the CPU baseline is an N = 1 figure


@pytest.mark.gpu
def test_bench_distributed_branch_runs_on_rccl_at_world_size_one(gpu):
    """the only code that calls init_process_group("nccl", device_id=...) had never executed anywhere.
    `bench.py --gpus 1 --force-dist` runs the distributed branch on RCCL at world size 1 -- barrier, the all_gather of
    {frames, ms}, the MAX-reduce of the elapsed time, all_gather_object of the device identities, destroy_process_group -- so the
    first multi-GPU run is not the first RCCL run.  The line says which backend ran and which device every rank was on."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--clips", "64", "--steps", "5", "--warmup", "1",
                        "--no-cpu-baseline", "--no-host-io", "--no-traffic", "--no-speech", "--no-legs"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 1 and line["dist"]["distinct_devices"] == 1
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["parity_max_abs_diff"] <= 1e-4
    r0 = line["per_rank"][0]
    assert r0["rank"] == 0 and r0["device_index"] == 0 and ("MI355" in r0["device_name"] or "gfx950" in r0["device_name"] or r0["device_name"])
    assert r0["frames_per_step"] == 64 * 998.0
    assert abs(line["value"] - 64 * 998 * 5 / (line["ms_per_step"] * 5e-3)) <= 1e-6 * line["value"]


RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch                               # before the HIP library: one HIP runtime per process (torch's), as in bench.py
import torch.distributed as dist
from mel_spec_amd.parallel import timed_steps
import mel_spec_amd as M
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
mel = M.HipMelSpectrogram(400, 160, 16000.0, 80, device=0)
pcm = torch.empty(8 * 16000, dtype=torch.float32, device=dev)
out = torch.empty(8 * mel.num_frames(16000) * 80, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
M.synth_pcm_device(pcm.data_ptr(), 16000, 16000, 0, 8, stream=stream)
calls = []
step = lambda: (calls.append(1), mel.compute_uniform_device(pcm.data_ptr(), 16000, 16000, 8, out.data_ptr(), stream=stream))
el = timed_steps(step, torch.cuda.synchronize, 4, 2, dist, dev)
assert len(calls) == 6 and 0.0 < el < 5.0
mine = torch.tensor([998.0, 0.25], dtype=torch.float64, device=dev)
allr = [torch.zeros_like(mine)]
dist.all_gather(allr, mine)
assert allr[0].tolist() == [998.0, 0.25]
objs = [None]
dist.all_gather_object(objs, {"device": torch.cuda.get_device_properties(dev).name})
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", dist.is_initialized(), objs[0]["device"], float(out.abs().sum().item()) > 0.0)
"""


@pytest.mark.gpu
def test_rccl_collectives_of_the_bench_protocol(gpu, tmp_path):
    """The collectives the bench uses, on backend "nccl" (= RCCL) at world size 1 next to the library's own launches on the same
    stream: barrier, all_gather of a 2-vector on the device, all_reduce(MAX) through mel_spec_amd.parallel.timed_steps (the helper
    bench.py times with), all_gather_object, destroy_process_group.  In a process of its own: torch must load its HIP runtime before
    the library does (as in bench.py); the test session has already loaded the library."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_OK")][0].split(maxsplit=2)
    assert line[1] == "False" and line[2].endswith("True")

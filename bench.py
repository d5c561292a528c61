#!/usr/bin/env python3
"""bench.py -- mel frames/s of the fused MI355X log-mel kernel (Whisper 400/160/80 @16 kHz).

A "step" is one pass of the hot path over one batch of synthetic PCM that is already resident in HBM.

  --config 2 (default)  BASELINE.json configs[1]: 1024 x 10 s f32 clips PER GPU -> 1 021 952 frames of 80 mels per
                        step and GPU (weak scaling: rank r owns clips [1024 r, 1024 (r+1)))
  --config 4            configs[3]: Whisper large-v3, 128 mels, 8192 x 30 s clips per GPU
  --config 5            configs[4]: 65 536 x 30 s clips, 80 mels, split over the ranks with
                        mel_spec_amd.parallel.shard_range (strong scaling: 8192 clips per rank at N = 8).  A rank whose
                        share does not fit its HBM walks it in sub-shards inside the step (the JSON says how many).

N > 1: one process per GPU over torch.distributed / RCCL.  `python bench.py --gpus N` re-executes itself under
torch.distributed.run when it was not launched by it; every rank owns its own clips (per-clip split, no data-path
collective); the timed region is bracketed by barrier + synchronize and the MAX over ranks is reported.  --gather also
times the optional consolidation of the per-rank outputs on rank 0 (RCCL send/recv over xGMI), reported separately.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     algorithmic bytes (640 B PCM in + 4*n_mels B out per frame) / kernel time vs 8 TB/s
  cpu_baseline the C oracle (a port of Spectrogram::compute_mel_spectrogram_cpu) on the host cores
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 16000.0
N_FFT, HOP = 400, 160
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CONFIGS = {             # name -> (clips, clip seconds, n_mels, scaling)
    2: (1024, 10, 80, "weak"),
    4: (8192, 30, 128, "weak"),
    5: (65536, 30, 80, "strong"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="default 2; with --gpus N > 1 and no --config the line also carries config 5's per-clip split (`config.cfg5`)")
    ap.add_argument("--no-speech", action="store_true", help="skip the real-input leg (`config.speech`: jfk_f32le.wav tiled to the config-2 batch)")
    ap.add_argument("--no-cfg5", action="store_true", help="N > 1 without --config: skip the extra config-5 leg")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra N = 1 legs next to `value` (`config.cfg3`, `config.cfg4`, `config.f64`, `config.mel_major`, "
                                                          "`config.cfg3_split`, `config.w512`, `config.nemo`, `config.nemo_f32`, `host_api_single_clip_ms`)")
    ap.add_argument("--clips", type=int, default=None, help="override the clip count (per GPU for weak, total for strong)")
    ap.add_argument("--clip-seconds", type=int, default=None)
    ap.add_argument("--n-mels", type=int, default=None)
    ap.add_argument("--precision", default="auto", choices=["auto", "f64", "f32"], help="melspec_set_precision (default: the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive host API figure (never `value`)")
    ap.add_argument("--gather", action="store_true", help="N > 1: also time the consolidation of the outputs on rank 0")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-measure roofline.traffic with rocprofv3 (two short PMC passes of this "
                    "script, ~1 min); the committed profiles/traffic.json is reported instead")
    ap.add_argument("--cfg5-clips", type=int, default=None, help="rehearsals only: shrink the config-5 leg's clip set")
    ap.add_argument("--one-device", action="store_true", help="rehearsal of the N > 1 code path on a 1-GPU box: every rank on GPU 0, ranks over gloo "
                                                               "(the figures it prints mean nothing: the ranks share one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: run the distributed branch anyway -- init_process_group(\"nccl\") = RCCL at world size 1, the "
                                                               "barrier, the all_gather of {frames, ms}, the MAX-reduce, destroy_process_group -- so that the first multi-GPU "
                                                               "run is not the first RCCL run (the line then carries `dist`)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: spawn the ranks (gloo), plan every rank's shard, run the timing protocol on a "
                                                            "sleep and print the JSON line -- the CPU test of the launch path")
    return ap.parse_args()


def respawn_under_torchrun(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def measure_traffic(config: int, timeout_s: float = 150.0):
    """HBM bytes per launch of the dominant kernel, measured now: one rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE; separate
    passes, with --kernel-trace only, as the guide prescribes) over a few steps of this script, 2 x FETCH_SIZE + WRITE_SIZE (KiB; the x 2 is
    the gfx950 correction calibrated in profiles/r01_calibration.txt).  Returns (bytes, note) or (None, why)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="melspec_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--config", str(config), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-host-io", "--no-traffic", "--no-speech", "--no-legs"]
            subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            per_kernel = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "whisper400" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        per_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not per_kernel:
                return None, f"no {counter} rows in the rocprofv3 output"
            # the dominant kernel: in the default mode every f32 launch is followed by the gated f64 launch, which moves nothing on this input
            got = max(per_kernel.values(), key=lambda v: sum(v) / len(v))
            vals[counter] = sum(got) / len(got) * 1024.0
        except Exception as e:          # a profiler that cannot run must not cost the bench line
            return None, f"rocprofv3 --pmc {counter} failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over 7 launches "
                                                            f"of this command, 2 x {vals['FETCH_SIZE'] / 1e6:.1f} MB + {vals['WRITE_SIZE'] / 1e6:.1f} MB per launch")


def cpu_baseline(clip_len: int, n_mels: int, target_s: float, pool: int) -> dict:
    """Time the oracle (port of compute_mel_spectrogram_cpu) on a bounded sample of the same
    synthetic workload, all host cores (clips split across OpenMP threads)."""
    import numpy as np
    from oracle import oracle as O
    cores = O.max_threads()
    probe = np.stack([O.synth_pcm(c, clip_len) for c in range(cores)])
    t0 = time.perf_counter()
    O.compute_mel_batch(probe, N_FFT, HOP, n_mels, SR, n_threads=cores)
    dt = max(time.perf_counter() - t0, 1e-4)
    n = int(min(pool, max(cores, round(cores * target_s / dt))))
    n -= n % cores or 0
    n = max(n, cores)
    clips = np.stack([O.synth_pcm(c, clip_len) for c in range(n)])
    reps, frames, t0 = 0, 0, time.perf_counter()
    while True:                       # repeat the sample until ~target_s of wall time has been timed
        out = O.compute_mel_batch(clips, N_FFT, HOP, n_mels, SR, n_threads=cores)
        reps += 1
        frames += out.shape[0] * out.shape[1]
        dt = time.perf_counter() - t0
        if dt >= target_s * 0.5 or reps >= 64:
            break
    # single-thread figure on a smaller slice, for the DESIGN.md table
    k = max(1, n // (2 * cores))
    t1 = time.perf_counter()
    o1 = O.compute_mel_batch(clips[:k], N_FFT, HOP, n_mels, SR, n_threads=1)
    dt1 = time.perf_counter() - t1
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": frames / dt, "unit": "mel frames/s", "cores": cores, "kind": "port", "cpu": cpu_model,
        "sample": f"{n} of the synthetic {clip_len / SR:.0f} s clips x {reps} passes ({frames} frames) in {dt:.2f} s, "
                  f"oracle/melspec_oracle.c (f64 restatement of Spectrogram::compute_mel_spectrogram_cpu), OpenMP over clips",
        "single_thread_frames_per_s": o1.shape[0] * o1.shape[1] / dt1,
    }


def dry_run(args) -> None:
    """The launch path without a GPU: ranks over gloo, the shard plan of every rank, the barrier / MAX-over-ranks protocol."""
    import torch
    import torch.distributed as dist
    from mel_spec_amd.parallel import shard_range, timed_steps
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    cfg_clips, cfg_seconds, cfg_mels, scaling = CONFIGS[args.config]
    total = args.clips or cfg_clips
    lo, hi = (rank * total, (rank + 1) * total) if scaling == "weak" else shard_range(total, rank, world)
    elapsed = timed_steps(lambda: time.sleep(0.002 * (rank + 1)), lambda: None, 3, 1, dist if world > 1 else None, None)
    # what the rank would hold resident for its share (run_workload's buffers): PCM, the mel output, the precision guard's note list
    # (one u64 per six-frame unit + a round of slack, whisper400.hip launch_ctx)
    clip_len = int((args.clip_seconds or cfg_seconds) * SR)
    n_mels = args.n_mels or cfg_mels
    fpc = (clip_len - N_FFT) // HOP + 1
    n_mine = hi - lo
    units = n_mine * ((fpc + 5) // 6)
    mine = torch.tensor([lo, hi, n_mine * clip_len * 4, n_mine * fpc * n_mels * 4, (units + 65536) * 8], dtype=torch.int64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    if rank == 0:
        line = {"dry_run": True, "n_gpus": world, "scaling": scaling, "config": args.config,
                "shards": [[int(t[0]), int(t[1])] for t in allr], "max_over_ranks_s": elapsed,
                "residency": [{"rank": r, "pcm_bytes": int(t[2]), "mel_bytes": int(t[3]), "guard_list_bytes": int(t[4]), "total_GB": float(int(t[2]) + int(t[3]) + int(t[4])) / 1e9}
                              for r, t in enumerate(allr)],
                "hbm_per_gpu_GB": 288.0}
        if world > 1 and not args.explicit_config and not args.no_cfg5:        # the extra leg of a plain `bench.py --gpus N`
            fpc5 = (CONFIGS[5][1] * int(SR) - N_FFT) // HOP + 1
            sh5 = [list(shard_range(CONFIGS[5][0], r, world)) for r in range(world)]
            line["cfg5"] = {"scaling": "strong", "shards": sh5, "per_rank_frames": [(b - a) * fpc5 for a, b in sh5]}
        if world == 1 and args.config == 2 and not args.no_speech:
            line["speech"] = {"input": "tests/golden/jfk_f32le.wav tiled to the config-2 batch"}
            line["speech128"] = {"input": "the same through the 128-mel bank (Whisper large-v3)"}
        if world == 1 and args.config == 2 and not args.no_legs:
            line["legs"] = list(LEG_NAMES) + ["host_api_single_clip_ms"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_workload(args, config: int, primary: bool, steps: int, warmup: int, M, torch, dist, dev, rank: int, world: int, local_rank: int) -> dict:
    """One config of BASELINE.json through the bench protocol: the rank's share resident in HBM, parity spot check, spin-up,
    `warmup` untimed + exactly `steps` timed steps between barrier + synchronize, MAX over ranks; HIP events around the same
    launches.  primary: the workload `value` is quoted on (the command line's overrides apply to it only)."""
    import numpy as np
    from mel_spec_amd.parallel import shard_range, timed_steps
    distributed = dist is not None            # also at world size 1 under --force-dist
    cfg_clips, cfg_seconds, cfg_mels, scaling = CONFIGS[config]
    n_mels = (args.n_mels if primary else None) or cfg_mels
    clip_seconds = (args.clip_seconds if primary else None) or cfg_seconds
    clip_len = int(clip_seconds * SR)
    total_or_per = (args.clips if primary else (args.cfg5_clips if config == 5 else None)) or cfg_clips
    if scaling == "weak":
        n_clips, first_clip = total_or_per, rank * total_or_per          # rank r owns clips [r*n, (r+1)*n)
    else:
        lo, hi = shard_range(total_or_per, rank, world)                  # contiguous per-clip split of the fixed set
        n_clips, first_clip = hi - lo, lo

    mel = M.HipMelSpectrogram(N_FFT, HOP, SR, n_mels, device=dev.index)
    mel.set_precision(args.precision)
    red_dev = None if args.one_device else dev            # where the tiny reduction tensors live (gloo reduces on the host)
    fpc = mel.num_frames(clip_len)

    # resident share, or sub-shards walked inside the step when the share does not fit (config 5 on few GPUs)
    free_b, _ = torch.cuda.mem_get_info(dev)
    bytes_per_clip = clip_len * 4 + fpc * n_mels * 4 + fpc * 4 / 6 * 1.05      # PCM + mel + the precision guard's queue
    sub = 1
    while n_clips / sub * bytes_per_clip > 0.9 * free_b:
        sub *= 2
    sub_clips = (n_clips + sub - 1) // sub
    pcm = torch.empty(sub_clips * clip_len, dtype=torch.float32, device=dev)
    out = torch.empty(sub_clips * fpc * n_mels, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    frames_per_step = fpc * n_clips

    def fill(shard: int) -> int:
        c0 = shard * sub_clips
        k = min(sub_clips, n_clips - c0)
        M.synth_pcm_device(pcm.data_ptr(), clip_len, clip_len, first_clip + c0, k, stream=stream)
        return k

    k0 = fill(0)
    torch.cuda.synchronize()

    def step():
        if sub == 1:
            mel.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, n_clips, out.data_ptr(), stream=stream)
        else:                       # the share in `sub` passes over one resident buffer (the refill is part of the step)
            for s in range(sub):
                k = fill(s)
                mel.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, k, out.data_ptr(), stream=stream)

    # parity spot check first (3 clips vs the oracle), so that nothing CPU-bound sits between the spin-up and
    # the timed region: the GPU drops back to its idle clocks within milliseconds of an empty queue
    mel.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, k0, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    parity, queued = None, None
    if rank == 0:
        from oracle import oracle as O
        worst = 0.0
        o3 = out[: k0 * fpc * n_mels].view(k0, fpc, n_mels)
        for c in sorted({0, k0 // 2, k0 - 1}):
            want = O.compute_mel_spectrogram_cpu(O.synth_pcm(first_clip + c, clip_len), N_FFT, HOP, n_mels, SR)
            worst = max(worst, float((o3[c].cpu().numpy() - want).__abs__().max()))
        parity = worst
        if worst > 1e-4:
            raise SystemExit(f"parity check failed before timing: max|diff| = {worst}")
        queued = mel.guard_last_count() if args.precision == "auto" else None
    if distributed:
        dist.barrier()

    # Untimed spin-up: the GPU leaves its idle power state only after tens of milliseconds of work
    # (measured: the first ~100 launches run ~15 % slower), so run launches for ~0.3 s, then the W warmup
    # steps, then straight into the timed region.  Nothing here is timed.
    spin_t0, spinup_steps = time.perf_counter(), 0
    while time.perf_counter() - spin_t0 < 0.3:
        for _ in range(20 if config == 2 else 1):
            step()
        torch.cuda.synchronize()
        spinup_steps += 20 if config == 2 else 1
    for _ in range(warmup):
        step()

    # timed region: barrier + synchronize on both sides, MAX over ranks (mel_spec_amd.parallel.timed_steps);
    # HIP events on the launch stream bracket the same K launches for the kernel-side figure.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    state = {"n": 0}

    def timed_step():
        if state["n"] == 0:
            ev0.record()
        step()
        state["n"] += 1
        if state["n"] == steps:
            ev1.record()

    elapsed = timed_steps(timed_step, torch.cuda.synchronize, steps, 0, dist if distributed else None, red_dev)
    my_kernel_ms = ev0.elapsed_time(ev1) / steps      # HIP events on the launch stream: a whole step (AUTO: the f32 kernel + the gated f64 launch)
    # dispersion of the figure: five more windows of the same K steps (at most 200), event-timed back to back behind the
    # timed region; `value` stays what the contract says (the K steps above), `value_ci` says how far a window of that length moves
    windows = []
    if primary:
        wsteps = max(1, min(steps, 200))
        for _ in range(5):
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record()
            for _ in range(wsteps):
                step()
            w1.record(); torch.cuda.synchronize()
            windows.append(frames_per_step * wsteps / (w0.elapsed_time(w1) * 1e-3))
    # the dominant kernel on its own: an event pair around the first kernel of each of 200 more calls (melspec_time_first_kernel), the
    # figure a kernel trace reports -- `roofline.achieved` is quoted on it; the step time above stays what `value` is made of
    first_ms = None
    if primary and sub == 1 and args.precision != "f64":
        torch.cuda.synchronize()
        first_ms = mel.time_first_kernel(pcm.data_ptr(), clip_len, clip_len, n_clips, out.data_ptr(), warmup=20, iters=200)
    kernel_ms = my_kernel_ms
    per_rank = [[float(frames_per_step), my_kernel_ms]]
    if distributed:
        mine = torch.tensor([float(frames_per_step), my_kernel_ms], dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)                   # per-rank {frames, ms}: tiny, the only collective of the run
        per_rank = [[float(t[0].item()), float(t[1].item())] for t in allr]
        kernel_ms = max(p[1] for p in per_rank)

    mel_name = mel.plain_kernel_name()
    res = dict(elapsed=elapsed, per_rank=per_rank, kernel_ms=kernel_ms, frames_per_step=frames_per_step, n_clips=n_clips, sub=sub,
               parity=parity, queued=queued, spinup_steps=spinup_steps, scaling=scaling, n_mels=n_mels, clip_seconds=clip_seconds,
               clip_len=clip_len, total_or_per=total_or_per, fpc=fpc, kernel=mel_name, mel=mel, out=out, pcm=pcm, stream=stream, first_ms=first_ms, windows=windows)
    return res


def speech_leg(M, torch, dev, stream, n_clips: int, clip_len: int, n_mels: int) -> dict:
    """The real-input figure next to `value` (never instead of it): the reference's own fixture, tests/golden/jfk_f32le.wav
    (/root/reference/testdata/jfk_f32le.wav, the signal its README numbers are quoted on), tiled with per-clip offsets to the
    config-2 batch, resident in HBM, default precision mode, same event-timed protocol.  Speech trips AUTO's guard on most
    frames, so the vote inside every launch hands the batch to the gated f64 kernel (DESIGN.md sections 4.9, 5)."""
    import numpy as np
    from oracle import oracle as O
    jfk = O.load_wav_f32(os.path.join(ROOT, "tests", "golden", "jfk_f32le.wav"))
    uniq = 64
    x = np.stack([np.resize(np.roll(jfk, -1237 * c), clip_len) for c in range(uniq)])
    pcm = torch.from_numpy(np.tile(x, (n_clips // uniq + 1, 1))[:n_clips].reshape(-1)).to(dev)
    mel = M.HipMelSpectrogram(N_FFT, HOP, SR, n_mels, device=dev.index)
    fpc = mel.num_frames(clip_len)
    out = torch.empty(n_clips * fpc * n_mels, dtype=torch.float32, device=dev)
    run = lambda: mel.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, n_clips, out.data_ptr(), stream=stream)
    regimes = []
    for _ in range(3):                      # every batch decides for itself (a vote inside its launch): the first one already runs on the f64 kernel
        mel.guard_last_count()
        run(); torch.cuda.synchronize()
        tripped = mel.guard_last_count()
        regimes.append(mel.auto_state()[0])
    o3 = out.view(n_clips, fpc, n_mels)
    worst = 0.0
    for c in (0, uniq - 1, n_clips - 1):
        want = O.compute_mel_spectrogram_cpu(x[c % uniq], N_FFT, HOP, n_mels, SR)
        worst = max(worst, float(np.abs(o3[c].cpu().numpy() - want).max()))
    if worst > 1e-4:
        raise SystemExit(f"speech leg: parity check failed, max|diff| = {worst}")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    k = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / k
    frames = n_clips * fpc
    gbs = frames * (HOP * 4 + n_mels * 4) / (ms * 1e-3) / 1e9
    res = {"input": f"tests/golden/jfk_f32le.wav tiled (per-clip offsets) to {n_clips} x {clip_len / SR:.0f} s, resident in HBM, precision mode auto",
           "ms": ms, "frames_per_s": frames / (ms * 1e-3), "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
           "frames_tripping_the_guard_per_step": tripped, "fraction_tripping": tripped / frames,
           "kernel_of_each_of_the_first_batches": ["gated f64 kernel (the launch's vote: heavy)" if h else "f32 kernel + f64 recompute of tripped frames" for h in regimes],
           "kernel": ("melspec::whisper400_six64_kernel<15, LensSix128, gated> (f64 FFT, six frames per wave, fifteen mel slots) behind the voting f32 launch" if n_mels == 128 else
                      "melspec::whisper400_six64_kernel<9, ., gated> (f64 FFT, six frames per wave) behind the voting f32 launch"), "parity_max_abs_diff": worst, "steps": k}
    v = valu_fields("speech128" if n_mels == 128 else "speech", res["frames_per_s"])
    if v is not None:
        res["valu"] = v
    mel.close()
    del pcm, out
    return res


LEG_NAMES = ("cfg3", "cfg3_split", "cfg4", "f64", "mel_major", "w512", "nemo", "nemo_f32")
F64_VECTOR_PEAK_TFLOPS = 78.6       # MI355X f64 vector peak (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs x 16 FMA lanes x 2 flops x 2.4 GHz


def _event_timed(torch, run, iters: int, spin_s: float = 0.15) -> float:
    """ms per launch: HIP events on torch's current stream (the stream every leg launches on) around `iters` launches, after a spin-up"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < spin_s:
        for _ in range(10):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


_ISA = None


def valu_fields(leg: str, frames_per_s: float):
    """The compute-side roofline of a leg (VERDICT r05 'missing' 4): VALU wave-instructions per frame of the kernel's unit loop, from the ISA
    histogram of the shipped library (tools/isa_legs.py -> profiles/isa_hist.json; static counts: one pass of the unit loop / frames per
    unit), and the f64 issue they amount to at the measured rate against the 78.6 TFLOP/s f64 vector peak (every f64 VALU instruction
    counted as one 64-lane FMA slot = 128 flops).  None when the table has no row for the leg."""
    global _ISA
    if _ISA is None:
        try:
            _ISA = json.load(open(os.path.join(ROOT, "profiles", "isa_hist.json")))
        except Exception:
            _ISA = {}
    row = (_ISA.get("legs") or {}).get(leg)
    if not row:
        return None
    fpu = float(row["frames_per_unit"])
    f64 = row.get("f64", 0) / fpu
    other = (row.get("valu32", 0) + row.get("cvt", 0) + row.get("dpp/lane", 0) + row.get("pk", 0)) / fpu
    out = {"f64_insts_per_frame": f64, "other_valu_per_frame": other, "lds_insts_per_frame": row.get("lds", 0) / fpu,
           "frac_of_f64_vector_peak": f64 * 128.0 * frames_per_s / (F64_VECTOR_PEAK_TFLOPS * 1e12),
           # a wave's v_fma_f64 measured alone issues every 6.2 cycles at two waves per SIMD, not every 4 (profiles/r04_mfma_f64_probe.txt): the same
           # count against THAT rate -- what the f64 pipe can be made to do on this part
           "frac_of_measured_f64_issue_rate": f64 * 128.0 * frames_per_s / (F64_VECTOR_PEAK_TFLOPS * 1e12) * (6.2 / 4.0),
           "kernel_symbol": row.get("kernel"), "isa_of_source_hash": _ISA.get("source_hash")}
    if row.get("note"):
        out["note"] = row["note"]
    try:
        from mel_spec_amd import build as hip_build
        if _ISA.get("source_hash") != hip_build.source_hash():
            out["stale"] = "profiles/isa_hist.json was made from other sources than this library (re-run tools/isa_legs.py)"
    except Exception:
        pass
    return out


def extra_legs(M, torch, dev, stream) -> dict:
    """The figures the driver would otherwise never see, NEXT TO `value`, never instead of it; N = 1, default run.
    Every leg: synthetic clips resident in HBM, a parity check of two clips against the oracle first, HIP events around >= 10 launches;
    {ms, frames_per_s, frac (algorithmic bytes / ms over 8 TB/s), valu (instructions per frame, fraction of the f64 vector peak), kernel,
    parity_max_abs_diff}.  A leg that fails -- a parity miss, an allocation that does not fit -- records {"error": ...} and the next
    one runs: nothing here can cost the line its `value`.  ~2 s of GPU time in all.
      cfg3       BASELINE configs[2]: Kaldi fbank (25 ms / 10 ms, 512-point FFT, 80 bins, pre-emphasis 0.97, Povey, CMN on), 1024 x 10 s
      cfg3_split the same batch through the additive split output (rows before CMN + means: no second pass over the rows)
      cfg4       BASELINE configs[3] at its stated size when the device has the room: Whisper large-v3, 128 mels, 8192 x 30 s (28.3 GB
                 resident); otherwise 1024 x 30 s with the reason in the record
      f64        configs[1] in MELSPEC_PRECISION_F64 (the f64 FFT on every frame: what speech costs without the vote)
      mel_major  configs[1] stored as interleave_frames(.., false, ..) = [mel][frames], the whisper.cpp layout (src/mel.rs:480-544)
      w512       Whisper at n_fft 512 / hop 160 / 80 mels -- the geometry of the reference's golden (src/rb.rs:134-179) -- default mode
      nemo / nemo_f32  the NeMo / Parakeet frontend, 128 mels, 1024 x 10 s: default mode (f64) and MELSPEC_PRECISION_F32 (the reference's f32)"""
    import gc
    import numpy as np
    from oracle import oracle as O
    legs = {}

    def record(leg, frames, bytes_per_frame, ms, kernel, parity, workload):
        gbs = frames * bytes_per_frame / (ms * 1e-3) / 1e9
        r = {"workload": workload, "ms": ms, "frames_per_s": frames / (ms * 1e-3), "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
             "kernel": kernel, "parity_max_abs_diff": parity}
        v = valu_fields(leg, r["frames_per_s"])
        if v is not None:
            r["valu"] = v
        return r

    def guarded(name, fn):
        try:
            legs[name] = fn()
        except BaseException as e:          # SystemExit of a parity miss included: the leg is lost, the line is not
            if isinstance(e, KeyboardInterrupt):
                raise
            legs[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
        gc.collect()
        try:
            torch.cuda.empty_cache()
        except Exception:
            pass

    n_clips, clip_len = 1024, 160000
    pcm = torch.empty(n_clips * clip_len, dtype=torch.float32, device=dev)
    M.synth_pcm_device(pcm.data_ptr(), clip_len, clip_len, 0, n_clips, stream=stream)

    def leg_cfg3():         # Kaldi fbank + CMN
        fb = M.Fbank(device=dev.index)
        try:
            fpc = fb.num_frames(clip_len)
            out = torch.empty(n_clips * fpc * 80, dtype=torch.float32, device=dev)
            run = lambda: fb.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, n_clips, out.data_ptr(), stream=stream)
            run(); torch.cuda.synchronize()
            o3 = out.view(n_clips, fpc, 80)
            worst = max(float(np.abs(o3[c].cpu().numpy() - O.fbank_compute(O.synth_pcm(c, clip_len))).max()) for c in (0, n_clips - 1))
            if worst > 1e-4:
                raise SystemExit(f"cfg3 leg: parity check failed, max|diff| = {worst}")
            return record("cfg3", n_clips * fpc, HOP * 4 + 80 * 4, _event_timed(torch, run, 100), fb.kernel_name() if hasattr(fb, "kernel_name") else
                          "melspec::fbank512_clip_kernel (f64 FFT, CMN inside)", worst,
                          "configs[2]: Kaldi fbank 80 bins + CMN on 1024 synthetic 10 s clips, resident in HBM")
        finally:
            fb.close()

    def leg_cfg3_split():   # the same batch as rows before CMN + the means (melspec_fbank_compute_uniform_device_split, additive, round 6)
        fb = M.Fbank(device=dev.index)
        try:
            fpc = fb.num_frames(clip_len)
            rows = torch.empty(n_clips * fpc * 80, dtype=torch.float32, device=dev)
            means = torch.empty(n_clips * 80, dtype=torch.float32, device=dev)
            run = lambda: fb.compute_uniform_device_split(pcm.data_ptr(), clip_len, clip_len, n_clips, rows.data_ptr(), means.data_ptr(), stream=stream)
            run(); torch.cuda.synchronize()
            got = (rows.view(n_clips, fpc, 80) - means.view(n_clips, 1, 80))
            worst = max(float(np.abs(got[c].cpu().numpy() - O.fbank_compute(O.synth_pcm(c, clip_len))).max()) for c in (0, n_clips - 1))
            if worst > 1e-4:
                raise SystemExit(f"cfg3_split leg: parity check failed, max|diff| = {worst}")
            return record("cfg3", n_clips * fpc, HOP * 4 + 80 * 4, _event_timed(torch, run, 100), "melspec::fbank512_clip_kernel (f64 FFT; rows before CMN + the clips' means)", worst,
                          "configs[2]'s batch as the split output {rows before CMN, means[clip][80]} for a consumer that folds the subtraction into its own read "
                          "(additive API; rows - means is bit for bit the fused output); parity checked on rows - means")
        finally:
            fb.close()

    want = {}

    def want80():
        if not want:
            want.update({c: O.compute_mel_spectrogram_cpu(O.synth_pcm(c, clip_len), N_FFT, HOP, 80, SR) for c in (0, n_clips - 1)})
        return want

    def leg_f64():          # configs[1] with the f64 FFT on every frame
        mel = M.HipMelSpectrogram(N_FFT, HOP, SR, 80, device=dev.index)
        try:
            fpc = mel.num_frames(clip_len)
            out = torch.empty(n_clips * fpc * 80, dtype=torch.float32, device=dev)
            mel.set_precision("f64")
            run = lambda: mel.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, n_clips, out.data_ptr(), stream=stream)
            run(); torch.cuda.synchronize()
            o3 = out.view(n_clips, fpc, 80)
            worst = max(float(np.abs(o3[c].cpu().numpy() - w).max()) for c, w in want80().items())
            if worst > 1e-4:
                raise SystemExit(f"f64 leg: parity check failed, max|diff| = {worst}")
            return record("f64", n_clips * fpc, HOP * 4 + 80 * 4, _event_timed(torch, run, 100), mel.plain_kernel_name(), worst,
                          "configs[1] (1024 x 10 s, 80 mels) with melspec_set_precision(f64): the f64 FFT on every frame")
        finally:
            mel.close()

    def leg_mel_major():    # configs[1] in the whisper.cpp layout
        mel = M.HipMelSpectrogram(N_FFT, HOP, SR, 80, device=dev.index)
        try:
            fpc = mel.num_frames(clip_len)
            W = mel.interleaved_width(clip_len, 0)
            outm = torch.empty(n_clips * 80 * W, dtype=torch.float32, device=dev)
            run = lambda: mel.compute_uniform_device_interleaved(pcm.data_ptr(), clip_len, clip_len, n_clips, outm.data_ptr(), False, 0, stream=stream)
            run(); torch.cuda.synchronize()
            om = outm.view(n_clips, 80, W)
            worst = max(float(np.abs(om[c].cpu().numpy() - O.interleave_frames(w, False, 0)).max()) for c, w in want80().items())
            if worst > 1e-4:
                raise SystemExit(f"mel-major leg: parity check failed, max|diff| = {worst}")
            return record("mel_major", n_clips * fpc, HOP * 4 + 80 * 4, _event_timed(torch, run, 100), "melspec::whisper400_six_kernel (mel-major store)", worst,
                          f"configs[1] stored mel-major [80][{W}] per clip (interleave_frames(.., false, 0), the whisper.cpp layout), default precision mode")
        finally:
            mel.close()

    def leg_single():       # the drop-in call on one clip, host memory in and out (PCIe-inclusive; README.md:117-123, src/cuda.rs:547-613)
        mel = M.HipMelSpectrogram(N_FFT, HOP, SR, 80, device=dev.index)
        try:
            single = {}
            for secs in (10, 60, 300):
                x = O.synth_pcm(1, int(secs * SR))
                mel.compute_mel_spectrogram(x)
                best, reps = 1e9, 10
                for _ in range(3):                  # best of three: the staging threads of a > 16 MiB call share the host with torch's pools
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        y = mel.compute_mel_spectrogram(x)
                    best = min(best, (time.perf_counter() - t0) / reps * 1e3)
                single[f"{secs}s"] = {"ms": best, "frames": int(y.shape[0])}
            return single
        finally:
            mel.close()

    def leg_w512():         # the geometry of the reference's value-level golden, plain batch, default mode
        mel = M.HipMelSpectrogram(512, HOP, SR, 80, device=dev.index)
        try:
            fpc = mel.num_frames(clip_len)
            out = torch.empty(n_clips * fpc * 80, dtype=torch.float32, device=dev)
            run = lambda: mel.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, n_clips, out.data_ptr(), stream=stream)
            run(); torch.cuda.synchronize()
            o3 = out.view(n_clips, fpc, 80)
            worst = max(float(np.abs(o3[c].cpu().numpy() - O.compute_mel_spectrogram_cpu(O.synth_pcm(c, clip_len), 512, HOP, 80, SR)).max()) for c in (0, n_clips - 1))
            if worst > 1e-4:
                raise SystemExit(f"w512 leg: parity check failed, max|diff| = {worst}")
            return record("w512", n_clips * fpc, HOP * 4 + 80 * 4, _event_timed(torch, run, 100), mel.plain_kernel_name(), worst,
                          "Whisper at n_fft 512 / hop 160 / 80 mels (the geometry of rust_jfk_golden.npy, src/rb.rs:134-179) on configs[1]'s 1024 x 10 s clips, default precision mode")
        finally:
            mel.close()

    def leg_nemo(mode, key):
        # the NeMo / Parakeet frontend (BatchLogMelSpectrogram, 128 mels, pre-emphasis 0.97): the default mode (f64 up to |X|^2,
        # 1e-4 from the f64 evaluation of the definition) and MELSPEC_PRECISION_F32, the reference's own arithmetic type for this frontend
        # (src/mel.rs:251-252,356-357), gated like tests/test_f32_512.py: within 4 x the distance of upstream's literal f32 arithmetic (the
        # oracle's f64 = False restatement) from the same f64 evaluation on the same clips
        fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24), device=dev.index)
        try:
            cols = fe.padded_frames(clip_len)
            out = torch.empty(n_clips * 128 * cols, dtype=torch.float32, device=dev)
            ocfg = O.blm_default_config(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24)
            wantn = {c: O.blm_compute(O.synth_pcm(c, clip_len), ocfg, True)[0] for c in (0, n_clips - 1)}
            room = max(float(np.abs(O.blm_compute(O.synth_pcm(c, clip_len), ocfg, False)[0] - wantn[c]).max()) for c in wantn)
            tol = 1e-4 if mode == "auto" else max(1e-4, 4.0 * room)
            run = lambda: fe.compute_uniform_device(pcm.data_ptr(), clip_len, clip_len, n_clips, out.data_ptr(), stream=stream)
            fe.set_precision(mode)
            run(); torch.cuda.synchronize()
            o3 = out.view(n_clips, 128, cols)
            worst = max(float(np.abs(o3[c].cpu().numpy() - wantn[c]).max()) for c in wantn)
            if worst > tol:
                raise SystemExit(f"{key} leg: parity check failed, max|diff| = {worst}")
            r = record(key, n_clips * cols, HOP * 4 + 128 * 4, _event_timed(torch, run, 100),
                       "melspec::fbank512_wave_kernel<float, 12 waves, NeMo> (f32, feature-major rows staged through LDS)" if fe.precision == "f32"
                       else "melspec::fbank512_wave_kernel<double, 8 waves, NeMo> (f64 FFT)", worst,
                       f"SURVEY 8(f) #1: BatchLogMelSpectrogram 128 mels, pre-emphasis 0.97, 1024 synthetic 10 s clips, feature-major [128][{cols}] per clip, precision {fe.precision}")
            r["reference_f32_max_abs_diff"] = room
            return r
        finally:
            fe.close()

    guarded("cfg3", leg_cfg3)
    guarded("cfg3_split", leg_cfg3_split)
    guarded("f64", leg_f64)
    guarded("mel_major", leg_mel_major)
    guarded("host_api_single_clip_ms", leg_single)
    guarded("w512", leg_w512)
    guarded("nemo", lambda: leg_nemo("auto", "nemo"))
    guarded("nemo_f32", lambda: leg_nemo("f32", "nemo_f32"))
    del pcm
    gc.collect(); torch.cuda.empty_cache()

    def leg_cfg4():         # BASELINE configs[3]: 128 mels, 30 s clips -- at its stated size when it fits
        clip_len4, full = 480000, CONFIGS[4][0]
        mel = M.HipMelSpectrogram(N_FFT, HOP, SR, 128, device=dev.index)
        try:
            fpc = mel.num_frames(clip_len4)
            per_clip = clip_len4 * 4 + fpc * 128 * 4 + fpc * 4 / 5 * 1.05           # PCM + mel + the guard's note list
            free_b, _ = torch.cuda.mem_get_info(dev)
            n4, why = full, None
            if full * per_clip > 0.85 * free_b:
                n4 = 1024
                why = f"scaled: the full {full} x 30 s needs {full * per_clip / 1e9:.1f} GB resident and the device has {free_b / 1e9:.1f} GB free"
            pcm4 = torch.empty(n4 * clip_len4, dtype=torch.float32, device=dev)
            out = torch.empty(n4 * fpc * 128, dtype=torch.float32, device=dev)
            M.synth_pcm_device(pcm4.data_ptr(), clip_len4, clip_len4, 0, n4, stream=stream)
            run = lambda: mel.compute_uniform_device(pcm4.data_ptr(), clip_len4, clip_len4, n4, out.data_ptr(), stream=stream)
            run(); torch.cuda.synchronize()
            o3 = out.view(n4, fpc, 128)
            worst = max(float(np.abs(o3[c].cpu().numpy() - O.compute_mel_spectrogram_cpu(O.synth_pcm(c, clip_len4), N_FFT, HOP, 128, SR)).max()) for c in (0, n4 // 2 + 1, n4 - 1))
            if worst > 1e-4:
                raise SystemExit(f"cfg4 leg: parity check failed, max|diff| = {worst}")
            r = record("cfg4", n4 * fpc, HOP * 4 + 128 * 4, _event_timed(torch, run, 10 if n4 == full else 50, spin_s=0.3), mel.plain_kernel_name(), worst,
                       f"configs[3]: Whisper large-v3, 128 mels, {n4} synthetic 30 s clips resident in HBM ({(n4 * per_clip) / 1e9:.1f} GB), default precision "
                       "mode (step = f32 kernel + the gated f64 launch)")
            r["clips"], r["full_size"] = n4, n4 == full
            if why:
                r["scaled_because"] = why
            return r
        finally:
            mel.close()

    guarded("cfg4", leg_cfg4)
    return legs


def main() -> None:
    args = parse_args()
    explicit_config = args.explicit_config = args.config is not None
    if args.config is None:
        args.config = 2
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        respawn_under_torchrun(args)          # does not return
    if args.dry_run:
        return dry_run(args)
    import numpy as np
    import torch
    import mel_spec_amd as M
    from mel_spec_amd.parallel import shard_range, timed_steps

    if args.force_dist and not launched:          # a one-rank job with its own rendezvous on the loopback
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    distributed = world > 1 or args.force_dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}

    steps = args.steps if args.steps is not None else (1000 if args.config == 2 else 20)
    warmup = args.warmup if args.warmup is not None else (100 if args.config == 2 else 3)
    w = run_workload(args, args.config, True, steps, warmup, M, torch, dist, dev, rank, world, local_rank)
    elapsed, per_rank, kernel_ms, frames_per_step = w["elapsed"], w["per_rank"], w["kernel_ms"], w["frames_per_step"]
    n_clips, sub, parity, queued, spinup_steps, scaling = w["n_clips"], w["sub"], w["parity"], w["queued"], w["spinup_steps"], w["scaling"]
    n_mels, clip_seconds, clip_len, total_or_per, fpc = w["n_mels"], w["clip_seconds"], w["clip_len"], w["total_or_per"], w["fpc"]
    mel, out, stream = w["mel"], w["out"], w["stream"]

    # which device every rank ran on: an N-GPU line must show N distinct devices
    props = torch.cuda.get_device_properties(dev)
    ident = {"rank": rank, "device_index": dev.index, "device_name": props.name,
             "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None}
    idents = [ident]
    if distributed:
        idents = [None] * world
        dist.all_gather_object(idents, ident)

    if world > 1 and not args.one_device:
        # N ranks on fewer than N devices is not an N-GPU measurement: refuse to print a line that looks like one
        distinct = len({(i or {}).get("uuid") or (i or {}).get("pci_bus_id") or (i or {}).get("device_index") for i in idents})
        if distinct != world:
            raise SystemExit(f"bench.py: {world} ranks ran on {distinct} distinct device(s) ({idents}); LOCAL_RANK -> device mapping is broken")

    gather = None
    if world > 1 and args.gather and sub == 1 and not args.one_device:
        # optional consolidation on rank 0: every peer sends its share over its own xGMI link
        n_out = fpc * n_clips * n_mels
        sizes = [int(p[0]) * n_mels for p in per_rank]
        bufs = [torch.empty(sz, dtype=torch.float32, device=dev) for sz in sizes[1:]] if rank == 0 else []
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if rank == 0:
            reqs = [dist.irecv(b, src=r + 1) for r, b in enumerate(bufs)]
        else:
            reqs = [dist.isend(out[:n_out], dst=0)]
        for r in reqs:
            r.wait()
        torch.cuda.synchronize(); dist.barrier()
        gdt = time.perf_counter() - t0
        if rank == 0:
            gb = sum(sizes[1:]) * 4 / 1e9
            gather = {"seconds": gdt, "gigabytes_into_rank0": gb, "GB_per_s": gb / gdt, "note": "RCCL send/recv, not part of `value`"}

    host_io = None
    if not args.no_host_io and rank == 0 and world == 1:          # like cpu_baseline: at N = 1 only
        from oracle import oracle as O
        # 64 host clips per call through melspec_compute_batch_host, caller-owned buffers reused from call to call (a fresh numpy
        # output per call measures the page faults of its first touch instead): pageable, then pinned (melspec_host_alloc)
        n_host = 64
        x = np.stack([O.synth_pcm(c, clip_len) for c in range(n_host)]).reshape(-1)
        offs = np.arange(n_host, dtype=np.uint64) * np.uint64(clip_len)
        lens = np.full(n_host, clip_len, np.uint64)
        frames_host = n_host * mel.num_frames(clip_len)
        out_h = np.empty(frames_host * n_mels, np.float32)
        reps = 10

        def host_rate(src, dst):
            mel.compute_batch_host(src, offs, lens, dst)
            best = 0.0
            for _ in range(3):                      # the staging memcpy threads share the host with whatever else runs: best of three
                t1 = time.perf_counter()
                for _ in range(reps):
                    mel.compute_batch_host(src, offs, lens, dst)
                best = max(best, reps * frames_host / (time.perf_counter() - t1))
            return best

        host_io = host_rate(x, out_h)
        pin_in, pin_out = M.HostBuffer(x.size), M.HostBuffer(out_h.size)
        pin_in.array[:] = x
        host_io_pinned = host_rate(pin_in.array, pin_out.array)
        pin_in.free(); pin_out.free()

    # the real-input leg (N = 1, the default workload) and, for N > 1 without --config, north_star's 65 536 x 30 s per-clip split
    speech = speech128 = None
    if rank == 0 and world == 1 and args.config == 2 and not args.no_speech and args.precision == "auto" and args.n_mels is None:
        def leg_or_error(fn):           # like the extra legs: a failing leg costs its own entry, never the line
            try:
                return fn()
            except BaseException as e:
                if isinstance(e, KeyboardInterrupt):
                    raise
                return {"error": f"{type(e).__name__}: {e}"[:400]}
        speech = leg_or_error(lambda: speech_leg(M, torch, dev, stream, n_clips, clip_len, n_mels))
        # the same real input through Whisper large-v3's bank (configs[3]'s 128 mels, on the config-2 batch): the kernel real large-v3 input runs on
        speech128 = leg_or_error(lambda: speech_leg(M, torch, dev, stream, n_clips, clip_len, 128))
    legs = None
    if rank == 0 and world == 1 and args.config == 2 and not args.no_legs and args.precision == "auto" and args.n_mels is None and args.clips is None and args.clip_seconds is None:
        try:
            legs = extra_legs(M, torch, dev, stream)
        except BaseException as e:              # its own set-up (the shared PCM buffer) failed: every leg is lost, the line is not
            if isinstance(e, KeyboardInterrupt):
                raise
            legs = {k: {"error": f"{type(e).__name__}: {e}"[:400]} for k in LEG_NAMES}
    cfg5 = None
    if world > 1 and not explicit_config and not args.no_cfg5:
        kernel_name = mel.plain_kernel_name()
        mel.close()
        del w, out
        torch.cuda.empty_cache()
        steps5, warmup5 = 10, 2
        w5 = run_workload(args, 5, False, steps5, warmup5, M, torch, dist, dev, rank, world, local_rank)
        frames5 = sum(p[0] for p in w5["per_rank"])
        cfg5 = {"workload": f"configs[4]: {w5['total_or_per']} synthetic {CONFIGS[5][1]} s f32 clips @16 kHz split per clip over {world} GPU(s) "
                            f"(mel_spec_amd.parallel.shard_range), Whisper n_fft={N_FFT} hop={HOP} n_mels={CONFIGS[5][2]}, resident in HBM",
                "scaling": "strong", "value": frames5 * steps5 / w5["elapsed"], "unit": "mel frames/s", "steps": steps5, "warmup": warmup5,
                "ms_per_step": w5["elapsed"] / steps5 * 1e3, "frames_per_step_all_ranks": frames5,
                "sub_shards_per_step": w5["sub"], "parity_max_abs_diff": w5["parity"],
                "per_rank": [{"frames_per_step": p[0], "kernel_ms": p[1]} for p in w5["per_rank"]],
                "roofline_frac": frames5 * (HOP * 4 + CONFIGS[5][2] * 4) / world / (w5["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        w5["mel"].close()
        mel = None
    else:
        kernel_name = mel.plain_kernel_name()

    if rank == 0:
        total_frames = sum(p[0] for p in per_rank) * steps
        value = total_frames / elapsed
        bytes_per_frame = HOP * 4 + n_mels * 4
        algo_bytes_per_launch = frames_per_step * bytes_per_frame
        dominant_ms = w.get("first_ms") if (world == 1 and cfg5 is None and w.get("first_ms")) else kernel_ms
        achieved = algo_bytes_per_launch / (dominant_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        traffic_kernel_ok = args.clips is None and args.clip_seconds is None and args.n_mels is None     # the child run repeats the default workload
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.config == 2:
            try:
                tj = json.load(open(tpath))
                if tj.get("clips") == n_clips and tj.get("clip_seconds") == clip_seconds and tj.get("n_mels") == n_mels:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = (f"committed rocprofv3 PMC passes of this command (profiles/traffic.json, source {tj.get('source')}): "
                                      "2 x FETCH_SIZE + WRITE_SIZE per launch; not re-measured in this run")
            except Exception:
                traffic = None
        if not args.no_traffic and world == 1 and args.config == 2 and traffic_kernel_ok:
            live, note = measure_traffic(args.config)
            if live is not None:
                traffic, traffic_source = live, note
            elif traffic_source is not None:
                traffic_source += f" ({note})"
        res = {
            "metric": "mel frames/sec/GPU (Whisper 400/160/80 @16 kHz); realtime x vs CPU ref",
            "value": value, "unit": "mel frames/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32" if args.precision != "f64" else "f64", "data": "synthetic",
            "config": {"workload": (f"configs[{args.config - 1}]: " + (
                           f"batched {n_clips} synthetic {clip_seconds} s f32 clips @16 kHz per GPU" if scaling == "weak" else
                           f"{total_or_per} synthetic {clip_seconds} s f32 clips @16 kHz split per clip over {world} GPU(s), {n_clips} on rank 0")
                           + f", Whisper n_fft={N_FFT} hop={HOP} n_mels={n_mels}, PCM and mel resident in HBM"),
                       "clips_per_gpu": n_clips, "frames_per_step_per_gpu": frames_per_step,
                       "sub_shards_per_step": sub,
                       "precision": f"melspec_set_precision({args.precision}): " + (
                           "f32 FFT; a vote inside the launch decides between the f64 recompute of the frames failing the error bound and the gated f64 kernel for the whole batch" if args.precision == "auto" else
                           ("f64 FFT on every frame" if args.precision == "f64" else "f32 FFT, no guard")),
                       "frames_recomputed_in_f64_per_step": queued,
                       "parallelism": f"per-clip split x{world}, no data-path collective"},
            "per_gpu_frames_per_s": value / world,
            "value_ci": ({"windows": len(w["windows"]), "steps_per_window": max(1, min(steps, 200)), "min": min(w["windows"]), "median": sorted(w["windows"])[len(w["windows"]) // 2],
                          "max": max(w["windows"]), "note": "rank 0's frames/s over five more event-timed windows of the same K steps (at most 200) right behind the timed region; `value` is the K steps of the contract"}
                         if (cfg5 is None and w.get("windows")) else None),
            "per_rank": [dict({"frames_per_step": p[0], "kernel_ms": p[1]}, **(idents[r] or {})) for r, p in enumerate(per_rank)],
            "realtime_x": value * (HOP / SR),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name,
                         "kernel_ms": dominant_ms,
                         "step_ms_events": kernel_ms,
                         "frac_of_the_whole_step": algo_bytes_per_launch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_note": ("average launch duration of the dominant kernel from a HIP event pair around it in each of 200 calls after the timed region "
                                            "(melspec_time_first_kernel; `step_ms_events` = HIP events around the K timed steps / K, which in the default mode also "
                                            "holds the gated f64 launch behind every f32 launch, ~5 us that return at once on this input)" if dominant_ms is not kernel_ms else
                                            "HIP events on the launch stream around the K timed steps / K"),
                         "algorithmic_bytes_per_launch": algo_bytes_per_launch,
                         "valu": valu_fields("value" if (args.config == 2 and n_mels == 80 and args.precision != "f64") else ("cfg4" if (n_mels == 128 and args.precision != "f64") else "none"),
                                             frames_per_step / (dominant_ms * 1e-3))},
            "parity_max_abs_diff": parity, "spinup_steps_untimed": spinup_steps,
        }
        if host_io is not None:
            res["host_api_frames_per_s_pcie_inclusive"] = host_io
            res["host_api_pinned_frames_per_s_pcie_inclusive"] = host_io_pinned
            res["host_api_note"] = ("64 host clips per call through melspec_compute_batch_host (H2D + kernels + D2H, chunked and overlapped), caller's buffers "
                                    "reused: pageable memory, and memory from melspec_host_alloc (pinned); never `value`")
        if speech is not None:
            res["config"]["speech"] = speech
        if speech128 is not None:
            res["config"]["speech128"] = speech128
        if legs is not None:
            res["host_api_single_clip_ms"] = legs.pop("host_api_single_clip_ms", None)
            res["host_api_single_clip_note"] = ("HipMelSpectrogram::compute_mel_spectrogram on ONE clip of 10 / 60 / 300 s, host memory in and out (PCIe-inclusive, never `value`): "
                                                "the shape of the reference's published figures (README.md:117-123) and of its #[ignore] benches (src/cuda.rs:547-613)")
            for k in LEG_NAMES:
                res["config"][k] = legs.get(k, {"error": "leg did not run"})
        if cfg5 is not None:
            res["config"]["cfg5"] = cfg5
        if gather is not None:
            res["gather_to_rank0"] = gather
        if distributed:
            res["dist"] = dict(dist_info, distinct_devices=len({(i or {}).get("uuid") or (i or {}).get("device_index") for i in idents}))
        if args.one_device:
            res["rehearsal"] = "every rank on GPU 0 over gloo: exercises the N > 1 code path on a 1-GPU box, the figures are not measurements"
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(clip_len, n_mels, args.cpu_seconds, 1024)
            res["cpu_baseline"] = cb
            res["realtime_x_vs_cpu"] = value / cb["value"]
        print(json.dumps(res), flush=True)

    if mel is not None:
        mel.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

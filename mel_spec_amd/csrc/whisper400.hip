// whisper400.hip -- launchers of the fused n_fft = 400 kernels (whisper400_kernels.hpp): launch_ctx picks the kernel a batch of a
// melspec_ctx runs on -- f32 with the precision guard and the vote, or f64 -- and launch_stft exports the spectrum (row a3).
#include "host_common.hpp"
#include "whisper400_kernels.hpp"
namespace melspec {
// emitted by melspec_runs.hip (compiled with its own scheduling strategy; see there)
extern template __global__ void whisper400_six_runs_kernel<kSixMaxSlots, LensSix80>(const FastParams);
extern template __global__ void whisper400_wave_runs_kernel<8, LensI80>(const FastParams);
extern template __global__ void whisper400_wave_runs_kernel<12, LensI128>(const FastParams);
extern template __global__ void whisper400_six_wide_runs_kernel<kSixWideSlots, LensSix128>(const FastParams);
}  // namespace melspec

namespace melspec {
namespace host {

// MELSPEC_PRECISION_AUTO: take in what the finished launches published (reporting only: melspec_auto_state)
void auto_poll(melspec_ctx *c) {
    FixState &fx = c->fix;
    if (!fx.host) return;
    const volatile unsigned long long *h = fx.host;
    const unsigned long long a = h[0], b = h[1];
    const uint32_t seq = static_cast<uint32_t>(a >> kStatShift);
    if (seq != static_cast<uint32_t>(b >> kStatShift) || seq == fx.seen_seq) return;     // a launch is publishing right now, or nothing new
    fx.seen_seq = seq;
    const unsigned long long tripped = a & kStatMask, frames = b & kStatMask & ~kStatFromGated;
    fx.heavy = (b & kStatFromGated) != 0;
    if (frames < kAutoMinFrames) return;
    fx.fraction = static_cast<double>(tripped) / static_cast<double>(frames);
}

int auto_sink(melspec_ctx *c, const BatchDesc &desc, hipStream_t stream, bool with_vote, FixSink &sink) {
    FixState &fx = c->fix;
    if (fx.used && fx.last_stream != stream) HIP_TRY(hipStreamSynchronize(fx.last_stream));
    const size_t need = (static_cast<size_t>(desc.n_units) + 65536) * sizeof(uint64_t);      // one note per unit + a round of slack
    if (need > fx.list.cap) {
        if (fx.used) HIP_TRY(hipStreamSynchronize(fx.last_stream));       // a launch in flight may still write the old list
        int rc = fx.list.ensure(need);
        if (rc) return rc;
    }
    sink.tab = static_cast<const double *>(fx.tab.p);
    sink.list = static_cast<uint64_t *>(fx.list.p);
    if (!fx.host) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&fx.host), 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(fx.host, 0, 64);
    }
    fx.used = true; fx.last_stream = stream;
    sink.count = static_cast<unsigned long long *>(fx.count.p);
    sink.acc = sink.count + 1;
    sink.host = fx.host;
    if (with_vote) {
        sink.vote = sink.count + 2;
        sink.decision = static_cast<unsigned *>(fx.verdicts.p);
    }
    return MELSPEC_OK;
}

// the launch-specific part of a guarded launch's statistics sink (the grid is only known where the launch is made)
FixSink sink_armed(melspec_ctx *c, FixSink sink, const BatchDesc &desc, unsigned grid) {
    if (!sink.acc) return sink;
    sink.frames = desc.stat_frames ? desc.stat_frames
                : desc.d_unit_prefix == nullptr ? static_cast<uint64_t>(desc.n_clips) * desc.frames_per_clip
                                                : desc.n_units * static_cast<uint64_t>(desc.frames_per_unit);   // device-planned ragged: upper bound
    sink.n_groups = grid;
    sink.seq = (c->fix.seq = (c->fix.seq + 1) & 0xffffffu) ? c->fix.seq : (c->fix.seq = 1);      // never 0: the host's "nothing seen yet"
    return sink;
}

PreciseParams precise_params(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat) {
    PreciseParams pp{};
    pp.b = desc;
    pp.stat = stat;
    pp.d_blob = static_cast<const uint32_t *>(c->d_blob64.p);
    pp.blob_words = static_cast<int>(c->pt.blob.size());
    pp.mel_off_words = c->pt.mel_off_words;
    pp.hop = c->hop_size;
    pp.n_mels = c->n_mels;
    pp.slots = c->ft.slots;
    return pp;
}

// the f64 kernel on the whole batch: MELSPEC_PRECISION_F64 (the plan is its own, kFPW frames per unit), or -- gate != nullptr -- AUTO's
// second launch, which runs only when the f32 launch in front of it voted "heavy" and walks THAT launch's plan (plain batches)
template <int NSLOTS, class Lens>
int launch_precise_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate, unsigned gate_value) {
    static std::atomic<uint64_t> attr_done{0};          // one bit per device: function attributes are per device
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_precise_kernel<NSLOTS, Lens, 0>, "hipFuncSetAttribute(whisper400_precise_kernel)");
        if (!rc) rc = allow_big_lds(&whisper400_precise_kernel<NSLOTS, Lens, 1>, "hipFuncSetAttribute(whisper400_precise_kernel, runs)");
        if (!rc) rc = allow_big_lds(&whisper400_precise_kernel<NSLOTS, Lens, 2>, "hipFuncSetAttribute(whisper400_precise_kernel, gated)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const bool walk = gate && !(desc.mel_major || desc.out_width != desc.frames_per_clip);      // gated layouts come with a plan of their own
    const uint64_t steps = walk ? (desc.n_units * static_cast<uint64_t>(desc.frames_per_unit) + kFPW - 1) / kFPW : desc.n_units;
    const uint64_t blocks = (steps + kPreciseWaves - 1) / kPreciseWaves;
    static const int per_cu = lab_int("MELSPEC_PRECISE_GRID_PER_CU", 1, 1, 4096);   // one workgroup is resident per CU
    const unsigned grid = grid_for_xcd(blocks, c->dev.cus, per_cu);
    FixSink armed = sink_armed(c, stat, desc, grid);
    if (gate) armed.frames |= kStatFromGated;
    PreciseParams pp = precise_params(c, desc, armed);
    pp.gate = gate; pp.gate_value = gate_value; pp.plan_fpu = desc.frames_per_unit;
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (gate && !layout)
        hipLaunchKernelGGL((whisper400_precise_kernel<NSLOTS, Lens, 2>), dim3(grid), dim3(kPreciseWaves * 64), c->precise_lds, stream, pp);
    else if (layout)
        hipLaunchKernelGGL((whisper400_precise_kernel<NSLOTS, Lens, 0>), dim3(grid), dim3(kPreciseWaves * 64), c->precise_lds, stream, pp);
    else
        hipLaunchKernelGGL((whisper400_precise_kernel<NSLOTS, Lens, 1>), dim3(grid), dim3(kPreciseWaves * 64), c->precise_lds, stream, pp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_precise(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate = nullptr, unsigned gate_value = 0) {
    if (c->ft.slots.n_slots <= 8)
        return c->lens_kind == 1 ? launch_precise_t<8, LensI80>(c, desc, stat, stream, gate, gate_value) : launch_precise_t<8, LensRuntime>(c, desc, stat, stream, gate, gate_value);
    return c->lens_kind == 2 ? launch_precise_t<12, LensI128>(c, desc, stat, stream, gate, gate_value) : launch_precise_t<12, LensRuntime>(c, desc, stat, stream, gate, gate_value);
}

// the f64 six-frame kernel on a plain batch planned in six-frame units: MELSPEC_PRECISION_F64, or -- gate != nullptr -- AUTO's second
// launch over the plan of the f32 launch in front of it
template <class Lens, int NS = kSixMaxSlots>
int launch_six64_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate, unsigned gate_value) {
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_six64_kernel<NS, Lens>, "hipFuncSetAttribute(whisper400_six64_kernel)");
        if constexpr (Lens::kStatic && NS == kSixMaxSlots)          // the layout form exists for the compile-time banks of up to 80 mels only (six64_layout_ok)
            if (!rc) rc = allow_big_lds(&whisper400_six64_layout_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six64_layout_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t blocks = (desc.n_units + kSix64Waves - 1) / kSix64Waves;
    static const int per_cu = lab_int("MELSPEC_SIX64_GRID_PER_CU", 1, 1, 4096);   // one 12-wave workgroup is resident per CU
    const unsigned grid = grid_for_xcd(blocks, c->dev.cus, per_cu);
    FixSink armed = sink_armed(c, stat, desc, grid);
    if (gate) armed.frames |= kStatFromGated;
    Six64Params pp{};
    pp.b = desc;
    pp.stat = armed;
    pp.d_blob = static_cast<const uint32_t *>(c->d_blob64x.p);
    pp.blob_words = static_cast<int>(c->t64.blob.size());
    pp.mel_off_words = c->t64.mel_off_words;
    pp.hop = c->hop_size;
    pp.n_mels = c->n_mels;
    pp.slots = c->ft6.slots;
    pp.gate = gate; pp.gate_value = gate_value;
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (layout) {
        if constexpr (Lens::kStatic && NS == kSixMaxSlots) hipLaunchKernelGGL((whisper400_six64_layout_kernel<kSixMaxSlots, Lens>), dim3(grid), dim3(kSix64Waves * 64), c->lds64x, stream, pp);
        else return fail(MELSPEC_ERR_INTERNAL, "whisper400_six64_layout_kernel has no run-time-lens form");      // launch_ctx never asks (six64_layout_ok)
    } else {
        hipLaunchKernelGGL((whisper400_six64_kernel<NS, Lens>), dim3(grid), dim3(kSix64Waves * 64), c->lds64x, stream, pp);
    }
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_six64(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate = nullptr, unsigned gate_value = 0) {
    if (c->six64_wide) return launch_six64_t<LensSix128, kSixWideSlots>(c, desc, stat, stream, gate, gate_value);
    return c->six_static == 1 ? launch_six64_t<LensSix80>(c, desc, stat, stream, gate, gate_value)
         : c->six_static == 2 ? launch_six64_t<LensSix64>(c, desc, stat, stream, gate, gate_value)
         : c->six_static == 3 ? launch_six64_t<LensSix40>(c, desc, stat, stream, gate, gate_value)
                              : launch_six64_t<LensRuntime>(c, desc, stat, stream, gate, gate_value);
}

FastParams fast_params(const BatchDesc &desc, const FastTables &ft, const DevBuf &blob, melspec_ctx *c, const FixSink &sink) {
    FastParams fp{};
    fp.b = desc;
    fp.d_blob = static_cast<const float *>(blob.p);
    fp.blob_len = static_cast<int>(ft.blob.size());
    fp.hop = c->hop_size;
    fp.n_mels = c->n_mels;
    fp.slice_floats = WaveLayout::slice_floats();
    fp.slots = ft.slots;
    fp.fix = sink;
    return fp;
}

// 5-frame f32 kernels: plain batches (uniform, ragged) on contiguous runs of units per wave, layouts round-robin
template <int NSLOTS, class Lens>
int launch_wave_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &sink, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_wave_kernel<NSLOTS, Lens>, "hipFuncSetAttribute(whisper400_wave_kernel)");
        if (!rc) rc = allow_big_lds(&whisper400_wave_runs_kernel<NSLOTS, Lens>, "hipFuncSetAttribute(whisper400_wave_runs_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t blocks = (desc.n_units + kWaveWaves - 1) / kWaveWaves;
    // two workgroups are resident per CU; 4 per CU measured best (8192 x 15..45 s x 128 mels: 9.17 vs 9.50 ms)
    static const int per_cu = lab_int("MELSPEC_GRID_PER_CU", 4, 1, 64);
    const unsigned grid = grid_for_xcd(blocks, c->dev.cus, per_cu);
    FixSink armed = sink_armed(c, sink, desc, grid);
    armed.vote_groups = std::min<unsigned>(grid, static_cast<unsigned>(c->dev.cus));           // workgroups that are certainly resident when the launch starts
    const FastParams fp = fast_params(desc, c->ft, c->d_blob, c, armed);
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (layout)
        hipLaunchKernelGGL((whisper400_wave_kernel<NSLOTS, Lens>), dim3(grid), dim3(kWaveWaves * 64), c->fast_lds, stream, fp);
    else
        hipLaunchKernelGGL((whisper400_wave_runs_kernel<NSLOTS, Lens>), dim3(grid), dim3(kWaveWaves * 64), c->fast_lds, stream, fp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_wave(melspec_ctx *c, const BatchDesc &desc, const FixSink &sink, hipStream_t stream) {
    if (c->ft.slots.n_slots <= 8)
        return c->lens_kind == 1 ? launch_wave_t<8, LensI80>(c, desc, sink, stream) : launch_wave_t<8, LensRuntime>(c, desc, sink, stream);
    return c->lens_kind == 2 ? launch_wave_t<12, LensI128>(c, desc, sink, stream) : launch_wave_t<12, LensRuntime>(c, desc, sink, stream);
}

// Which batches of the six-frame family run on TWELVE waves per CU (whisper400_six_wide_*: three waves per SIMD, 168 VGPRs): the 128-mel bank
// (fifteen slots) and the compile-time banks of 64 and 40 mels, whose slot lengths (two slots of ten intervals; slots of eleven and fourteen)
// made the sixteen-wave kernels reload spilled registers inside the unit loop (tools/isa_legs.py lists such kernels).  Measured at
// 1024 x 10 s, sixteen -> twelve waves (profiles/r06_wide_layouts.txt): 64 mels plain 0.381-0.386 -> 0.309-0.314 ms, mel-major 0.491-0.501
// -> 0.350-0.357; 40 mels plain 0.2979 -> 0.2951, mel-major 0.373 -> 0.335.  The 80-mel bank does not spill at sixteen and stays there
// (twelve: plain 0.298 -> 0.306, mel-major 0.340 -> 0.350), as do the run-time banks.
bool twelve_waves_for(const melspec_ctx *c, bool /*layout*/) {
    return c->six_wide32 || (c->six && (c->six_static == 2 || c->six_static == 3));
}

template <class Lens>
int launch_six_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &sink, hipStream_t stream) {
    constexpr bool kTwelve = std::is_same_v<Lens, LensSix64> || std::is_same_v<Lens, LensSix40>;       // twelve_waves_for
    constexpr int kWaves = kTwelve ? kSixWideWaves : kSixWaves;
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc;
        if constexpr (kTwelve) {
            rc = allow_big_lds(&whisper400_six_wide_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six_wide_kernel<9, .>)");
            if (!rc) rc = allow_big_lds(&whisper400_six_wide_runs_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six_wide_runs_kernel<9, .>)");
        } else {
            rc = allow_big_lds(&whisper400_six_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six_kernel)");
            if (!rc) rc = allow_big_lds(&whisper400_six_runs_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six_runs_kernel)");
        }
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t blocks = (desc.n_units + kWaves - 1) / kWaves;
    static const int per_cu = lab_int("MELSPEC_SIX_GRID_PER_CU", 1, 1, 4096);     // one workgroup per CU
    const dim3 grid(grid_for_xcd(blocks, c->dev.cus, per_cu)), block(kWaves * 64);
    FixSink armed = sink_armed(c, sink, desc, grid.x);
    armed.vote_groups = std::min<unsigned>(grid.x, static_cast<unsigned>(c->dev.cus));        // the workgroups resident when the launch starts (one per CU)
    const FastParams fp = fast_params(desc, c->ft6, c->d_blob6, c, armed);
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if constexpr (kTwelve) {
        if (layout) hipLaunchKernelGGL((whisper400_six_wide_kernel<kSixMaxSlots, Lens>), grid, block, c->lds6, stream, fp);
        else hipLaunchKernelGGL((whisper400_six_wide_runs_kernel<kSixMaxSlots, Lens>), grid, block, c->lds6, stream, fp);
        HIP_TRY(hipGetLastError());
        return MELSPEC_OK;
    } else {
    // plain batches, uniform and ragged, take the run-per-wave kernel (no division per unit, the clip record in scalar registers, a
    // wave re-reads its own frame-tail halo): cfg2 0.3105 -> 0.3055 ms, 8192 x 30 s 7.55 -> 7.42 ms against the round-robin deal
    if (layout) hipLaunchKernelGGL((whisper400_six_kernel<kSixMaxSlots, Lens>), grid, block, c->lds6, stream, fp);
    else hipLaunchKernelGGL((whisper400_six_runs_kernel<kSixMaxSlots, Lens>), grid, block, c->lds6, stream, fp);
    HIP_TRY(hipGetLastError());
#ifdef MELSPEC_LAB_STAMPS
    // tools/tail_probe.py: the 200th plain launch's per-wave end stamps and per-workgroup start stamps, as one line per workgroup
    static int stamp_calls = 0;
    if (!layout && fp.fix.list && lab_int("MELSPEC_LAB_STAMPS", 0, 0, 1) && ++stamp_calls == 200) {
        HIP_TRY(hipStreamSynchronize(stream));
        const size_t n = static_cast<size_t>(grid.x) * kSixWaves + grid.x;
        std::vector<uint64_t> st(n);
        HIP_TRY(hipMemcpy(st.data(), fp.fix.list + desc.n_units + 4096, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull;
        for (unsigned g = 0; g < grid.x; ++g) t0 = std::min(t0, st[static_cast<size_t>(grid.x) * kSixWaves + g]);
        for (unsigned g = 0; g < grid.x; ++g) {
            std::fprintf(stderr, "STAMP wg %u start %llu ends", g, static_cast<unsigned long long>(st[static_cast<size_t>(grid.x) * kSixWaves + g] - t0));
            for (int w = 0; w < kSixWaves; ++w) std::fprintf(stderr, " %llu", static_cast<unsigned long long>(st[static_cast<size_t>(g) * kSixWaves + w] - t0));
            std::fprintf(stderr, "\n");
        }
    }
#endif
    return MELSPEC_OK;
    }
}

// the fifteen-slot f32 kernels on twelve waves: Whisper large-v3's 128-mel bank, plain batches (runs) and layouts (rounds)
int launch_six_wide(melspec_ctx *c, const BatchDesc &desc, const FixSink &sink, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_six_wide_runs_kernel<kSixWideSlots, LensSix128>, "hipFuncSetAttribute(whisper400_six_wide_runs_kernel)");
        if (!rc) rc = allow_big_lds(&whisper400_six_wide_kernel<kSixWideSlots, LensSix128>, "hipFuncSetAttribute(whisper400_six_wide_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t blocks = (desc.n_units + kSixWideWaves - 1) / kSixWideWaves;
    const dim3 grid(grid_for_xcd(blocks, c->dev.cus, 1)), block(kSixWideWaves * 64);          // one twelve-wave workgroup per CU
    FixSink armed = sink_armed(c, sink, desc, grid.x);
    armed.vote_groups = std::min<unsigned>(grid.x, static_cast<unsigned>(c->dev.cus));
    const FastParams fp = fast_params(desc, c->ft6w, c->d_blob6w, c, armed);
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (layout) hipLaunchKernelGGL((whisper400_six_wide_kernel<kSixWideSlots, LensSix128>), grid, block, c->lds6w, stream, fp);
    else hipLaunchKernelGGL((whisper400_six_wide_runs_kernel<kSixWideSlots, LensSix128>), grid, block, c->lds6w, stream, fp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_ctx(melspec_ctx *c, const BatchDesc &desc_in, hipStream_t stream) {
    if (desc_in.n_units == 0) return MELSPEC_OK;
    BatchDesc desc = desc_in;
    const bool layout_batch = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (desc.sync_rounds < 0) {
        // measured (profiles/r01_variants.txt): six-frame kernel, 16 waves: four waves 4 apart; precise kernel, 8 waves:
        // consecutive pairs; 5-frame kernel, two 8-wave workgroups per CU: pairs 4 apart
        // (the groups that work are the waves of one SIMD: sixteen waves -> fours 4 apart, twelve -> threes 4 apart: the wide kernel's
        // mel-major store at 128 mels 0.532 ms with fours, 0.405-0.424 with threes, profiles/r06_wide_layouts.txt)
        // (the 80-mel layouts on twelve waves, built: 0.350 ms with threes or consecutive pairs against 0.339-0.342 on sixteen)
        if (c->fast && desc.frames_per_unit == kSixFrames) desc.sync_rounds = twelve_waves_for(c, true) ? 19 : 20;
        else if (c->fast && c->precision == MELSPEC_PRECISION_F64) desc.sync_rounds = 2;
        else if (c->fast) desc.sync_rounds = 18;
        else desc.sync_rounds = 1;
    }
    if (!c->fast && c->fast512 && desc.frames_per_unit == kFbFPW) return launch_whisper512(c, desc, stream);
    if (!c->fast) return launch_generic(c->gt, desc, c->hop_size, 0, 1, 1, 0.0, 0.0, c->dev.cus, stream);
    if (c->precision == MELSPEC_PRECISION_F64) {
        if ((layout_batch ? six64_layout_ok(c) : c->six64) && desc.frames_per_unit == kSixFrames && desc.d_unit_prefix == nullptr) {
            // mel-major stores of the twelve-wave kernel, measured (tools/mm64_sync_probe.py, 1024 x 10 s): consecutive pairs 0.491 ms, none 0.493,
            // pairs four apart 0.496, fours 0.512, fours one from each SIMD (the f32 kernel's best) 0.520, workgroup barrier 0.533
            if (layout_batch && desc_in.sync_rounds < 0) desc.sync_rounds = 2;
            return launch_six64(c, desc, FixSink{}, stream);
        }
        if (c->six64 && !layout_batch && desc.frames_per_unit == kSixFrames) return launch_six64(c, desc, FixSink{}, stream);
        return launch_precise(c, desc, FixSink{}, stream);
    }
    FixSink sink{};
    bool vote = false;
    if (c->precision == MELSPEC_PRECISION_AUTO) {
        // The vote (FixSink::vote): plain batches and the padded / mel-major layouts (whose sample is the head of the batch: they deal
        // their units round-robin).  Not where the mel kernel also leaves the image extremes for the TGA quantiser (d_unit_ext: the two
        // kernels' units differ): PCM -> TGA keeps the f32 kernel + recompute tail whatever the input.
        vote = c->fix.adaptive && (!layout_batch || desc.d_unit_ext == nullptr);
        int rc = auto_sink(c, desc, stream, vote, sink);
        if (rc) return rc;
    }
    int rc;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (c->first_kernel_events) {
        HIP_TRY(hipEventCreate(&pe0)); HIP_TRY(hipEventCreate(&pe1));
        c->first_kernel_events->push_back(pe0); c->first_kernel_events->push_back(pe1);
        HIP_TRY(hipEventRecord(pe0, stream));
    }
    if (c->six && desc.frames_per_unit == kSixFrames)
        rc = c->six_static == 1 ? launch_six_t<LensSix80>(c, desc, sink, stream) : c->six_static == 2 ? launch_six_t<LensSix64>(c, desc, sink, stream)
           : c->six_static == 3 ? launch_six_t<LensSix40>(c, desc, sink, stream) : launch_six_t<LensRuntime>(c, desc, sink, stream);
    else if (c->six_wide32 && desc.frames_per_unit == kSixFrames)
        rc = launch_six_wide(c, desc, sink, stream);
    else
        rc = launch_wave(c, desc, sink, stream);
    if (pe1) HIP_TRY(hipEventRecord(pe1, stream));
    if (rc || !vote) return rc;
    // AUTO's second launch: returns at its first instruction unless the launch above voted "heavy" (its number is c->fix.seq)
    const unsigned gate_value = (c->fix.seq & 0xffffffu) << 2 | kVoteDecided | kVoteHeavy;
    FixSink stat{};
    stat.count = sink.count; stat.acc = sink.acc; stat.host = sink.host;
    if (six64_layout_ok(c) && layout_batch && desc.frames_per_unit == kSixFrames && desc.d_unit_prefix == nullptr) {
        if (desc_in.sync_rounds < 0) desc.sync_rounds = 2;
        return launch_six64(c, desc, stat, stream, sink.decision, gate_value);          // the layouts on the six-frame f64 kernel: the f32 launch's own plan
    }
    if (layout_batch && desc.frames_per_unit != kFPW) {
        // the layouts' f64 kernel deals units of its own size: the same (uniform) batch planned for five frames per unit
        BatchPlan p5 = plan_uniform(desc.pcm, desc.out, desc.clip_stride, desc.frames_per_clip, desc.n_clips, c->n_mels, kFPW, desc.out_width, desc.mel_major != 0);
        if (p5.desc.sync_rounds < 0) p5.desc.sync_rounds = 2;          // the precise kernel's measured grouping (consecutive pairs)
        return launch_precise(c, p5.desc, stat, stream, sink.decision, gate_value);
    }
    if (layout_batch && desc_in.sync_rounds < 0) desc.sync_rounds = 2;
    if (c->six64 && !layout_batch && desc.frames_per_unit == kSixFrames) return launch_six64(c, desc, stat, stream, sink.decision, gate_value);
    if (c->six64_wide && !layout_batch && desc.d_unit_prefix == nullptr) {
        // 128 mels: the f32 launch walked five-frame units, the gated kernel deals six -- the same uniform batch planned again (arithmetic only;
        // a ragged batch's plan lives in device arrays made for five-frame units: those stay on the precise kernel)
        const BatchPlan p6 = plan_uniform(desc.pcm, desc.out, desc.clip_stride, desc.frames_per_clip, desc.n_clips, c->n_mels, kSixFrames);
        return launch_six64(c, p6.desc, stat, stream, sink.decision, gate_value);
    }
    return launch_precise(c, desc, stat, stream, sink.decision, gate_value);
}

int launch_stft(melspec_ctx *c, const BatchDesc &desc, int bins, int dtype, hipStream_t s) {
    if (desc.n_units == 0) return MELSPEC_OK;
    const int words = bins * 2 * (dtype == MELSPEC_STFT_F64 ? 2 : 1);
    if (c->fast) {
        static std::atomic<uint64_t> attr_done{0};
        if (!device_done(attr_done)) {
            int rc = allow_big_lds(&whisper400_stft_kernel<float>, "hipFuncSetAttribute(whisper400_stft_kernel<float>)");
            if (!rc) rc = allow_big_lds(&whisper400_stft_kernel<double>, "hipFuncSetAttribute(whisper400_stft_kernel<double>)");
            if (rc) return rc;
            mark_device_done(attr_done);
        }
        StftParams p{};
        p.b = desc;
        p.d_blob = static_cast<const uint32_t *>(c->d_blob64s.p);
        p.blob_words = PreciseBlob::kCount * 2;                    // the f64 tables only, not the mel section behind them
        p.hop = c->hop_size; p.bins = bins; p.words_per_frame = words;
        const size_t lds = static_cast<size_t>(p.blob_words) * 4 + static_cast<size_t>(kPreciseWaves) * PreciseLayout::slice_doubles() * sizeof(double);
        const uint64_t blocks = (desc.n_units + kPreciseWaves - 1) / kPreciseWaves;
        const unsigned grid = grid_for_xcd(blocks, c->dev.cus, 1);
        if (dtype == MELSPEC_STFT_F64) hipLaunchKernelGGL(whisper400_stft_kernel<double>, dim3(grid), dim3(kPreciseWaves * 64), lds, s, p);
        else hipLaunchKernelGGL(whisper400_stft_kernel<float>, dim3(grid), dim3(kPreciseWaves * 64), lds, s, p);
        HIP_TRY(hipGetLastError());
        return MELSPEC_OK;
    }
    return launch_generic_stft(c, desc, bins, dtype, s);
}

}  // namespace host
}  // namespace melspec

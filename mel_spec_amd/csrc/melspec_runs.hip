// melspec_runs.hip -- the run-per-wave f32 Whisper kernels (plain frame-major batches: config 2, 4, 5), as a translation unit of
// their own so that they can be compiled with their own instruction-scheduling strategy.
//
// hipcc's default scheduler balances register pressure against latency for the whole file; `-mllvm -amdgpu-sched-strategy=max-ilp`
// schedules for instruction-level parallelism first.  Measured same-box per kernel (tools/ab_run.py, profiles/r03_sched.txt): the
// 5-frame run kernel with the compile-time 128-mel bank (config 4) -6.2 %, the six-frame run kernel (config 2, 5) -0.7 %; the
// run-time-lens variants of the same kernels +0.1...+2 %, the f64 kernels +5...+11 %, the layout kernel of the six-frame family
// +25 % -- so the flag cannot be given to the library, only to these specialisations.  There is no
// source-level spelling of the per-function attribute ("amdgpu-sched-strategy"), hence the file: whisper400.hip declares these
// specialisations `extern template`, their device code and host stubs are emitted here, and mel_spec_amd/build.py compiles every
// unit to an object with its own flags and links them into libmelspec_hip.so.
#include "whisper400_kernels.hpp"

namespace melspec {

template __global__ void whisper400_six_runs_kernel<kSixMaxSlots, LensSix80>(const FastParams);
// NOT dispatched any more (the 40-mel bank runs whisper400_six_wide_runs_kernel since round 6) and kept on purpose: with this instantiation
// in the unit the compiler emits the LensSix80 kernel above as the 5552 instructions every measurement of the headline was made on;
// without it 5573, and config 2 is 1.1-1.3 % slower (same-box A/B both ways, profiles/r06_wide_layouts.txt).  tools/isa_compare.py before
// touching this list.
template __global__ void whisper400_six_runs_kernel<kSixMaxSlots, LensSix40>(const FastParams);
template __global__ void whisper400_wave_runs_kernel<8, LensI80>(const FastParams);
template __global__ void whisper400_wave_runs_kernel<12, LensI128>(const FastParams);
template __global__ void whisper400_six_wide_runs_kernel<kSixWideSlots, LensSix128>(const FastParams);

}  // namespace melspec

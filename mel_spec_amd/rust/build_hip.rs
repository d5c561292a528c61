// Branch to add to the reference's build.rs (next to the CARGO_FEATURE_CUDA branch, build.rs:9-49).
// NOT COMPILED HERE.  The kernels are already inside libmelspec_hip.so, so Cargo only has to link it.
fn link_hip_backend() {
    if std::env::var("CARGO_FEATURE_HIP").is_err() {
        return;
    }
    println!("cargo:rerun-if-env-changed=MELSPEC_HIP_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    // directory holding libmelspec_hip.so (built by `python -m mel_spec_amd.build`: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
    // -fPIC -c of every unit in csrc/*.hip -- melspec_runs.hip with -mllvm -amdgpu-sched-strategy=max-ilp -- then hipcc -shared)
    let dir = std::env::var("MELSPEC_HIP_DIR").expect("set MELSPEC_HIP_DIR to the directory of libmelspec_hip.so");
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=melspec_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
}

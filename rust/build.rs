// build.rs — feature `hip`: build libcuddle_hip.so from the HIP sources with hipcc and link it.
// (Cargo.toml: [features] hip = []; [build-dependencies] none.)
fn main() {
    if std::env::var("CARGO_FEATURE_HIP").is_err() {
        return;
    }
    let out = std::path::PathBuf::from(std::env::var("OUT_DIR").unwrap());
    let src = std::env::var("CUDDLE_HIP_SRC").unwrap_or_else(|_| "cuddle-hip/vectordb_amd/csrc".to_string());
    let hipcc = std::env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let status = std::process::Command::new(hipcc)
        .args([
            "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
            // distances reproduce the NEON accumulation order bit for bit: no contraction, correctly rounded division
            "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-o",
        ])
        .arg(out.join("libcuddle_hip.so"))
        .arg(format!("{src}/cuddle_hip.hip"))
        .status()
        .expect("hipcc not found");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=cuddle_hip");
    println!("cargo:rerun-if-changed={src}");
}

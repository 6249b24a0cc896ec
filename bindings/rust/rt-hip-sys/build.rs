// Builds librt_hip.so from the HIP source with hipcc (gfx950 only) and links it.
// RT_HIP_ROOT names the checkout that holds raytracer_amd/csrc and include/ (default: three
// directories up, i.e. this crate living in bindings/rust/rt-hip-sys of that checkout).
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let manifest = PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
    let root = std::env::var("RT_HIP_ROOT").map(PathBuf::from).unwrap_or_else(|_| manifest.join("../../.."));
    let out = PathBuf::from(std::env::var("OUT_DIR").unwrap());
    let src = root.join("raytracer_amd/csrc/rt_hip.hip");
    let lib = out.join("librt_hip.so");
    let hipcc = std::env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let status = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"])
        .arg("-o")
        .arg(&lib)
        .arg(&src)
        .status()
        .expect("hipcc not found (set HIPCC)");
    assert!(status.success(), "hipcc failed on {}", src.display());
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rerun-if-changed={}", src.display());
    println!("cargo:rerun-if-changed={}", root.join("raytracer_amd/csrc/rt_device.h").display());
    println!("cargo:rerun-if-changed={}", root.join("include/rt_hip.h").display());
    println!("cargo:rerun-if-env-changed=RT_HIP_ROOT");
}

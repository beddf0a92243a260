// Builds librt355.so from the HIP sources of this repository with hipcc for gfx950 and links it.
// RT355_SRC overrides the source directory (default: ../../raytracer_amd/csrc relative to this crate).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let src = env::var("RT355_SRC").map(PathBuf::from).unwrap_or_else(|_| manifest.join("../../raytracer_amd/csrc"));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let so = out.join("librt355.so");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let ok = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o"])
        .arg(&so)
        .arg(src.join("rt355.hip"))
        .arg(src.join("scene_build.cpp"))
        .arg(src.join("rt355_multi.cpp"))
        .args(["-ldl", "-lpthread"])
        .status()
        .expect("hipcc not found (set HIPCC)")
        .success();
    assert!(ok, "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt355");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    for f in ["rt355.hip", "scene_build.cpp", "rt355_multi.cpp", "kernels.h", "rt355_device.h"] {
        println!("cargo:rerun-if-changed={}", src.join(f).display());
    }
    println!("cargo:rerun-if-env-changed=RT355_SRC");
}

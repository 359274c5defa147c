// Links libsylow_hip.so (built by `make -C sylow_amd/csrc` with hipcc --offload-arch=gfx950).
// SYLOW_HIP_LIB_DIR = directory holding libsylow_hip.so (default: ../../sylow_amd relative to this crate).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("SYLOW_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../sylow_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=sylow_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=SYLOW_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/sylow_hip.h");
}

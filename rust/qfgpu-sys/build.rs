// Links crypto_amd/lib/libqfgpu.so.  QFGPU_LIB_DIR overrides the search path (default: the repository
// layout, two directories above this crate).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("QFGPU_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../crypto_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=qfgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=QFGPU_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/qfgpu.h");
}

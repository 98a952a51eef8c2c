//! Links libg16_amd.so (built by `make -C circom_compat_amd/csrc` in the repository root).
//! G16_AMD_LIB_DIR overrides the search path; the default is the in-tree location.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("G16_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../circom_compat_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=g16_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=G16_AMD_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/g16_amd.h");
    println!("cargo:rerun-if-changed=../../include/g16_loaders.h");
}

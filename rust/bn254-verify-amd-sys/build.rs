// Links libbn254_verify_amd.so (built by snark-bn254-verifier_amd/csrc/Makefile with hipcc for gfx950).  BN254_VERIFY_AMD_LIB_DIR overrides where it is looked for.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("BN254_VERIFY_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../snark-bn254-verifier_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=bn254_verify_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BN254_VERIFY_AMD_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/bn254_verify.h");
}

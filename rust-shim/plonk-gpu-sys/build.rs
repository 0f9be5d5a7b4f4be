// Tells rustc where libark_plonk_amd.so lives: ARK_PLONK_AMD_LIB_DIR, or ../../ark_plonk_amd relative to this crate
// (the in-tree build output of `python -m ark_plonk_amd.build`).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("ARK_PLONK_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ark_plonk_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ark_plonk_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=ARK_PLONK_AMD_LIB_DIR");
}

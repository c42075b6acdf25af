// build.rs for the reference crate (sabrinahirani/PLONK, Cargo.toml:1-18) once it links libplonk_hip.
//
// Place at the crate root, add `build = "build.rs"` under [package] in Cargo.toml, and point PLONK_HIP_ROOT at a checkout
// of this repository.  The library is built by its own Makefile (hipcc --offload-arch=gfx950) and linked dynamically.
// NOT COMPILED in this repository's build environment: no Rust toolchain exists there.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("PLONK_HIP_ROOT").expect("PLONK_HIP_ROOT = path of the plonk-mi355x checkout"));
    let csrc = root.join("plonk_amd").join("csrc");
    let status = Command::new("make").arg("-C").arg(&csrc).arg("-j8").status().expect("make not found");
    assert!(status.success(), "building libplonk_hip.so failed");
    // the whole-proof entry point behind src/prover_fast.rs (include/plonk_host.h), plain g++
    let status = Command::new("make").arg("-C").arg(root.join("plonk_amd").join("host")).status().expect("make not found");
    assert!(status.success(), "building libplonk_host.so failed");
    let lib = root.join("plonk_amd").join("lib");
    println!("cargo:rustc-link-search=native={}", lib.display());
    println!("cargo:rustc-link-lib=dylib=plonk_hip");
    println!("cargo:rustc-link-lib=dylib=plonk_host");
    println!("cargo:rerun-if-changed={}", root.join("include").join("plonk_host.h").display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib.display());
    println!("cargo:rerun-if-changed={}", root.join("include").join("plonk_hip.h").display());
    println!("cargo:rerun-if-env-changed=PLONK_HIP_ROOT");
}

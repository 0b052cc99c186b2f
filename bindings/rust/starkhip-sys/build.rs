// Links libstarkhip.so.  STARKHIP_LIB_DIR points at the directory that holds it (the repository's starky_bls12_381_amd/ after
// `make`); the default assumes this crate sits at <repo>/bindings/rust/starkhip-sys.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("STARKHIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../starky_bls12_381_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=starkhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=STARKHIP_LIB_DIR");
}

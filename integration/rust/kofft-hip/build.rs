// Link against the C-ABI library built by `make -C kofft_amd/csrc` (set KOFFT_HIP_LIB_DIR to its directory).
fn main() {
    let dir = std::env::var("KOFFT_HIP_LIB_DIR").unwrap_or_else(|_| "../../../kofft_amd/lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=kofft_hip");
    println!("cargo:rerun-if-env-changed=KOFFT_HIP_LIB_DIR");
}

// Links libphastft_hip.so; PHASTFT_HIP_LIB_DIR points at phastft_amd/lib of this repository.
fn main() {
    let dir = std::env::var("PHASTFT_HIP_LIB_DIR").unwrap_or_else(|_| "../../phastft_amd/lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=phastft_hip");
    println!("cargo:rerun-if-env-changed=PHASTFT_HIP_LIB_DIR");
}

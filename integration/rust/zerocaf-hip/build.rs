// Points the linker at libzerocaf_hip.so (set ZEROCAF_HIP_LIB_DIR to dusk_zerocaf_amd/).
fn main() {
    let dir = std::env::var("ZEROCAF_HIP_LIB_DIR").expect("set ZEROCAF_HIP_LIB_DIR to the directory holding libzerocaf_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=zerocaf_hip");
    println!("cargo:rerun-if-env-changed=ZEROCAF_HIP_LIB_DIR");
}

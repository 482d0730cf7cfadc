//! Locates libaic_hip.so: `AIC_HIP_LIB_DIR` (the directory holding the library built by
//! `make -C all_is_cubes_amd/csrc` in the MI355X repository), else the system linker path.
fn main() {
    println!("cargo:rerun-if-env-changed=AIC_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("AIC_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=aic_hip");
}

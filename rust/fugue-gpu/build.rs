// Links libfugue_amd.so (built by `python fugue_amd/build.py`: hipcc --offload-arch=gfx950).
fn main() {
    let dir = std::env::var("FUGUE_AMD_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../fugue_amd/lib").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=fugue_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=FUGUE_AMD_LIB_DIR");
}

// Links librodio_hip.so (built by `python rodio_amd/build.py`, hipcc --offload-arch=gfx950).
// RODIO_HIP_DIR = the directory that holds it (default: ../../rodio_amd relative to this crate).
fn main() {
    let dir = std::env::var("RODIO_HIP_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../rodio_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rodio_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RODIO_HIP_DIR");
}

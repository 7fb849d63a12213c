// Links libmi355rt.so.  MI355RT_LIB_DIR = the directory holding it (raytracer_amd/ of this repository after
// `python -c "import __graft_entry__ as g; g.build()"`); the .so must also be on the loader path at run time.
fn main() {
    if let Ok(dir) = std::env::var("MI355RT_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=mi355rt");
    println!("cargo:rerun-if-env-changed=MI355RT_LIB_DIR");
}

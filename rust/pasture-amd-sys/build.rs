// UNCOMPILED sketch.  Point PASTURE_AMD_LIB_DIR at <repo>/pasture_amd (where libpasture_amd.so is built).
fn main() {
    if let Ok(dir) = std::env::var("PASTURE_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=pasture_amd");
}

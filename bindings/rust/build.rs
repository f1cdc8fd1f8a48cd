// build.rs
fn main() {
    println!("cargo:rustc-link-search=native={}", std::env::var("POSEIDON_MI355X_LIB_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=poseidon_mi355x");   // libposeidon_mi355x.so (links libamdhip64; librccl is dlopen-ed when a group is formed)
}

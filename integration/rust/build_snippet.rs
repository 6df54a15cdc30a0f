// build.rs of the crate: link the engine (CALLABLE_HIP_DIR = decodingustools_amd/lib of this repository)
println!("cargo:rustc-link-search=native={}", std::env::var("CALLABLE_HIP_DIR").unwrap());
println!("cargo:rustc-link-lib=dylib=callable_hip");

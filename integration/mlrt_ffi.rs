//! integration/mlrt_ffi.rs - the Rust side of the drop-in boundary: what a maintainer of mu-lambda-raytracer adds as
//! `src/mlrt_ffi.rs` to call libmlrt_hip.so where `do_tracing` builds a `Renderer` and calls `rt.render(..)`
//! (/root/reference/src/main.rs:147-157).
//!
//! NOT COMPILED, NOT TESTED: the build image of this repository has no rustc / cargo (and no crates).  The file is a
//! field-for-field `#[repr(C)]` mirror of include/mlrt.h plus the handful of calls the seam needs; tests/test_abi.py checks
//! the struct sizes it assumes against the header through ctypes, and tests/c_abi_consumer.c is the same call sequence in C,
//! compiled and run on the GPU box.  build.rs: `println!("cargo:rustc-link-lib=dylib=mlrt_hip");`
#![allow(dead_code)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_void};

pub const MLRT_OK: i32 = 0;
pub const MLRT_PRECISION_FP32: i32 = 0; // fast mixed-precision mode
pub const MLRT_PRECISION_FP64: i32 = 1; // the reference's arithmetic (vec.rs:5-8): the mode a drop-in uses
pub const MLRT_BG_BLACK: i32 = 0;
pub const MLRT_BG_GRADIENT: i32 = 1;
pub const MLRT_GATHER_AUTO: i32 = 0;

#[repr(C)] pub struct MlrtMaterial { pub kind: i32, pub texture: i32, pub albedo: [f64; 3], pub param: f64 }              // materials.rs, volumes.rs:67-83
#[repr(C)] pub struct MlrtTexture { pub kind: i32, pub index: i32, pub index2: i32, pub _pad: i32, pub color: [f64; 3], pub scale: f64 }
#[repr(C)] pub struct MlrtImage { pub width: i32, pub height: i32, pub rgb: *const u8 }                                   // image_texture.rs
#[repr(C)] pub struct MlrtPerlin { pub ranvec: *const f64, pub perm: *const i32 }                                         // textures.rs:51-74
#[repr(C)] pub struct MlrtSphere { pub center: [f64; 3], pub radius: f64, pub material: i32, pub _pad: i32 }              // shapes.rs:25-42
#[repr(C)] pub struct MlrtBox { pub min: [f64; 3], pub max: [f64; 3], pub material: i32, pub _pad: i32 }                 // shapes.rs:166-198
#[repr(C)] pub struct MlrtRect { pub a0: i32, pub a1: i32, pub a0_v0: f64, pub a0_v1: f64, pub a1_v0: f64, pub a1_v1: f64,
                                 pub plane: f64, pub material: i32, pub _pad: i32 }                                       // aarects.rs:19-43
#[repr(C)] pub struct MlrtPrimRef { pub kind: i32, pub index: i32 }
#[repr(C)] pub struct MlrtBvhNode { pub bounds_min: [f64; 3], pub bounds_max: [f64; 3], pub left: i32, pub right: i32 }   // bhv.rs:109-112
#[repr(C)] pub struct MlrtBvh { pub root: i32, pub _pad: i32 }
#[repr(C)] pub struct MlrtTransform { pub has_rotate: i32, pub axis: i32, pub sin_theta: f64, pub cos_theta: f64,
                                      pub has_translate: i32, pub _pad: i32, pub offset: [f64; 3] }                       // transforms.rs
#[repr(C)] pub struct MlrtObject { pub kind: i32, pub bvh: i32, pub prim: MlrtPrimRef, pub transform: i32, pub phase_material: i32,
                                   pub neg_inv_density: f64 }
#[repr(C)] pub struct MlrtScene {
    pub n_spheres: i32, pub n_boxes: i32, pub n_rects: i32, pub n_prim_refs: i32, pub n_bvh_nodes: i32, pub n_bvhs: i32,
    pub n_transforms: i32, pub n_objects: i32, pub n_materials: i32, pub n_textures: i32, pub n_images: i32, pub n_perlins: i32,
    pub spheres: *const MlrtSphere, pub boxes: *const MlrtBox, pub rects: *const MlrtRect, pub prim_refs: *const MlrtPrimRef,
    pub bvh_nodes: *const MlrtBvhNode, pub bvhs: *const MlrtBvh, pub transforms: *const MlrtTransform, pub objects: *const MlrtObject,
    pub materials: *const MlrtMaterial, pub textures: *const MlrtTexture, pub images: *const MlrtImage, pub perlins: *const MlrtPerlin,
}
#[repr(C)] pub struct MlrtCamera { pub origin: [f64; 3], pub lower_left_corner: [f64; 3], pub horizontal: [f64; 3], pub vertical: [f64; 3],
                                   pub u: [f64; 3], pub v: [f64; 3], pub lens_radius: f64 }                               // camera.rs:3-12
#[repr(C)] pub struct MlrtParams { pub image_width: i32, pub image_height: i32, pub samples_per_pixel: i32, pub max_depth: i32, pub seed: u64,
                                   pub background_kind: i32, pub tracer: i32 /* 0 RecursiveRayTracer, 1 SingleLightSourceRayTracer */,
                                   pub background_top: [f64; 3], pub background_bottom: [f64; 3], pub light_source: [f64; 3], pub light_intensity: f64 }
#[repr(C)] pub struct MlrtContext { _private: [u8; 0] }
#[repr(C)] pub struct MlrtMulti { _private: [u8; 0] }
pub type MlrtProgress = Option<extern "C" fn(rows_done: i32, rows_total: i32, user: *mut c_void)>;

extern "C" {
    pub fn mlrt_last_error() -> *const c_char;
    pub fn mlrt_device_count() -> i32;
    pub fn mlrt_context_create(scene: *const MlrtScene, device: i32, precision: i32, out: *mut *mut MlrtContext) -> i32;
    pub fn mlrt_context_render(ctx: *mut MlrtContext, camera: *const MlrtCamera, params: *const MlrtParams, rows: *const i32, n_rows: i32,
                               rgb_out: *mut i32, progress: MlrtProgress, user: *mut c_void) -> i32;
    pub fn mlrt_context_destroy(ctx: *mut MlrtContext);
    // all GPUs of the node: rows dealt round-robin, one RCCL gather (include/mlrt.h, mlrt_multi_*)
    pub fn mlrt_multi_create(scene: *const MlrtScene, devices: *const i32, n_devices: i32, precision: i32, options: *const c_void,
                             gather_mode: i32, out: *mut *mut MlrtMulti) -> i32;
    pub fn mlrt_multi_render(multi: *mut MlrtMulti, camera: *const MlrtCamera, params: *const MlrtParams, rgb_out: *mut i32,
                             progress: MlrtProgress, user: *mut c_void) -> i32;
    pub fn mlrt_multi_destroy(multi: *mut MlrtMulti);
}

fn last_error() -> String { unsafe { CStr::from_ptr(mlrt_last_error()).to_string_lossy().into_owned() } }

pub type RGB = (i32, i32, i32); // raytrace.rs:57

/// Replaces `Renderer::new_with_rng(camera, world, background, params, RecursiveRayTracer{max_depth}, rngator)` + `rt.render(logger)`
/// (main.rs:147-157).  `scene` is the flattened world, `devices` the GPUs to spread the rows over.
/// Errors are panics, as everywhere at this seam in the reference.
pub fn render(scene: &MlrtScene, camera: &MlrtCamera, params: &MlrtParams, devices: &[i32]) -> Vec<Vec<RGB>> {
    let (w, h) = (params.image_width as usize, params.image_height as usize);
    let mut flat = vec![0i32; w * h * 3]; // Vec<Vec<RGB>> flattened, row 0 = the bottom row (quirk Q13)
    unsafe {
        if devices.len() > 1 {
            let mut m = std::ptr::null_mut();
            if mlrt_multi_create(scene, devices.as_ptr(), devices.len() as i32, MLRT_PRECISION_FP64, std::ptr::null(), MLRT_GATHER_AUTO, &mut m) != MLRT_OK { panic!("{}", last_error()); }
            let rc = mlrt_multi_render(m, camera, params, flat.as_mut_ptr(), None, std::ptr::null_mut());
            mlrt_multi_destroy(m);
            if rc != MLRT_OK { panic!("{}", last_error()); }
        } else {
            let mut ctx = std::ptr::null_mut();
            if mlrt_context_create(scene, *devices.first().unwrap_or(&0), MLRT_PRECISION_FP64, &mut ctx) != MLRT_OK { panic!("{}", last_error()); }
            let rc = mlrt_context_render(ctx, camera, params, std::ptr::null(), 0, flat.as_mut_ptr(), None, std::ptr::null_mut());
            mlrt_context_destroy(ctx);
            if rc != MLRT_OK { panic!("{}", last_error()); }
        }
    }
    flat.chunks(w * 3).map(|row| row.chunks(3).map(|p| (p[0], p[1], p[2])).collect()).collect()
}

//! `extern "C"` view of `include/rt355.h`, one to one: the same struct layouts (`#[repr(C)]`, fields in the
//! header's order) and every function the header declares.  `tests/test_host.py` parses this file and checks
//! both against the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT355_VERSION: c_int = 1;
pub const RT355_OK: c_int = 0;
pub const RT355_EINVAL: c_int = -1;
pub const RT355_EHIP: c_int = -2;
pub const RT355_ECAPACITY: c_int = -3;
pub const RT355_ENOMEM: c_int = -4;
pub const RT355_MAX_GEN: usize = 8;
pub const RT355_MAX_LAUNCH: usize = 48;
pub const RT355_NOALLOC_BYTES: usize = 64 << 20;
pub const RT355_CTX_ONE_STREAM: c_int = 1;
pub const RT355_MAX_BATCH: usize = 64;

// enum rt355_kind: position in `enum Shape` (src/shape.rs:11-20)
pub const RT355_SPHERE: i32 = 0;
pub const RT355_PLANE: i32 = 1;
pub const RT355_CUBE: i32 = 2;
pub const RT355_CYLINDER: i32 = 3;
pub const RT355_CONE: i32 = 4;
pub const RT355_GROUP: i32 = 5;
pub const RT355_TRIANGLE: i32 = 6;
pub const RT355_SMOOTH_TRIANGLE: i32 = 7;
// enum rt355_pattern: position in `enum Pattern` (src/pattern.rs:6-12) + 1; 0 = material.pattern None
pub const RT355_PATTERN_NONE: i32 = 0;
pub const RT355_PATTERN_STRIPE: i32 = 1;
pub const RT355_PATTERN_GRADIENT: i32 = 2;
pub const RT355_PATTERN_RING: i32 = 3;
pub const RT355_PATTERN_CHECKER: i32 = 4;
pub const RT355_PATTERN_TEST: i32 = 5;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt355_material {
    pub color: [f32; 3],
    pub ambient: f32,
    pub diffuse: f32,
    pub specular: f32,
    pub shininess: f32,
    pub reflective: f32,
    pub refractive: f32,
    pub transparency: f32,
    pub pattern_kind: i32,
    pub c1: [f32; 3],
    pub c2: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt355_shape_desc {
    pub kind: i32,
    pub m: [f32; 16],
    pub mat: rt355_material,
    pub min: f32,
    pub max: f32,
    pub closed: i32,
    pub p1: [f32; 3],
    pub p2: [f32; 3],
    pub p3: [f32; 3],
    pub n1: [f32; 3],
    pub n2: [f32; 3],
    pub n3: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt355_light_desc {
    pub position: [f32; 3],
    pub intensity: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt355_camera_desc {
    pub hsize: f32,
    pub vsize: f32,
    pub fov: f32,
    pub transform: [f32; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt355_stats {
    pub radiance_rays: [u64; RT355_MAX_GEN],
    pub hits: [u64; RT355_MAX_GEN],
    pub total_radiance: u64,
    pub total_shadow: u64,
    pub frame_ms: f64,
    pub n_launch: i32,
    pub launch_kind: [i32; RT355_MAX_LAUNCH],
    pub launch_gen: [i32; RT355_MAX_LAUNCH],
    pub launch_ms: [f32; RT355_MAX_LAUNCH],
    pub launch_box_tests: [u64; RT355_MAX_LAUNCH],
    pub launch_prim_tests: [u64; RT355_MAX_LAUNCH],
    pub launch_max_box_tests: [u64; RT355_MAX_LAUNCH],
    pub launch_max_wave_clocks: [u64; RT355_MAX_LAUNCH],
    pub launch_flops: [f64; RT355_MAX_LAUNCH],
}

pub enum rt355_ctx {}
pub enum rt355_scene {}
pub enum rt355_multi {}

extern "C" {
    pub fn rt355_last_error() -> *const c_char;
    pub fn rt355_version() -> c_int;
    pub fn rt355_ctx_create(out: *mut *mut rt355_ctx, device_id: c_int, real_bits: c_int) -> c_int;
    pub fn rt355_ctx_create_ex(out: *mut *mut rt355_ctx, device_id: c_int, real_bits: c_int, flags: c_int) -> c_int;
    pub fn rt355_ctx_destroy(ctx: *mut rt355_ctx);
    pub fn rt355_ctx_set_pool(ctx: *mut rt355_ctx, rays_per_pixel: f64) -> c_int;
    pub fn rt355_ctx_set_exact_cones(ctx: *mut rt355_ctx, mode: c_int) -> c_int;
    pub fn rt355_ctx_set_counting(ctx: *mut rt355_ctx, mode: c_int) -> c_int;
    pub fn rt355_ctx_set_timing(ctx: *mut rt355_ctx, on: c_int) -> c_int;
    pub fn rt355_scene_upload(ctx: *mut rt355_ctx, shapes: *const rt355_shape_desc, n_shapes: usize,
                              light: *const rt355_light_desc, out: *mut *mut rt355_scene) -> c_int;
    pub fn rt355_scene_destroy(scene: *mut rt355_scene);
    pub fn rt355_scene_info(scene: *const rt355_scene, out: *mut u64) -> c_int;
    pub fn rt355_render(ctx: *mut rt355_ctx, scene: *const rt355_scene, cam: *const rt355_camera_desc,
                        max_depth: c_int, rgb_out: *mut f32, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_render_u8(ctx: *mut rt355_ctx, scene: *const rt355_scene, cam: *const rt355_camera_desc,
                           max_depth: c_int, rgb_out: *mut u8, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_host_alloc(bytes: usize) -> *mut c_void;
    pub fn rt355_host_free(p: *mut c_void);
    pub fn rt355_render_device(ctx: *mut rt355_ctx, scene: *const rt355_scene, cam: *const rt355_camera_desc,
                               max_depth: c_int, rgb_out_dev: *mut f32, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_render_device_noalloc(ctx: *mut rt355_ctx, scene: *const rt355_scene, cam: *const rt355_camera_desc,
                                       max_depth: c_int, rgb_out_dev: *mut f32, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_render_tiles(ctx: *mut rt355_ctx, scene: *const rt355_scene, cam: *const rt355_camera_desc,
                              max_depth: c_int, rank: c_int, n_ranks: c_int, tiles_out_dev: *mut f32,
                              stats: *mut rt355_stats) -> c_int;
    pub fn rt355_render_tiles_begin(ctx: *mut rt355_ctx, scene: *const rt355_scene, cam: *const rt355_camera_desc,
                                    max_depth: c_int, rank: c_int, n_ranks: c_int, tiles_out_dev: *mut f32) -> c_int;
    pub fn rt355_render_device_begin(ctx: *mut rt355_ctx, scene: *const rt355_scene, cam: *const rt355_camera_desc,
                                     max_depth: c_int, rgb_out_dev: *mut f32) -> c_int;
    pub fn rt355_frame_end(ctx: *mut rt355_ctx, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_render_tiles_batch_begin(ctx: *mut rt355_ctx, scene: *const rt355_scene, cams: *const rt355_camera_desc,
                                          n_frames: c_int, max_depth: c_int, rank: c_int, n_ranks: c_int,
                                          tiles_out_dev: *mut f32) -> c_int;
    pub fn rt355_render_device_batch_begin(ctx: *mut rt355_ctx, scene: *const rt355_scene, cams: *const rt355_camera_desc,
                                           n_frames: c_int, max_depth: c_int, rgb_out_dev: *mut f32) -> c_int;
    pub fn rt355_render_batch(ctx: *mut rt355_ctx, scene: *const rt355_scene, cams: *const rt355_camera_desc,
                              n_frames: c_int, max_depth: c_int, rgb_out: *mut f32, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_tiles_per_rank(cam: *const rt355_camera_desc, n_ranks: c_int) -> i64;
    pub fn rt355_untile(ctx: *mut rt355_ctx, cam: *const rt355_camera_desc, n_ranks: c_int,
                        gathered_dev: *const f32, rgb_out_dev: *mut f32) -> c_int;
    pub fn rt355_untile_u8(ctx: *mut rt355_ctx, cam: *const rt355_camera_desc, n_ranks: c_int,
                           gathered_dev: *const u8, rgb_out_dev: *mut u8) -> c_int;
    pub fn rt355_untile_batch(ctx: *mut rt355_ctx, cam: *const rt355_camera_desc, n_ranks: c_int, n_frames: c_int,
                              gathered_dev: *const f32, rgb_out_dev: *mut f32) -> c_int;
    pub fn rt355_untile_batch_u8(ctx: *mut rt355_ctx, cam: *const rt355_camera_desc, n_ranks: c_int, n_frames: c_int,
                                 gathered_dev: *const u8, rgb_out_dev: *mut u8) -> c_int;
    pub fn rt355_multi_create(out: *mut *mut rt355_multi, device_ids: *const c_int, n_devices: c_int,
                              real_bits: c_int) -> c_int;
    pub fn rt355_multi_destroy(multi: *mut rt355_multi);
    pub fn rt355_multi_upload(multi: *mut rt355_multi, shapes: *const rt355_shape_desc, n_shapes: usize,
                              light: *const rt355_light_desc) -> c_int;
    pub fn rt355_multi_render(multi: *mut rt355_multi, cam: *const rt355_camera_desc, max_depth: c_int,
                              rgb_out: *mut f32, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_multi_render_batch(multi: *mut rt355_multi, cams: *const rt355_camera_desc, n_frames: c_int,
                                    max_depth: c_int, rgb_out: *mut f32, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_multi_gather_kind(multi: *const rt355_multi) -> *const c_char;
    pub fn rt355_color_at(ctx: *mut rt355_ctx, scene: *const rt355_scene, rays: *const f32, n: usize,
                          max_depth: c_int, rgb_out: *mut f32, stats: *mut rt355_stats) -> c_int;
    pub fn rt355_trace_sample(ctx: *mut rt355_ctx, scene: *const rt355_scene, rays: *const f32, n: usize,
                              t_out: *mut f64, shape_index_out: *mut i32) -> c_int;
    pub fn rt355_is_shadowed(ctx: *mut rt355_ctx, scene: *const rt355_scene, points: *const f32, n: usize,
                             out: *mut u8) -> c_int;
    pub fn rt355_to_u8(ctx: *mut rt355_ctx, rgb: *const f32, n: usize, out: *mut u8) -> c_int;
    pub fn rt355_to_u8_device(ctx: *mut rt355_ctx, rgb_dev: *const f32, n: usize, out_dev: *mut u8) -> c_int;
    pub fn rt355_ctx_wait_stream(ctx: *mut rt355_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn rt355_stream_wait_ctx(ctx: *mut rt355_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn rt355_ctx_set_async(ctx: *mut rt355_ctx, on: c_int) -> c_int;
    pub fn rt355_ctx_sync(ctx: *mut rt355_ctx) -> c_int;
    pub fn rt355_scene_check(shapes: *const rt355_shape_desc, n_shapes: usize, real_bits: c_int,
                             exact_cones: c_int, out: *mut u64) -> c_int;
    pub fn rt355_flop_table(out: *mut f64);
}

//! `Camera::render` (src/camera.rs:74-88) on MI355X through `rt355-sys`.
//!
//! `Camera::render_gpu(&world)` has the signature and result of `Camera::render(&world)`.  `GpuRenderer` keeps
//! the device context and the uploaded `World` between frames (scene upload -- inverses, BVH -- runs once).
use crate::{camera::Camera, canvas::Canvas, color::Color, material::Material, shape::Shape, world::World};
use nalgebra::{Matrix4, Vector4};
use rt355_sys::*;
use std::{ffi::CStr, ptr};

fn v3(v: &Vector4<f32>) -> [f32; 3] {
    [v.x, v.y, v.z]
}
/// nalgebra stores `Matrix4` column-major, which is what `rt355_shape_desc::m` expects.
fn m16(m: &Matrix4<f32>) -> [f32; 16] {
    let mut a = [0f32; 16];
    a.copy_from_slice(m.as_slice());
    a
}
fn last_error() -> String {
    unsafe { CStr::from_ptr(rt355_last_error()).to_string_lossy().into_owned() }
}

fn material(m: &Material) -> rt355_material {
    let black = Color::rgb(0., 0., 0.);
    let (kind, c1, c2) = m.pattern.map(|p| p.gpu_desc()).unwrap_or((RT355_PATTERN_NONE, black, black));
    rt355_material {
        color: [m.color.r, m.color.g, m.color.b],
        ambient: m.ambient,
        diffuse: m.diffuse,
        specular: m.specular,
        shininess: m.shininess,
        reflective: m.reflective,
        refractive: m.refractive,
        transparency: m.transparency,
        pattern_kind: kind,
        c1: [c1.r, c1.g, c1.b],
        c2: [c2.r, c2.g, c2.b],
    }
}

/// One entry of `World.shapes` -> one descriptor, same index (the index is the hit identity).
fn shape(s: &Shape) -> rt355_shape_desc {
    let z = [0f32; 3];
    let mut d = rt355_shape_desc {
        kind: RT355_SPHERE,
        m: m16(&s.transformation()),
        mat: material(s.material()),
        min: f32::NEG_INFINITY,
        max: f32::INFINITY,
        closed: 0,
        p1: z, p2: z, p3: z, n1: z, n2: z, n3: z,
    };
    match s {
        Shape::Sphere(_) => d.kind = RT355_SPHERE,
        Shape::Plane(_) => d.kind = RT355_PLANE,
        Shape::Cube(_) => d.kind = RT355_CUBE,
        Shape::Cylinder(c) => { d.kind = RT355_CYLINDER; d.min = c.min; d.max = c.max; d.closed = c.closed as i32 }
        Shape::Cone(c) => { d.kind = RT355_CONE; d.min = c.min; d.max = c.max; d.closed = c.closed as i32 }
        Shape::Group(_) => d.kind = RT355_GROUP, // inert in World::intersect (src/shape.rs:376); keeps indices aligned
        Shape::Triangle(t) => { d.kind = RT355_TRIANGLE; d.p1 = v3(&t.p1); d.p2 = v3(&t.p2); d.p3 = v3(&t.p3) }
        Shape::SmoothTriangle(t) => {
            d.kind = RT355_SMOOTH_TRIANGLE;
            d.p1 = v3(&t.p1); d.p2 = v3(&t.p2); d.p3 = v3(&t.p3);
            d.n1 = v3(&t.n1); d.n2 = v3(&t.n2); d.n3 = v3(&t.n3);
        }
    }
    d
}

fn descs(world: &World) -> (Vec<rt355_shape_desc>, rt355_light_desc) {
    let shapes = world.shapes.iter().map(shape).collect();
    let l = &world.pointlight;
    (shapes, rt355_light_desc { position: v3(&l.position), intensity: [l.intensity.r, l.intensity.g, l.intensity.b] })
}

fn camera_desc(c: &Camera) -> rt355_camera_desc {
    let (h, v, fov, t) = c.gpu_desc();
    rt355_camera_desc { hsize: h, vsize: v, fov, transform: m16(&t) }
}

fn canvas_of(cam: &rt355_camera_desc, rgb: &[f32]) -> Canvas {
    let pixels = rgb.chunks_exact(3).map(|p| Color::rgb(p[0], p[1], p[2])).collect();
    Canvas::new(cam.hsize as usize, cam.vsize as usize, pixels) // as src/camera.rs:83-87
}

enum Backend {
    One { ctx: *mut rt355_ctx, scene: *mut rt355_scene },
    Multi(*mut rt355_multi),
}

/// A device context (or one per GPU) with one uploaded `World`.
pub struct GpuRenderer {
    backend: Backend,
}

impl GpuRenderer {
    /// `real_bits`: 64 (double, the measured configuration) or 32 (the reference's own precision).
    pub fn new(device: i32, real_bits: i32, world: &World) -> Result<Self, String> {
        let (shapes, light) = descs(world);
        unsafe {
            let (mut ctx, mut scene) = (ptr::null_mut(), ptr::null_mut());
            if rt355_ctx_create(&mut ctx, device, real_bits) != RT355_OK {
                return Err(last_error());
            }
            if rt355_scene_upload(ctx, shapes.as_ptr(), shapes.len(), &light, &mut scene) != RT355_OK {
                let e = last_error();
                rt355_ctx_destroy(ctx);
                return Err(e);
            }
            Ok(GpuRenderer { backend: Backend::One { ctx, scene } })
        }
    }

    /// The frame's 32x32 tiles spread over `devices` (tile t on devices[t % n]), gathered to the first.
    pub fn new_multi(devices: &[i32], real_bits: i32, world: &World) -> Result<Self, String> {
        let (shapes, light) = descs(world);
        unsafe {
            let mut m = ptr::null_mut();
            if rt355_multi_create(&mut m, devices.as_ptr(), devices.len() as i32, real_bits) != RT355_OK {
                return Err(last_error());
            }
            if rt355_multi_upload(m, shapes.as_ptr(), shapes.len(), &light) != RT355_OK {
                let e = last_error();
                rt355_multi_destroy(m);
                return Err(e);
            }
            Ok(GpuRenderer { backend: Backend::Multi(m) })
        }
    }

    /// `Camera::render` with the recursion limit of src/camera.rs:79.
    pub fn render(&self, camera: &Camera) -> Result<Canvas, String> {
        let cam = camera_desc(camera);
        let mut rgb = vec![0f32; cam.hsize as usize * cam.vsize as usize * 3];
        let rc = unsafe {
            match self.backend {
                Backend::One { ctx, scene } => rt355_render(ctx, scene, &cam, 5, rgb.as_mut_ptr(), ptr::null_mut()),
                Backend::Multi(m) => rt355_multi_render(m, &cam, 5, rgb.as_mut_ptr(), ptr::null_mut()),
            }
        };
        if rc != RT355_OK {
            return Err(last_error());
        }
        Ok(canvas_of(&cam, &rgb))
    }
}

impl GpuRenderer {
    /// `Camera::render` for every camera of a path (all of one image size).  The frames go through launch chains
    /// of up to RT355_MAX_BATCH frames whose rays are traced together: a GPU that one 1080p frame leaves half
    /// idle is filled by four, and eight GPUs that each render an eighth of a frame by eight frames.
    pub fn render_path(&self, cameras: &[Camera]) -> Result<Vec<Canvas>, String> {
        let mut canvases = Vec::with_capacity(cameras.len());
        for part in cameras.chunks(RT355_MAX_BATCH) {
            let cams: Vec<rt355_camera_desc> = part.iter().map(camera_desc).collect();
            let per = cams[0].hsize as usize * cams[0].vsize as usize * 3;
            let mut rgb = vec![0f32; per * cams.len()];
            let n = cams.len() as i32;
            let rc = unsafe {
                match self.backend {
                    Backend::One { ctx, scene } => rt355_render_batch(ctx, scene, cams.as_ptr(), n, 5, rgb.as_mut_ptr(), ptr::null_mut()),
                    Backend::Multi(m) => rt355_multi_render_batch(m, cams.as_ptr(), n, 5, rgb.as_mut_ptr(), ptr::null_mut()),
                }
            };
            if rc != RT355_OK {
                return Err(last_error());
            }
            canvases.extend(cams.iter().zip(rgb.chunks_exact(per)).map(|(c, px)| canvas_of(c, px)));
        }
        Ok(canvases)
    }
}

impl Drop for GpuRenderer {
    fn drop(&mut self) {
        unsafe {
            match self.backend {
                Backend::One { ctx, scene } => {
                    rt355_scene_destroy(scene);
                    rt355_ctx_destroy(ctx);
                }
                Backend::Multi(m) => rt355_multi_destroy(m),
            }
        }
    }
}

impl Camera {
    /// Drop-in for `Camera::render` (src/camera.rs:74-88): same arguments, same `Canvas`.  Where the reference
    /// panics (singular transformation, NaN), this returns the library's message.
    pub fn render_gpu(&self, world: &World) -> Result<Canvas, String> {
        GpuRenderer::new(0, 64, world)?.render(self)
    }
}

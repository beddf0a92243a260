// `rendering_world_with_camera` (src/camera.rs:155-169) through the GPU: cargo test --features gpu
#![cfg(feature = "gpu")]
use approx::assert_relative_eq;
use nalgebra::Vector4;
use raytracer::{camera::Camera, color::Color, epsilon::EPSILON, tuple::TupleOperation, world::World};
use std::f32::consts::PI;

#[test]
fn rendering_world_with_camera_on_the_gpu() {
    let world = World::new_default();
    let transformation = Vector4::point(0.0, 0.0, -5.0)
        .view_transformation(&Vector4::point(0.0, 0.0, 0.0), &Vector4::vector(0.0, 1.0, 0.0));
    let camera = Camera::new(11.0, 11.0, PI / 2.0, transformation);
    let canvas = camera.render_gpu(&world).expect("rt355");
    assert_relative_eq!(
        canvas.pixels.get(5 * 11 + 5).unwrap().vector(),
        Color::rgb(0.38066, 0.47583, 0.2855).vector(),
        epsilon = EPSILON
    );
    // and pixel by pixel against the reference's own CPU path
    let cpu = camera.render(&world);
    assert_eq!(canvas.pixels.len(), cpu.pixels.len());
    for (a, b) in canvas.pixels.iter().zip(cpu.pixels.iter()) {
        assert_relative_eq!(a.vector(), b.vector(), epsilon = EPSILON);
    }
}

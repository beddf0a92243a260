// Append to src/camera.rs (the struct's fields are private, src/camera.rs:11-19):
#[cfg(feature = "gpu")]
impl Camera {
    /// (hsize, vsize, field_of_view, transform): the arguments of `Camera::new` (src/camera.rs:23-28).
    pub(crate) fn gpu_desc(&self) -> (f32, f32, f32, nalgebra::Matrix4<f32>) {
        (self.horizontal_size, self.vertical_size, self.field_of_view, self.transform)
    }
}

// Append to src/pattern.rs (the colour fields are private, src/pattern.rs:15-38):
#[cfg(feature = "gpu")]
impl Pattern {
    /// (kind as `enum rt355_pattern`, first colour, second colour).  The pattern's own transformation is not
    /// passed on: `pattern_at` never reads it (src/pattern.rs:94).
    pub(crate) fn gpu_desc(&self) -> (i32, Color, Color) {
        let black = Color::rgb(0., 0., 0.);
        match self {
            Pattern::Stripe(p) => (1, p.c1, p.c2),
            Pattern::Gradient(p) => (2, p.c1, p.c2),
            Pattern::Ring(p) => (3, p.c1, p.c2),
            Pattern::Checker(p) => (4, p.c1, p.c2),
            Pattern::Test(_) => (5, black, black),
        }
    }
}

//! dump_golden.rs -- golden vectors FROM THE REFERENCE ITSELF, for whoever has cargo.
//!
//! The build container of curvis_amd has no Rust toolchain, so its CPU oracle (oracle/curvis_oracle.c) is a
//! restatement that nothing reference-made pins beyond the reference's own few known answers.  This example
//! closes that gap in one command.  It links the UNMODIFIED reference crate (`curvis`), renders the small scenes
//! of curvis_amd's committed fixtures through the reference's public API and writes raw little-endian files
//! that `tests/test_reference_golden.py` compares with the oracle (every arithmetic flavour) and with the GPU.
//!
//!   cp tools/ref_dump/dump_golden.rs  <reference checkout>/examples/dump_golden.rs
//!   python tools/ref_dump/make_inputs.py                       # writes tools/ref_dump/inputs/*.png
//!   cd <reference checkout> && cargo run --release --example dump_golden -- \
//!        <curvis_amd>/tools/ref_dump/inputs  <curvis_amd>/tests/golden/ref
//!
//! What is written, per scene (W x H pixels, row-major, pixel (i, j) at index j * W + i):
//!   <scene>.rgb        W*H*3 u8   RelativisticSystem::render_image (or render_image_efficient) on the checker skies
//!   <scene>.index.rgb  W*H*3 u8   the same render on "index" skies whose texel (x, y) has the colour
//!                                 (x & 255, (x >> 8) | (y >> 8) << 4 | 64 | 128 * (sky == negative), y & 255):
//!                                 the raw texel index and the escape side of every ray, read through the
//!                                 reference's own private pixel_indexes_x_y_from_theta_phi_of_image
//!   <scene>.state      W*H records of 72 bytes: final position (4 f64), final covariant momentum (4 f64),
//!                                 Euler steps taken (u32), escape code (i32: +1, -1, 0 = not escaped) -- brute scenes
//!                                 only; the 10-line loop of the reference's PRIVATE escape_photon
//!                                 (src/systems.rs:115-139) is repeated here around the reference's own
//!                                 update_relativistic_object, because the private function cannot be called
//!   <scene>.samples    u64 n, then n x (alpha, escape angle, escape space) f64 -- efficient scenes only: the call
//!                                 render_image_efficient makes to doubly_sample_function (src/systems.rs:437-486)
//! Nothing here is derived from curvis_amd's code: every number comes out of the reference crate.

use std::env;
use std::f64::consts::PI;
use std::fs;
use std::path::{Path, PathBuf};

use nalgebra as na;

use curvis::sampling::sampling::doubly_sample_function;
use curvis::systems::systems::{EscapeAngle, RelativisticSystem};
use curvis::{
    compute_escape_angle, load_image_as_spherical_image, Camera, Covariance, DiagonalSphericalMetric,
    EllisMetric, InterstellarMetric, RelativisticObject, RelativisticVector,
};

const MAX_RADIUS: f64 = 100.0; // settings/defaults/simulation_settings.toml
const DELTA: f64 = 0.05;

struct Scene {
    name: &'static str,
    res: (u32, u32),
    pos: [f64; 4],
    cap: u32,
}

fn camera(s: &Scene) -> Camera {
    Camera::new(
        RelativisticVector::new(
            na::Vector4::new(s.pos[0], s.pos[1], s.pos[2], s.pos[3]),
            Covariance::Contravariant,
        ),
        na::Vector3::new(-1.0, 0.0, 0.0),
        na::Vector3::new(0.0, 0.0, 1.0),
        15.0,
        43.0,
        s.res.0,
        s.res.1,
    )
}

fn system<M: DiagonalSphericalMetric>(metric: M, s: &Scene, inputs: &Path, index: bool) -> RelativisticSystem<M> {
    let (p, n) = if index { ("index_pos.png", "index_neg.png") } else { ("pos.png", "neg.png") };
    RelativisticSystem::new(
        metric,
        load_image_as_spherical_image(&inputs.join(p), None, None),
        load_image_as_spherical_image(&inputs.join(n), None, None),
        camera(s),
    )
}

fn write(out: &Path, name: &str, ext: &str, bytes: &[u8]) {
    let path: PathBuf = out.join(format!("{}.{}", name, ext));
    fs::write(&path, bytes).expect("cannot write output file");
    println!("wrote {} ({} bytes)", path.display(), bytes.len());
}

/// the loop of the reference's private escape_photon (src/systems.rs:115-139), with a step counter
fn escape<M: DiagonalSphericalMetric>(metric: &M, photon: &mut RelativisticObject, cap: u32) -> (u32, i32) {
    let mut steps = 0u32;
    for _ in 0..cap {
        metric.update_relativistic_object(photon, DELTA);
        steps += 1;
        if photon.x(1) > MAX_RADIUS {
            return (steps, 1);
        } else if photon.x(1) < -MAX_RADIUS {
            return (steps, -1);
        }
    }
    (steps, 0)
}

fn brute<M: DiagonalSphericalMetric>(make: &dyn Fn() -> M, s: &Scene, inputs: &Path, out: &Path) {
    for index in [false, true] {
        let sys = system(make(), s, inputs, index);
        let img = sys.render_image(s.cap, MAX_RADIUS, DELTA);
        write(out, s.name, if index { "index.rgb" } else { "rgb" }, &img.to_rgb8().into_raw());
    }
    let metric = make();
    let cam = camera(s);
    let mut state: Vec<u8> = Vec::new();
    for j in 0..s.res.1 {
        for i in 0..s.res.0 {
            // RelativisticSystem::camera_pixels_x_y_to_photon (src/systems.rs:531-534), which is private too
            let direction = cam.outward_vector_on_world_space_from_x_y(i, j);
            let mut photon = metric.new_photon(cam.position(), direction);
            let (steps, code) = escape(&metric, &mut photon, s.cap);
            for k in 0..4 {
                state.extend_from_slice(&photon.x(k).to_le_bytes());
            }
            for k in 0..4 {
                state.extend_from_slice(&photon.p(k).to_le_bytes());
            }
            state.extend_from_slice(&steps.to_le_bytes());
            state.extend_from_slice(&code.to_le_bytes());
        }
    }
    write(out, s.name, "state", &state);
}

fn efficient<M: DiagonalSphericalMetric>(make: &dyn Fn() -> M, s: &Scene, inputs: &Path, out: &Path) {
    // the CLI's wiring (src/main.rs:46-47, :106-107): sampling_initial_nums = 100 for BOTH alpha_nums and
    // max_iterations_sampling; thresholds 1e-5 / 1e-5 as in curvis_amd's fixtures
    for index in [false, true] {
        let sys = system(make(), s, inputs, index);
        let img = sys.render_image_efficient(s.cap, MAX_RADIUS, DELTA, 100, 100, 1e-5, 1e-5);
        write(out, s.name, if index { "index.rgb" } else { "rgb" }, &img.to_rgb8().into_raw());
    }
    let metric = make();
    let l_cam = s.pos[1];
    let (alphas, angles, spaces) = doubly_sample_function(-0.1 * PI, 1.1 * PI, 100, 100, 1e-5, 1e-5, |alpha| {
        match compute_escape_angle(&metric, l_cam, alpha, DELTA, s.cap, MAX_RADIUS) {
            EscapeAngle::PositiveSpace(e) => (e, 1.0),
            EscapeAngle::NegativeSpace(e) => (e, -1.0),
            EscapeAngle::NotEscaped => (f64::NAN, f64::NAN),
        }
    });
    let mut bytes: Vec<u8> = Vec::new();
    bytes.extend_from_slice(&(alphas.len() as u64).to_le_bytes());
    for k in 0..alphas.len() {
        bytes.extend_from_slice(&alphas[k].to_le_bytes());
        bytes.extend_from_slice(&angles[k].to_le_bytes());
        bytes.extend_from_slice(&spaces[k].to_le_bytes());
    }
    write(out, s.name, "samples", &bytes);
}

fn main() {
    let args: Vec<String> = env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: dump_golden <inputs dir (tools/ref_dump/inputs)> <output dir (tests/golden/ref)>");
        std::process::exit(2);
    }
    let inputs = PathBuf::from(&args[1]);
    let out = PathBuf::from(&args[2]);
    fs::create_dir_all(&out).expect("cannot create the output directory");
    let half_pi = PI / 2.0;
    let ellis = || EllisMetric::new(1.0);
    let inter = || InterstellarMetric::new(0.1, 1e-4, 1.0);
    // the scenes of curvis_amd/tests/golden/make_golden.py
    brute(&ellis, &Scene { name: "brute_ellis_default_64x36", res: (64, 36), pos: [0.0, 5.0, half_pi, 0.0], cap: 4096 }, &inputs, &out);
    brute(&inter, &Scene { name: "brute_interstellar_default_64x36", res: (64, 36), pos: [0.0, 5.0, half_pi, 0.0], cap: 8192 }, &inputs, &out);
    brute(&ellis, &Scene { name: "brute_ellis_orbit_l3_48x27", res: (48, 27), pos: [0.0, 3.0, half_pi, 1.0], cap: 4096 }, &inputs, &out);
    efficient(&ellis, &Scene { name: "efficient_ellis_default_96x54", res: (96, 54), pos: [0.0, 5.0, half_pi, 0.0], cap: 40000 }, &inputs, &out);
    efficient(&inter, &Scene { name: "efficient_interstellar_default_64x36", res: (64, 36), pos: [0.0, 5.0, half_pi, 0.0], cap: 40000 }, &inputs, &out);
    println!("done: commit tests/golden/ref/ and run `python -m pytest tests/test_reference_golden.py`");
}

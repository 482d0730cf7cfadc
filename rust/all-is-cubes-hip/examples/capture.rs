//! Captures a template scene of the reference for the MI355X repository (BASELINE configs 1, 2, 4, 5):
//!
//! ```text
//! AIC_DUMP=atrium-1080p.aic  cargo run --release --example capture -- atrium    1920 1080
//! AIC_DUMP=demo-city-256.aic cargo run --release --example capture -- demo-city  256  256
//! ```
//!
//! builds `UniverseTemplate::{Atrium, DemoCity}` with `seed: Some(0)` exactly as `test-renderers/cases/src/lib.rs:1054-1066`
//! does, renders ONE frame of it through `HipRtRenderer` from the default character's spawn with
//! `GraphicsOptions::default()` minus bloom, and prints what the receiving side checks: the frame size, `cubes_traced`, and
//! an FNV-1a hash of the RGBA8 bytes. With `AIC_DUMP` set, `libaic_hip.so` records every call it received; copy the file to
//! `tests/golden/` of the MI355X repository: `tests/test_replay.py` holds it to the parity bar against the CPU oracle and
//! `python bench.py --workload replay:tests/golden/atrium-1080p.aic` benchmarks it (README of tests/golden/).
//!
//! The same binary with `--cpu` renders the frame with the reference's own `RtRenderer` instead and prints the same line:
//! the two lines are the end-to-end parity check of the shim on a machine that has both toolchains.
//!
//! NOTE: like the rest of this crate, written against all-is-cubes 0.10.0 and never compiled (no Rust toolchain where it
//! was written).

use std::str::FromStr as _;
use std::sync::Arc;

use all_is_cubes::listen;
use all_is_cubes::util::yield_progress_for_testing;
use all_is_cubes_content::{TemplateParameters, UniverseTemplate};
use all_is_cubes_hip::HipRtRenderer;
use all_is_cubes_render::HeadlessRenderer as _;
use all_is_cubes_render::camera::{GraphicsOptions, Layers, StandardCameras, UiViewState, Viewport};
use all_is_cubes_render::raytracer::RtRenderer;

fn fnv1a(bytes: impl Iterator<Item = u8>) -> u64 {
    bytes.fold(0xcbf2_9ce4_8422_2325, |h, b| (h ^ u64::from(b)).wrapping_mul(0x0000_0100_0000_01b3))
}

fn main() {
    let args: Vec<String> = std::env::args().skip(1).collect();
    let cpu = args.iter().any(|a| a == "--cpu");
    let positional: Vec<&String> = args.iter().filter(|a| !a.starts_with("--")).collect();
    let template_name = positional.first().map_or("atrium", |s| s.as_str());
    let width: u32 = positional.get(1).map_or(1920, |s| s.parse().expect("width"));
    let height: u32 = positional.get(2).map_or(1080, |s| s.parse().expect("height"));

    let template = UniverseTemplate::from_str(template_name).expect("template name (atrium, demo-city, ...)");
    let universe = async_io::block_on(template.build(yield_progress_for_testing(), TemplateParameters { seed: Some(0), size: None }))
        .expect("template build");
    let character = universe.get_default_character().expect("template has a character");

    let mut options = GraphicsOptions::default();
    options.bloom_intensity = all_is_cubes::math::zo32(0.0); // the raytracers have no bloom (Flaws::NO_BLOOM)
    options.debug_info_text = false;
    let cameras = StandardCameras::new(
        listen::constant(Arc::new(options)),
        listen::constant(Viewport::with_scale(1.0, [width, height])),
        listen::constant(Some(character)),
        listen::constant(Arc::new(UiViewState::default())),
    );

    let (data, cubes_traced): (Vec<[u8; 4]>, String) = if cpu {
        let mut renderer = RtRenderer::new(cameras, Box::new(core::convert::identity), listen::constant(Default::default()));
        renderer.update(Layers::splat(universe.read_ticket()), None).expect("update");
        let image = renderer.draw_rgba(|_| String::new());
        (image.data, format!("{}", image.info))
    } else {
        let mut renderer = HipRtRenderer::new(cameras, Box::new(core::convert::identity), 0).expect("an MI355X");
        renderer.update(Layers::splat(universe.read_ticket()), None).expect("update");
        let image = async_io::block_on(renderer.draw("")).expect("draw");
        (image.data, format!("{}", image.info))
    };
    println!(
        "{template_name} {width}x{height} {} rgba8-fnv1a {:016x} info: {cubes_traced}",
        if cpu { "RtRenderer (CPU reference)" } else { "HipRtRenderer (MI355X)" },
        fnv1a(data.iter().flatten().copied()),
    );
    if let Ok(path) = std::env::var("AIC_DUMP") {
        if !cpu {
            println!("calls recorded in {path}: copy it to tests/golden/ of the MI355X repository");
        }
    }
}

//! `HeadlessRenderer` for all-is-cubes on an AMD MI355X: the CPU raytracer of `all-is-cubes-render`
//! (`RtRenderer` / `SpaceRaytracer`) replaced by the HIP kernel behind `libaic_hip.so`.
//!
//! This crate holds on the host exactly what `RtRenderer` holds -- the `StandardCameras`, the size policy and one
//! change listener per layer (the `SpaceChange` filter of raytracer/updating.rs:200-219) -- and forwards snapshots and
//! deltas through the C ABI instead of mutating a `SpaceRaytracer`:
//!
//! | reference                                              | here                                   |
//! |--------------------------------------------------------|----------------------------------------|
//! | `RtRenderer::new` (renderer.rs:65-81)                  | `HipRtRenderer::new` -> `aic_create`   |
//! | `UpdatingSpaceRaytracer::update`, everything (121-131) | `aic_upload_space`                     |
//! | ... blocks (139-153)                                   | `aic_replace_blocks`                   |
//! | ... cubes (155-166)                                    | `aic_update_cubes`                     |
//! | `RtRenderer::draw_rgba` (renderer.rs:282-308)          | `aic_render`                           |
//!
//! With `AIC_DUMP=path` in the environment the library records every call it receives; the recording replays on
//! a machine without the Rust toolchain (`python -m all_is_cubes_amd.replay path` in the MI355X repository) -- that is
//! how a real Atrium / DemoCity scene reaches its benchmark and parity tests.
//!
//! NOTE: written against all-is-cubes 0.10.0; it has not been compiled in the repository that carries it (no Rust
//! toolchain there). Expect to adjust imports to the workspace it is dropped into.

#![warn(missing_docs)]

use std::ffi::CStr;
use std::ptr::NonNull;
use std::sync::{Arc, Mutex};

use all_is_cubes::character::Cursor;
use all_is_cubes::listen::{self, Listen as _};
use all_is_cubes::math::{Cube, Rgba, ZeroOne};
use all_is_cubes::space::{BlockIndex, Space, SpaceChange};
use all_is_cubes::universe::{Handle, ReadTicket};
use all_is_cubes_render::camera::{
    AntialiasingOption, Camera, FogOption, GraphicsOptions, Layers, LightingOption, StandardCameras, ToneMappingOperator,
    TransparencyOption, Viewport,
};
use all_is_cubes_render::{Flaws, HeadlessRenderer, RenderError, Rendering};
use futures_core::future::BoxFuture;

pub mod ffi;
mod flatten;

use flatten::{FlatSpace, flatten_block, gather_cubes};

/// What `SrtTodo` is in the reference (updating.rs:174-219): the changes not yet forwarded to the device.
#[derive(Debug, Default)]
struct Todo {
    everything: bool,
    blocks: std::collections::HashSet<BlockIndex>,
    cubes: std::collections::HashSet<Cube>,
}

impl listen::Store<SpaceChange> for Todo {
    fn receive(&mut self, messages: &[SpaceChange]) {
        for message in messages {
            match *message {
                SpaceChange::EveryBlock => {
                    self.everything = true;
                    self.blocks.clear();
                    self.cubes.clear();
                }
                SpaceChange::CubeLight { cube, .. } | SpaceChange::CubeBlock { cube, .. } => {
                    self.cubes.insert(cube);
                }
                SpaceChange::BlockIndex(index) | SpaceChange::BlockEvaluation(index) => {
                    self.blocks.insert(index);
                }
                SpaceChange::Physics => {}
            }
        }
    }
}

struct LayerSync {
    space: Handle<Space>,
    todo: listen::StoreLock<Todo>,
    listening: bool,
    n_blocks: usize,
}

/// Renderer-specific diagnostic information: the device's `RaytraceInfo` sums and kernel time.
#[derive(Clone, Copy, Debug)]
pub struct HipInfo(pub ffi::aic_frame_info);
impl core::fmt::Display for HipInfo {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "{} cubes traced, kernel {:.3} ms", self.0.cubes_traced, self.0.kernel_ms)
    }
}

/// The MI355X raytracer behind the reference's renderer interface.
pub struct HipRtRenderer {
    ctx: NonNull<ffi::aic_ctx>,
    cameras: StandardCameras,
    size_policy: Box<dyn Fn(Viewport) -> Viewport + Send + Sync>,
    layers: Layers<Option<LayerSync>>,
    had_cursor: bool,
}

// SAFETY: the context is only used through `&mut self`; libaic_hip keeps no thread affinity besides hipSetDevice,
// which every entry point performs itself.
unsafe impl Send for HipRtRenderer {}

fn check(ctx: *const ffi::aic_ctx, rc: core::ffi::c_int) -> Result<(), RenderError> {
    if rc == ffi::AIC_OK {
        return Ok(());
    }
    // SAFETY: aic_last_error returns a NUL-terminated string owned by the context
    let message = unsafe { CStr::from_ptr(ffi::aic_last_error(ctx)) }.to_string_lossy().into_owned();
    match rc {
        // the reference asserts on the same conditions (e.g. output length, renderer.rs:193-197)
        ffi::AIC_ERR_INVALID => panic!("libaic_hip rejected a call: {message}"),
        // RenderError has no device variant yet (lib.rs:46-54 "TODO: add errors for out of memory, lost GPU");
        // until it does, a lost device is reported the way an unreadable scene is
        _ => {
            log::error!("libaic_hip: {message}");
            Err(RenderError::Read(all_is_cubes::universe::HandleError::from_message(message)))
        }
    }
}

fn options_of(o: &GraphicsOptions) -> ffi::aic_options {
    let (transparency, threshold) = match o.transparency {
        TransparencyOption::Surface => (0, 0.5),
        TransparencyOption::Volumetric => (1, 0.5),
        TransparencyOption::Threshold(t) => (2, t.into_inner()),
        _ => (1, 0.5),
    };
    let (lighting, bounce_samples) = match o.lighting_display {
        LightingOption::None => (0, 0),
        LightingOption::Flat => (1, 0),
        LightingOption::Coarse => (2, 0),
        LightingOption::Linear => (3, 0),
        LightingOption::Smoothstep => (4, 0),
        LightingOption::Bounce { samples } => (5, i32::from(samples)), // reported as Flaws::UNSUPPORTED by the device
        _ => (3, 0),
    };
    ffi::aic_options {
        fog: match o.fog { FogOption::None => 0, FogOption::Abrupt => 1, FogOption::Compromise => 2, FogOption::Physical => 3, _ => 1 },
        transparency,
        threshold,
        lighting,
        bounce_samples,
        antialiasing: match o.antialiasing { AntialiasingOption::None => 0, AntialiasingOption::IfCheap => 1, AntialiasingOption::Always => 2, _ => 0 },
        debug_pixel_cost: i32::from(o.debug_pixel_cost),
        tone_mapping: match o.tone_mapping { ToneMappingOperator::Clamp => 0, ToneMappingOperator::Reinhard => 1, _ => 0 },
        maximum_intensity: o.maximum_intensity.into_inner(),
        bloom_intensity: o.bloom_intensity.into_inner(),
        view_distance: o.view_distance.into_inner(),
    }
}

fn camera_of(camera: &Camera) -> ffi::aic_camera {
    ffi::aic_camera {
        // euclid's Transform3D::to_array is m11, m12, ..., m44: the order aic_camera documents
        inverse_projection_view: camera.inverse_projection_view().to_array(),
        exposure: camera.exposure().into_inner(),
        reserved: 0,
    }
}

impl HipRtRenderer {
    /// As `RtRenderer::new` (renderer.rs:65-81), on HIP device `device_id` (negative: the current device).
    ///
    /// # Errors
    /// Fails if no usable MI355X is present: there is no CPU fallback.
    pub fn new(
        cameras: StandardCameras,
        size_policy: Box<dyn Fn(Viewport) -> Viewport + Send + Sync>,
        device_id: i32,
    ) -> Result<Self, String> {
        let mut status = 0;
        // SAFETY: plain FFI call; a null return is handled below
        let ctx = unsafe { ffi::aic_create(device_id, &mut status) };
        let ctx = NonNull::new(ctx).ok_or_else(|| format!("aic_create failed with status {status} (no usable MI355X?)"))?;
        assert_eq!(unsafe { ffi::aic_abi_version() }, ffi::AIC_ABI_VERSION);
        Ok(Self { ctx, cameras, size_policy, layers: Layers::default(), had_cursor: false })
    }

    fn sync_layer(
        ctx: *mut ffi::aic_ctx,
        layer: core::ffi::c_int,
        slot: &mut Option<LayerSync>,
        space: Option<&Handle<Space>>,
        read_ticket: ReadTicket<'_>,
        options: &GraphicsOptions,
    ) -> Result<(), RenderError> {
        // the Option-synchronisation of RtRenderer::update (renderer.rs:110-141)
        match (space, &mut *slot) {
            (Some(space), Some(sync)) if *space == sync.space => {}
            (Some(space), s) => {
                *s = Some(LayerSync {
                    space: space.clone(),
                    todo: listen::StoreLock::new(Todo { everything: true, ..Todo::default() }),
                    listening: false,
                    n_blocks: 0,
                });
            }
            (None, s) => {
                if s.take().is_some() {
                    check(ctx, unsafe { ffi::aic_clear_space(ctx, layer) })?;
                }
            }
        }
        let Some(sync) = slot else { return Ok(()) };
        let space = sync.space.read(read_ticket).map_err(RenderError::Read)?;
        if !sync.listening {
            space.listen(sync.todo.listener());
            sync.listening = true;
        }
        let todo = &mut *sync.todo.lock();
        let block_data = space.block_data();
        if core::mem::take(&mut todo.everything) {
            // == SpaceRaytracer::new (sr.rs:64-88)
            let flat = FlatSpace::new(&space);
            check(ctx, unsafe { ffi::aic_upload_space(ctx, layer, &flat.desc()) })?;
            todo.blocks.clear();
            todo.cubes.clear();
        } else {
            // appended blocks, then re-evaluated ones (updating.rs:139-153): one batched call
            let mut indices: Vec<u32> = (sync.n_blocks..block_data.len()).map(|i| i as u32).collect();
            indices.extend(todo.blocks.drain().map(u32::from).filter(|&i| (i as usize) < sync.n_blocks));
            if !indices.is_empty() {
                let flat: Vec<_> = indices.iter().map(|&i| flatten_block(&block_data[i as usize])).collect();
                let descs: Vec<_> = flat.iter().map(|b| b.desc).collect();
                let voxels: Vec<_> = flat.iter().map(|b| b.voxels.as_ptr()).collect();
                let palettes: Vec<_> = flat.iter().map(|b| b.palette.as_ptr()).collect();
                check(ctx, unsafe {
                    ffi::aic_replace_blocks(ctx, layer, indices.len() as u32, indices.as_ptr(), descs.as_ptr(), voxels.as_ptr(), palettes.as_ptr())
                })?;
            }
            if !todo.cubes.is_empty() {
                let (xyz, idx, light) = gather_cubes(&space, todo.cubes.drain());
                check(ctx, unsafe { ffi::aic_update_cubes(ctx, layer, idx.len() as u32, xyz.as_ptr(), idx.as_ptr(), light.as_ptr()) })?;
            }
        }
        sync.n_blocks = block_data.len();
        check(ctx, unsafe { ffi::aic_set_options(ctx, layer, &options_of(options)) })
    }

    /// `RtRenderer::update` (renderer.rs:96-141).
    pub fn update_scene(&mut self, read_tickets: Layers<ReadTicket<'_>>, cursor: Option<&Cursor>) -> Result<(), RenderError> {
        self.had_cursor = cursor.is_some();
        self.cameras.update(read_tickets);
        let ctx = self.ctx.as_ptr();
        let world_options = self.cameras.graphics_options().clone();
        let ui_options = self.cameras.ui_view_state().graphics_options.clone();
        let world_space = self.cameras.world_space().get();
        Self::sync_layer(ctx, ffi::AIC_LAYER_WORLD, &mut self.layers.world, world_space.as_ref(), read_tickets.world, &world_options)?;
        let ui_space = self.cameras.ui_space().cloned();
        Self::sync_layer(ctx, ffi::AIC_LAYER_UI, &mut self.layers.ui, ui_space.as_ref(), read_tickets.ui, &ui_options)
    }

    /// `RtRenderer::draw_rgba` (renderer.rs:282-308).
    pub fn draw_rgba(&mut self, info_text: &str) -> Result<Rendering, RenderError> {
        let viewport = (self.size_policy)(self.cameras.viewport()); // renderer.rs:226-233
        let size = viewport.framebuffer_size;
        let mut data = vec![[0u8; 4]; (size.width as usize) * (size.height as usize)];
        let backdrop = self.cameras.ui_view_state().backdrop;
        let frame = ffi::aic_frame_desc {
            width: size.width,
            height: size.height,
            world: camera_of(&self.cameras.cameras().world),
            ui: camera_of(&self.cameras.cameras().ui),
            backdrop: if backdrop == Rgba::TRANSPARENT {
                [0.0; 4]
            } else {
                [backdrop.red().into_inner(), backdrop.green().into_inner(), backdrop.blue().into_inner(), backdrop.alpha().into_inner()]
            },
            partition: ffi::aic_partition::default(),
            flags: 0,
            reserved: 0,
        };
        let mut info = ffi::aic_frame_info::default();
        if !data.is_empty() {
            // zero-area viewports produce an empty image (cases viewport_zero, cases/src/lib.rs:1167-1212)
            let ctx = self.ctx.as_ptr();
            check(ctx, unsafe { ffi::aic_render(ctx, &frame, data.as_mut_ptr().cast(), 0, &mut info) })?;
        }
        let options = self.cameras.graphics_options();
        if options.debug_info_text && !info_text.is_empty() {
            // renderer.rs:659-683: the 2-D glyph blit stays on the host; the raytracer crate's function is private, so a
            // maintainer either exposes it or copies its 25 lines here
            log::trace!("info text not drawn by all-is-cubes-hip: {info_text}");
        }
        let mut flaws = Flaws::empty();
        if info.flaws & ffi::AIC_FLAW_UNSUPPORTED != 0 {
            flaws |= Flaws::UNSUPPORTED;
        }
        if options.bloom_intensity != ZeroOne::ZERO {
            flaws |= Flaws::NO_BLOOM; // renderer.rs:293-297
        }
        if self.had_cursor {
            flaws |= Flaws::NO_CURSOR; // renderer.rs:298-300
        }
        Ok(Rendering { size, data, flaws, info: Arc::new(HipInfo(info)) })
    }

    /// The cameras this renderer draws (as `RtRenderer::cameras`).
    pub fn cameras(&self) -> &StandardCameras {
        &self.cameras
    }

    /// Runs the light updater ON THE DEVICE against the world space as last uploaded: `Mutation::fast_evaluate_light`
    /// (if `fast`) and `Mutation::evaluate_light(epsilon, ..)` (space.rs:1496-1540) with
    /// `LightPhysics::Rays { maximum_distance }`. The device's light volume -- what the following frames trace -- is
    /// updated in place; the host `Space`'s light is not touched. Batches of 32 in the table order of the reference's
    /// queue on x86-64, i.e. the texels `Space::evaluate_light` itself would produce with feature "auto-threads".
    /// Returns the number of cube updates.
    ///
    /// # Errors
    /// As [`HeadlessRenderer::draw`] for device failures.
    pub fn evaluate_light(&mut self, maximum_distance: u8, fast: bool, epsilon: u8) -> Result<u64, RenderError> {
        let params = ffi::aic_light_params {
            maximum_distance: i32::from(maximum_distance),
            fast: i32::from(fast),
            epsilon: i32::from(epsilon),
            batch: 32,
            queue_order: if cfg!(target_arch = "x86_64") { 16 } else { 8 }, // hashbrown's Group::WIDTH on the host the goldens came from
            n_queue: -1,
            lanes_per_cube: 0,
            reserved: 0,
            queue_cubes: core::ptr::null(),
            queue_priorities: core::ptr::null(),
            max_updates: 0,
        };
        let mut info = ffi::aic_light_info::default();
        // SAFETY: ctx is a live context; params and info outlive the call
        check(self.ctx.as_ptr(), unsafe { ffi::aic_evaluate_light(self.ctx.as_ptr(), ffi::AIC_LAYER_WORLD, &params, &mut info) })?;
        Ok(info.updates)
    }
}

impl Drop for HipRtRenderer {
    fn drop(&mut self) {
        // SAFETY: the context came from aic_create and is dropped exactly once
        unsafe { ffi::aic_destroy(self.ctx.as_ptr()) }
    }
}

impl HeadlessRenderer for HipRtRenderer {
    fn update(&mut self, read_tickets: Layers<ReadTicket<'_>>, cursor: Option<&Cursor>) -> Result<(), RenderError> {
        self.update_scene(read_tickets, cursor)
    }

    fn draw<'a>(&'a mut self, info_text: &'a str) -> BoxFuture<'a, Result<Rendering, RenderError>> {
        // `draw` must not touch the universe (headless.rs:36-38): everything it needs was forwarded by `update`
        Box::pin(async move { self.draw_rgba(info_text) })
    }
}

/// A mutex-wrapped renderer for call sites that need `Sync` (e.g. the recording thread, record.rs:97-113).
pub type SharedHipRtRenderer = Arc<Mutex<HipRtRenderer>>;

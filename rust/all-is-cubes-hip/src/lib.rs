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
//! | `RtRenderer::draw_rgba` (renderer.rs:282-308)          | `aic_render` / `aic_multi_render`      |
//! | `draw_info_text` (renderer.rs:659-683)                 | the same few lines, on the host        |
//!
//! One renderer may span several GPUs of a node ([`HipRtRenderer::with_devices`]): the scene is replicated, every device
//! traces its interleaved 8-row strips and device 0 assembles the frame (`aic_create_multi`, csrc/aic_multi.cpp).
//! [`HipRtRenderer::set_device_light`] hands the light itself over to the device: block edits are queued there
//! (`aic_light_cubes_changed`) and [`HipRtRenderer::evaluate_light_budgeted`] advances the light between frames.
//!
//! With `AIC_DUMP=path` in the environment the library records every call it receives; the recording replays on
//! a machine without the Rust toolchain (`python -m all_is_cubes_amd.replay path` in the MI355X repository) -- that is
//! how a real Atrium / DemoCity scene reaches its benchmark and parity tests.
//!
//! NOTE: written against all-is-cubes 0.10.0. THIS CRATE HAS NEVER BEEN COMPILED: the repository that carries it has no Rust
//! toolchain (tests/test_rust_shim.py only keeps `ffi.rs` in step with include/aic_hip.h). Expect to adjust imports and
//! signatures to the workspace it is dropped into; the C++ host mirror (all_is_cubes_amd/host/) is the tested twin of this
//! file, function for function.

#![warn(missing_docs)]

use std::ffi::CStr;
use std::ptr::NonNull;
use std::sync::{Arc, Mutex};

use all_is_cubes::character::Cursor;
use all_is_cubes::listen::{self, Listen as _};
use all_is_cubes::math::{Cube, Rgba, ZeroOne};
use all_is_cubes::space::{BlockIndex, Space, SpaceChange};
use all_is_cubes::text;
use all_is_cubes::universe;
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

/// One context per HIP device (`aic_create`), or the single-process multi-device context (`aic_create_multi`).
#[derive(Clone, Copy)]
enum Device {
    One(NonNull<ffi::aic_ctx>),
    Many(NonNull<ffi::aic_multi>),
}

impl Device {
    fn check(self, rc: core::ffi::c_int) -> Result<(), RenderError> {
        if rc == ffi::AIC_OK {
            return Ok(());
        }
        // SAFETY: both return a NUL-terminated string owned by the context
        let message = unsafe {
            CStr::from_ptr(match self {
                Device::One(c) => ffi::aic_last_error(c.as_ptr()),
                Device::Many(m) => ffi::aic_multi_last_error(m.as_ptr()),
            })
        }
        .to_string_lossy()
        .into_owned();
        match rc {
            // the reference asserts on the same conditions (e.g. output length, renderer.rs:193-197)
            ffi::AIC_ERR_INVALID => panic!("libaic_hip rejected a call: {message}"),
            // RenderError has no device variant yet (lib.rs:46-54 "TODO: add errors for out of memory, lost GPU");
            // until it does, a lost device is reported the way an unreadable scene is. (Whether `HandleError` can be
            // built from a message outside its crate is unchecked -- this file was never compiled: a maintainer adds the
            // RenderError variant that TODO asks for.)
            _ => {
                log::error!("libaic_hip: {message}");
                Err(RenderError::Read(all_is_cubes::universe::HandleError::from_message(message)))
            }
        }
    }
    /// The context the light updater runs on (a sequential relaxation: it does not shard; csrc/aic_multi.cpp).
    #[allow(dead_code)]
    fn first_ctx(self) -> *mut ffi::aic_ctx {
        match self {
            Device::One(c) => c.as_ptr(),
            // SAFETY: a live multi context has at least one device
            Device::Many(m) => unsafe { ffi::aic_multi_context(m.as_ptr(), 0) },
        }
    }
    // SAFETY (every method below): the context is live, the borrowed arguments outlive the call, the library copies them
    fn upload_space(self, layer: core::ffi::c_int, desc: &ffi::aic_space_desc) -> Result<(), RenderError> {
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_upload_space(c.as_ptr(), layer, desc),
                Device::Many(m) => ffi::aic_multi_upload_space(m.as_ptr(), layer, desc),
            }
        })
    }
    fn clear_space(self, layer: core::ffi::c_int) -> Result<(), RenderError> {
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_clear_space(c.as_ptr(), layer),
                Device::Many(m) => ffi::aic_multi_clear_space(m.as_ptr(), layer),
            }
        })
    }
    fn replace_blocks(
        self,
        layer: core::ffi::c_int,
        indices: &[u32],
        descs: &[ffi::aic_block_desc],
        voxels: &[*const u16],
        palettes: &[*const f32],
    ) -> Result<(), RenderError> {
        let n = indices.len() as u32;
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_replace_blocks(c.as_ptr(), layer, n, indices.as_ptr(), descs.as_ptr(), voxels.as_ptr(), palettes.as_ptr()),
                Device::Many(m) => ffi::aic_multi_replace_blocks(m.as_ptr(), layer, n, indices.as_ptr(), descs.as_ptr(), voxels.as_ptr(), palettes.as_ptr()),
            }
        })
    }
    /// `light` = None: block indices only (the light of these cubes is the device's business, see `set_device_light`).
    fn update_cubes(self, layer: core::ffi::c_int, xyz: &[i32], idx: &[u16], light: Option<&[u8]>) -> Result<(), RenderError> {
        let n = idx.len() as u32;
        let light = light.map_or(core::ptr::null(), <[u8]>::as_ptr);
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_update_cubes(c.as_ptr(), layer, n, xyz.as_ptr(), idx.as_ptr(), light),
                Device::Many(m) => ffi::aic_multi_update_cubes(m.as_ptr(), layer, n, xyz.as_ptr(), idx.as_ptr(), light),
            }
        })
    }
    fn set_options(self, layer: core::ffi::c_int, options: &ffi::aic_options) -> Result<(), RenderError> {
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_set_options(c.as_ptr(), layer, options),
                Device::Many(m) => ffi::aic_multi_set_options(m.as_ptr(), layer, options),
            }
        })
    }
    fn render(self, frame: &ffi::aic_frame_desc, out: &mut [[u8; 4]], info: &mut ffi::aic_frame_info) -> Result<(), RenderError> {
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_render(c.as_ptr(), frame, out.as_mut_ptr().cast(), 0, info),
                Device::Many(m) => ffi::aic_multi_render(m.as_ptr(), frame, out.as_mut_ptr().cast(), 0, info),
            }
        })
    }
    fn evaluate_light(self, layer: core::ffi::c_int, params: &ffi::aic_light_params, info: &mut ffi::aic_light_info) -> Result<(), RenderError> {
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_evaluate_light(c.as_ptr(), layer, params, info),
                // runs on device 0 and hands the resulting volume to the others
                Device::Many(m) => ffi::aic_multi_evaluate_light(m.as_ptr(), layer, params, info),
            }
        })
    }
    /// `aic_evaluate_light_submit` / `_wait`: the update on the context's worker thread, published by the wait (single device only:
    /// the multi-device context runs the updater on device 0 and hands the volume over, which is a blocking step).
    fn evaluate_light_submit(self, layer: core::ffi::c_int, params: &ffi::aic_light_params) -> Result<bool, RenderError> {
        match self {
            // SAFETY: live handle; the library copies `params` before it returns
            Device::One(c) => self.check(unsafe { ffi::aic_evaluate_light_submit(c.as_ptr(), layer, params) }).map(|()| true),
            Device::Many(_) => Ok(false),
        }
    }
    fn evaluate_light_wait(self, layer: core::ffi::c_int, info: &mut ffi::aic_light_info) -> Result<(), RenderError> {
        match self {
            // SAFETY: live handle, `info` is a valid out-pointer
            Device::One(c) => self.check(unsafe { ffi::aic_evaluate_light_wait(c.as_ptr(), layer, info) }),
            Device::Many(_) => Ok(()),
        }
    }
    /// `aic_light_cubes_changed` for cubes whose block `update_cubes` just changed (updater.rs:135-173).
    fn light_cubes_changed(self, layer: core::ffi::c_int, xyz: &[i32], n: u32) -> Result<(), RenderError> {
        // SAFETY: live handle, `xyz` holds 3 * n coordinates and outlives the call
        self.check(unsafe {
            match self {
                Device::One(c) => ffi::aic_light_cubes_changed(c.as_ptr(), layer, n, xyz.as_ptr(), QUEUE_ORDER),
                Device::Many(m) => ffi::aic_multi_light_cubes_changed(m.as_ptr(), layer, n, xyz.as_ptr(), QUEUE_ORDER),
            }
        })
    }
    fn destroy(self) {
        // SAFETY: called once, from Drop
        unsafe {
            match self {
                Device::One(c) => ffi::aic_destroy(c.as_ptr()),
                Device::Many(m) => ffi::aic_destroy_multi(m.as_ptr()),
            }
        }
    }
}

/// hashbrown's `Group::WIDTH` on the host the reference's goldens came from: the order of equal-priority light updates
const QUEUE_ORDER: core::ffi::c_int = if cfg!(target_arch = "x86_64") { 16 } else { 8 };

/// The MI355X raytracer behind the reference's renderer interface.
pub struct HipRtRenderer {
    device: Device,
    cameras: StandardCameras,
    size_policy: Box<dyn Fn(Viewport) -> Viewport + Send + Sync>,
    layers: Layers<Option<LayerSync>>,
    had_cursor: bool,
    /// `Some(maximum_distance)`: the world layer's light is computed on the device (see `set_device_light`)
    device_light: Option<u8>,
}

// SAFETY: the contexts are only used through `&mut self`; libaic_hip keeps no thread affinity besides hipSetDevice,
// which every entry point performs itself.
unsafe impl Send for HipRtRenderer {}

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
        LightingOption::Bounce { samples } => (5, i32::from(samples)), // traced on the device (secondary rays inside the SHADE event)
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
        Self::with_devices(cameras, size_policy, &[device_id])
    }

    /// One renderer over several GPUs of the node: the scene is replicated on each, every device traces its interleaved
    /// 8-row strips of a frame, the strips go to `device_ids[0]` over the direct links and are assembled there
    /// (`aic_create_multi`). With one id this is [`Self::new`].
    ///
    /// # Errors
    /// Fails if any of the devices is not a usable MI355X.
    pub fn with_devices(
        cameras: StandardCameras,
        size_policy: Box<dyn Fn(Viewport) -> Viewport + Send + Sync>,
        device_ids: &[i32],
    ) -> Result<Self, String> {
        // SAFETY: plain FFI calls; null returns are handled
        assert_eq!(unsafe { ffi::aic_abi_version() }, ffi::AIC_ABI_VERSION);
        let mut status = 0;
        let device = match device_ids {
            [] => return Err("no device given".into()),
            [one] => Device::One(
                NonNull::new(unsafe { ffi::aic_create(*one, &mut status) })
                    .ok_or_else(|| format!("aic_create failed with status {status} (no usable MI355X?)"))?,
            ),
            many => Device::Many(
                NonNull::new(unsafe { ffi::aic_create_multi(many.len() as core::ffi::c_int, many.as_ptr(), &mut status) })
                    .ok_or_else(|| format!("aic_create_multi failed with status {status}"))?,
            ),
        };
        Ok(Self { device, cameras, size_policy, layers: Layers::default(), had_cursor: false, device_light: None })
    }

    /// Hands the world space's light over to the device: from now on `update` forwards block changes
    /// WITHOUT the host's light texels and queues the changed cubes in the device's own update queue
    /// (`LightStorage::modified_cube_needs_update`, updater.rs:135-173, as `aic_light_cubes_changed`), and
    /// [`Self::evaluate_light_budgeted`] advances the device's light between frames. The host `Space`'s light is
    /// neither read nor written. `None` returns to forwarding the host's light.
    pub fn set_device_light(&mut self, maximum_distance: Option<u8>) {
        self.device_light = maximum_distance;
    }

    fn sync_layer(
        device: Device,
        device_light: bool,
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
                    device.clear_space(layer)?;
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
            device.upload_space(layer, &flat.desc())?;
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
                device.replace_blocks(layer, &indices, &descs, &voxels, &palettes)?;
            }
            if !todo.cubes.is_empty() {
                let (xyz, idx, light) = gather_cubes(&space, todo.cubes.drain());
                if device_light && layer == ffi::AIC_LAYER_WORLD {
                    // block indices only; the device relights what changed. (A CubeLight message of the host Space lands
                    // here too and is harmless: the cube's block index is rewritten with the value it already has.)
                    device.update_cubes(layer, &xyz, &idx, None)?;
                    // (several devices: the updater runs on the first; `aic_multi_light_cubes_changed` scatters the texels it
                    //  wrote there into the others' volumes, so the next draw traces the same light on every device)
                    device.light_cubes_changed(layer, &xyz, idx.len() as u32)?;
                } else {
                    device.update_cubes(layer, &xyz, &idx, Some(&light))?;
                }
            }
        }
        sync.n_blocks = block_data.len();
        device.set_options(layer, &options_of(options))
    }

    /// `RtRenderer::update` (renderer.rs:96-141).
    pub fn update_scene(&mut self, read_tickets: Layers<ReadTicket<'_>>, cursor: Option<&Cursor>) -> Result<(), RenderError> {
        self.had_cursor = cursor.is_some();
        self.cameras.update(read_tickets);
        let (device, device_light) = (self.device, self.device_light.is_some());
        let world_options = self.cameras.graphics_options().clone();
        let ui_options = self.cameras.ui_view_state().graphics_options.clone();
        let world_space = self.cameras.world_space().get();
        Self::sync_layer(device, device_light, ffi::AIC_LAYER_WORLD, &mut self.layers.world, world_space.as_ref(), read_tickets.world, &world_options)?;
        let ui_space = self.cameras.ui_space().cloned();
        Self::sync_layer(device, device_light, ffi::AIC_LAYER_UI, &mut self.layers.ui, ui_space.as_ref(), read_tickets.ui, &ui_options)
    }

    /// The frame descriptor of the current cameras and the viewport it is drawn for.
    fn frame_desc(&self) -> (Viewport, ffi::aic_frame_desc) {
        let viewport = (self.size_policy)(self.cameras.viewport()); // renderer.rs:226-233
        let size = viewport.framebuffer_size;
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
            tuning: 0, // the library's choices (tile queues, kernel variant); `HipInfo` reports what ran
        };
        (viewport, frame)
    }

    /// `RtRenderer::draw_rgba` (renderer.rs:282-308).
    pub fn draw_rgba(&mut self, info_text: &str) -> Result<Rendering, RenderError> {
        let (viewport, frame) = self.frame_desc();
        let size = viewport.framebuffer_size;
        let mut data = vec![[0u8; 4]; (size.width as usize) * (size.height as usize)];
        let mut info = ffi::aic_frame_info::default();
        if !data.is_empty() {
            // zero-area viewports produce an empty image (cases viewport_zero, cases/src/lib.rs:1167-1212)
            self.device.render(&frame, &mut data, &mut info)?;
        }
        self.wrap_rendering(viewport, data, info, info_text)
    }

    /// The finished pixels as the trait's `Rendering`: info text, flaws, info.
    fn wrap_rendering(&self, viewport: Viewport, mut data: Vec<[u8; 4]>, info: ffi::aic_frame_info, info_text: &str) -> Result<Rendering, RenderError> {
        let size = viewport.framebuffer_size;
        let options = self.cameras.graphics_options();
        if options.debug_info_text && !info_text.is_empty() && !data.is_empty() {
            // renderer.rs:205-217: the encoder's black and white (exposure and tone mapping applied, as the kernel's encoder does)
            let camera = &self.cameras.cameras().world;
            let paint = [
                camera.post_process_color(Rgba::BLACK).to_srgb8(),
                camera.post_process_color(Rgba::WHITE).to_srgb8(),
            ];
            draw_info_text(&mut data, viewport, &paint, info_text);
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

    /// Queues the current view on `slot` (0..`AIC_MULTI_MAX_IN_FLIGHT`) and returns at once: the streaming half of `draw_rgba` for a caller that renders
    /// frame after frame (record.rs:97-113 steps the universe, updates and draws in a loop) -- `update_scene` for frame n + 1 and its `begin_frame` may run
    /// while frame n is still being traced and assembled. On a multi-device renderer this is `aic_multi_render_submit`; a single-device renderer has no
    /// host-target streaming entry point (`aic_render_submit` writes device memory), so the frame is drawn here and merely handed over by `finish_frame`.
    ///
    /// # Errors
    /// As [`HeadlessRenderer::draw`] for device failures; a busy slot is an error.
    pub fn begin_frame(&mut self, slot: u32) -> Result<PendingFrame, RenderError> {
        let (viewport, frame) = self.frame_desc();
        let size = viewport.framebuffer_size;
        let mut data = vec![[0u8; 4]; (size.width as usize) * (size.height as usize)];
        let mut info = ffi::aic_frame_info::default();
        let streamed = match self.device {
            // SAFETY: live handle; `data`'s heap buffer is not moved or freed before `finish_frame` (PendingFrame owns it), and holds width * height pixels
            Device::Many(m) if !data.is_empty() => {
                self.device.check(unsafe { ffi::aic_multi_render_submit(m.as_ptr(), &frame, data.as_mut_ptr().cast(), 0, slot) })?;
                true
            }
            _ => {
                if !data.is_empty() {
                    self.device.render(&frame, &mut data, &mut info)?;
                }
                false
            }
        };
        Ok(PendingFrame { slot, streamed, viewport, data, info })
    }

    /// Waits for the frame `begin_frame` queued and wraps it as `draw_rgba` would.
    ///
    /// # Errors
    /// As [`HeadlessRenderer::draw`] for device failures.
    pub fn finish_frame(&mut self, mut pending: PendingFrame, info_text: &str) -> Result<Rendering, RenderError> {
        if pending.streamed {
            if let Device::Many(m) = self.device {
                // SAFETY: live handle, `info` is a valid out-pointer; the library writes `pending.data` before this returns
                self.device.check(unsafe { ffi::aic_multi_render_wait(m.as_ptr(), pending.slot, &mut pending.info) })?;
            }
        }
        self.wrap_rendering(pending.viewport, pending.data, pending.info, info_text)
    }

    /// The cameras this renderer draws (as `RtRenderer::cameras`).
    pub fn cameras(&self) -> &StandardCameras {
        &self.cameras
    }

    fn light_params(maximum_distance: u8, fast: bool, epsilon: u8, n_queue: i32, max_updates: u64) -> ffi::aic_light_params {
        ffi::aic_light_params {
            maximum_distance: i32::from(maximum_distance),
            fast: i32::from(fast),
            epsilon: i32::from(epsilon),
            batch: 32, // update_light_from_queue's batch with feature "auto-threads" (updater.rs:212-252)
            queue_order: QUEUE_ORDER,
            n_queue,
            lanes_per_cube: 0,
            hooks: 0,
            queue_cubes: core::ptr::null(),
            queue_priorities: core::ptr::null(),
            max_updates,
        }
    }

    /// Runs the light updater ON THE DEVICE against the world space as last uploaded: `Mutation::fast_evaluate_light`
    /// (if `fast`) and `Mutation::evaluate_light(epsilon, ..)` (space.rs:1496-1540) with
    /// `LightPhysics::Rays { maximum_distance }`. The device's light volume -- what the following frames trace -- is
    /// updated in place; the host `Space`'s light is not touched. Batches of 32 in the table order of the reference's
    /// queue, i.e. the texels `Space::evaluate_light` itself would produce with feature "auto-threads".
    /// Returns the number of cube updates.
    ///
    /// # Errors
    /// As [`HeadlessRenderer::draw`] for device failures.
    pub fn evaluate_light(&mut self, maximum_distance: u8, fast: bool, epsilon: u8) -> Result<u64, RenderError> {
        // n_queue = -1: without `fast`, start from every cube whose texel is Uninitialized (Priority::UNINIT)
        let params = Self::light_params(maximum_distance, fast, epsilon, -1, 0);
        let mut info = ffi::aic_light_info::default();
        self.device.evaluate_light(ffi::AIC_LAYER_WORLD, &params, &mut info)?;
        Ok(info.updates)
    }

    /// With [`Self::set_device_light`] on: continues the device's own update queue (what `update` queued for the blocks
    /// that changed, and what earlier calls left) for at most `max_updates` cube updates -- the per-frame slice of
    /// `Space::step`'s light update (updater.rs:181-290), on the device. Returns (updates done, queue entries left).
    ///
    /// # Errors
    /// As [`HeadlessRenderer::draw`] for device failures.
    pub fn evaluate_light_budgeted(&mut self, max_updates: u64) -> Result<(u64, u32), RenderError> {
        let maximum_distance = self.device_light.expect("evaluate_light_budgeted needs set_device_light(Some(..))");
        // n_queue = 0: nothing new from the caller; the layer's queue is continued
        let params = Self::light_params(maximum_distance, false, 1, 0, max_updates);
        let mut info = ffi::aic_light_info::default();
        self.device.evaluate_light(ffi::AIC_LAYER_WORLD, &params, &mut info)?;
        Ok((info.updates, info.queue_left))
    }
}

impl HipRtRenderer {
    /// [`Self::evaluate_light_budgeted`] without blocking: the update runs on the library's worker thread beside the frames drawn
    /// meanwhile (which see the light as it stood); [`Self::finish_light_update`] -- or the next `update` -- publishes it. On a
    /// multi-device renderer the blocking call is made instead.
    ///
    /// # Errors
    /// As [`HeadlessRenderer::draw`] for device failures.
    pub fn start_light_update(&mut self, max_updates: u64) -> Result<(), RenderError> {
        let maximum_distance = self.device_light.expect("start_light_update needs set_device_light(Some(..))");
        let params = Self::light_params(maximum_distance, false, 1, 0, max_updates);
        if !self.device.evaluate_light_submit(ffi::AIC_LAYER_WORLD, &params)? {
            let mut info = ffi::aic_light_info::default();
            self.device.evaluate_light(ffi::AIC_LAYER_WORLD, &params, &mut info)?;
        }
        Ok(())
    }

    /// Waits for the update [`Self::start_light_update`] began and makes its light current. Returns (updates done, queue entries left).
    ///
    /// # Errors
    /// As [`HeadlessRenderer::draw`] for device failures.
    pub fn finish_light_update(&mut self) -> Result<(u64, u32), RenderError> {
        let mut info = ffi::aic_light_info::default();
        self.device.evaluate_light_wait(ffi::AIC_LAYER_WORLD, &mut info)?;
        Ok((info.updates, info.queue_left))
    }
}

/// A frame in flight ([`HipRtRenderer::begin_frame`]): the host buffer the library fills, and what wrapping it needs. Dropping it without
/// `finish_frame` leaves the slot busy until the renderer is dropped (the library waits for it then).
pub struct PendingFrame {
    slot: u32,
    streamed: bool,
    viewport: Viewport,
    data: Vec<[u8; 4]>,
    info: ffi::aic_frame_info,
}

/// `draw_info_text` of the reference (raytracer/renderer.rs:659-683, private there): the info text over the finished frame,
/// `FontSystem16` glyphs at offset (5, 5), outline in `paint[0]`, glyph pixels in `paint[1]`.
fn draw_info_text(output: &mut [[u8; 4]], viewport: Viewport, paint: &[[u8; 4]; 2], info_text: &str) {
    let size = viewport.framebuffer_size;
    let font = universe::Builtin::FontSystem16.read::<text::FontDef>();
    font.draw_str_monospaced(info_text, |pixel, value| {
        let (x, y) = (pixel.x + 5, pixel.y + 5);
        if x >= 0 && y >= 0 && (x as u32) < size.width && (y as u32) < size.height {
            output[y as usize * size.width as usize + x as usize] = paint[match value {
                text::Value::Outline => 0,
                text::Value::Foreground => 1,
            }];
        }
    });
}

impl Drop for HipRtRenderer {
    fn drop(&mut self) {
        self.device.destroy();
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

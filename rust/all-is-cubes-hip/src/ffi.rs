//! Raw bindings to libaic_hip.so: one item per declaration of `include/aic_hip.h` (AIC_ABI_VERSION 3), same
//! names, same field order. Kept in step with the header by `tests/test_rust_shim.py` of the MI355X repository.
#![allow(non_camel_case_types, missing_docs, clippy::missing_safety_doc)]

use core::ffi::{c_char, c_int, c_void};

pub const AIC_ABI_VERSION: c_int = 3;
pub const AIC_OK: c_int = 0;
pub const AIC_ERR_INVALID: c_int = 1;
pub const AIC_ERR_NO_DEVICE: c_int = 2;
pub const AIC_ERR_OOM: c_int = 3;
pub const AIC_ERR_DEVICE: c_int = 4;
pub const AIC_ERR_UNSUPPORTED: c_int = 5;
pub const AIC_LAYER_WORLD: c_int = 0;
pub const AIC_LAYER_UI: c_int = 1;
pub const AIC_BLOCK_ONE: u32 = 1;
pub const AIC_BLOCK_AIR: u32 = 2;
pub const AIC_FLAW_UNSUPPORTED: u32 = 1;
pub const AIC_FLAW_NO_BLOOM: u32 = 2;
pub const AIC_FRAME_COUNTERS: u32 = 1;
pub const AIC_FRAME_AUX: u32 = 2;
pub const AIC_FRAME_PIXEL_CENTERS: u32 = 4;
pub const AIC_FRAME_OUT_LINEAR: u32 = 8;
pub const AIC_FRAME_OUT_COLORBUF: u32 = 16;
pub const AIC_FRAME_NO_FEEDBACK: u32 = 32;
pub const AIC_MAX_IN_FLIGHT: u32 = 32;
pub const AIC_MULTI_MAX_IN_FLIGHT: u32 = 8;
pub const AIC_TUNE_QUEUES_SHIFT: u32 = 0;
pub const AIC_TUNE_SUPER_SHIFT: u32 = 4;
pub const AIC_TUNE_VARIANT_SHIFT: u32 = 9;
pub const AIC_VARIANT_AUTO: u32 = 0;
pub const AIC_VARIANT_PLAIN: u32 = 1;
pub const AIC_VARIANT_EXCHANGING: u32 = 2;
pub const AIC_VARIANT_RECORDING: u32 = 3;
pub const AIC_LIGHT_HOOK_SESSION: i32 = 1;
pub const AIC_LIGHT_HOOK_POOL_SHIFT: i32 = 8;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aic_block_desc {
    pub resolution: i32,
    pub vlo: [i32; 3],
    pub vsize: [i32; 3],
    pub vox_off: u32,
    pub pal_off: u32,
    pub pal_len: u32,
    pub flags: u32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aic_space_desc {
    pub lo: [i32; 3],
    pub size: [i32; 3],
    pub block_index: *const u16,
    pub light: *const u8,
    pub n_blocks: u32,
    pub blocks: *const aic_block_desc,
    pub voxels: *const u16,
    pub n_voxels: u64,
    pub palette: *const f32,
    pub n_palette: u64,
    pub sky_kind: i32,
    pub sky: [[f32; 3]; 8],
    pub block_sky: [[u8; 4]; 7],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aic_options {
    pub fog: i32,
    pub transparency: i32,
    pub threshold: f32,
    pub lighting: i32,
    pub bounce_samples: i32,
    pub antialiasing: i32,
    pub debug_pixel_cost: i32,
    pub tone_mapping: i32,
    pub maximum_intensity: f32,
    pub bloom_intensity: f32,
    pub view_distance: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aic_camera {
    pub inverse_projection_view: [f64; 16],
    pub exposure: f32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aic_partition {
    pub strip_rows: u32,
    pub n_parts: u32,
    pub part: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aic_frame_desc {
    pub width: u32,
    pub height: u32,
    pub world: aic_camera,
    pub ui: aic_camera,
    pub backdrop: [f32; 4],
    pub partition: aic_partition,
    pub flags: u32,
    pub tuning: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aic_frame_info {
    pub cubes_traced: u64,
    pub n_outer: u64,
    pub n_inner: u64,
    pub n_hits: u64,
    pub n_light: u64,
    pub kernel_ms: f32,
    pub total_ms: f32,
    pub rows_rendered: u32,
    pub flaws: u32,
    pub variant: u32,
    pub tile_queues: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aic_pixel_aux {
    pub hit: i32,
    pub cube: [i32; 3],
    pub voxel: [i32; 3],
    pub resolution: i32,
    pub face: i32,
    pub block_index: i32,
    pub cubes_traced: u32,
    pub layer: u32,
    pub t_distance: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aic_rc_step {
    pub cube: [i32; 3],
    pub face: i32,
    pub t_distance: f64,
    pub intersection_point: [f64; 3],
}

#[repr(C)]
pub struct aic_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct aic_multi {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aic_light_params {
    pub maximum_distance: i32,
    pub fast: i32,
    pub epsilon: i32,
    pub batch: i32,
    pub queue_order: i32,
    pub n_queue: i32,
    pub lanes_per_cube: i32,
    pub hooks: i32,
    pub queue_cubes: *const i32,
    pub queue_priorities: *const i32,
    pub max_updates: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aic_light_info {
    pub updates: u64,
    pub batches: u64,
    pub cost: u64,
    pub device_ms: f64,
    pub total_ms: f64,
    pub queue_left: u32,
    pub pad: u32,
    pub bundles_visited: u64,
}

unsafe extern "C" {
    pub fn aic_create(device_id: c_int, status: *mut c_int) -> *mut aic_ctx;
    pub fn aic_destroy(ctx: *mut aic_ctx);
    pub fn aic_last_error(ctx: *const aic_ctx) -> *const c_char;
    pub fn aic_abi_version() -> c_int;
    pub fn aic_device_name(ctx: *const aic_ctx, buf: *mut c_char, buf_len: u32) -> c_int;
    pub fn aic_upload_space(ctx: *mut aic_ctx, layer: c_int, space: *const aic_space_desc) -> c_int;
    pub fn aic_clear_space(ctx: *mut aic_ctx, layer: c_int) -> c_int;
    pub fn aic_update_cubes(ctx: *mut aic_ctx, layer: c_int, n: u32, xyz: *const i32, block_index: *const u16, light: *const u8) -> c_int;
    pub fn aic_update_light_volume(ctx: *mut aic_ctx, layer: c_int, light: *const u8) -> c_int;
    pub fn aic_replace_block(ctx: *mut aic_ctx, layer: c_int, index: u32, desc: *const aic_block_desc, voxels: *const u16, palette: *const f32) -> c_int;
    pub fn aic_replace_blocks(ctx: *mut aic_ctx, layer: c_int, n: u32, indices: *const u32, descs: *const aic_block_desc, voxels: *const *const u16, palettes: *const *const f32) -> c_int;
    pub fn aic_compact(ctx: *mut aic_ctx, layer: c_int) -> c_int;
    pub fn aic_set_options(ctx: *mut aic_ctx, layer: c_int, options: *const aic_options) -> c_int;
    pub fn aic_render(ctx: *mut aic_ctx, frame: *const aic_frame_desc, out_rgba8: *mut c_void, out_is_device: c_int, info: *mut aic_frame_info) -> c_int;
    pub fn aic_render_submit(ctx: *mut aic_ctx, frame: *const aic_frame_desc, out_device: *mut c_void, slot: u32) -> c_int;
    pub fn aic_render_wait(ctx: *mut aic_ctx, slot: u32, info: *mut aic_frame_info) -> c_int;
    pub fn aic_render_submit_batch(ctx: *mut aic_ctx, n_frames: u32, frames: *const aic_frame_desc, out_devices: *const *mut c_void, slot: u32) -> c_int;
    pub fn aic_render_wait_batch(ctx: *mut aic_ctx, slot: u32, n_frames: u32, infos: *mut aic_frame_info) -> c_int;
    pub fn aic_trace_patches(ctx: *mut aic_ctx, frame: *const aic_frame_desc, n: u32, rects: *const f64, out_rgba8: *mut c_void, aux: *mut aic_pixel_aux, info: *mut aic_frame_info) -> c_int;
    pub fn aic_partition_rows(height: u32, partition: *const aic_partition) -> u32;
    pub fn aic_assemble_strips(ctx: *mut aic_ctx, gathered_device: *const c_void, out_device: *mut c_void, width: u32, height: u32, strip_rows: u32, n_parts: u32) -> c_int;
    pub fn aic_assemble_strips_async(ctx: *mut aic_ctx, gathered_device: *const c_void, out_device: *mut c_void, width: u32, height: u32, strip_rows: u32, n_parts: u32) -> c_int;
    pub fn aic_assemble_strips_on(ctx: *mut aic_ctx, gathered_device: *const c_void, out_device: *mut c_void, width: u32, height: u32, strip_rows: u32, n_parts: u32, hip_stream: *mut c_void) -> c_int;
    pub fn aic_read_aux(ctx: *mut aic_ctx, out: *mut aic_pixel_aux, n_records: u64) -> c_int;
    pub fn aic_synchronize(ctx: *mut aic_ctx) -> c_int;
    pub fn aic_stream(ctx: *mut aic_ctx) -> *mut c_void;
    pub fn aic_wait_event(ctx: *mut aic_ctx, hip_event: *mut c_void) -> c_int;
    pub fn aic_stream_wait_frame(ctx: *mut aic_ctx, slot: u32, hip_stream: *mut c_void) -> c_int;
    pub fn aic_ortho_image_size(lo: *const i32, size: *const i32, resolution: c_int, width: *mut u32, height: *mut u32) -> c_int;
    pub fn aic_render_orthographic(ctx: *mut aic_ctx, layer: c_int, resolution: c_int, out_rgba8: *mut c_void, out_is_device: c_int, width: *mut u32, height: *mut u32, info: *mut aic_frame_info) -> c_int;
    pub fn aic_create_multi(n_devices: c_int, device_ids: *const c_int, status: *mut c_int) -> *mut aic_multi;
    pub fn aic_destroy_multi(m: *mut aic_multi);
    pub fn aic_multi_device_count(m: *const aic_multi) -> c_int;
    pub fn aic_multi_context(m: *mut aic_multi, i: c_int) -> *mut aic_ctx;
    pub fn aic_multi_last_error(m: *const aic_multi) -> *const c_char;
    pub fn aic_multi_upload_space(m: *mut aic_multi, layer: c_int, space: *const aic_space_desc) -> c_int;
    pub fn aic_multi_clear_space(m: *mut aic_multi, layer: c_int) -> c_int;
    pub fn aic_multi_update_cubes(m: *mut aic_multi, layer: c_int, n: u32, xyz: *const i32, block_index: *const u16, light: *const u8) -> c_int;
    pub fn aic_multi_update_light_volume(m: *mut aic_multi, layer: c_int, light: *const u8) -> c_int;
    pub fn aic_multi_replace_blocks(m: *mut aic_multi, layer: c_int, n: u32, indices: *const u32, descs: *const aic_block_desc, voxels: *const *const u16, palettes: *const *const f32) -> c_int;
    pub fn aic_multi_set_options(m: *mut aic_multi, layer: c_int, options: *const aic_options) -> c_int;
    pub fn aic_multi_render(m: *mut aic_multi, frame: *const aic_frame_desc, out_rgba8: *mut c_void, out_is_device: c_int, info: *mut aic_frame_info) -> c_int;
    pub fn aic_multi_render_submit(m: *mut aic_multi, frame: *const aic_frame_desc, out_rgba8: *mut c_void, out_is_device: c_int, slot: u32) -> c_int;
    pub fn aic_multi_render_wait(m: *mut aic_multi, slot: u32, info: *mut aic_frame_info) -> c_int;
    pub fn aic_probe_raycast(ctx: *mut aic_ctx, origin: *const f64, direction: *const f64, use_bounds: c_int, lo: *const i32, hi: *const i32, include_exit: c_int, max_steps: u32, out: *mut aic_rc_step, n_out: *mut u32, ended: *mut c_int) -> c_int;
    pub fn aic_probe_powf(ctx: *mut aic_ctx, x: *const f32, y: *const f32, n: u32, out: *mut f32) -> c_int;
    pub fn aic_probe_light_lut(ctx: *mut aic_ctx, out: *mut f32) -> c_int;
    // light propagation on the device (Space::evaluate_light / fast_evaluate_light)
    pub fn aic_evaluate_light(ctx: *mut aic_ctx, layer: c_int, params: *const aic_light_params, info: *mut aic_light_info) -> c_int;
    pub fn aic_evaluate_light_submit(ctx: *mut aic_ctx, layer: c_int, params: *const aic_light_params) -> c_int;
    pub fn aic_evaluate_light_wait(ctx: *mut aic_ctx, layer: c_int, info: *mut aic_light_info) -> c_int;
    pub fn aic_evaluate_light_poll(ctx: *mut aic_ctx, layer: c_int, done: *mut c_int) -> c_int;
    pub fn aic_light_cubes_changed(ctx: *mut aic_ctx, layer: c_int, n: u32, xyz: *const i32, queue_order: c_int) -> c_int;
    pub fn aic_read_light_volume(ctx: *mut aic_ctx, layer: c_int, out: *mut u8) -> c_int;
    pub fn aic_read_light_cubes(ctx: *mut aic_ctx, layer: c_int, n: u32, xyz: *const i32, out: *mut u8) -> c_int;
    pub fn aic_light_chart(weights: *mut f32, children: *mut u32, depth: *mut u32) -> u32;
    pub fn aic_probe_derived(ctx: *mut aic_ctx, layer: c_int, out: *mut f32, out_opaque: *mut u8) -> c_int;
    pub fn aic_probe_log2f(ctx: *mut aic_ctx, x: *const f32, n: u32, out: *mut f32) -> c_int;
    pub fn aic_multi_evaluate_light(m: *mut aic_multi, layer: c_int, params: *const aic_light_params, info: *mut aic_light_info) -> c_int;
    pub fn aic_multi_light_cubes_changed(m: *mut aic_multi, layer: c_int, n: u32, xyz: *const i32, queue_order: c_int) -> c_int;
}

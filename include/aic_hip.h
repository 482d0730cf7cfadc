/*
 * aic_hip.h -- C ABI of the MI355X-native voxel raytracer (libaic_hip.so).
 *
 * This is the drop-in boundary for ONE path of kpreid/all-is-cubes v0.10.0: the CPU
 * raytracer behind `HeadlessRenderer` / `RtRenderer`
 * (all-is-cubes-render/src/headless.rs:17-44, src/raytracer/renderer.rs:35-356).
 * The reference has no FFI for this path (all-is-cubes-render is #![forbid(unsafe_code)],
 * lib.rs:21); these entry points are what a new `all-is-cubes-hip` crate implementing
 * `HeadlessRenderer` would bind (see INTEGRATION.md for the Rust `extern "C"` block).
 * Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *  - plain C, plain pointers and sizes; no C++ / torch types.
 *  - every function returning `int` returns an AIC_* status; aic_last_error() gives text.
 *  - host buffers passed in are copied before the call returns; the caller keeps ownership.
 *  - a context is bound to one HIP device (one process per GPU); it may be moved between
 *    threads but used by one thread at a time (`&mut self`; RtRenderer is Send+Sync,
 *    renderer.rs:696-698).
 *  - all grids are Z-major: index = ((x-lo.x)*size.y + (y-lo.y))*size.z + (z-lo.z)
 *    (all-is-cubes-base/src/math/vol.rs:988-1023).
 */
#ifndef AIC_HIP_H
#define AIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIC_ABI_VERSION 3

/* status codes */
#define AIC_OK 0
#define AIC_ERR_INVALID 1      /* bad argument (maps to a panic/assert in the reference) */
#define AIC_ERR_NO_DEVICE 2    /* no usable HIP device */
#define AIC_ERR_OOM 3          /* device allocation failed => Flaws::OUT_OF_MEMORY-style degradation */
#define AIC_ERR_DEVICE 4       /* HIP runtime error / device lost => RenderError (lib.rs:46-54) */
#define AIC_ERR_UNSUPPORTED 5

/* layers (camera/stdcam.rs:21-28 `Layers { world, ui }`) */
#define AIC_LAYER_WORLD 0
#define AIC_LAYER_UI 1

/* Flaws bits reported in aic_frame_info.flaws (all-is-cubes-render/src/flaws.rs:20-80) */
#define AIC_FLAW_UNSUPPORTED 1u   /* an option was substituted (Flaws::UNSUPPORTED). Reported by nothing at present: LightingOption::Bounce, which
                                   * rounds 1-3 rendered as Linear (graphics_options.rs:460-467), is traced as the reference does since round 4 */
#define AIC_FLAW_NO_BLOOM 2u      /* renderer.rs:293-297 */

/* block descriptor flags */
#define AIC_BLOCK_ONE 1u  /* EvoxelsInner::One(voxel): resolution 1, palette[pal_off] is the voxel */
#define AIC_BLOCK_AIR 2u  /* block == AIR (TracingCubeData.always_invisible, sr.rs:543-564) */

/* One Space block-palette entry = `TracingBlock.voxels: Evoxels`
 * (sr.rs:568-587; all-is-cubes/src/block/eval/voxel_storage.rs:199-227). 48 bytes. */
typedef struct aic_block_desc {
    int32_t resolution;  /* 1..128, power of two (resolution.rs:18-31) */
    int32_t vlo[3];      /* stored voxel volume: lower corner ... */
    int32_t vsize[3];    /* ... and size; may be smaller than resolution^3 (outside => AIR) */
    uint32_t vox_off;    /* offset in u16 elements into aic_space_desc.voxels */
    uint32_t pal_off;    /* offset in entries into aic_space_desc.palette */
    uint32_t pal_len;
    uint32_t flags;      /* AIC_BLOCK_* */
    int32_t reserved;
} aic_block_desc;

/* Full snapshot of one Space = the inputs of `SpaceRaytracer::new` (sr.rs:64-88). */
typedef struct aic_space_desc {
    int32_t lo[3];
    int32_t size[3];
    const uint16_t *block_index; /* [n cubes] `Space` contents (space.rs:77) */
    const uint8_t *light;        /* [n cubes][4] PackedLight::as_texel r,g,b,status (light/data.rs:160-170) */
    uint32_t n_blocks;
    const aic_block_desc *blocks;
    const uint16_t *voxels;      /* pool of palette indices, Z-major per block */
    uint64_t n_voxels;
    const float *palette;        /* pool [n_palette][8]: Evoxel color rgba, emission rgb, pad. Every component must be what the
                                  * reference's Rgba / Rgb can hold (PositiveSign<f32> x 3 + ZeroOne<f32>, math/color.rs:288-314): not NaN,
                                  * not negative, alpha <= 1 -- anything else is rejected with AIC_ERR_INVALID by aic_upload_space /
                                  * aic_replace_block(s) (since round 3; the kernel's compositing and its table powf rely on it). A block
                                  * with pal_len > 0 (or AIC_BLOCK_ONE) needs a non-null palette, one with voxels a non-null voxel pool. */
    uint64_t n_palette;
    int32_t sky_kind;            /* 0 Sky::Uniform(sky[0]); 1 Sky::Octants (sky.rs:16-21) */
    float sky[8][3];             /* Rgb = PositiveSign<f32> x 3: not NaN, not negative (rejected with AIC_ERR_INVALID otherwise; -0.0 is kept as +0.0, as in
                                  * the reference's type, restricted_number.rs:283 -- palette components likewise) */
    uint8_t block_sky[7][4];     /* Sky::for_blocks(): faces nx,ny,nz,px,py,pz then mean, as texels (sky.rs:45-82) */
} aic_space_desc;

/* The subset of GraphicsOptions the raytracer honours (camera/graphics_options.rs:28-152). */
typedef struct aic_options {
    int32_t fog;              /* FogOption: 0 None, 1 Abrupt, 2 Compromise, 3 Physical */
    int32_t transparency;     /* TransparencyOption: 0 Surface, 1 Volumetric, 2 Threshold */
    float threshold;          /* Threshold(t) */
    int32_t lighting;         /* LightingOption: 0 None, 1 Flat, 2 Coarse, 3 Linear, 4 Smoothstep, 5 Bounce */
    int32_t bounce_samples;
    int32_t antialiasing;     /* AntialiasingOption: 0 None, 1 IfCheap, 2 Always */
    int32_t debug_pixel_cost;
    int32_t tone_mapping;     /* ToneMappingOperator: 0 Clamp, 1 Reinhard */
    float maximum_intensity;  /* may be +inf */
    float bloom_intensity;    /* only reported back as AIC_FLAW_NO_BLOOM */
    double view_distance;     /* already `repair()`ed: clamped to 1..10000 (graphics_options.rs:194-198) */
} aic_options;

/* One Camera as the kernel needs it (camera/camera_struct.rs:43-77). */
typedef struct aic_camera {
    double inverse_projection_view[16]; /* euclid order m11..m44; Camera.inverse_projection_view (412-416) */
    float exposure;                     /* Camera::exposure() (365-367): a PositiveSign<f32> -- a frame with a negative or NaN exposure is rejected */
    int32_t reserved;
} aic_camera;

/* Which rows of the frame this context renders: rows are grouped into strips of
 * `strip_rows`; strip s belongs to part (s % n_parts). n_parts = 1 renders everything.
 * (Image rows are independent work items in the reference too: renderer.rs:537-555.) */
typedef struct aic_partition {
    uint32_t strip_rows;
    uint32_t n_parts;
    uint32_t part;
    uint32_t reserved;
} aic_partition;

typedef struct aic_frame_desc {
    uint32_t width, height;      /* framebuffer size after size_policy (renderer.rs:226-233) */
    aic_camera world;            /* used if a world space is uploaded */
    aic_camera ui;               /* used if a UI space is uploaded */
    float backdrop[4];           /* UiViewState.backdrop; all-zero = none (renderer.rs:235-253) */
    aic_partition partition;
    uint32_t flags;              /* AIC_FRAME_* */
    uint32_t tuning;             /* 0 = the library's choices. Otherwise (measurement and tests; ABI 3 -- until then these were environment
                                  * variables the library read per frame: AIC_TILE_QUEUES, AIC_SUPER_SHIFT, AIC_XCHG_TILES):
                                  *   bits 0-3   number of tile queues the frame's work tiles are dealt to, 1..8 (0: one per XCD)
                                  *   bits 4-8   super-block edge in macro tiles, log2, PLUS ONE (0: the largest power of two within an eighth of
                                  *              the image height)
                                  *   bits 9-10  AIC_VARIANT_*: which production variant of the trace kernel runs (0: by the frame's size)
                                  * aic_frame_info.variant / .tile_queues report what ran. Any value gives the same image. */
} aic_frame_desc;

#define AIC_TUNE_QUEUES_SHIFT 0
#define AIC_TUNE_SUPER_SHIFT 4
#define AIC_TUNE_VARIANT_SHIFT 9
/* production variants of the trace kernel (aic_frame_desc.tuning to ask for one, aic_frame_info.variant to learn which ran) */
#define AIC_VARIANT_AUTO 0       /* request only: the exchanging variant for a frame of several tiles per resident wave */
#define AIC_VARIANT_PLAIN 1      /* persistent waves, per-lane refill */
#define AIC_VARIANT_EXCHANGING 2 /* ... and lanes exchanged between the waves of a workgroup through a pool of parked rays */
#define AIC_VARIANT_RECORDING 3  /* report only: the variant that writes aic_pixel_aux records / the four byte counters (AIC_FRAME_AUX, AIC_FRAME_COUNTERS) */

#define AIC_FRAME_COUNTERS 1u /* also accumulate n_outer/n_inner/n_hits/n_light */
#define AIC_FRAME_AUX 2u      /* also write the per-pixel aic_pixel_aux records */
/* float output instead of sRGB RGBA8: the output buffer holds [rows][width] float[4] (16 bytes per pixel) */
#define AIC_FRAME_OUT_LINEAR 8u    /* Rgba::from(ColorBuf) (raytracer_components.rs:141-163): linear r,g,b,a before exposure and
                                    * tone mapping -- what a float framebuffer wants */
#define AIC_FRAME_OUT_COLORBUF 16u /* the ColorBuf itself: premultiplied light r,g,b and transmittance
                                    * (raytracer_components.rs:20-39; raytrace_to_texture.rs:638-655 reads exactly this) */
#define AIC_FRAME_NO_FEEDBACK 32u  /* neither use nor record the tile-cost feedback (the order in which this context hands out
                                    * work tiles, learnt from its previous frame of the same shape and camera): a "cold" single
                                    * frame, as the first frame of any sequence is */
#define AIC_FRAME_PIXEL_CENTERS 4u /* one ray through each pixel centre, Viewport::normalize_fb_x/_y (viewport.rs:89-99),
                                    * as the text renderer casts them (sr.rs:400-472); default: the image path's patch
                                    * centres / antialiasing points (renderer.rs:424-451) */

/* `RaytraceInfo` / `ImageInfo` (sr.rs:508-537; renderer.rs:617-647) + kernel timing. */
typedef struct aic_frame_info {
    uint64_t cubes_traced; /* RaytraceInfo.cubes_traced summed over the rendered pixels */
    uint64_t n_outer;      /* in-bounds cube-grid lookups */
    uint64_t n_inner;      /* in-bounds voxel lookups */
    uint64_t n_hits;       /* surfaces converted to light (Surface::to_light returned Some) */
    uint64_t n_light;      /* light texel fetches */
    float kernel_ms;       /* HIP-event time of the trace kernel on the context's stream */
    float total_ms;        /* launch + read-back wall time of the call */
    uint32_t rows_rendered;
    uint32_t flaws;        /* AIC_FLAW_* */
    uint32_t variant;      /* AIC_VARIANT_*: the trace kernel variant that traced the frame's world pass (ABI 3) */
    uint32_t tile_queues;  /* tile queues the frame's work tiles were dealt to (0: the single counter -- patch batches, orthographic views) */
} aic_frame_info;

/* Optional per-pixel record (N4 "other accumulators": first-hit id and depth): the first
 * Hit carrying a Position that reached the accumulator (raytracer/hit.rs:23-123). */
typedef struct aic_pixel_aux {
    int32_t hit;           /* 0 none, 1 surface */
    int32_t cube[3];
    int32_t voxel[3];
    int32_t resolution;
    int32_t face;          /* Face7 discriminant */
    int32_t block_index;   /* index into the block table of the layer named by `layer` */
    uint32_t cubes_traced; /* steps taken by this pixel's rays */
    uint32_t layer;        /* AIC_LAYER_WORLD / AIC_LAYER_UI: the layer the first hit belongs to */
    double t_distance;
} aic_pixel_aux;

typedef struct aic_ctx aic_ctx;

/* --- lifetime ------------------------------------------------------------------------ */
/* replaces: RtRenderer::new (renderer.rs:65-81). device_id < 0 selects the current device. */
aic_ctx *aic_create(int device_id, int *status);
void aic_destroy(aic_ctx *ctx);
const char *aic_last_error(const aic_ctx *ctx);
int aic_abi_version(void);
/* name of the HIP device the context is bound to (for Info / logging) */
int aic_device_name(const aic_ctx *ctx, char *buf, uint32_t buf_len);

/* --- scene snapshot: RtRenderer::update -> UpdatingSpaceRaytracer::update -------------- */
/* replaces: SpaceRaytracer::new on SpaceChange::EveryBlock / first update
 * (updating.rs:107-126; sr.rs:64-88, prepare_cubes 543-549).
 * Limits (AIC_ERR_INVALID beyond them): the kernel addresses the cube grid + voxel volumes (2 bytes per element) and the light volume (4 bytes per cube)
 * by 32-bit byte offsets, so cubes + voxels < 2^31 and cubes < 2^30 (a 1024^3 space is one cube too many); at most 65536 blocks. */
int aic_upload_space(aic_ctx *ctx, int layer, const aic_space_desc *space);
/* replaces: `rts.<layer> = None` (renderer.rs:134-135). */
int aic_clear_space(aic_ctx *ctx, int layer);
/* replaces: the `todo.cubes` scatter of UpdatingSpaceRaytracer::update (updating.rs:146-166)
 * fed by SpaceChange::{CubeBlock, CubeLight} (updating.rs:201-219). block_index or light may be
 * NULL to leave that field unchanged. xyz are absolute cube coordinates. */
int aic_update_cubes(aic_ctx *ctx, int layer, uint32_t n, const int32_t *xyz, const uint16_t *block_index,
                     const uint8_t *light);
/* replaces: a bulk SpaceChange::CubeLight burst after a light-propagation step (config 5):
 * re-uploads the whole light volume ([n cubes][4]). */
int aic_update_light_volume(aic_ctx *ctx, int layer, const uint8_t *light);
/* replaces: the `todo.blocks` re-evaluation of UpdatingSpaceRaytracer::update (updating.rs:128-145)
 * for SpaceChange::{BlockIndex, BlockEvaluation}: replaces or appends (index == n_blocks) one
 * palette entry. `voxels`/`palette` hold only this block's data (desc offsets are ignored). */
int aic_replace_block(aic_ctx *ctx, int layer, uint32_t index, const aic_block_desc *desc, const uint16_t *voxels,
                      const float *palette);
/* The same for a batch of blocks under ONE wait for the frames in flight and one synchronisation (a tick that
 * re-evaluates many animated blocks: updating.rs:128-145 runs once per changed block index). A block whose new voxel
 * volume and palette fit its current ranges is overwritten in place; one that outgrew them is appended, and the pools
 * are re-packed on the device once the ranges left behind exceed a quarter of a pool. */
int aic_replace_blocks(aic_ctx *ctx, int layer, uint32_t n, const uint32_t *indices, const aic_block_desc *descs,
                       const uint16_t *const *voxels, const float *const *palettes);
/* Re-pack the layer's voxel and palette pools now (done automatically by aic_replace_block(s) past a threshold). */
int aic_compact(aic_ctx *ctx, int layer);
/* replaces: the graphics_options DynSource (updating.rs:24,68-73). */
int aic_set_options(aic_ctx *ctx, int layer, const aic_options *options);

/* --- drawing: RtRenderer::draw_rgba / HeadlessRenderer::draw --------------------------- */
/* replaces: RtRenderer::draw_rgba -> trace_scene_to_image_impl (renderer.rs:282-308, 516-556):
 * out_rgba8 receives the rows selected by frame->partition, compacted in increasing row order,
 * as sRGB RGBA8 row-major (headless.rs:52-67). `out_is_device` != 0: out_rgba8 is a device
 * pointer on the context's device (no read-back; used for the RCCL gather). */
int aic_render(aic_ctx *ctx, const aic_frame_desc *frame, void *out_rgba8, int out_is_device, aic_frame_info *info);
/* Streaming pair for frame sequences (the reference's recording loop renders frame after frame,
 * all-is-cubes-desktop/src/record.rs:97-113): aic_render_submit queues a frame on slot
 * 0..AIC_MAX_IN_FLIGHT-1 and returns at once; aic_render_wait blocks until that slot's frame is in
 * `out_device` and reports it. With several frames in flight the next frame's trace starts filling the
 * GPU while the previous frame's last rays finish; a frame of many tiles that is submitted while others are in flight is launched on a part of the
 * chip (a third with three others in flight, a quarter from four on), so that the frames in flight are resident side by side -- same pixels and counts,
 * a higher frame rate, a longer life of each frame; a frame submitted with nothing else in flight takes the whole chip. out_device must be a device pointer; the
 * AIC_FRAME_AUX flag is ignored here (use aic_render). Scene updates that change what a frame in flight
 * reads (aic_upload_space, aic_update_cubes, aic_replace_blocks, aic_set_options, aic_compact) wait for every
 * frame in flight first; aic_update_light_volume and aic_evaluate_light do NOT: they write the other half of the
 * light double buffer beside the frames in flight (which keep the half they were given) and only wait for a frame
 * still reading that other half.
 * Slots 0..7 are made with the context; a higher slot gets its stream and events when it is first used.
 * "The frame is in out_device" is all aic_render_wait (and aic_render with a device target) waits for: behind the frame the slot's
 * stream still prepares the slot's next frame (the frame's cost record becomes a tile order, counters are cleared -- microseconds of
 * work private to the slot). Work the caller orders behind the frame with aic_stream_wait_frame, or issues after the wait returns,
 * never has to wait for that. aic_render_wait does NOT drain the slot's stream: call aic_synchronize before handing aic_stream() to
 * other code that assumes it idle.
 * Slots are HIP streams; the runtime runs streams that share one of its GPU_MAX_HW_QUEUES hardware queues (4 when unset) one behind the other.
 * The library sets GPU_MAX_HW_QUEUES=8 when it is loaded unless the caller has set it (AIC_KEEP_HW_QUEUES=1: hands off); a host that has used
 * HIP before loading the library sets it itself before its first HIP call (INTEGRATION.md "Hardware queues"). */
#define AIC_MAX_IN_FLIGHT 32u
int aic_render_submit(aic_ctx *ctx, const aic_frame_desc *frame, void *out_device, uint32_t slot);
int aic_render_wait(aic_ctx *ctx, uint32_t slot, aic_frame_info *info);
/* Several frames in ONE launch (ABI 3): n_frames = 1, 2, 4 or 8 frames of the same size, partition, flags and tuning, each with its own cameras, backdrop
 * and output buffer (out_devices[i], device memory), under the scene and options as they stand -- the frames a recording loop knows ahead of time
 * (record.rs:97-113 steps a camera path over a scene that does not change), or a rank's shares of consecutive frames of a multi-GPU stream. The
 * persistent grid is split between the frames (every workgroup traces one of them, each frame has its own tile queues and cost record), so they are
 * resident side by side by construction: n small frames fill the chip like one large one -- one ramp, one tail, and the lane-exchanging variant where
 * a frame alone would be too small for it -- instead of n launches that share the hardware queues as the runtime sees fit. The batch occupies `slot`
 * like one frame: aic_render_wait reports the sums of its frames' counts, aic_render_wait_batch each frame's; aic_stream_wait_frame orders a foreign
 * stream behind all of them. Every frame's pixels and counts are those of the same frame submitted alone. */
int aic_render_submit_batch(aic_ctx *ctx, uint32_t n_frames, const aic_frame_desc *frames, void *const *out_devices, uint32_t slot);
int aic_render_wait_batch(aic_ctx *ctx, uint32_t slot, uint32_t n_frames, aic_frame_info *infos);
/* replaces: RtScene::trace_patch (renderer.rs:418-451) for a batch of pixel rectangles -- the call
 * all-is-cubes-gpu's raytrace_to_texture makes for its incremental pixel batches
 * (raytrace_to_texture.rs:603-633). rects = [n][4] {min.x, min.y, max.x, max.y} in normalized device
 * coordinates; each is traced like one image pixel (its centre, or the four antialiasing points).
 * out_rgba8 = [n] encoded pixels and aux (may be NULL) = [n] first-hit records, both host memory.
 * Only the cameras, backdrop and flags of `frame` are used. */
int aic_trace_patches(aic_ctx *ctx, const aic_frame_desc *frame, uint32_t n, const double *rects, void *out_rgba8,
                      aic_pixel_aux *aux, aic_frame_info *info);
/* number of rows / first rows a partition selects (host-side helper for buffer sizing) */
uint32_t aic_partition_rows(uint32_t height, const aic_partition *partition);
/* scatter compacted strips gathered from n_parts contexts back into a full frame, on device:
 * gathered = [n_parts][max_rows_per_part][width] pixels, out = [height][width]. */
int aic_assemble_strips(aic_ctx *ctx, const void *gathered_device, void *out_device, uint32_t width, uint32_t height,
                        uint32_t strip_rows, uint32_t n_parts);
/* The same de-interleave queued on the context's stream WITHOUT waiting for it: for an exchange loop that only enqueues
 * (aic_stream_wait_frame / aic_wait_event order the device side). The image is complete once the stream is (aic_synchronize, or an
 * event of the caller's recorded behind it through aic_stream). */
int aic_assemble_strips_async(aic_ctx *ctx, const void *gathered_device, void *out_device, uint32_t width, uint32_t height,
                              uint32_t strip_rows, uint32_t n_parts);
/* ... or on a stream of the caller's on the context's device (hip_stream: a hipStream_t; NULL = the context's own): the exchange step of a pipeline that
 * keeps its copies and de-interleaves off the streams the frames are traced on (aic_multi_render_submit does). */
int aic_assemble_strips_on(aic_ctx *ctx, const void *gathered_device, void *out_device, uint32_t width, uint32_t height,
                           uint32_t strip_rows, uint32_t n_parts, void *hip_stream);
/* read back the aux records of the last aic_render issued with AIC_FRAME_AUX
 * ([rows_rendered][width]). */
int aic_read_aux(aic_ctx *ctx, aic_pixel_aux *out, uint64_t n_records);
/* blocks until all work queued on the context's stream is complete */
int aic_synchronize(aic_ctx *ctx);
/* the context's HIP stream (hipStream_t) for callers that time or order work against it */
void *aic_stream(aic_ctx *ctx);
/* Everything the context queues from now on (on any of its streams) waits for `hip_event` (a hipEvent_t recorded by
 * the caller on a stream of its own, e.g. after an RCCL gather that reads or writes buffers the next frames touch):
 * orders foreign work before the context's without blocking the host. */
int aic_wait_event(aic_ctx *ctx, void *hip_event);
/* The mirror of aic_wait_event: `hip_stream` (a hipStream_t of the caller's, e.g. the stream an RCCL gather of the frame's strips is
 * issued from) waits ON THE DEVICE for the frame submitted on `slot` -- hipStreamWaitEvent on the event aic_render_submit recorded
 * behind the slot's trace. The host does not block and the slot stays occupied (aic_render_wait still collects it, later, off
 * the exchange step's path). A slot with no frame in flight is a no-op. Together the two calls hand a frame from the trace to
 * the gather and the ring slot back to the trace without a host round trip (the reference has no counterpart: its image loop,
 * renderer.rs:516-556, returns a finished image to one caller). */
int aic_stream_wait_frame(aic_ctx *ctx, uint32_t slot, void *hip_stream);

/* --- orthographic views (icons, previews) -------------------------------------------------- */
/* replaces: raytracer::ortho::render_orthographic (all-is-cubes-render/src/raytracer/ortho.rs:30-88): five pixel-perfect
 * axis-aligned views of the layer's space (top, left, front, right, bottom: MultiOrthoCamera, ortho.rs:142-200) at
 * `resolution` pixels per cube, traced with GraphicsOptions::UNALTERED_COLORS by trace_axis_aligned_ray
 * (sr.rs:126-133) and encoded with Rgba::to_srgb8; pixels between the views are transparent. out_rgba8 =
 * [*height][*width] RGBA8 (NULL: only report the size). info->cubes_traced counts every pixel's trace; the
 * reference's per-row pixel cache (ortho.rs:103-131) skips repeated voxel columns and so reports fewer steps for
 * the same image. */
int aic_ortho_image_size(const int32_t lo[3], const int32_t size[3], int resolution, uint32_t *width, uint32_t *height);
int aic_render_orthographic(aic_ctx *ctx, int layer, int resolution, void *out_rgba8, int out_is_device, uint32_t *width,
                            uint32_t *height, aic_frame_info *info);

/* --- several devices in one process ------------------------------------------------------ */
/* No reference counterpart: the reference's renderer is one object, and so is this -- an aic_multi owns one context
 * per device (ids may repeat), replicates the scene calls on all of them, and renders a frame by dealing 8-row
 * strips round-robin to the devices (rows are independent work items in the reference: renderer.rs:537-555),
 * copying each device's compact strips to device_ids[0] over its direct xGMI link (hipMemcpyPeerAsync) and
 * de-interleaving there. The multi-PROCESS form of the same partition (one rank per GPU, RCCL gather) is what
 * bench.py drives; both produce the single-device frame byte for byte. RGBA8 output only. */
typedef struct aic_multi aic_multi;
aic_multi *aic_create_multi(int n_devices, const int *device_ids, int *status);
void aic_destroy_multi(aic_multi *m);
int aic_multi_device_count(const aic_multi *m);
aic_ctx *aic_multi_context(aic_multi *m, int i);
const char *aic_multi_last_error(const aic_multi *m);
int aic_multi_upload_space(aic_multi *m, int layer, const aic_space_desc *space);
int aic_multi_clear_space(aic_multi *m, int layer);
int aic_multi_update_cubes(aic_multi *m, int layer, uint32_t n, const int32_t *xyz, const uint16_t *block_index, const uint8_t *light);
int aic_multi_update_light_volume(aic_multi *m, int layer, const uint8_t *light);
int aic_multi_replace_blocks(aic_multi *m, int layer, uint32_t n, const uint32_t *indices, const aic_block_desc *descs,
                             const uint16_t *const *voxels, const float *const *palettes);
int aic_multi_set_options(aic_multi *m, int layer, const aic_options *options);
/* out_rgba8: [height][width] RGBA8; out_is_device != 0: a device pointer on device_ids[0] */
int aic_multi_render(aic_multi *m, const aic_frame_desc *frame, void *out_rgba8, int out_is_device, aic_frame_info *info);
/* The streaming pair of the multi-device context (ABI 3), as aic_render_submit / aic_render_wait are of one context's: submit queues the frame on
 * slot 0..AIC_MULTI_MAX_IN_FLIGHT-1 of every device and returns at once -- each device traces its strips, device 0's transfer stream waits for the
 * shares on the device, copies them over the peers' links and de-interleaves --; wait blocks until that slot's frame is in out_rgba8 (a device pointer on
 * device_ids[0], or host memory: then the read-back happens in the wait) and reports it. With two or more slots in use the next frame's traces run
 * under this frame's copies: the only way a caller that renders frame after frame (all-is-cubes-desktop/src/record.rs:97-113) keeps N devices busy.
 * Scene calls (aic_multi_upload_space ...) wait for the frames in flight as the single-context ones do. aic_multi_render is submit + wait on slot 0. */
#define AIC_MULTI_MAX_IN_FLIGHT 8u
int aic_multi_render_submit(aic_multi *m, const aic_frame_desc *frame, void *out_rgba8, int out_is_device, uint32_t slot);
int aic_multi_render_wait(aic_multi *m, uint32_t slot, aic_frame_info *info);

/* --- device-side probes used by the parity tests (not part of the render path) --------- */
/* runs Raycaster::new(origin,dir)[.within(lo,hi,include_exit)] on the device, one ray,
 * and writes up to max_steps {cube[3], face, t_distance} records (raycast.rs:239-284). */
typedef struct aic_rc_step {
    int32_t cube[3];
    int32_t face;
    double t_distance;
    double intersection_point[3];
} aic_rc_step;
int aic_probe_raycast(aic_ctx *ctx, const double origin[3], const double direction[3], int use_bounds,
                      const int32_t lo[3], const int32_t hi[3], int include_exit, uint32_t max_steps,
                      aic_rc_step *out, uint32_t *n_out, int *ended);
/* the device's f32::powf as apply_transmittance uses it (raytracer_components.rs:215-258): out[i] = x[i]^y[i] */
int aic_probe_powf(aic_ctx *ctx, const float *x, const float *y, uint32_t n, float *out);
/* the device's PackedLight decode table (light/data.rs:301-354) */
int aic_probe_light_lut(aic_ctx *ctx, float out[256]);

/* ---- light propagation on the device ----
 * The light volume that the tracer reads is produced in the reference by Space's light updater
 * (all-is-cubes/src/space/light/updater.rs). These entry points run that updater against the uploaded space: the
 * priority queue and the apply step on the host, Space::compute_light (updater.rs:368-417) for a batch of queue
 * entries at a time on the device. The light volume is updated IN PLACE on the device (no aic_update_light_volume
 * round trip); aic_read_light_volume copies it out. */
typedef struct aic_light_params {
    int32_t maximum_distance; /* LightPhysics::Rays { maximum_distance } (space.rs: u8) */
    int32_t fast;             /* nonzero: Mutation::fast_evaluate_light first (updater.rs:537-581): discards the current
                                 light and queue, estimates sky light per column, queues every cube near a surface */
    int32_t epsilon;          /* Mutation::evaluate_light(epsilon, ..) (space.rs:1496-1527): stop when the queue's highest
                                 priority is at most Priority::from_difference(epsilon) */
    int32_t batch;            /* queue entries computed against one light state before any is applied: 32 = the reference
                                 built with "auto-threads" (updater.rs:231-268), 1 = without; larger = throughput */
    int32_t queue_order;      /* order of equal-priority updates: 16 or 8 = the table order of the reference's queue
                                 (hashbrown Group::WIDTH of the build target: 16 on x86-64, 8 elsewhere); 0 = first in, first out */
    int32_t n_queue;          /* when !fast: entries to ADD to the layer's queue, inserted in order (what
                                 modified_cube_needs_update, updater.rs:135-173, enqueues after a change); < 0: every cube
                                 whose texel is Uninitialized, at Priority::UNINIT */
    int32_t lanes_per_cube;   /* how compute_light is mapped to the device: 256 (or 0 = default) / 64: a block of four waves / one
                                 wave per cube walks the ray-bundle tree level by level and the contributions are added in the
                                 reference's order; 1 = one lane per cube (the plain restatement). Same results, bit for bit */
    int32_t hooks;            /* 0 in normal use. Test / experiment hooks (ABI 3; environment variables read per call until then): bit 0 = serve a call's
                                 small batches from one session kernel fed through pinned host memory (AIC_LIGHT_HOOK_SESSION; measured neutral,
                                 profiles/r04_experiments.txt J); bits 8-23 = chunks the layer's dependency pool STARTS with (0: eight per cube),
                                 to force the grow-and-recompute path */
    const int32_t *queue_cubes;      /* [n_queue][3] */
    const int32_t *queue_priorities; /* [n_queue], 0..255 */
    uint64_t max_updates;     /* stop once this many cubes were updated (checked between batches); 0 = run until done */
} aic_light_params;
#define AIC_LIGHT_HOOK_SESSION 1
#define AIC_LIGHT_HOOK_POOL_SHIFT 8
typedef struct aic_light_info {
    uint64_t updates;     /* cubes computed and applied (LightUpdatesInfo::update_count) */
    uint64_t batches;     /* device launches */
    uint64_t cost;        /* sum of ComputedLight::cost */
    double device_ms;     /* time inside the gather kernel */
    double total_ms;      /* wall time of the call */
    uint32_t queue_left;  /* entries left in the queue (at or below epsilon, or cut off by max_updates) */
    uint32_t pad;
    uint64_t bundles_visited;  /* ray-tree bundles visited by the device walk (walk_ray_tree calls that passed the weight and
                                * distance checks, updater.rs:427-530): the unit of work of the light updater's roofline (ABI 2) */
} aic_light_info;
int aic_evaluate_light(aic_ctx *ctx, int layer, const aic_light_params *params, aic_light_info *info);
/* The same update WITHOUT blocking the caller (ABI 3): aic_evaluate_light_submit copies `params` (and its queue arrays), hands the update to a worker thread
 * the context owns and returns; the update works on a spare light volume, a copy of the current one, so every frame submitted while it runs reads the
 * light as it stood. aic_evaluate_light_wait blocks until the update is done, PUBLISHES it -- the frames submitted from then on read the updated volume --
 * and reports it. A sim + render loop (Space::step's light budget, space.rs:1496-1540, then a frame) calls submit, submits its frame, and waits at the top
 * of the next step: the update's host half (queue, apply) and its launches run beside the frame instead of in front of it. One update at a time per
 * context; every other scene or light call (aic_update_cubes, aic_light_cubes_changed, aic_read_light_volume, aic_evaluate_light ...) first finishes and
 * publishes a pending update, whose report stays available to aic_evaluate_light_wait. The volume after the wait is byte for byte what the blocking call
 * produces. Nothing submitted for `layer`: the wait returns zeros. */
int aic_evaluate_light_submit(aic_ctx *ctx, int layer, const aic_light_params *params);
int aic_evaluate_light_wait(aic_ctx *ctx, int layer, aic_light_info *info);
/* *done = 0 while a submitted update of `layer` is still running, 1 otherwise (finished and waiting to be collected, or none): a loop that must not stall
 * on the light (frames at their own pace, the light a step or two behind) asks before it waits. Publishes nothing. */
int aic_evaluate_light_poll(aic_ctx *ctx, int layer, int *done);
/* After aic_update_cubes changed the blocks of these cubes: what LightStorage::modified_cube_needs_update (updater.rs:135-173) does
 * for each -- a cube that is now opaque for light gets PackedLight::OPAQUE at once, any other is queued at Priority::NEWLY_VISIBLE,
 * and so are the neighbours that are not opaque towards it. The queue is the layer's own and lives until the next
 * aic_upload_space: a following aic_evaluate_light with fast = 0 and n_queue = 0 drains it (in several calls if max_updates is
 * set: a per-frame budget, as Space::step gives its light updater). */
int aic_light_cubes_changed(aic_ctx *ctx, int layer, uint32_t n, const int32_t *xyz, int queue_order);
/* the layer's current light volume, [n cubes][4] PackedLight texels, Z-major */
int aic_read_light_volume(aic_ctx *ctx, int layer, uint8_t *out);
/* the texels of n cubes, out[n][4] (LightStorage::get, all-is-cubes/src/space/light/data.rs, over a list): served from the
 * updater's host mirror of the volume when that is current. AIC_ERR_INVALID for a cube outside the space. */
int aic_read_light_cubes(aic_ctx *ctx, int layer, uint32_t n, const int32_t *xyz, uint8_t *out);
/* The propagation chart (chart/generator.rs): returns the node count; fills weights[n][6] and children[n][6] when both
 * are non-null, and the depth of the tree. Host-only: needs no device or context. */
uint32_t aic_light_chart(float *weights, uint32_t *children, uint32_t *depth);
/* test probes: the per-block derived properties the updater reads (derived.rs:78-240): out[n_blocks][32] = colour rgba,
 * 6 face colours rgba (nx ny nz px py pz), emission rgb, visible; out_opaque[n_blocks][6]. And the device's f32::log2
 * as PackedLight::scalar_in uses it (data.rs:214-218). */
int aic_probe_derived(aic_ctx *ctx, int layer, float *out, uint8_t *out_opaque);
int aic_probe_log2f(aic_ctx *ctx, const float *x, uint32_t n, float *out);
/* aic_evaluate_light on the first device; the resulting light volume is handed to the others (the updater does not shard) */
int aic_multi_evaluate_light(aic_multi *m, int layer, const aic_light_params *params, aic_light_info *info);
/* aic_light_cubes_changed on the device that runs the light updater (device 0). The texels it writes there (PackedLight::OPAQUE at
 * the cubes that are now opaque) are scattered into the other devices' volumes before it returns -- n texels, not the volume -- so
 * all devices always trace the same light volume; if that hand-over fails, the whole volume goes over at the next
 * aic_multi_evaluate_light or aic_multi_render, whichever comes first. */
int aic_multi_light_cubes_changed(aic_multi *m, int layer, uint32_t n, const int32_t *xyz, int queue_order);

#ifdef __cplusplus
}
#endif
#endif /* AIC_HIP_H */

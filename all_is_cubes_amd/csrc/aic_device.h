// aic_device.h -- device-side data layout shared by the host ABI code and the kernels.
//
// HBM layout of one uploaded Space ("layer"):
//   pool    : u16 pool. [0, n cubes): the cube grid = block index per cube, Z-major (same
//             linearisation as the reference's Vol); then, per recursive block, a Z-major volume
//             of *device voxel codes* (see below). One pool so that both DDA levels index memory
//             the same way: pool[volume offset + (x*sy + y)*sz + z].
//   cls     : 2-bit class per block index (invisible single voxel / visible single voxel /
//             recursive), copied to LDS by the trace kernel so that classifying a cube costs no
//             second global load.
//   light   : u32  [n cubes]  PackedLight texel r | g<<8 | b<<16 | status<<24
//   blocks  : DevBlock [n blocks], 64 B each (one cache line per block entry)
//   voxels  : (inside pool) per block a Z-major volume of *device voxel codes*: the block's
//             palette is reordered at upload so invisible entries (alpha == 0 && emission == 0,
//             surface.rs:395) come first; code < DevBlock.n_invisible means "invisible voxel"
//             and needs no palette fetch.
//   palette : DevPaletteEntry pool, 32 B each (rgba f32x4, emission f32x3, pad)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aic {

struct DevBlock {
    float color[4];       // single-voxel colour (valid if kind == 0)
    float emission[3];
    uint32_t kind;        // low byte: 0 = single voxel (R1 / Evoxels::One), else the resolution (2..128);
                          // bit 31 (single voxel only): the voxel is invisible (alpha == 0 && emission == 0)
    uint32_t vlo_packed;  // stored voxel volume lower corner  x | y<<8 | z<<16  (each 0..127)
    uint32_t vsize_packed;  // stored voxel volume size      x | y<<8 | z<<16  (each 1..128)
    uint32_t vox_off;     // u16 units into DevLayer.pool (already past the cube grid)
    uint32_t pal_off;     // entries into the palette pool
    uint32_t n_invisible; // device voxel codes below this are invisible
    uint32_t pad[3];
};
static_assert(sizeof(DevBlock) == 64, "DevBlock is one 64-byte line");

struct DevPaletteEntry {
    float color[4];
    float emission[3];
    float pad;
};
static_assert(sizeof(DevPaletteEntry) == 32, "palette entry is 32 B");

struct DevOptions {
    int32_t fog;
    int32_t transparency;
    float threshold;
    int32_t lighting;       // LightingOption: 0 None, 1 Flat, 2 Coarse, 3 Linear, 4 Smoothstep, 5 Bounce
    int32_t antialiasing;
    int32_t debug_pixel_cost;
    int32_t tone_mapping;
    float maximum_intensity;
    int32_t bounce_samples;  // LightingOption::Bounce { samples } (u8)
    int32_t pad_;
    double view_distance;
};

struct DevLayer {
    const uint16_t *pool;    // u16 pool: the cube grid (block indices) at offset 0, then every block's
                             // voxel volume (device voxel codes); DevBlock.vox_off is an offset into it
    const uint32_t *cls;     // 2 bits per block index: 0 invisible single voxel, 1 visible single voxel, 2 recursive
    const uint32_t *light;
    const DevBlock *blocks;
    const DevPaletteEntry *palette;
    int32_t lo[3];
    int32_t size[3];
    uint32_t n_blocks;
    int32_t present;        // 0: no space uploaded for this layer
    int32_t air_index;      // block index flagged AIR, or -1
    int32_t sky_kind;
    float sky[8][3];
    uint32_t block_sky[7];  // texels: nx ny nz px py pz mean
    DevOptions opt;
    double inv[16];         // camera inverse_projection_view
    float exposure;
    uint32_t cls_in_code;   // cube-grid entries carry their block's class in the top two bits (below)
};

// Cube-grid entries of the pool when DevLayer.cls_in_code != 0 (block table of at most 16384 entries):
//   bits 0-13 block index, bits 14-15 the block's class (0 invisible single voxel, 1 visible single
//   voxel, 2 recursive) -- so the trace kernel classifies a cube with no second lookup.
// Larger block tables store the plain 16-bit index and the kernel consults the class table in LDS.
static constexpr uint32_t kCubeClassShift = 14u;
static constexpr uint32_t kCubeIndexMask = (1u << kCubeClassShift) - 1u;

constexpr uint32_t kMaxTileQueues = 8;  // one per XCD

struct DevCounters {
    unsigned long long cubes_traced;
    unsigned long long n_outer;
    unsigned long long n_inner;
    unsigned long long n_hits;
    unsigned long long n_light;
    unsigned long long bailed;    // waves of the exchanging variants that gave up waiting for rays in transit (spun_out): never, short of a bug -- the host
                                  // turns a non-zero count into AIC_ERR_DEVICE, because such a wave abandons its rays and pixels stay unwritten
    unsigned long long prof[48];  // AIC_PROFILE builds only
    uint32_t tile_next;           // dynamic tile dispenser of the persistent trace kernel
    uint32_t waves_done;          // waves of the world pass that have added their sums: the last one hands the sums to the host (DevFrame::host_counters)
    uint32_t tile_next_q[kMaxTileQueues][16];  // the same per tile queue (DevFrame::n_queues), a cache line each: [q][0] counts
#ifdef AIC_PROFILE
    uint32_t wave_prof[2048][4];  // per wave: start, first saw the queue dry, end (cycle counter), pixels taken
#endif
};

struct DevAux {  // == aic_pixel_aux
    int32_t hit;
    int32_t cube[3];
    int32_t voxel[3];
    int32_t resolution;
    int32_t face;
    int32_t block_index;
    uint32_t cubes_traced;
    uint32_t layer;  // layer of the first hit (0 world, 1 UI)
    double t_distance;
};

// One view of raytracer::ortho::MultiOrthoCamera (ortho.rs:142-200): an image rectangle, the pixel -> world transform
// (euclid row-vector 4x4, pixel centres included) and the unit direction of its axis-aligned rays.
struct DevOrthoView {
    uint32_t x0, y0, w, h;
    double m[16];
    double dir[3];
};

// What differs between the frames ONE launch of the trace kernel traces (aic_render_submit_batch: up to kMaxSub frames that share the scene, the options,
// the image shape and the partition -- a rank's shares of consecutive frames of a multi-GPU stream, a camera path over a still scene). Every persistent workgroup
// belongs to one of them for its whole life (sub-frame = workgroup index mod n_sub, n_sub a power of two), so everything here is wave-uniform: scalar loads from the
// kernel-argument segment at a uniform offset. Each sub-frame has its own counters, tile queues and cost record: the launch is n_sub independent frames that
// are resident together by construction, not n_sub kernels that HIP may or may not run side by side. A plain frame is n_sub = 1.
struct DevSub {
    double inv[16];          // the camera of the pass this launch runs (world pass: the world camera's inverse_projection_view; UI pre-pass: the UI camera's)
    float backdrop[4];
    float exposure;          // world Camera::exposure()
    int32_t has_backdrop;
    uint32_t *out;           // [local_rows][width] RGBA8 (or float4: out_mode)
    float4 *acc_buf;         // [samples][local_rows][width] ColorBuf {light rgb, transmittance}: written by the UI pre-pass, read by the world pass
    DevCounters *counters;
    // pinned host memory for the frame's five sums (cubes_traced, n_outer, n_inner, n_hits, n_light) and the `bailed` count, written by the last wave of the world pass
    // to finish: no copy launch behind the trace (a blit kernel that, with frames streamed, waits ~0.2 ms for a CU to have room). Null: the host copies.
    unsigned long long *host_counters;
    // cost feedback (aic_trace.hip order_tiles_kernel): tile_order[k] = k-th macro tile to hand out, longest
    // rays of the previous frame first (null: index order); tile_cost[macro tile] receives this frame's longest ray
    const uint32_t *tile_order;
    uint32_t *tile_cost;
    const uint32_t *queue_start;  // XCD-local tile queues: see DevFrame::n_queues
};
constexpr uint32_t kMaxSub = 8;

struct DevFrame {
    DevLayer layer;          // the layer this launch traces, BY VALUE: kernarg fields are fetched with
                             // scalar loads into SGPRs (a pointer to a device-memory struct costs a
                             // dependent vector load in front of every lookup)
    int32_t layer_transparency, layer_lighting;  // host-side copy, selects the kernel variant
    uint32_t width, height;
    int32_t antialias;       // world camera options: AntialiasingOption::Always
    float maximum_intensity; // world options (encoder)
    int32_t tone_mapping;
    uint32_t strip_rows, n_parts, part;
    uint32_t local_rows;     // rows this launch renders
    uint32_t tiles_x, tiles_y;  // tile grid over (width, local_rows)
    uint32_t tile;              // tile edge in pixels: 8 (default) or 16
    uint32_t macro;             // tiles are handed out macro x macro at a time (a power of two): neighbours stay together
    uint32_t macros_x, macros_y;  // macro-tile grid; tile_order / tile_cost are indexed by macro tile
    uint32_t n_cus;          // compute units of the device (sizes the persistent grid)
    uint32_t tiles_per_wave; // host-side: tiles a wave should get on average when the frame is smaller than the chip (1 = latency, 4 = streamed frames)
    int32_t pass;            // 0: final pass (world layer + encode); 1: UI pre-pass
    int32_t use_init;        // final pass: start each sample from acc_buf (written by the UI pre-pass)
    int32_t pixel_centers;   // AIC_FRAME_PIXEL_CENTERS
    int32_t out_mode;        // 0 sRGB RGBA8 (4 B/pixel); 1 linear Rgba f32x4; 2 ColorBuf f32x4 (16 B/pixel)
    const DevOrthoView *ortho;  // aic_render_orthographic: the views (device memory), else null
    int32_t ortho_n;
    const double *patches;   // aic_trace_patches: [n_patches][4] NDC rectangles replacing the pixel grid (pixel i = row-major index)
    uint32_t n_patches;
    DevAux *aux;             // [local_rows][width] or null (the recording variants; single frames only)
    // XCD-local tile queues (0: one queue for the whole chip, counters->tile_next). tile_order is then n_queues segments, segment q =
    // positions queue_start[q] .. queue_start[q+1] of it (queue_start: device array of n_queues + 1), each costliest first. A macro
    // tile belongs to the queue of the super-block it lies in (order_tiles_kernel), a workgroup starts on the queue of the XCD it runs on.
    uint32_t n_queues;
    uint32_t n_sub;          // frames this launch traces: 1, 2, 4 or 8 (DevSub)
    DevSub sub[kMaxSub];
    const float *light_lut;  // 256 floats
    const float *srgb_thr;   // 256 floats: srgb_thr[k] = smallest linear value whose sRGB8 encoding is >= k
    // Pixel-edge tables (host-made, per frame shape): edge_x[x] = x / width * 2 - 1 for x = 0..width, edge_y[y] = -(y / height * 2 - 1) for
    // y = 0..height -- Viewport's pixel edges (viewport.rs:104-113) evaluated once in the reference's own f64 operations, so that starting a
    // ray (and re-deriving its origin, below) costs two 16-byte loads instead of four f64 divisions. Null for patch batches and orthographic views.
    const double *edge_x, *edge_y;
    // Per-ray state the exchanging trace variants keep in global memory (aic_trace.hip "lane exchange"): the four ColorBuf::mean sums of the
    // ray's pixel, 16 bytes per (workgroup, LDS column), ONLY in frames traced with antialiasing (null otherwise). A ray's origin is not kept
    // at all by those variants: ENTER re-derives it from the pixel and the camera matrix (round 6; rounds 5 kept origin and direction here,
    // 64 bytes per column, which tripled C3's HBM-side traffic). Sized by the host from trace_ray_cold_bytes(); `ray_cold_groups` workgroups fit.
    uint4 *ray_cold;
    uint32_t ray_cold_groups;
    // How the kernel derives a ray from its pixel (aic_trace.hip ray_of_pixel), decided by the launcher from the fields above so that the common cameras take ONE scalar
    // fetch and branch instead of a chain of five dependent ones (round 6): 0 = the layer holds a space, pixel grid, edge tables, one part; 1 = the same with the strip
    // partition (n_parts > 1); 2 = anything else (pixel centres, patch rectangles, orthographic views, no tables, no space).
    uint32_t ray_mode;
    uint32_t exchange;       // host-side: launch the exchanging variant (aic_trace.hip "lane exchange") -- a frame with several tiles per persistent wave; a frame of
                             // about one tile per wave (a rank's share at N >= 4, small images) runs the variant without the pool, which it would only pay for
};

// order_tiles_kernel's jobs: one workgroup each -- the cost record to read (null: index order), the order and the queues' starts to write, and words to clear
// (the frame's counters) once the record has been read
struct OrderJobs {
    const uint32_t *cost[kMaxSub];
    uint32_t *order[kMaxSub];
    uint32_t *queue_start[kMaxSub];
    uint32_t *clear_words[kMaxSub];
};

// Largest work tile edge in pixels (DevFrame.tile is 8 by default, 16 with AIC_TILE=16). Row strips of the multi-GPU partition are a multiple of the
// DEFAULT tile (8 rows since the end of round 6): a tile works on the part's local rows, which are contiguous whatever the strips are, so a larger tile
// or a macro tile that spans two strips is correct, only less local.
constexpr int kTile = 16;
constexpr int kClsWords = 4096;  // 65536 blocks x 2 bits

}  // namespace aic

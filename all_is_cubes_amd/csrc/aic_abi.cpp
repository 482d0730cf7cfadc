// aic_abi.cpp -- host side of the C ABI declared in include/aic_hip.h.
//
// Owns the device-resident snapshot of up to two Spaces (world + UI), i.e. what the
// reference keeps in `RtRenderer.rts: Layers<Option<UpdatingSpaceRaytracer>>`
// (all-is-cubes-render/src/raytracer/renderer.rs:35-54), and launches the kernels of
// aic_trace.hip on a private HIP stream. No CPU rendering path exists here: every entry
// point fails with AIC_ERR_NO_DEVICE / AIC_ERR_DEVICE when HIP is unusable.

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <string>
#include <utility>
#include <vector>

#include "../../include/aic_hip.h"
#include "aic_device.h"

namespace aic {
void launch_trace_image(const DevFrame &F, bool diag, hipStream_t stream);
size_t trace_ray_cold_bytes(uint32_t n_cus, uint32_t *groups);
void launch_scatter_cubes(uint16_t *grid, uint32_t *light, const int32_t *xyz, const uint16_t *bi, const uint32_t *lt,
                          uint32_t n, const int lo[3], const int size[3], const uint32_t *cls, hipStream_t stream);
void launch_probe_powf(const float *x, const float *y, float *out, uint32_t n, hipStream_t stream);
void launch_order_tiles(const uint32_t *cost, uint32_t *order, uint32_t n_tiles, uint32_t macros_x, uint32_t sb_shift, uint32_t n_queues, uint32_t *queue_start,
                        hipStream_t stream, bool clear_cost = false, uint32_t *clear_words = nullptr, uint32_t n_clear_words = 0);
// the same for the frames of a batch, one workgroup each, in ONE launch (OrderJobs: aic_device.h)
void launch_order_tiles_jobs(const OrderJobs &jobs, uint32_t n_jobs, uint32_t n_tiles, uint32_t macros_x, uint32_t sb_shift, uint32_t n_queues, hipStream_t stream,
                             bool clear_cost, uint32_t n_clear_words);
void launch_tag_cubes(uint16_t *grid, size_t n, const uint32_t *cls, int from_tagged, int to_tagged, hipStream_t stream);
void launch_assemble_strips(const uint32_t *gathered, uint32_t *out, uint32_t w, uint32_t h, uint32_t strip_rows,
                            uint32_t n_parts, uint32_t max_rows, hipStream_t stream);
void launch_probe_raycast(const double *od, int use_bounds, const int *lohi, int include_exit, uint32_t max_steps,
                          double *out_rec, uint32_t *n_out, int *ended, hipStream_t stream);
}  // namespace aic

using namespace aic;

static_assert(sizeof(aic_pixel_aux) == sizeof(DevAux), "aux record layout");

// Frame slots are HIP streams, and the runtime deals streams onto GPU_MAX_HW_QUEUES hardware queues; streams that share a queue run their kernels one
// behind the other. With the variable unset a streamed C2 frame costs 0.86 ms instead of 0.35 (tools/hw_queues.py, profiles/r06_experiments.txt W):
// the launches of four frames in flight never overlap, and the part-grid sizing of streamed frames (submit_frames) then leaves two thirds of the chip
// idle. The runtime reads the variable when it starts -- at the process's first HIP call --, so a default is put in place when this library is loaded,
// which for a host that links it is before main(). A value the caller has set is left alone; AIC_KEEP_HW_QUEUES=1 leaves the runtime's own default.
// A host that has used HIP before it loads this library has to set the variable itself.
__attribute__((constructor)) static void aic_default_hw_queues() {
    const char *keep = std::getenv("AIC_KEEP_HW_QUEUES");
    if (keep && keep[0] && keep[0] != '0') return;
    setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
}
static_assert(sizeof(aic_block_desc) == 48, "aic_block_desc is 48 bytes");

struct LightState;                       // aic_light_host.inc
void light_state_free(LightState *s);
// keeps the light updater's host mirrors in step with an aic_update_cubes call (no-op if they were not current)
void light_state_cubes_updated(LightState *s, uint64_t version_before, uint64_t version_after, uint32_t n, const int32_t *xyz, const uint16_t *block_index,
                               const uint8_t *light);

namespace {

constexpr uint64_t kMaxPoolElems = 0x7ffffff0ull;  // u16 elements of cube grid + voxel volumes (32-bit byte offsets in the kernel)
constexpr uint64_t kMaxLightTexels = 0x3ffffff0ull; // cubes of a space: the SHADE event addresses light texels by 32-bit byte offsets (aic_lightmath.h)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;    // elements in use
    size_t cap = 0;  // elements allocated
    hipError_t ensure(size_t count, bool keep = false, hipStream_t stream = nullptr) {
        if (count <= cap) { n = count; return hipSuccess; }
        size_t new_cap = count + count / 4 + 16;
        T *np_ = nullptr;
        hipError_t e = hipMalloc((void **)&np_, new_cap * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(np_, p, cap * sizeof(T), hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                (void)hipFree(np_);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = np_;
        cap = new_cap;
        n = count;  // only now: a failed grow leaves the buffer as it was
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = cap = 0;
    }
};

struct Layer {
    bool present = false;
    int32_t lo[3] = {0, 0, 0}, size[3] = {0, 0, 0};
    DevBuf<uint16_t> pool;   // [0, n_cubes): cube grid; then the voxel volumes of the recursive blocks
    DevBuf<uint32_t> cls;    // 2-bit block classes
    std::vector<uint32_t> host_cls;
    DevBuf<uint32_t> light;
    // Spare light volumes: a new volume (aic_update_light_volume, aic_evaluate_light beside frames in flight) is made in one NO frame in flight reads and
    // becomes current for the frames submitted from then on. Two halves (rounds 2-5) made every update wait for the frame submitted two updates before --
    // a sim + render loop with four frames in flight was held to two --, so there are as many spares as it takes (up to kLightSpares), made on demand.
    static constexpr int kLightSpares = 7;
    DevBuf<uint32_t> light_spare[kLightSpares];
    DevBuf<DevBlock> blocks;
    DevBuf<DevPaletteEntry> palette;
    std::vector<DevBlock> host_blocks;  // mirror of the block table (for replace/append)
    // per block: elements of the voxel pool / palette pool its current ranges can hold (so that a re-evaluated block is
    // written in place when it fits, updating.rs:128-145), and what replaced blocks left behind (compacted past a threshold)
    // vox_base / pal_base: where the block's reserved ranges start. Kept apart from host_blocks[i].vox_off / pal_off, which are
    // 0 while the block has no voxels (an atom written over a voxel block keeps its reservation for the next re-evaluation).
    std::vector<uint32_t> vox_cap, pal_cap, vox_base, pal_base;
    uint64_t garbage_vox = 0, garbage_pal = 0;
    int32_t air_index = -1;
    int32_t sky_kind = 0;
    float sky[8][3] = {};
    uint32_t block_sky[7] = {};
    aic_options opt;
    bool opt_set = false;
    bool cls_in_code = false;  // cube-grid entries carry the block class in bits 14-15 (aic_device.h)
    uint64_t version = 0;      // bumped by every scene mutation; the light updater's host mirrors follow it
    uint64_t upload_serial = 0;  // bumped by aic_upload_space only: the light update queue lives as long as one upload
    LightState *lstate = nullptr;
    size_t n_cubes() const { return (size_t)size[0] * (size_t)size[1] * (size_t)size[2]; }
    void release() {
        pool.release(); cls.release(); light.release(); blocks.release(); palette.release();
        for (auto &sp : light_spare) sp.release();
        host_blocks.clear(); host_cls.clear(); vox_cap.clear(); pal_cap.clear(); vox_base.clear(); pal_base.clear();
        garbage_vox = garbage_pal = 0;
        present = false;
        version++;
        light_state_free(lstate);
        lstate = nullptr;
    }
};

aic_options default_options() {
    // GraphicsOptions::default() (graphics_options.rs:256-280)
    aic_options o;
    std::memset(&o, 0, sizeof(o));
    o.fog = 1;
    o.transparency = 1;
    o.threshold = 0.5f;
    o.lighting = 3;
    o.maximum_intensity = INFINITY;
    o.bloom_intensity = 0.125f;
    o.view_distance = 200.0;
    return o;
}

}  // namespace

struct aic_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t upload_stream = nullptr;  // light-volume uploads run beside the frames in flight
    Layer layers[2];
    DevBuf<float> lut;
    DevBuf<float> srgb_thr;
    DevBuf<uint32_t> out;      // internal RGBA8 target when the caller wants a host copy
    DevBuf<DevAux> aux;
    DevBuf<unsigned char> staging;  // scratch for scatter updates / probes
    DevBuf<DevOrthoView> ortho_views;  // aic_render_orthographic
    uint64_t aux_records = 0;
    bool streaming_submit = false;  // set around aic_render_submit: frames meant to overlap are sized for throughput, synchronous ones for latency
    // frames in flight: slot 0 runs on `stream` (and serves the synchronous aic_render), slot 1 on a
    // second stream so that a submitted frame's trace can start while the previous one drains
    // What each frame of a slot owns. A plain frame is sub-frame 0; aic_render_submit_batch traces up to kMaxSub frames in one launch (DevSub) and each has
    // its own counters, tile queues and cost record.
    struct SubSlot {
        DevBuf<DevCounters> counters;
        DevCounters *host_counters = nullptr;      // pinned
        // What a frame needs cleared or ordered is enqueued BEHIND the previous frame of the slot, not ahead of this one (round 4): the counters
        // are cleared again right after they were copied out, the cost record is turned into the next frame's tile order and cleared as soon as the
        // trace that wrote it is done. A frame alone then starts with its trace launch; before, three small launches (~25 us) stood in front of it.
        bool counters_clean = false;       // the device counters are zero (cleared behind the slot's last frame)
        size_t cost_clean_n = 0;           // this many entries of tile_cost are zero
        bool record_ready = false;         // tile_order / queue_start hold the cost order of the frame described by cost_sig / cost_cam / order_key
        uint32_t order_key[6] = {0, 0, 0, 0, 0, 0};  // cost_sig + number of queues + super-block shift
        DevBuf<float4> acc;  // UI pre-pass accumulators
        // cost feedback: the longest ray of every tile of the sub-frame's last frame, and the tile order made from it
        DevBuf<uint32_t> tile_cost, tile_order, queue_start;
        uint32_t cost_sig[4] = {0, 0, 0, 0};  // width, local rows, partition of the frame tile_cost describes
        double cost_cam[16] = {0};            // ... and its world camera
        void release() {
            counters.release(); acc.release(); tile_cost.release(); tile_order.release(); queue_start.release();
            if (host_counters) (void)hipHostFree(host_counters);
            host_counters = nullptr;
            counters_clean = record_ready = false; cost_clean_n = 0;
        }
    };
    struct FrameSlot {
        hipStream_t stream = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the trace launch(es): the frame's kernel time
        hipEvent_t ev2 = nullptr;                  // behind the copy of the counters to `host_counters`: what aic_render_wait waits for
        SubSlot sub[kMaxSub];
        uint32_t n_sub = 1;                // frames of the batch in flight
        bool static_ready = false;         // tile_static / queue_static hold the index order for static_key (a matter of the frame's shape: shared by a batch's frames)
        uint32_t static_key[6] = {0, 0, 0, 0, 0, 0};
        DevBuf<uint32_t> tile_static, queue_static;
        DevBuf<uint4> ray_cold;  // the exchanging trace kernels' antialiasing sums in global memory (DevFrame::ray_cold; antialiased frames only)
        DevBuf<double> edges;    // DevFrame::edge_x / edge_y of the slot's frame shape: width + 1, then height + 1 doubles
        uint32_t edges_w = 0, edges_h = 0;
        bool busy = false;
        bool diag = false;  // the slot's frame ran the aux-recording kernel variant
        uint32_t variant = 0, tile_queues = 0;  // what aic_frame_info reports of the slot's frame
        const void *light_used[2] = {nullptr, nullptr};  // per layer: the light buffer the slot's frame reads
        uint32_t flaws = 0, local_rows = 0;
        size_t npix = 0;
        std::chrono::steady_clock::time_point t_begin;
    } slots[AIC_MAX_IN_FLIGHT];
    std::string err;
    std::FILE *dump = nullptr;  // AIC_DUMP=path: every scene / options / frame argument is appended here
    char devname[256] = {0};
    uint32_t n_cus = 256;
    // aic_evaluate_light_submit / _wait: ONE light update at a time runs on a worker thread the context owns, against a spare light volume (Layer::light_spare);
    // the frames submitted meanwhile read the volume as it stood. The update is PUBLISHED -- the spare becomes the layer's volume -- by aic_evaluate_light_wait,
    // or by whatever call next needs the scene still (every scene or light call finishes a pending update first: light_job_finish).
    struct LightJob {
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        bool has_work = false, quit = false;   // (under mu)
        bool running = false;                   // (under mu) the worker is inside the update
        bool pending = false;                   // an update was submitted and has not been published yet (caller's thread only)
        bool result_ready = false;              // rc / info / err describe an update aic_evaluate_light_wait has not reported yet
        int layer = 0, spare = -1;
        aic_light_params params;
        std::vector<int32_t> queue_cubes, queue_priorities;
        int rc = AIC_OK;
        aic_light_info info;
        std::string err;
    } ljob;
    // measurement switches, read from the environment ONCE, when the context is made -- nothing on the frame path reads the environment
    struct Switches {
        int tile = 0, macro = 0;         // AIC_TILE, AIC_MACRO: work-tile edge in pixels (8 | 16), tiles per macro tile edge
        bool feedback = true;            // AIC_TILE_FEEDBACK=0: no cost-feedback tile order
        bool wait_whole_stream = false;  // AIC_WAIT_WHOLE_STREAM=1: aic_render_wait drains the slot's stream (rounds 1-3)
        uint32_t tiles_per_wave = 0;     // AIC_TILES_PER_WAVE: grid sizing of streamed frames smaller than the chip
        std::string wave_prof;           // AIC_WAVE_PROF (-DAIC_PROFILE builds): file for the per-wave clocks
    } sw;
};

namespace {

// (the light worker's failures go to its job's own string, not to the context's last error, which the caller's thread may be writing: see LightJob)
thread_local std::string *tl_err_sink = nullptr;
int fail(aic_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    if (c) {
        std::string &dst = tl_err_sink ? *tl_err_sink : c->err;
        dst = what;
        if (e != hipSuccess) {
            dst += ": ";
            dst += hipGetErrorString(e);
        }
    }
    return code;
}
int light_job_finish(aic_ctx *c);  // aic_light_host.inc
// ---- call recorder: AIC_DUMP=<path> makes a context append every argument it is given --
// scene snapshots, deltas, options, frame descriptors -- to <path>, verbatim, so that a scene produced by
// the reference (which cannot be generated here) can be captured where the Rust shim runs and replayed
// anywhere (all_is_cubes_amd/replay.py). Layout: "AICDUMP1", then records {u32 tag, u32 layer, u64 bytes, payload}.
enum DumpTag : uint32_t { DUMP_UPLOAD = 1, DUMP_CLEAR = 2, DUMP_CUBES = 3, DUMP_LIGHT = 4, DUMP_BLOCK = 5, DUMP_OPTIONS = 6, DUMP_FRAME = 7 };
struct DumpPart { const void *p; size_t n; };
void dump_record(aic_ctx *c, uint32_t tag, uint32_t layer, std::initializer_list<DumpPart> parts) {
    if (!c || !c->dump) return;
    uint64_t total = 0;
    for (const DumpPart &d : parts) total += d.p ? d.n : 0;
    std::fwrite(&tag, 4, 1, c->dump);
    std::fwrite(&layer, 4, 1, c->dump);
    std::fwrite(&total, 8, 1, c->dump);
    for (const DumpPart &d : parts)
        if (d.p && d.n) std::fwrite(d.p, 1, d.n, c->dump);
    std::fflush(c->dump);
}

int hip_fail(aic_ctx *c, const char *what, hipError_t e) {
    return fail(c, e == hipErrorOutOfMemory ? AIC_ERR_OOM : AIC_ERR_DEVICE, what, e);
}
#define HIP_TRY(ctx, expr)                                    \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return hip_fail(ctx, #expr, e_); \
    } while (0)

// A light volume of `n` cubes that no frame in flight reads: an existing spare, else a new one, else -- every spare in use by a frame -- the first one, once
// the frame reading it is done. *index says which (the caller swaps it with Layer::light when its contents are complete).
int take_light_spare(aic_ctx *c, Layer &l, int layer, size_t n, int *index) {
    auto in_use = [&](const DevBuf<uint32_t> &b) {
        for (uint32_t i = 0; i < AIC_MAX_IN_FLIGHT; i++)
            if (c->slots[i].busy && b.p && c->slots[i].light_used[layer] == (const void *)b.p) return true;
        return false;
    };
    int pick = -1;
    for (int k = 0; k < Layer::kLightSpares && pick < 0; k++)
        if (l.light_spare[k].p && l.light_spare[k].cap >= n && !in_use(l.light_spare[k])) pick = k;
    for (int k = 0; k < Layer::kLightSpares && pick < 0; k++)
        if (!l.light_spare[k].p || (l.light_spare[k].cap < n && !in_use(l.light_spare[k]))) pick = k;  // (a buffer of an earlier, smaller space is re-made)
    if (pick < 0) {
        pick = 0;
        for (uint32_t i = 0; i < AIC_MAX_IN_FLIGHT; i++)
            if (c->slots[i].busy && c->slots[i].light_used[layer] == (const void *)l.light_spare[0].p) {
                const hipError_t e = hipStreamSynchronize(c->slots[i].stream);
                if (e != hipSuccess) return fail(c, AIC_ERR_DEVICE, "waiting for a frame that reads a spare light volume", e);
            }
    }
    const hipError_t e = l.light_spare[pick].ensure(n);
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? AIC_ERR_OOM : AIC_ERR_DEVICE, "alloc light (spare volume)", e);
    *index = pick;
    return AIC_OK;
}

// Every scene mutation waits for the frames in flight: they read the buffers it is about to change.
int quiesce(aic_ctx *c) {
    { const int rc = light_job_finish(c); if (rc != AIC_OK) return rc; }  // (a light update in flight reads the scene too, and its result is published first)
    for (uint32_t i = 0; i < AIC_MAX_IN_FLIGHT; i++)
        if (c->slots[i].busy) HIP_TRY(c, hipStreamSynchronize(c->slots[i].stream));
    return AIC_OK;
}

bool valid_layer(int l) { return l == AIC_LAYER_WORLD || l == AIC_LAYER_UI; }
bool valid_resolution(int r) { return r >= 1 && r <= 128 && (r & (r - 1)) == 0; }

inline uint32_t block_class(const DevBlock &b) { return (b.kind & 255u) ? 2u : ((b.kind & 0x80000000u) ? 0u : 1u); }
inline void set_class(std::vector<uint32_t> &cls, uint32_t index, uint32_t c2) {
    if (cls.size() <= index / 16u) cls.resize(index / 16u + 1u, 0u);
    cls[index / 16u] = (cls[index / 16u] & ~(3u << ((index & 15u) * 2u))) | (c2 << ((index & 15u) * 2u));
}

inline bool invisible(const float *e) { return e[3] == 0.f && e[4] == 0.f && e[5] == 0.f && e[6] == 0.f; }

// Builds the device form of one block: palette reordered "invisible entries first", voxel
// codes remapped accordingly (aic_device.h). Appends to vox/pal pools at the given offsets.
int convert_block(aic_ctx *c, const aic_block_desc &d, const uint16_t *voxels, const float *palette, uint32_t vox_off,
                  uint32_t pal_off, DevBlock *out, std::vector<uint16_t> *vox_out, std::vector<DevPaletteEntry> *pal_out) {
    std::memset(out, 0, sizeof(*out));
    if (!valid_resolution(d.resolution)) return fail(c, AIC_ERR_INVALID, "block resolution must be a power of two in 1..128");
    const bool one = (d.flags & AIC_BLOCK_ONE) != 0;
    size_t nvox = one ? 1 : (size_t)d.vsize[0] * (size_t)d.vsize[1] * (size_t)d.vsize[2];
    for (int a = 0; a < 3 && !one; a++) {
        if (d.vsize[a] < 0 || d.vlo[a] < 0 || d.vlo[a] + d.vsize[a] > d.resolution)
            return fail(c, AIC_ERR_INVALID, "block voxel bounds exceed GridAab::for_block(resolution)");
    }
    if (d.pal_len == 0 && nvox > 0) return fail(c, AIC_ERR_INVALID, "block has voxels but an empty palette");
    if (!palette && (one || d.pal_len > 0)) return fail(c, AIC_ERR_INVALID, "block has a palette length but no palette");
    if (!voxels && !one && nvox > 0) return fail(c, AIC_ERR_INVALID, "block has a voxel volume but no voxels");
    // Evoxel colours are the reference's Rgba = PositiveSign<f32> x 3 + ZeroOne<f32> (math/color.rs:288-314): components that type
    // cannot hold (NaN, negative, alpha above 1) are rejected here, and the kernel's powf / compositing rely on it
    for (uint32_t i = 0; i < (one ? 1u : d.pal_len) && palette; i++) {
        const float *e = palette + 8 * (size_t)i;
        for (int k = 0; k < 7; k++)
            if (!(e[k] >= 0.f)) return fail(c, AIC_ERR_INVALID, "palette entry has a negative or NaN component");
        if (!(e[3] <= 1.f)) return fail(c, AIC_ERR_INVALID, "palette entry has alpha above 1");
    }
    if (one || d.resolution == 1) {
        // Evoxels::single_voxel() (voxel_storage.rs:364-385)
        const float *e = nullptr;
        static const float air[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (one) e = palette;
        else {
            // indices.get([0,0,0]) or Evoxel::AIR
            bool inb = nvox > 0 && d.vlo[0] <= 0 && d.vlo[1] <= 0 && d.vlo[2] <= 0 && d.vlo[0] + d.vsize[0] > 0 &&
                       d.vlo[1] + d.vsize[1] > 0 && d.vlo[2] + d.vsize[2] > 0;
            if (inb) {
                size_t i = ((size_t)(0 - d.vlo[0]) * d.vsize[1] + (size_t)(0 - d.vlo[1])) * d.vsize[2] + (size_t)(0 - d.vlo[2]);
                uint16_t idx = voxels[i];
                if (idx >= d.pal_len) return fail(c, AIC_ERR_INVALID, "voxel index exceeds palette");
                e = palette + 8 * (size_t)idx;
            } else e = air;
        }
        out->kind = invisible(e) ? 0x80000000u : 0u;
        // (+ 0.0f: a PositiveSign never holds -0.0 -- restricted_number.rs:283 -- and the kernel's ps_mul relies on the sign bit)
        for (int k = 0; k < 4; k++) out->color[k] = e[k] + 0.0f;
        for (int k = 0; k < 3; k++) out->emission[k] = e[4 + k] + 0.0f;
        return AIC_OK;
    }
    // recursive block
    std::vector<uint32_t> remap(d.pal_len);
    uint32_t n_inv = 0;
    for (uint32_t i = 0; i < d.pal_len; i++)
        if (invisible(palette + 8 * (size_t)i)) n_inv++;
    uint32_t next_inv = 0, next_vis = n_inv;
    size_t pal_base = pal_out->size();
    pal_out->resize(pal_base + d.pal_len);
    for (uint32_t i = 0; i < d.pal_len; i++) {
        const float *e = palette + 8 * (size_t)i;
        uint32_t dst = invisible(e) ? next_inv++ : next_vis++;
        remap[i] = dst;
        DevPaletteEntry &pe = (*pal_out)[pal_base + dst];
        for (int k = 0; k < 4; k++) pe.color[k] = e[k] + 0.0f;
        for (int k = 0; k < 3; k++) pe.emission[k] = e[4 + k] + 0.0f;
        pe.pad = 0.f;
    }
    size_t vox_base = vox_out->size();
    vox_out->resize(vox_base + nvox);
    for (size_t i = 0; i < nvox; i++) {
        uint16_t idx = voxels[i];
        if (idx >= d.pal_len) return fail(c, AIC_ERR_INVALID, "voxel index exceeds palette");
        (*vox_out)[vox_base + i] = (uint16_t)remap[idx];
    }
    out->kind = (uint32_t)d.resolution;
    out->vlo_packed = (uint32_t)d.vlo[0] | ((uint32_t)d.vlo[1] << 8) | ((uint32_t)d.vlo[2] << 16);
    out->vsize_packed = (uint32_t)d.vsize[0] | ((uint32_t)d.vsize[1] << 8) | ((uint32_t)d.vsize[2] << 16);
    out->vox_off = vox_off;
    out->pal_off = pal_off;
    out->n_invisible = n_inv;
    return AIC_OK;
}

void fill_dev_layer(const aic_ctx *c, const Layer &l, const aic_camera &cam, DevLayer *d, uint32_t *flaws) {
    std::memset(d, 0, sizeof(*d));
    d->present = l.present ? 1 : 0;
    d->pool = l.pool.p;
    d->cls = l.cls.p;
    d->n_blocks = (uint32_t)l.host_blocks.size();
    d->light = l.light.p;
    d->blocks = l.blocks.p;
    d->palette = l.palette.p;
    for (int a = 0; a < 3; a++) { d->lo[a] = l.lo[a]; d->size[a] = l.size[a]; }
    d->air_index = l.air_index;
    d->sky_kind = l.sky_kind;
    std::memcpy(d->sky, l.sky, sizeof(d->sky));
    std::memcpy(d->block_sky, l.block_sky, sizeof(d->block_sky));
    const aic_options o = l.opt_set ? l.opt : default_options();
    d->opt.fog = o.fog;
    d->opt.transparency = o.transparency;
    d->opt.threshold = o.threshold;
    d->opt.lighting = o.lighting;  // 5 = Bounce: traced as the reference does (surface.rs:119-166; aic_trace.hip bounce_secondary_ray)
    d->opt.bounce_samples = o.bounce_samples;
    d->opt.antialiasing = o.antialiasing;
    d->opt.debug_pixel_cost = o.debug_pixel_cost;
    d->opt.tone_mapping = o.tone_mapping;
    d->opt.maximum_intensity = o.maximum_intensity;
    d->opt.view_distance = o.view_distance;
    std::memcpy(d->inv, cam.inverse_projection_view, sizeof(d->inv));
    d->exposure = cam.exposure + 0.0f;
    d->cls_in_code = l.cls_in_code ? 1u : 0u;
    (void)c;
}

}  // namespace

extern "C" {

int aic_abi_version(void) { return AIC_ABI_VERSION; }

aic_ctx *aic_create(int device_id, int *status) {
    int st_dummy;
    if (!status) status = &st_dummy;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *status = AIC_ERR_NO_DEVICE;
        return nullptr;
    }
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess) device_id = 0;
    }
    if (device_id >= n) {
        *status = AIC_ERR_INVALID;
        return nullptr;
    }
    if (hipSetDevice(device_id) != hipSuccess) {
        *status = AIC_ERR_DEVICE;
        return nullptr;
    }
    aic_ctx *c = new aic_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) std::snprintf(c->devname, sizeof(c->devname), "%s (%s)", prop.name, prop.gcnArchName);
    {
        auto env_int = [](const char *name, int dflt) { const char *e = std::getenv(name); return e ? std::atoi(e) : dflt; };
        c->sw.tile = env_int("AIC_TILE", 0);
        c->sw.macro = env_int("AIC_MACRO", 0);
        c->sw.feedback = env_int("AIC_TILE_FEEDBACK", 1) != 0;
        c->sw.wait_whole_stream = env_int("AIC_WAIT_WHOLE_STREAM", 0) != 0;
        c->sw.tiles_per_wave = (uint32_t)std::max(0, env_int("AIC_TILES_PER_WAVE", 0));
        if (const char *p = std::getenv("AIC_WAVE_PROF")) c->sw.wave_prof = p;
    }
    if (const char *path = std::getenv("AIC_DUMP")) {
        static int n_dumps = 0;  // one file per context: <path>, <path>.1, <path>.2 ...
        std::string pth = path;
        if (n_dumps > 0) pth += "." + std::to_string(n_dumps);
        n_dumps++;
        c->dump = std::fopen(pth.c_str(), "wb");
        if (c->dump) std::fwrite("AICDUMP1", 1, 8, c->dump);
    }
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    // the first eight slots are made with the context (as AIC_MAX_IN_FLIGHT = 8 always was: HIP deals streams onto its hardware queues in creation order); slots
    // 8..31 -- a rank's many small shares of a multi-GPU frame -- get their stream and events on first use (ensure_slot)
    for (uint32_t i = 0; ok && i < 8u && i < AIC_MAX_IN_FLIGHT; i++) {
        aic_ctx::FrameSlot &fs = c->slots[i];
        if (i == 0) fs.stream = c->stream;
        else ok = hipStreamCreateWithFlags(&fs.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreate(&fs.ev0) == hipSuccess && hipEventCreate(&fs.ev1) == hipSuccess && hipEventCreate(&fs.ev2) == hipSuccess && fs.sub[0].counters.ensure(1) == hipSuccess;
    }
    // created after the frame streams: HIP deals streams onto a few hardware queues in creation order
    // (4 by default), and two frame slots sharing a queue would serialise their kernels
    ok = ok && hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) == hipSuccess;
    // PackedLight decode table: PACKED_LIGHT_SCALAR_LOOKUP_TABLE is *defined* as
    // exp2f((v - 144) / 10) with 0 -> 0 (light/data.rs:239-249, 301-354); correctly rounded.
    float lut[256];
    lut[0] = 0.f;
    for (int v = 1; v < 256; v++) {
        float arg = ((float)v - 144.0f) / 10.0f;
        lut[v] = (float)std::exp2((double)arg);
    }
    ok = ok && c->lut.ensure(256) == hipSuccess && hipMemcpy(c->lut.p, lut, sizeof(lut), hipMemcpyHostToDevice) == hipSuccess;
    // sRGB8 encode thresholds: thr[k] = the smallest f32 c with to_srgb8(c) >= k, found by bisection
    // on the reference formula (color.rs:1038-1054) evaluated here on the host with powf.
    {
        auto enc = [](float cc) -> int {
            float v = cc <= 0.0031308f ? cc * (323.f / 25.f) : (211.f * std::pow(cc, 5.f / 12.f) - 11.f) / 200.f;
            float r = std::round(v * 255.f);
            if (!(r > 0.f)) return 0;
            return r >= 255.f ? 255 : (int)r;
        };
        float thr[256];
        thr[0] = 0.f;
        for (int k = 1; k < 256; k++) {
            uint32_t lo_b = 0u, hi_b = 0x40000000u;  // (0.0, 2.0]: enc(lo) < k <= enc(hi)
            while (hi_b - lo_b > 1u) {
                const uint32_t mid = lo_b + (hi_b - lo_b) / 2u;
                float f;
                std::memcpy(&f, &mid, 4);
                if (enc(f) >= k) hi_b = mid; else lo_b = mid;
            }
            // the bisection assumes enc() is monotone; libm's powf need not be at the last bit, so settle the threshold on
            // the LOWEST float of the neighbourhood that encodes to >= k while its predecessor does not
            for (int back = 0; back < 8 && hi_b > 1u; back++) {
                uint32_t probe = hi_b - 1u;
                float f;
                std::memcpy(&f, &probe, 4);
                bool lower_found = false;
                for (uint32_t d = 0; d < 8u && probe > d; d++) {
                    const uint32_t q = probe - d;
                    std::memcpy(&f, &q, 4);
                    if (enc(f) >= k) { hi_b = q; lower_found = true; }
                }
                if (!lower_found) break;
            }
            std::memcpy(&thr[k], &hi_b, 4);
        }
        ok = ok && c->srgb_thr.ensure(256) == hipSuccess && hipMemcpy(c->srgb_thr.p, thr, sizeof(thr), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        *status = AIC_ERR_DEVICE;
        aic_destroy(c);
        return nullptr;
    }
    for (auto &l : c->layers) l.opt = default_options();
    *status = AIC_OK;
    return c;
}

void aic_destroy(aic_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)light_job_finish(c);  // a light update in flight ends (and is published) before anything it uses is freed ...
    if (c->ljob.worker.joinable()) {  // ... and the worker leaves
        { std::lock_guard<std::mutex> lk(c->ljob.mu); c->ljob.quit = true; }
        c->ljob.cv.notify_all();
        c->ljob.worker.join();
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (uint32_t i = 0; i < AIC_MAX_IN_FLIGHT; i++) {
        aic_ctx::FrameSlot &fs = c->slots[i];
        if (fs.stream) (void)hipStreamSynchronize(fs.stream);
        for (auto &sb : fs.sub) sb.release();
        fs.ray_cold.release(); fs.edges.release();
        if (fs.ev0) (void)hipEventDestroy(fs.ev0);
        if (fs.ev1) (void)hipEventDestroy(fs.ev1);
        if (fs.ev2) (void)hipEventDestroy(fs.ev2);
        fs.tile_static.release(); fs.queue_static.release();
        if (i > 0 && fs.stream) (void)hipStreamDestroy(fs.stream);
    }
    for (auto &l : c->layers) l.release();
    c->lut.release(); c->srgb_thr.release(); c->out.release(); c->aux.release(); c->staging.release(); c->ortho_views.release();
    if (c->dump) std::fclose(c->dump);
    if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *aic_last_error(const aic_ctx *c) { return c ? c->err.c_str() : "no context"; }

int aic_device_name(const aic_ctx *c, char *buf, uint32_t buf_len) {
    if (!c || !buf || !buf_len) return AIC_ERR_INVALID;
    std::snprintf(buf, buf_len, "%s", c->devname);
    return AIC_OK;
}

int aic_upload_space(aic_ctx *c, int layer, const aic_space_desc *s) {
    if (!c || !s || !valid_layer(layer)) return fail(c, AIC_ERR_INVALID, "aic_upload_space: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    { int qrc = quiesce(c); if (qrc != AIC_OK) return qrc; }
    for (int a = 0; a < 3; a++)
        if (s->size[a] < 0 || (int64_t)s->lo[a] + s->size[a] > 2147483647LL) return fail(c, AIC_ERR_INVALID, "space bounds out of range");
    Layer &l = c->layers[layer];
    const size_t n = (size_t)s->size[0] * (size_t)s->size[1] * (size_t)s->size[2];
    if (n && (!s->block_index || !s->light)) return fail(c, AIC_ERR_INVALID, "space has cubes but no block_index/light");
    if (s->n_blocks > 65536) return fail(c, AIC_ERR_INVALID, "more than 65536 blocks");
    // the trace kernel addresses the pool with 32-bit BYTE offsets: at most 2^31 u16 elements (4 GiB)
    if (n + s->n_voxels > kMaxPoolElems) return fail(c, AIC_ERR_INVALID, "space too large (cube grid + voxel pool must stay within 4 GiB)");
    if (n > kMaxLightTexels) return fail(c, AIC_ERR_INVALID, "space too large (the light volume, 4 bytes per cube, must stay within 4 GiB)");
    if (s->n_blocks && !s->blocks) return fail(c, AIC_ERR_INVALID, "blocks is null");
    // Sky colours are Rgb = PositiveSign<f32> x 3 (sky.rs:20-41): what that type cannot hold is rejected (as for palette entries, make_dev_block)
    for (int o = 0; o < 8; o++)
        for (int k = 0; k < 3; k++)
            if (!(s->sky[o][k] >= 0.f)) return fail(c, AIC_ERR_INVALID, "sky colour has a negative or NaN component");

    // block table + pools
    std::vector<DevBlock> blocks(s->n_blocks);
    std::vector<uint32_t> vox_cap(s->n_blocks, 0u), pal_cap(s->n_blocks, 0u), vox_base(s->n_blocks, 0u), pal_base(s->n_blocks, 0u);
    std::vector<uint16_t> vox;
    std::vector<DevPaletteEntry> pal;
    vox.reserve((size_t)s->n_voxels);
    pal.reserve((size_t)s->n_palette);
    int32_t air_index = -1;
    for (uint32_t i = 0; i < s->n_blocks; i++) {
        const aic_block_desc &d = s->blocks[i];
        const bool one = (d.flags & AIC_BLOCK_ONE) != 0;
        size_t nvox = one ? 1 : (size_t)(d.vsize[0] > 0 ? d.vsize[0] : 0) * (size_t)(d.vsize[1] > 0 ? d.vsize[1] : 0) * (size_t)(d.vsize[2] > 0 ? d.vsize[2] : 0);
        if ((uint64_t)d.vox_off + nvox > s->n_voxels || (uint64_t)d.pal_off + d.pal_len > s->n_palette)
            return fail(c, AIC_ERR_INVALID, "block voxel/palette range exceeds the pools");
        // blocks may alias or overlap ranges of the input pools, so the converted pools can outgrow n_voxels / n_palette:
        // check the running sizes before every conversion (offsets are stored in 32 bits)
        if (n + vox.size() + nvox > kMaxPoolElems || pal.size() + d.pal_len > 0xfffffff0ull) return fail(c, AIC_ERR_INVALID, "pool too large");
        const size_t vb = vox.size(), pb = pal.size();
        int rc = convert_block(c, d, s->voxels ? s->voxels + d.vox_off : nullptr, s->palette ? s->palette + 8 * (size_t)d.pal_off : nullptr,
                               (uint32_t)(n + vox.size()), (uint32_t)pal.size(), &blocks[i], &vox, &pal);
        if (rc != AIC_OK) return rc;
        vox_cap[i] = (uint32_t)(vox.size() - vb);
        pal_cap[i] = (uint32_t)(pal.size() - pb);
        vox_base[i] = (uint32_t)(n + vb);
        pal_base[i] = (uint32_t)pb;
        if ((d.flags & AIC_BLOCK_AIR) && air_index < 0) air_index = (int32_t)i;
    }
    // validate cube indices on the host copy (the reference indexes `blocks[...]` with a bounds check)
    for (size_t i = 0; i < n; i++)
        if (s->block_index[i] >= s->n_blocks) return fail(c, AIC_ERR_INVALID, "cube block index out of range");

    hipError_t e;
    std::vector<uint32_t> cls;
    for (uint32_t i = 0; i < s->n_blocks; i++) set_class(cls, i, block_class(blocks[i]));
    if (cls.empty()) cls.push_back(0u);
    if ((e = l.pool.ensure(n + vox.size())) != hipSuccess) return hip_fail(c, "alloc pool", e);
    if ((e = l.cls.ensure(cls.size())) != hipSuccess) return hip_fail(c, "alloc classes", e);
    if ((e = l.light.ensure(n)) != hipSuccess) return hip_fail(c, "alloc light", e);
    if ((e = l.blocks.ensure(blocks.size())) != hipSuccess) return hip_fail(c, "alloc blocks", e);
    if ((e = l.palette.ensure(pal.size())) != hipSuccess) return hip_fail(c, "alloc palette", e);
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(l.pool.p, s->block_index, n * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(l.light.p, s->light, n * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (!blocks.empty()) HIP_TRY(c, hipMemcpyAsync(l.blocks.p, blocks.data(), blocks.size() * sizeof(DevBlock), hipMemcpyHostToDevice, c->stream));
    if (!vox.empty()) HIP_TRY(c, hipMemcpyAsync(l.pool.p + n, vox.data(), vox.size() * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(l.cls.p, cls.data(), cls.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    const bool cls_in_code = s->n_blocks <= kCubeIndexMask + 1u;
    if (cls_in_code) launch_tag_cubes(l.pool.p, n, l.cls.p, 0, 1, c->stream);
    if (!pal.empty()) HIP_TRY(c, hipMemcpyAsync(l.palette.p, pal.data(), pal.size() * sizeof(DevPaletteEntry), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // host buffers may be released on return

    for (int a = 0; a < 3; a++) { l.lo[a] = s->lo[a]; l.size[a] = s->size[a]; }
    l.host_blocks.swap(blocks);
    l.host_cls.swap(cls);
    l.vox_cap.swap(vox_cap);
    l.pal_cap.swap(pal_cap);
    l.vox_base.swap(vox_base);
    l.pal_base.swap(pal_base);
    l.garbage_vox = l.garbage_pal = 0;
    l.air_index = air_index;
    l.sky_kind = s->sky_kind;
    for (int o = 0; o < 8; o++)
        for (int k = 0; k < 3; k++) l.sky[o][k] = s->sky[o][k] + 0.0f;  // (-0.0 becomes +0.0, as in a PositiveSign)
    for (int f = 0; f < 7; f++)
        l.block_sky[f] = (uint32_t)s->block_sky[f][0] | ((uint32_t)s->block_sky[f][1] << 8) | ((uint32_t)s->block_sky[f][2] << 16) |
                         ((uint32_t)s->block_sky[f][3] << 24);
    l.cls_in_code = cls_in_code;
    l.present = true;
    l.version++;
    l.upload_serial++;
    if (c->dump) {
        struct { int32_t lo[3], size[3]; uint32_t n_blocks; int32_t sky_kind; uint64_t n_voxels, n_palette; float sky[8][3]; uint8_t block_sky[7][4]; } h;
        std::memset(&h, 0, sizeof(h));
        for (int a = 0; a < 3; a++) { h.lo[a] = s->lo[a]; h.size[a] = s->size[a]; }
        h.n_blocks = s->n_blocks; h.sky_kind = s->sky_kind; h.n_voxels = s->n_voxels; h.n_palette = s->n_palette;
        std::memcpy(h.sky, s->sky, sizeof(h.sky));
        std::memcpy(h.block_sky, s->block_sky, sizeof(h.block_sky));
        dump_record(c, DUMP_UPLOAD, (uint32_t)layer,
                    {{&h, sizeof(h)}, {s->block_index, n * 2}, {s->light, n * 4}, {s->blocks, (size_t)s->n_blocks * sizeof(aic_block_desc)},
                     {s->voxels, (size_t)s->n_voxels * 2}, {s->palette, (size_t)s->n_palette * 32}});
    }
    return AIC_OK;
}

int aic_clear_space(aic_ctx *c, int layer) {
    if (!c || !valid_layer(layer)) return fail(c, AIC_ERR_INVALID, "aic_clear_space: bad argument");
    { int qrc = quiesce(c); if (qrc != AIC_OK) return qrc; }
    c->layers[layer].present = false;
    dump_record(c, DUMP_CLEAR, (uint32_t)layer, {});
    return AIC_OK;
}

int aic_update_cubes(aic_ctx *c, int layer, uint32_t n, const int32_t *xyz, const uint16_t *block_index, const uint8_t *light) {
    if (!c || !valid_layer(layer) || (n && !xyz)) return fail(c, AIC_ERR_INVALID, "aic_update_cubes: bad argument");
    Layer &l = c->layers[layer];
    if (!l.present) return fail(c, AIC_ERR_INVALID, "aic_update_cubes: no space uploaded for this layer");
    if (!n) return AIC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    { int qrc = quiesce(c); if (qrc != AIC_OK) return qrc; }
    if (block_index)
        for (uint32_t i = 0; i < n; i++)
            if (block_index[i] >= l.host_blocks.size()) return fail(c, AIC_ERR_INVALID, "cube block index out of range");
    const size_t b_xyz = (size_t)n * 12, b_bi = block_index ? (((size_t)n * 2 + 15) & ~(size_t)15) : 0, b_lt = light ? (size_t)n * 4 : 0;
    hipError_t e = c->staging.ensure(b_xyz + b_bi + b_lt + 64);
    if (e != hipSuccess) return hip_fail(c, "alloc staging", e);
    unsigned char *base = c->staging.p;
    HIP_TRY(c, hipMemcpyAsync(base, xyz, b_xyz, hipMemcpyHostToDevice, c->stream));
    if (block_index) HIP_TRY(c, hipMemcpyAsync(base + b_xyz, block_index, (size_t)n * 2, hipMemcpyHostToDevice, c->stream));
    if (light) HIP_TRY(c, hipMemcpyAsync(base + b_xyz + b_bi, light, b_lt, hipMemcpyHostToDevice, c->stream));
    l.version++;
    light_state_cubes_updated(l.lstate, l.version - 1, l.version, n, xyz, block_index, light);
    launch_scatter_cubes(l.pool.p, l.light.p, (const int32_t *)base, block_index ? (const uint16_t *)(base + b_xyz) : nullptr,
                         light ? (const uint32_t *)(base + b_xyz + b_bi) : nullptr, n, l.lo, l.size, l.cls_in_code ? l.cls.p : nullptr, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    {
        const uint32_t h[4] = {n, block_index ? 1u : 0u, light ? 1u : 0u, 0u};
        dump_record(c, DUMP_CUBES, (uint32_t)layer, {{h, sizeof(h)}, {xyz, (size_t)n * 12}, {block_index, (size_t)n * 2}, {light, (size_t)n * 4}});
    }
    return AIC_OK;
}

int aic_update_light_volume(aic_ctx *c, int layer, const uint8_t *light) {
    if (!c || !valid_layer(layer) || !light) return fail(c, AIC_ERR_INVALID, "aic_update_light_volume: bad argument");
    Layer &l = c->layers[layer];
    if (!l.present) return fail(c, AIC_ERR_INVALID, "aic_update_light_volume: no space uploaded for this layer");
    HIP_TRY(c, hipSetDevice(c->device));
    { const int rc = light_job_finish(c); if (rc != AIC_OK) return rc; }
    // Double-buffered: the new volume goes into the buffer no frame in flight is reading, on its own
    // stream, and becomes current for the frames submitted from now on -- a streaming loop that
    // re-lights every frame (BASELINE config 5) keeps its frames overlapped.
    const size_t n = l.n_cubes();
    int spare = 0;
    { const int rc = take_light_spare(c, l, layer, n, &spare); if (rc != AIC_OK) return rc; }
    if (n) HIP_TRY(c, hipMemcpyAsync(l.light_spare[spare].p, light, n * 4, hipMemcpyHostToDevice, c->upload_stream));
    HIP_TRY(c, hipStreamSynchronize(c->upload_stream));
    std::swap(l.light, l.light_spare[spare]);
    l.version++;
    {
        const uint64_t h = n;
        dump_record(c, DUMP_LIGHT, (uint32_t)layer, {{&h, sizeof(h)}, {light, n * 4}});
    }
    return AIC_OK;
}

namespace {

// Re-packs the voxel and palette pools of a layer on the device, dropping the ranges that replaced blocks left
// behind: the cube grid stays at offset 0, every recursive block's ranges are copied (device to device) to the
// next free offset of fresh pools, and the block table follows. Called with no frame in flight.
int compact_pools(aic_ctx *c, Layer &l) {
    l.version++;
    const size_t n = l.n_cubes();
    uint64_t need_vox = n, need_pal = 0;
    for (size_t i = 0; i < l.host_blocks.size(); i++) { need_vox += l.vox_cap[i]; need_pal += l.pal_cap[i]; }
    DevBuf<uint16_t> np;
    DevBuf<DevPaletteEntry> npal;
    hipError_t e;
    if ((e = np.ensure((size_t)need_vox)) != hipSuccess) return hip_fail(c, "compact pool", e);
    if ((e = npal.ensure((size_t)(need_pal ? need_pal : 1))) != hipSuccess) { np.release(); return hip_fail(c, "compact palette", e); }
    // the new offsets go into copies of the tables: the layer keeps pointing at its old pools until every copy has landed
    std::vector<DevBlock> blocks = l.host_blocks;
    std::vector<uint32_t> vbase = l.vox_base, pbase = l.pal_base;
    uint64_t vo = n, po = 0;
    e = n ? hipMemcpyAsync(np.p, l.pool.p, n * sizeof(uint16_t), hipMemcpyDeviceToDevice, c->stream) : hipSuccess;
    for (size_t i = 0; e == hipSuccess && i < blocks.size(); i++) {
        DevBlock &b = blocks[i];
        if (l.vox_cap[i]) {
            e = hipMemcpyAsync(np.p + vo, l.pool.p + l.vox_base[i], (size_t)l.vox_cap[i] * sizeof(uint16_t), hipMemcpyDeviceToDevice, c->stream);
            if ((b.kind & 255u) != 0u) b.vox_off = (uint32_t)vo;  // a block that is a single voxel at the moment keeps offset 0 (its reservation moves all the same)
            vbase[i] = (uint32_t)vo;
            vo += l.vox_cap[i];
        }
        if (e == hipSuccess && l.pal_cap[i]) {
            e = hipMemcpyAsync(npal.p + po, l.palette.p + l.pal_base[i], (size_t)l.pal_cap[i] * sizeof(DevPaletteEntry), hipMemcpyDeviceToDevice, c->stream);
            if ((b.kind & 255u) != 0u) b.pal_off = (uint32_t)po;
            pbase[i] = (uint32_t)po;
            po += l.pal_cap[i];
        }
    }
    if (e == hipSuccess && !blocks.empty())
        e = hipMemcpyAsync(l.blocks.p, blocks.data(), blocks.size() * sizeof(DevBlock), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        // nothing was committed: put the old block table back on the device (best effort) and drop the new pools
        if (!l.host_blocks.empty()) (void)hipMemcpy(l.blocks.p, l.host_blocks.data(), l.host_blocks.size() * sizeof(DevBlock), hipMemcpyHostToDevice);
        np.release();
        npal.release();
        return hip_fail(c, "compact pools", e);
    }
    l.pool.release();
    l.palette.release();
    l.pool = np;
    l.palette = npal;
    l.pool.n = (size_t)vo;
    l.palette.n = (size_t)po;
    l.host_blocks.swap(blocks);
    l.vox_base.swap(vbase);
    l.pal_base.swap(pbase);
    l.garbage_vox = l.garbage_pal = 0;
    return AIC_OK;
}

// One BlockEvaluation / BlockIndex change (updating.rs:128-145), issued on c->stream without synchronising.
int replace_one(aic_ctx *c, int layer, Layer &l, uint32_t index, const aic_block_desc *desc, const uint16_t *voxels, const float *palette) {
    l.version++;
    if (!desc) return fail(c, AIC_ERR_INVALID, "aic_replace_block: null descriptor");
    if (index > l.host_blocks.size() || index >= 65536) return fail(c, AIC_ERR_INVALID, "aic_replace_block: index out of range");
    const bool exists = index < l.host_blocks.size();
    std::vector<uint16_t> vox;
    std::vector<DevPaletteEntry> pal;
    DevBlock db;
    // convert first (offsets patched below), then decide where the data goes
    int rc = convert_block(c, *desc, voxels, palette, 0u, 0u, &db, &vox, &pal);
    if (rc != AIC_OK) return rc;
    // in place when the block's current ranges can hold the new data (the reference replaces the block in place);
    // otherwise append and leave the old ranges as garbage for compact_pools
    const bool fits = exists && vox.size() <= l.vox_cap[index] && pal.size() <= l.pal_cap[index];
    uint32_t vox_off, pal_off;
    hipError_t e;
    if (fits) {
        // the reserved ranges, not host_blocks[index].vox_off / pal_off: those are 0 after a replacement that had no voxels
        vox_off = l.vox_base[index];
        pal_off = l.pal_base[index];
    } else {
        vox_off = (uint32_t)l.pool.n;
        pal_off = (uint32_t)l.palette.n;
        if ((uint64_t)vox_off + vox.size() > kMaxPoolElems || (uint64_t)pal_off + pal.size() > 0xfffffff0ull)
            return fail(c, AIC_ERR_INVALID, "voxel pool too large");
        if (!vox.empty() && (e = l.pool.ensure(vox_off + vox.size(), true, c->stream)) != hipSuccess) return hip_fail(c, "grow pool", e);
        if (!pal.empty() && (e = l.palette.ensure(pal_off + pal.size(), true, c->stream)) != hipSuccess) return hip_fail(c, "grow palette", e);
        if (exists) { l.garbage_vox += l.vox_cap[index]; l.garbage_pal += l.pal_cap[index]; }
    }
    if (!vox.empty()) {
        db.vox_off = vox_off;
        HIP_TRY(c, hipMemcpy(l.pool.p + vox_off, vox.data(), vox.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (!pal.empty()) {
        db.pal_off = pal_off;
        HIP_TRY(c, hipMemcpy(l.palette.p + pal_off, pal.data(), pal.size() * sizeof(DevPaletteEntry), hipMemcpyHostToDevice));
    }
    bool class_changed = false;
    if (!exists) {
        l.host_blocks.push_back(db);
        l.vox_cap.push_back((uint32_t)vox.size());
        l.pal_cap.push_back((uint32_t)pal.size());
        l.vox_base.push_back(vox_off);
        l.pal_base.push_back(pal_off);
        if ((e = l.blocks.ensure(l.host_blocks.size(), true, c->stream)) != hipSuccess) return hip_fail(c, "grow blocks", e);
    } else {
        class_changed = block_class(l.host_blocks[index]) != block_class(db);
        l.host_blocks[index] = db;
        if (!fits) {
            l.vox_cap[index] = (uint32_t)vox.size(); l.pal_cap[index] = (uint32_t)pal.size();
            l.vox_base[index] = vox_off; l.pal_base[index] = pal_off;
        }
    }
    HIP_TRY(c, hipMemcpy(l.blocks.p + index, &db, sizeof(db), hipMemcpyHostToDevice));
    set_class(l.host_cls, index, block_class(db));
    if ((e = l.cls.ensure(l.host_cls.size(), true, c->stream)) != hipSuccess) return hip_fail(c, "grow classes", e);
    HIP_TRY(c, hipMemcpy(l.cls.p + index / 16u, &l.host_cls[index / 16u], sizeof(uint32_t), hipMemcpyHostToDevice));
    if (l.cls_in_code) {
        // the cube grid carries class bits: drop them if the table outgrew 14-bit indices, refresh
        // them if an existing block changed class (cubes already holding this index must follow)
        if (l.host_blocks.size() > kCubeIndexMask + 1u) {
            launch_tag_cubes(l.pool.p, l.n_cubes(), l.cls.p, 1, 0, c->stream);
            l.cls_in_code = false;
        } else if (class_changed) {
            launch_tag_cubes(l.pool.p, l.n_cubes(), l.cls.p, 1, 1, c->stream);
        }
        HIP_TRY(c, hipGetLastError());
    }
    if (desc->flags & AIC_BLOCK_AIR) {
        if (l.air_index < 0) l.air_index = (int32_t)index;
    } else if (l.air_index == (int32_t)index) {
        l.air_index = -1;
    }
    {
        const bool one = (desc->flags & AIC_BLOCK_ONE) != 0;
        const uint64_t nvox = one ? 0 : (uint64_t)(desc->vsize[0] > 0 ? desc->vsize[0] : 0) * (uint64_t)(desc->vsize[1] > 0 ? desc->vsize[1] : 0) * (uint64_t)(desc->vsize[2] > 0 ? desc->vsize[2] : 0);
        const uint64_t h[2] = {index, nvox};
        dump_record(c, DUMP_BLOCK, (uint32_t)layer, {{h, sizeof(h)}, {desc, sizeof(*desc)}, {voxels, (size_t)nvox * 2}, {palette, (size_t)desc->pal_len * 32}});
    }
    return AIC_OK;
}

}  // namespace

int aic_replace_blocks(aic_ctx *c, int layer, uint32_t n, const uint32_t *indices, const aic_block_desc *descs, const uint16_t *const *voxels,
                       const float *const *palettes) {
    if (!c || !valid_layer(layer) || (n && (!indices || !descs || !voxels || !palettes))) return fail(c, AIC_ERR_INVALID, "aic_replace_blocks: bad argument");
    Layer &l = c->layers[layer];
    if (!l.present) return fail(c, AIC_ERR_INVALID, "aic_replace_block: no space uploaded for this layer");
    HIP_TRY(c, hipSetDevice(c->device));
    { int qrc = quiesce(c); if (qrc != AIC_OK) return qrc; }  // once for the whole batch
    int rc = AIC_OK;
    for (uint32_t k = 0; k < n && rc == AIC_OK; k++) rc = replace_one(c, layer, l, indices[k], &descs[k], voxels[k], palettes[k]);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (rc != AIC_OK) return rc;
    // garbage left by blocks that outgrew their ranges: re-pack once it is worth it (more than a quarter of the pool,
    // and more than 256 KB), instead of growing without bound until the next full upload
    const bool big_vox = l.garbage_vox * 4 > l.pool.n && l.garbage_vox > (1u << 17);
    const bool big_pal = l.garbage_pal * 4 > l.palette.n && l.garbage_pal > (1u << 13);
    if (big_vox || big_pal) return compact_pools(c, l);
    return AIC_OK;
}

int aic_compact(aic_ctx *c, int layer) {
    if (!c || !valid_layer(layer)) return fail(c, AIC_ERR_INVALID, "aic_compact: bad argument");
    Layer &l = c->layers[layer];
    if (!l.present) return AIC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    { int qrc = quiesce(c); if (qrc != AIC_OK) return qrc; }
    return compact_pools(c, l);
}

int aic_replace_block(aic_ctx *c, int layer, uint32_t index, const aic_block_desc *desc, const uint16_t *voxels, const float *palette) {
    if (!desc) return fail(c, AIC_ERR_INVALID, "aic_replace_block: bad argument");
    return aic_replace_blocks(c, layer, 1u, &index, desc, &voxels, &palette);
}

int aic_set_options(aic_ctx *c, int layer, const aic_options *o) {
    if (!c || !valid_layer(layer) || !o) return fail(c, AIC_ERR_INVALID, "aic_set_options: bad argument");
    if (o->fog < 0 || o->fog > 3 || o->transparency < 0 || o->transparency > 2 || o->lighting < 0 || o->lighting > 5 ||
        o->antialiasing < 0 || o->antialiasing > 2 || o->tone_mapping < 0 || o->tone_mapping > 1 || !(o->view_distance > 0.0))
        return fail(c, AIC_ERR_INVALID, "aic_set_options: enum or view_distance out of range");
    c->layers[layer].opt = *o;
    c->layers[layer].opt_set = true;
    dump_record(c, DUMP_OPTIONS, (uint32_t)layer, {{o, sizeof(*o)}});
    return AIC_OK;
}

uint32_t aic_partition_rows(uint32_t height, const aic_partition *p) {
    if (!p || p->n_parts <= 1 || p->strip_rows == 0) return height;
    if (p->part >= p->n_parts) return 0;
    uint32_t rows = 0;
    const uint32_t n_strips = (height + p->strip_rows - 1) / p->strip_rows;
    for (uint32_t s = p->part; s < n_strips; s += p->n_parts) {
        uint32_t r0 = s * p->strip_rows;
        uint32_t r1 = r0 + p->strip_rows;
        if (r1 > height) r1 = height;
        rows += r1 - r0;
    }
    return rows;
}

namespace {

// Do two cameras see nearly the same picture? Compares the rays through three NDC points.
bool cameras_close(const double a[16], const double b[16]) {
    auto unproject = [](const double m[16], double x, double y, double z, double out[3]) {  // euclid row-vector convention
        const double w = x * m[3] + y * m[7] + z * m[11] + m[15];
        for (int k = 0; k < 3; k++) out[k] = (x * m[k] + y * m[4 + k] + z * m[8 + k] + m[12 + k]) / w;
    };
    static const double pts[3][2] = {{0.0, 0.0}, {1.0, 1.0}, {-1.0, -1.0}};
    for (const auto &p : pts) {
        double na[3], fa[3], nb[3], fb[3];
        unproject(a, p[0], p[1], 0.0, na); unproject(a, p[0], p[1], 1.0, fa);
        unproject(b, p[0], p[1], 0.0, nb); unproject(b, p[0], p[1], 1.0, fb);
        double da[3], db[3], la = 0, lb = 0, dot = 0, move = 0;
        for (int k = 0; k < 3; k++) {
            da[k] = fa[k] - na[k]; db[k] = fb[k] - nb[k];
            la += da[k] * da[k]; lb += db[k] * db[k]; dot += da[k] * db[k];
            move += (na[k] - nb[k]) * (na[k] - nb[k]);
        }
        if (!(la > 0) || !(lb > 0)) return false;
        if (!(dot / std::sqrt(la * lb) >= 0.99985)) return false;  // cos(1 degree)
        if (!(move <= 0.0625)) return false;
    }
    return true;
}

// Queues the frames of one launch on a slot's stream: counters reset, optional UI pre-pass, the trace. No waiting. k = 1: a plain frame. k = 2, 4, 8
// (aic_render_submit_batch): frames of the same shape, partition and flags under the scene and options as they stand, each with its own cameras, backdrop,
// output buffer, counters and cost record -- traced side by side by ONE launch per pass (DevSub), every persistent workgroup bound to one of them.
int submit_frames(aic_ctx *c, uint32_t k, const aic_frame_desc *frames, uint32_t *const *out_devices, uint32_t slot, bool allow_aux, const double *patches = nullptr,
                  uint32_t n_patches = 0, const DevOrthoView *ortho = nullptr, int32_t ortho_n = 0) {
    aic_ctx::FrameSlot &fs = c->slots[slot];
    const aic_frame_desc *f = &frames[0];
    fs.t_begin = std::chrono::steady_clock::now();
    fs.n_sub = k;
    aic_partition part = f->partition;
    if (part.n_parts <= 1 || part.strip_rows == 0) {
        part.n_parts = 1;
        part.part = 0;
        part.strip_rows = f->height ? f->height : 1;
    }
    if (part.part >= part.n_parts) return fail(c, AIC_ERR_INVALID, "aic_render: partition.part >= n_parts");
    const uint32_t local_rows = aic_partition_rows(f->height, &part);
    const size_t npix = (size_t)f->width * local_rows;
    for (uint32_t j = 0; j < k; j++)
        if (npix && !out_devices[j]) return fail(c, AIC_ERR_INVALID, "aic_render: output buffer is null");
    if (!patches)
        for (uint32_t j = 0; j < k; j++) dump_record(c, DUMP_FRAME, slot, {{&frames[j], sizeof(frames[j])}});
    if (f->width > 65535u || local_rows > 65535u) return fail(c, AIC_ERR_INVALID, "aic_render: frame dimensions above 65535 are not supported");
    for (uint32_t j = 0; j < k; j++)  // Camera::exposure() is a PositiveSign<f32> (camera_struct.rs:365-367)
        if (!(frames[j].world.exposure >= 0.f) || !(frames[j].ui.exposure >= 0.f)) return fail(c, AIC_ERR_INVALID, "aic_render: exposure is negative or NaN");

    uint32_t flaws = 0;
    DevFrame F;
    std::memset(&F, 0, sizeof(F));
    DevLayer hl[2];
    fill_dev_layer(c, c->layers[AIC_LAYER_WORLD], f->world, &hl[0], &flaws);
    fill_dev_layer(c, c->layers[AIC_LAYER_UI], f->ui, &hl[1], &flaws);
    {
        const aic_options &wo = c->layers[AIC_LAYER_WORLD].opt;
        if (wo.bloom_intensity != 0.0f) flaws |= AIC_FLAW_NO_BLOOM;  // renderer.rs:293-297
    }
    F.width = f->width;
    F.height = f->height;
    F.n_sub = k;
    // the encoder and the sampling pattern follow the WORLD camera's options (renderer.rs:283-291, 426)
    F.pixel_centers = (f->flags & AIC_FRAME_PIXEL_CENTERS) && !patches ? 1 : 0;
    F.out_mode = (f->flags & AIC_FRAME_OUT_LINEAR) ? 1 : ((f->flags & AIC_FRAME_OUT_COLORBUF) ? 2 : 0);
    F.patches = patches;
    F.n_patches = n_patches;
    F.ortho = ortho;
    F.ortho_n = ortho_n;
    if (ortho_n) {
        // render_orthographic traces with GraphicsOptions::UNALTERED_COLORS and no UI layer (ortho.rs:44, 103-131)
        hl[1].present = 0;
        DevOptions &o = hl[0].opt;
        o.fog = 0; o.transparency = 1; o.lighting = 0; o.antialiasing = 0; o.debug_pixel_cost = 0; o.tone_mapping = 0;
        o.maximum_intensity = INFINITY; o.view_distance = 200.0;
        hl[0].exposure = 1.0f;
        flaws = 0;
    }
    F.antialias = (hl[0].opt.antialiasing == 2 && !F.pixel_centers) ? 1 : 0;
    F.maximum_intensity = hl[0].opt.maximum_intensity;
    F.tone_mapping = hl[0].opt.tone_mapping;
    F.strip_rows = part.strip_rows;
    F.n_parts = part.n_parts;
    F.part = part.part;
    F.local_rows = local_rows;
    {
        // 8x8-pixel tiles: one wave-full of pixels per fetch from the tile counter. Measured against
        // 16x16 (AIC_TILE=16): -24 % frame time at 1080p, -11 % at 4K -- the coarser tiles left the
        // 2048 persistent waves with ~4 work items each and a long unbalanced tail.
        const int forced = c->sw.tile;
        F.tile = (forced == 8 || forced == 16) ? (uint32_t)forced : 8u;
        F.tiles_x = (f->width + F.tile - 1) / F.tile;
        F.tiles_y = (local_rows + F.tile - 1) / F.tile;
        const int macro_env = c->sw.macro;
        F.macro = (macro_env == 1 || macro_env == 2 || macro_env == 4 || macro_env == 8 || macro_env == 16) ? (uint32_t)macro_env : 2u;
        F.macros_x = (F.tiles_x + F.macro - 1) / F.macro;
        F.macros_y = (F.tiles_y + F.macro - 1) / F.macro;
    }
    F.light_lut = c->lut.p;
    F.n_cus = c->n_cus;
    F.tiles_per_wave = c->sw.tiles_per_wave ? c->sw.tiles_per_wave : (c->streaming_submit ? 4u : 1u);
    if (!c->sw.tiles_per_wave && c->streaming_submit && k == 1u) {
        // A streamed frame with several tiles per resident wave (a whole 1080p frame has 7.9) that is submitted while others are in flight gets a PART of the
        // chip: a third with three others queued, a quarter from four on -- up to 32 tiles per wave, which a 4K frame has on the whole chip. Full-grid launches
        // run one behind the other and only their tails overlap; part-grid launches are resident side by side, each wave refills its lanes three or four
        // times as often before its frame runs dry, and a frame's ramp and tail are paid on a part of the chip while the others are in full swing: C2 streamed
        // 0.376 -> 0.341 ms with four frames in flight, 0.338 with eight (profiles/r06_experiments.txt R). A frame submitted with nothing else in flight (the
        // first of a stream, or a host that submits slower than the device traces) still takes the whole chip; small frames keep their four tiles per wave.
        uint32_t busy_others = 0;
        for (uint32_t i = 0; i < AIC_MAX_IN_FLIGHT; i++) busy_others += (i != slot && c->slots[i].busy) ? 1u : 0u;
        const uint32_t resident_waves = (uint32_t)c->n_cus * 16u;
        const uint32_t tpw_full = (F.tiles_x * F.tiles_y + resident_waves - 1u) / resident_waves;
        if (tpw_full >= 4u && busy_others > 0u) {
            const uint32_t t = tpw_full * (busy_others < 4u ? busy_others : 4u);
            F.tiles_per_wave = t < 32u ? t : (tpw_full > 32u ? tpw_full : 32u);
        }
    }
    F.srgb_thr = c->srgb_thr.p;

    const bool want_aux = allow_aux && (f->flags & AIC_FRAME_AUX) != 0;
    const bool diag = want_aux || (f->flags & AIC_FRAME_COUNTERS) != 0;
    fs.diag = diag;
    fs.flaws = flaws;
    fs.light_used[0] = hl[0].light;
    fs.light_used[1] = hl[1].light;
    fs.local_rows = local_rows;
    fs.npix = npix;
    fs.variant = fs.tile_queues = 0;
    if (allow_aux) c->aux_records = 0;
    if (!npix) return AIC_OK;
    hipError_t e;
    if (want_aux) {
        if ((e = c->aux.ensure(npix)) != hipSuccess) return hip_fail(c, "alloc aux", e);
        F.aux = c->aux.p;
    }
    const bool ui = hl[1].present != 0;
    const uint32_t n_tiles = F.macros_x * F.macros_y;  // the feedback works on macro tiles
    // XCD-local tile queues (aic_trace.hip order_tiles_kernel): one per XCD (32 CUs each on this part), a macro tile in the queue of the
    // 2^sb_shift-macro-tile super-block it lies in. One queue (aic_frame_desc.tuning) is the single dispenser of rounds 1-3.
    // (aic_frame_desc.tuning; environment variables read per frame until round 5)
    const int queues_env = (int)((f->tuning >> AIC_TUNE_QUEUES_SHIFT) & 15u);
    const int super_env = (int)((f->tuning >> AIC_TUNE_SUPER_SHIFT) & 31u) - 1;
    uint32_t n_queues = queues_env > 0 ? (uint32_t)queues_env : (uint32_t)c->n_cus / 32u;
    if (n_queues > kMaxTileQueues) n_queues = kMaxTileQueues;
    if (n_queues < 2u || patches || ortho_n || !n_tiles) n_queues = 0u;
    const uint32_t macro_log2 = F.macro >= 16 ? 4u : (F.macro >= 8 ? 3u : (F.macro >= 4 ? 2u : (F.macro >= 2 ? 1u : 0u)));
    const uint32_t tile_log2 = F.tile >= 16 ? 4u : 3u;
    // default super-block edge: the largest power of two within an eighth of the (local) image height -- 128 pixels at 1080p, 256 at 4K, ~135 blocks
    // either way: larger blocks fetch less (s256: 9.0 GB per frame with one dispenser, 5.2 at 128 pixels, 3.6 at 512) but leave a queue with too few
    // blocks to even out a scene that is part sky (profiles/r04_experiments.txt K)
    uint32_t sb_px_log2 = 0;
    while ((2u << sb_px_log2) <= std::max(1u, local_rows / 8u)) sb_px_log2++;
    const uint32_t sb_shift = super_env >= 0 ? (uint32_t)std::min(super_env, 12) : (sb_px_log2 > macro_log2 + tile_log2 ? sb_px_log2 - macro_log2 - tile_log2 : 0u);
    // tile order for a frame from the cost the sub-frame's previous frame recorded, if that frame
    // had the same shape (else index order); the record was turned into an order, and cleared, behind that frame
    const bool use_feedback = c->sw.feedback && n_tiles && !patches && !ortho_n && !(f->flags & AIC_FRAME_NO_FEEDBACK);
    const uint32_t sig[4] = {f->width, local_rows, (part.n_parts << 16) | part.part, (part.strip_rows << 8) | (F.macro << 4) | (F.tile >> 3)};
    uint32_t order_key[6] = {sig[0], sig[1], sig[2], sig[3], n_queues, sb_shift};
    const bool ordered = use_feedback || n_queues;  // the frames take their tiles through an order (else: the plain counter, index order)
    F.n_queues = ordered ? n_queues : 0u;
    if (ordered && n_queues) {
        // no record to go by: index order inside each queue, made once per frame shape
        bool any_static = false;
        for (uint32_t j = 0; j < k; j++) {
            aic_ctx::SubSlot &sb = fs.sub[j];
            const bool same = use_feedback && sb.record_ready && std::memcmp(order_key, sb.order_key, sizeof(order_key)) == 0 && sb.tile_order.n >= n_tiles &&
                              cameras_close(frames[j].world.inverse_projection_view, sb.cost_cam);
            any_static = any_static || !same;
        }
        if (any_static && (!fs.static_ready || std::memcmp(order_key, fs.static_key, sizeof(order_key)) != 0 || fs.tile_static.n < n_tiles)) {
            fs.static_ready = false;
            if ((e = fs.tile_static.ensure(n_tiles)) != hipSuccess || (e = fs.queue_static.ensure(kMaxTileQueues + 1)) != hipSuccess) return hip_fail(c, "alloc tile queues", e);
            launch_order_tiles(nullptr, fs.tile_static.p, n_tiles, F.macros_x, sb_shift, n_queues, fs.queue_static.p, fs.stream);
            HIP_TRY(c, hipGetLastError());
            std::memcpy(fs.static_key, order_key, sizeof(order_key));
            fs.static_ready = true;
        }
    }
    for (uint32_t j = 0; j < k; j++) {
        aic_ctx::SubSlot &sb = fs.sub[j];
        DevSub &S = F.sub[j];
        std::memcpy(S.backdrop, frames[j].backdrop, sizeof(S.backdrop));
        S.has_backdrop = !(frames[j].backdrop[0] == 0.f && frames[j].backdrop[1] == 0.f && frames[j].backdrop[2] == 0.f && frames[j].backdrop[3] == 0.f);
        S.exposure = ortho_n ? 1.0f : frames[j].world.exposure + 0.0f;  // (PositiveSign: checked above; -0.0 becomes +0.0)
        S.out = out_devices[j];
        if ((e = sb.counters.ensure(1)) != hipSuccess) return hip_fail(c, "alloc frame counters", e);
        S.counters = sb.counters.p;
        if (!sb.host_counters) {  // (on a sub-frame's first frame: most contexts only ever use slot 0, and a context is cheap to make and drop)
            HIP_TRY(c, hipHostMalloc((void **)&sb.host_counters, sizeof(DevCounters), hipHostMallocDefault));
            std::memset(sb.host_counters, 0, sizeof(DevCounters));
        }
        if (!sb.counters_clean) HIP_TRY(c, hipMemsetAsync(sb.counters.p, 0, sizeof(DevCounters), fs.stream));
        sb.counters_clean = false;
        if (ordered) {
            bool same = use_feedback && sb.record_ready && std::memcmp(order_key, sb.order_key, sizeof(order_key)) == 0 && sb.tile_order.n >= n_tiles;
            if (same) {
                // the record only predicts this frame if the camera has barely moved since: the view rays
                // through the centre and two corners within a degree, the eye within a quarter cube.
                // (A stale order is worse than none.)
                same = cameras_close(frames[j].world.inverse_projection_view, sb.cost_cam);
            }
            if (same) {
                S.tile_order = sb.tile_order.p;
                S.queue_start = n_queues ? sb.queue_start.p : nullptr;
            } else if (n_queues) {
                S.tile_order = fs.tile_static.p;
                S.queue_start = fs.queue_static.p;
            }
            if (use_feedback) {
                const uint32_t *const before = sb.tile_cost.p;
                if ((e = sb.tile_cost.ensure(n_tiles)) != hipSuccess) return hip_fail(c, "alloc tile feedback", e);
                if (sb.tile_cost.p != before) sb.cost_clean_n = 0;
                if (sb.cost_clean_n < n_tiles) HIP_TRY(c, hipMemsetAsync(sb.tile_cost.p, 0, (size_t)n_tiles * sizeof(uint32_t), fs.stream));
                sb.cost_clean_n = 0;  // (this frame writes it)
                S.tile_cost = sb.tile_cost.p;
                std::memcpy(sb.cost_sig, sig, sizeof(sig));
            }
        }
        if (ui) {
            const size_t samples = F.antialias ? 4 : 1;
            if ((e = sb.acc.ensure(samples * npix)) != hipSuccess) return hip_fail(c, "alloc accumulators", e);
            S.acc_buf = sb.acc.p;
        }
    }
    if (!patches && !ortho_n) {
        // Viewport's pixel edges (viewport.rs:104-113), once per frame shape: x / width * 2 - 1 and -(y / height * 2 - 1) in the reference's own f64
        // operations (this file is built with -ffp-contract=off, like the kernels), so that the kernel reads them instead of dividing per ray
        if (fs.edges_w != f->width || fs.edges_h != f->height || !fs.edges.p) {
            std::vector<double> host((size_t)f->width + 1 + (size_t)f->height + 1);
            for (uint32_t x = 0; x <= f->width; x++) host[x] = ((double)x) / (double)f->width * 2.0 - 1.0;
            for (uint32_t y = 0; y <= f->height; y++) host[(size_t)f->width + 1 + y] = -(((double)y) / (double)f->height * 2.0 - 1.0);
            fs.edges_w = fs.edges_h = 0;
            if ((e = fs.edges.ensure(host.size())) != hipSuccess) return hip_fail(c, "alloc pixel edges", e);
            // (the slot's stream may still be reading the table of the previous shape: the copy is ordered behind it, and staged before the call returns)
            HIP_TRY(c, hipMemcpyAsync(fs.edges.p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, fs.stream));
            HIP_TRY(c, hipStreamSynchronize(fs.stream));
            fs.edges_w = f->width; fs.edges_h = f->height;
        }
        F.edge_x = fs.edges.p;
        F.edge_y = fs.edges.p + (size_t)f->width + 1;
    }
    if (!diag) {
        // The production variants exist with and without the lane exchange (aic_trace.hip). The exchanging ones keep a pixel's antialiasing sums in global
        // memory -- one region per persistent workgroup, the same for the UI pre-pass and the world pass of a frame (they follow one another on the
        // stream) -- which only a frame traced with antialiasing needs.
        uint32_t groups = 0;
        const size_t bytes = trace_ray_cold_bytes(c->n_cus, &groups);
        if (bytes) {
            // the exchange (a pool of parked rays per workgroup: more rays in flight than lanes, lanes traded between waves) pays when a wave refills its lanes
            // several times over: from 5 tiles per resident wave for a frame alone (the whole 1080p frame has 7.9 and gains 6.5 %, a rank's share at
            // N = 2 has 4 and gains nothing, at N = 4 / 8 two / one and loses 3-6 % -- profiles/r05_rank_share.txt); the UI pre-pass follows the world pass
            // -- and from 1.9 for frames that are streamed (aic_render_submit), where a share of C3 at N = 8 (4 tiles per wave) gains 9 % by it and, since the
            // scheduler round was trimmed, a share of C2 at N = 4 (2 tiles per wave) 1.8 %; at one tile per wave (N = 8) the plain variant stays 2.5 % ahead).
            // A batch's frames share the resident grid: each has 1 / k of the waves.
            const uint32_t asked = (f->tuning >> AIC_TUNE_VARIANT_SHIFT) & 3u;  // (AIC_XCHG_TILES, an environment variable read per frame, until round 5)
            const double resident_waves = (double)c->n_cus * 16.0 / (double)k;
            const double need = c->streaming_submit ? 1.9 : 5.0;
            F.exchange = asked == AIC_VARIANT_EXCHANGING ? 1u : (asked == AIC_VARIANT_PLAIN ? 0u : (((double)F.tiles_x * (double)F.tiles_y >= need * resident_waves) ? 1u : 0u));
            if (F.exchange && F.antialias) {
                if ((e = fs.ray_cold.ensure(bytes / sizeof(uint4))) != hipSuccess) return hip_fail(c, "alloc ray state", e);
                F.ray_cold = fs.ray_cold.p;
                F.ray_cold_groups = groups;
            }
        }
    }
    HIP_TRY(c, hipEventRecord(fs.ev0, fs.stream));
    if (ui) {
        F.pass = 1;
        F.use_init = 0;
        F.layer = hl[1];
        F.layer_transparency = hl[1].opt.transparency;
        F.layer_lighting = hl[1].opt.lighting;
        DevSub keep[kMaxSub];
        std::memcpy(keep, F.sub, sizeof(keep));
        const uint32_t queues_keep = F.n_queues;
        for (uint32_t j = 0; j < k; j++) {
            std::memcpy(F.sub[j].inv, frames[j].ui.inverse_projection_view, sizeof(F.sub[j].inv));
            F.sub[j].tile_order = nullptr;  // the feedback describes the world pass
            F.sub[j].tile_cost = nullptr;
            F.sub[j].queue_start = nullptr;
            F.sub[j].host_counters = nullptr;  // (the world pass hands over the sums of both)
        }
        F.n_queues = 0u;
        launch_trace_image(F, diag, fs.stream);
        std::memcpy(F.sub, keep, sizeof(keep));
        F.n_queues = queues_keep;
        for (uint32_t j = 0; j < k; j++) HIP_TRY(c, hipMemsetAsync(&fs.sub[j].counters.p->tile_next, 0, sizeof(uint32_t), fs.stream));
        F.use_init = 1;
    }
    F.pass = 0;
    for (uint32_t j = 0; j < k; j++) {
        std::memcpy(F.sub[j].inv, ortho_n ? hl[0].inv : frames[j].world.inverse_projection_view, sizeof(F.sub[j].inv));
#ifndef AIC_PROFILE
        F.sub[j].host_counters = reinterpret_cast<unsigned long long *>(fs.sub[j].host_counters);  // (DevCounters begins with the five sums)
#endif
    }
    F.layer = hl[0];
    F.layer_transparency = hl[0].opt.transparency;
    F.layer_lighting = hl[0].opt.lighting;
    // (Bounce lighting has no exchanging variant: aic_trace.hip launch_trace)
    fs.variant = diag ? AIC_VARIANT_RECORDING : ((F.exchange && hl[0].opt.lighting != 5 && (F.ray_cold || !F.antialias)) ? AIC_VARIANT_EXCHANGING : AIC_VARIANT_PLAIN);
    fs.tile_queues = F.n_queues;
    launch_trace_image(F, diag, fs.stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(fs.ev1, fs.stream));
    // behind the trace: the frame's sums are in pinned host memory -- written by the last wave of the trace itself, or (profile builds: the whole
    // counter block) copied there; ev2 is what a wait waits for ...
#ifdef AIC_PROFILE
    for (uint32_t j = 0; j < k; j++) HIP_TRY(c, hipMemcpyAsync(fs.sub[j].host_counters, fs.sub[j].counters.p, sizeof(DevCounters), hipMemcpyDeviceToHost, fs.stream));
#endif
    HIP_TRY(c, hipEventRecord(fs.ev2, fs.stream));
    fs.busy = true;  // the frame is in flight and a wait can collect it; nothing below can fail the call any more
    // ... and the slot is made ready for its next frame: this frame's cost record becomes the tile order of the next frame of the same view, the
    // record and the counters are cleared. A failure here costs the next frame its head start (it clears and orders for itself), not this frame its result.
    bool cleared = false;
    if (use_feedback && n_tiles) {
        OrderJobs jobs;
        std::memset(&jobs, 0, sizeof(jobs));
        bool ok = true;
        for (uint32_t j = 0; j < k; j++) {
            aic_ctx::SubSlot &sb = fs.sub[j];
            sb.record_ready = false;
            sb.cost_clean_n = 0;
            ok = ok && sb.tile_order.ensure(n_tiles) == hipSuccess && sb.queue_start.ensure(kMaxTileQueues + 1) == hipSuccess;
            jobs.cost[j] = sb.tile_cost.p; jobs.order[j] = sb.tile_order.p; jobs.queue_start[j] = sb.queue_start.p;
            jobs.clear_words[j] = reinterpret_cast<uint32_t *>(sb.counters.p);
        }
        if (ok) {
            // (one launch, a workgroup per frame: orders, then clears the record it has read and the counters)
            launch_order_tiles_jobs(jobs, k, n_tiles, F.macros_x, sb_shift, n_queues ? n_queues : 1u, fs.stream, true, (uint32_t)(sizeof(DevCounters) / 4));
            if (hipGetLastError() == hipSuccess) {
                cleared = true;
                for (uint32_t j = 0; j < k; j++) {
                    aic_ctx::SubSlot &sb = fs.sub[j];
                    sb.cost_clean_n = n_tiles;
                    std::memcpy(sb.order_key, order_key, sizeof(order_key));
                    std::memcpy(sb.cost_cam, frames[j].world.inverse_projection_view, sizeof(sb.cost_cam));
                    sb.record_ready = true;
                }
            }
        }
    }
    if (!cleared) {
        cleared = true;
        for (uint32_t j = 0; j < k; j++) cleared = (hipMemsetAsync(fs.sub[j].counters.p, 0, sizeof(DevCounters), fs.stream) == hipSuccess) && cleared;
    }
    for (uint32_t j = 0; j < k; j++) fs.sub[j].counters_clean = cleared;
    if (want_aux) c->aux_records = npix;
    return AIC_OK;
}
int submit_frame(aic_ctx *c, const aic_frame_desc *f, uint32_t *out_device, uint32_t slot, bool allow_aux, const double *patches = nullptr,
                 uint32_t n_patches = 0, const DevOrthoView *ortho = nullptr, int32_t ortho_n = 0) {
    uint32_t *const outs[1] = {out_device};
    return submit_frames(c, 1, f, outs, slot, allow_aux, patches, n_patches, ortho, ortho_n);
}

// Waits for a slot's frame (or batch of frames) and reports it: `info` = the frame's, or the batch's sums; `infos` (may be null) = each frame's.
int wait_frame(aic_ctx *c, uint32_t slot, aic_frame_info *info, bool whole_stream = false, aic_frame_info *infos = nullptr, uint32_t n_infos = 0) {
    aic_ctx::FrameSlot &fs = c->slots[slot];
    if (info) std::memset(info, 0, sizeof(*info));
    for (uint32_t j = 0; infos && j < n_infos; j++) std::memset(&infos[j], 0, sizeof(infos[j]));
    float kernel_ms = 0.f;
    const uint32_t k = fs.n_sub ? fs.n_sub : 1u;
    if (fs.busy) {
        fs.busy = false;  // released whatever happens below: a frame that failed must not block its slot for good
        // the frame and its counters (ev2), not the slot's housekeeping behind them -- unless the caller has enqueued a copy of its own behind the frame
        if (whole_stream || c->sw.wait_whole_stream) HIP_TRY(c, hipStreamSynchronize(fs.stream));  // (the switch: a measurement)
        else HIP_TRY(c, hipEventSynchronize(fs.ev2));
        HIP_TRY(c, hipEventElapsedTime(&kernel_ms, fs.ev0, fs.ev1));
        for (uint32_t j = 0; j < k; j++) {
            DevCounters hc;
            std::memcpy(&hc, fs.sub[j].host_counters, sizeof(hc));
            if (hc.bailed) return fail(c, AIC_ERR_DEVICE, "trace kernel: a wave gave up waiting for rays in transit between waves; the frame has unwritten pixels");
            if (info) {
                info->cubes_traced += hc.cubes_traced;
                info->n_outer += hc.n_outer;
                info->n_inner += hc.n_inner;
                info->n_hits += hc.n_hits;
                info->n_light += hc.n_light;
            }
            if (infos && j < n_infos) {
                infos[j].cubes_traced = hc.cubes_traced; infos[j].n_outer = hc.n_outer; infos[j].n_inner = hc.n_inner;
                infos[j].n_hits = hc.n_hits; infos[j].n_light = hc.n_light;
            }
#ifdef AIC_PROFILE
            { static const char *names[40] = {"max_lifetime","max_until_dry","cyc_until_dry","cyc_lifetime","shade_ph","shade_ln","enter_ph","enter_ln","ray_ph","ray_ln","step_iters","step_lanes","cyc_step","cyc_shade_rest","cyc_enter","cyc_newray","cyc_finish","cyc_refill","cyc_shade_light","cyc_sched","fast_iters","fast_lanes","trips","trip_lanes","pass_hl_lanes","pass_fast_eligible","leave_blocks","leave_lanes","apply_blocks","apply_lanes","pass_needed_lanes","xchg_rounds","xchg_lanes","cyc_xchg","xchg_picked","xchg_parked","idle_spins","xchg_claims_lost","xchg_empty","-"};
              // (only the production variant's frames: the aux-recording variant is another kernel, at half the occupancy)
              if (!fs.diag) for (int i = 0; i < 39; i++) std::fprintf(stderr, "PROF %s %llu\n", names[i], hc.prof[i]);
              if (const char *path = (fs.diag || c->sw.wave_prof.empty()) ? nullptr : c->sw.wave_prof.c_str()) {
                  if (FILE *fp = std::fopen(path, "w")) {
                      for (int w = 0; w < 2048; w++) std::fprintf(fp, "%u %u %u %u\n", hc.wave_prof[w][0], hc.wave_prof[w][1], hc.wave_prof[w][2], hc.wave_prof[w][3]);
                      std::fclose(fp);
                  }
              } }
#endif
        }
    }
    const float total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - fs.t_begin).count();
    auto fill = [&](aic_frame_info *o) {
        o->kernel_ms = kernel_ms;
        o->rows_rendered = fs.local_rows;
        o->flaws = fs.flaws;
        o->variant = fs.npix ? fs.variant : 0u;
        o->tile_queues = fs.npix ? fs.tile_queues : 0u;
        o->total_ms = total_ms;
    };
    if (info) fill(info);
    for (uint32_t j = 0; infos && j < n_infos && j < k; j++) fill(&infos[j]);
    return AIC_OK;
}

}  // namespace

int aic_render(aic_ctx *c, const aic_frame_desc *f, void *out_rgba8, int out_is_device, aic_frame_info *info) {
    if (!c || !f) return fail(c, AIC_ERR_INVALID, "aic_render: bad argument");
    if (info) std::memset(info, 0, sizeof(*info));
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->slots[0].busy) return fail(c, AIC_ERR_INVALID, "aic_render: a submitted frame still occupies slot 0 (aic_render_wait it first)");
    uint32_t *target = (uint32_t *)out_rgba8;
    const size_t px_words = (f->flags & (AIC_FRAME_OUT_LINEAR | AIC_FRAME_OUT_COLORBUF)) ? 4 : 1;  // 16 or 4 bytes per pixel
    if (!out_is_device && out_rgba8) {
        aic_partition part = f->partition;
        if (part.n_parts <= 1 || part.strip_rows == 0) { part.n_parts = 1; part.part = 0; part.strip_rows = f->height ? f->height : 1; }
        const size_t npix = (size_t)f->width * (part.part < part.n_parts ? aic_partition_rows(f->height, &part) : 0);
        hipError_t e;
        if (npix && (e = c->out.ensure(npix * px_words)) != hipSuccess) return hip_fail(c, "alloc output", e);
        target = c->out.p;
    }
    int rc = submit_frame(c, f, target, 0, true);
    if (rc != AIC_OK) return rc;
    if (!out_is_device && c->slots[0].npix)
        HIP_TRY(c, hipMemcpyAsync(out_rgba8, c->out.p, c->slots[0].npix * px_words * 4, hipMemcpyDeviceToHost, c->slots[0].stream));
    return wait_frame(c, 0, info, !out_is_device);
}

namespace {

// raytracer::ortho::OrthoCamera::new / MultiOrthoCamera::new (ortho.rs:147-184, 213-282), on the host: five pixel-perfect
// views (top, left, front, right, bottom) laid out around the front view. Matrices are euclid's row-vector 4x4.
struct Mat4h { double m[16]; };
Mat4h mat_identity() { Mat4h t{}; t.m[0] = t.m[5] = t.m[10] = t.m[15] = 1.0; return t; }
Mat4h mat_then(const Mat4h &a, const Mat4h &b) {  // Transform3D::then
    Mat4h o{};
    for (int r = 0; r < 4; r++)
        for (int cc = 0; cc < 4; cc++)
            o.m[r * 4 + cc] = a.m[r * 4 + 0] * b.m[0 * 4 + cc] + a.m[r * 4 + 1] * b.m[1 * 4 + cc] + a.m[r * 4 + 2] * b.m[2 * 4 + cc] + a.m[r * 4 + 3] * b.m[3 * 4 + cc];
    return o;
}
void ortho_view(int resolution, const int32_t lo[3], const int32_t size[3], int viewed_face /* 1 NX .. 6 PZ */, DevOrthoView *out) {
    const int axis = (viewed_face - 1) % 3;
    const int32_t ub[3] = {lo[0] + size[0], lo[1] + size[1], lo[2] + size[2]};
    out->w = (uint32_t)(axis == 0 ? size[2] : size[0]) * (uint32_t)resolution;
    out->h = (uint32_t)(axis == 1 ? size[2] : size[1]) * (uint32_t)resolution;
    double ot[3];
    // images of +X, +Y, +Z under the view's grid rotation (Face::clockwise / counterclockwise, face.rs:440-465)
    int bx[3], by[3], bz[3];
    auto set = [](int v[3], int x, int y, int z) { v[0] = x; v[1] = y; v[2] = z; };
    switch (viewed_face) {
        case 1: ot[0] = lo[0]; ot[1] = ub[1]; ot[2] = lo[2]; set(bx, 0, 0, 1); set(by, 0, 1, 0); set(bz, -1, 0, 0); break;   // NX: PY.clockwise()
        case 2: ot[0] = lo[0]; ot[1] = lo[1]; ot[2] = ub[2]; set(bx, 1, 0, 0); set(by, 0, 0, 1); set(bz, 0, -1, 0); break;   // NY: PX.clockwise()
        case 3: ot[0] = ub[0]; ot[1] = ub[1]; ot[2] = lo[2]; set(bx, -1, 0, 0); set(by, 0, 1, 0); set(bz, 0, 0, -1); break;  // NZ: 180 degrees about Y
        case 4: ot[0] = ub[0]; ot[1] = ub[1]; ot[2] = ub[2]; set(bx, 0, 0, -1); set(by, 0, 1, 0); set(bz, 1, 0, 0); break;   // PX: PY.counterclockwise()
        case 5: ot[0] = lo[0]; ot[1] = ub[1]; ot[2] = lo[2]; set(bx, 1, 0, 0); set(by, 0, 0, -1); set(bz, 0, 1, 0); break;   // PY: PX.counterclockwise()
        default: ot[0] = lo[0]; ot[1] = ub[1]; ot[2] = ub[2]; set(bx, 1, 0, 0); set(by, 0, 1, 0); set(bz, 0, 0, 1); break;   // PZ: identity
    }
    // translation(0.5, 0.5, 0).then_scale(1, -1, 1).then(scale 1/resolution).then(rotation).then_translate(origin)
    Mat4h t = mat_identity();
    t.m[12] = 0.5; t.m[13] = 0.5;
    Mat4h flip = mat_identity();
    flip.m[5] = -1.0;
    t = mat_then(t, flip);
    Mat4h sc = mat_identity();
    sc.m[0] = sc.m[5] = sc.m[10] = 1.0 / (double)resolution;
    t = mat_then(t, sc);
    Mat4h rot{};
    for (int a = 0; a < 3; a++) { rot.m[0 + a] = bx[a]; rot.m[4 + a] = by[a]; rot.m[8 + a] = bz[a]; }
    rot.m[15] = 1.0;
    t = mat_then(t, rot);
    Mat4h tr = mat_identity();
    tr.m[12] = ot[0]; tr.m[13] = ot[1]; tr.m[14] = ot[2];
    t = mat_then(t, tr);
    std::memcpy(out->m, t.m, sizeof(t.m));
    // transform_vector3d((0, 0, -1)) reduced to its axis (TryFrom<Ray> for AaRay keeps only the direction's axis and sign)
    for (int a = 0; a < 3; a++) { const double d = -t.m[8 + a]; out->dir[a] = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0); }
}
void multi_ortho(int resolution, const int32_t lo[3], const int32_t size[3], DevOrthoView v[5], uint32_t *w, uint32_t *h) {
    ortho_view(resolution, lo, size, 5, &v[0]);  // top
    ortho_view(resolution, lo, size, 1, &v[1]);  // left
    ortho_view(resolution, lo, size, 6, &v[2]);  // front
    ortho_view(resolution, lo, size, 4, &v[3]);  // right
    ortho_view(resolution, lo, size, 2, &v[4]);  // bottom
    v[0].x0 = v[1].w + 1; v[0].y0 = 0;
    v[1].x0 = 0; v[1].y0 = v[0].h + 1;
    v[2].x0 = v[1].w + 1; v[2].y0 = v[0].h + 1;
    v[3].x0 = v[1].w + v[2].w + 2; v[3].y0 = v[0].h + 1;
    v[4].x0 = v[1].w + 1; v[4].y0 = v[0].h + v[2].h + 2;
    *w = *h = 0;
    for (int i = 0; i < 5; i++) {
        if (v[i].x0 + v[i].w > *w) *w = v[i].x0 + v[i].w;
        if (v[i].y0 + v[i].h > *h) *h = v[i].y0 + v[i].h;
    }
}
bool valid_ortho_resolution(int r) { return r >= 1 && r <= 128 && (r & (r - 1)) == 0; }

}  // namespace

int aic_ortho_image_size(const int32_t lo[3], const int32_t size[3], int resolution, uint32_t *width, uint32_t *height) {
    if (!lo || !size || !width || !height || !valid_ortho_resolution(resolution) || size[0] < 0 || size[1] < 0 || size[2] < 0) return AIC_ERR_INVALID;
    DevOrthoView v[5];
    multi_ortho(resolution, lo, size, v, width, height);
    return AIC_OK;
}

int aic_render_orthographic(aic_ctx *c, int layer, int resolution, void *out_rgba8, int out_is_device, uint32_t *width, uint32_t *height,
                            aic_frame_info *info) {
    if (!c || !valid_layer(layer) || !valid_ortho_resolution(resolution)) return fail(c, AIC_ERR_INVALID, "aic_render_orthographic: bad argument");
    if (info) std::memset(info, 0, sizeof(*info));
    Layer &l = c->layers[layer];
    if (!l.present) return fail(c, AIC_ERR_INVALID, "aic_render_orthographic: no space uploaded for this layer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->slots[0].busy) return fail(c, AIC_ERR_INVALID, "aic_render_orthographic: a submitted frame still occupies slot 0");
    DevOrthoView v[5];
    uint32_t w = 0, h = 0;
    multi_ortho(resolution, l.lo, l.size, v, &w, &h);
    if (width) *width = w;
    if (height) *height = h;
    if (!out_rgba8 || !w || !h) return AIC_OK;  // size query
    hipError_t e;
    if ((e = c->ortho_views.ensure(5)) != hipSuccess) return hip_fail(c, "alloc ortho views", e);
    HIP_TRY(c, hipMemcpy(c->ortho_views.p, v, sizeof(v), hipMemcpyHostToDevice));
    aic_frame_desc f;
    std::memset(&f, 0, sizeof(f));
    f.width = w;
    f.height = h;
    static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(f.world.inverse_projection_view, ident, sizeof(ident));
    std::memcpy(f.ui.inverse_projection_view, ident, sizeof(ident));
    f.world.exposure = f.ui.exposure = 1.0f;
    uint32_t *target = (uint32_t *)out_rgba8;
    const size_t npix = (size_t)w * h;
    if (!out_is_device) {
        if ((e = c->out.ensure(npix)) != hipSuccess) return hip_fail(c, "alloc output", e);
        target = c->out.p;
    }
    // the views are traced as the WORLD layer of the launch, whatever layer holds the space
    const bool swap_layers = layer != AIC_LAYER_WORLD;
    if (swap_layers) std::swap(c->layers[AIC_LAYER_WORLD], c->layers[layer]);
    int rc = submit_frame(c, &f, target, 0, false, nullptr, 0, c->ortho_views.p, 5);
    if (swap_layers) std::swap(c->layers[AIC_LAYER_WORLD], c->layers[layer]);
    if (rc != AIC_OK) return rc;
    if (!out_is_device) HIP_TRY(c, hipMemcpyAsync(out_rgba8, c->out.p, npix * 4, hipMemcpyDeviceToHost, c->slots[0].stream));
    return wait_frame(c, 0, info, !out_is_device);
}

int aic_trace_patches(aic_ctx *c, const aic_frame_desc *f, uint32_t n, const double *rects, void *out_rgba8, aic_pixel_aux *aux,
                      aic_frame_info *info) {
    if (!c || !f || (n && (!rects || !out_rgba8))) return fail(c, AIC_ERR_INVALID, "aic_trace_patches: bad argument");
    if (info) std::memset(info, 0, sizeof(*info));
    if (!n) return AIC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->slots[0].busy) return fail(c, AIC_ERR_INVALID, "aic_trace_patches: a submitted frame still occupies slot 0 (aic_render_wait it first)");
    // the batch is laid out as an image of up to 2048 columns; pixel i of that image traces rects[i]
    aic_frame_desc g = *f;
    g.width = n < 2048u ? n : 2048u;
    g.height = (n + g.width - 1u) / g.width;
    if (g.height > 65535u) return fail(c, AIC_ERR_INVALID, "aic_trace_patches: more than 2048 x 65535 rectangles in one call");
    g.partition = aic_partition{0, 1, 0, 0};
    g.flags = (f->flags & (AIC_FRAME_COUNTERS | AIC_FRAME_OUT_LINEAR | AIC_FRAME_OUT_COLORBUF)) | (aux ? AIC_FRAME_AUX : 0u);
    const size_t npix = (size_t)g.width * g.height;
    hipError_t e;
    const size_t px_words = (f->flags & (AIC_FRAME_OUT_LINEAR | AIC_FRAME_OUT_COLORBUF)) ? 4 : 1;
    if ((e = c->out.ensure(npix * px_words)) != hipSuccess) return hip_fail(c, "alloc output", e);
    if ((e = c->staging.ensure((size_t)n * 32)) != hipSuccess) return hip_fail(c, "alloc staging", e);
    HIP_TRY(c, hipMemcpyAsync(c->staging.p, rects, (size_t)n * 32, hipMemcpyHostToDevice, c->slots[0].stream));
    int rc = submit_frame(c, &g, c->out.p, 0, true, (const double *)c->staging.p, n);
    if (rc != AIC_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_rgba8, c->out.p, (size_t)n * px_words * 4, hipMemcpyDeviceToHost, c->slots[0].stream));
    rc = wait_frame(c, 0, info, true);
    if (rc != AIC_OK) return rc;
    if (aux) {
        HIP_TRY(c, hipMemcpy(aux, c->aux.p, (size_t)n * sizeof(aic_pixel_aux), hipMemcpyDeviceToHost));
    }
    if (info) info->rows_rendered = n;
    return AIC_OK;
}

// a frame slot's stream, events and counter block (slots past the eighth: on first use)
static int ensure_slot(aic_ctx *c, uint32_t slot) {
    aic_ctx::FrameSlot &fs = c->slots[slot];
    if (fs.stream && fs.ev0 && fs.ev1 && fs.ev2 && fs.sub[0].counters.p) return AIC_OK;
    if (!fs.stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&fs.stream, hipStreamNonBlocking));
        // "Everything the context queues afterwards waits for the event" (aic_wait_event) must hold for a stream made later too: the context's first
        // stream received every such wait, so the new stream is ordered behind where that one stands now (a first frame on slot >= 8
        // after an aic_wait_event raced the gather still reading its strip buffer)
        hipEvent_t ev = nullptr;
        HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e1 = hipEventRecord(ev, c->stream);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(fs.stream, ev, 0);
        (void)hipEventDestroy(ev);  // (released once the recorded work is done)
        if (e1 != hipSuccess) return hip_fail(c, "order a new slot's stream behind the context's waits", e1);
    }
    if (!fs.ev0) HIP_TRY(c, hipEventCreate(&fs.ev0));
    if (!fs.ev1) HIP_TRY(c, hipEventCreate(&fs.ev1));
    if (!fs.ev2) HIP_TRY(c, hipEventCreate(&fs.ev2));
    hipError_t e = fs.sub[0].counters.ensure(1);
    if (e != hipSuccess) return hip_fail(c, "alloc frame counters", e);
    return AIC_OK;
}

int aic_render_submit(aic_ctx *c, const aic_frame_desc *f, void *out_device, uint32_t slot) {
    if (!c || !f || slot >= AIC_MAX_IN_FLIGHT) return fail(c, AIC_ERR_INVALID, "aic_render_submit: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->slots[slot].busy) return fail(c, AIC_ERR_INVALID, "aic_render_submit: slot busy (aic_render_wait it first)");
    { const int rs = ensure_slot(c, slot); if (rs != AIC_OK) return rs; }
    c->streaming_submit = true;
    const int rc = submit_frame(c, f, (uint32_t *)out_device, slot, false);
    c->streaming_submit = false;
    return rc;
}

int aic_render_submit_batch(aic_ctx *c, uint32_t n_frames, const aic_frame_desc *frames, void *const *out_devices, uint32_t slot) {
    if (!c || !frames || !out_devices || slot >= AIC_MAX_IN_FLIGHT) return fail(c, AIC_ERR_INVALID, "aic_render_submit_batch: bad argument");
    if (n_frames != 1u && n_frames != 2u && n_frames != 4u && n_frames != 8u) return fail(c, AIC_ERR_INVALID, "aic_render_submit_batch: 1, 2, 4 or 8 frames per launch");
    for (uint32_t j = 1; j < n_frames; j++) {
        const aic_frame_desc &a = frames[0], &b = frames[j];
        if (a.width != b.width || a.height != b.height || a.flags != b.flags || a.tuning != b.tuning || std::memcmp(&a.partition, &b.partition, sizeof(a.partition)) != 0)
            return fail(c, AIC_ERR_INVALID, "aic_render_submit_batch: the frames of a batch share size, partition, flags and tuning");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->slots[slot].busy) return fail(c, AIC_ERR_INVALID, "aic_render_submit_batch: slot busy (aic_render_wait it first)");
    { const int rs = ensure_slot(c, slot); if (rs != AIC_OK) return rs; }
    uint32_t *outs[kMaxSub];
    for (uint32_t j = 0; j < n_frames; j++) outs[j] = (uint32_t *)out_devices[j];
    c->streaming_submit = true;
    const int rc = submit_frames(c, n_frames, frames, outs, slot, false);
    c->streaming_submit = false;
    return rc;
}

int aic_render_wait(aic_ctx *c, uint32_t slot, aic_frame_info *info) {
    if (!c || slot >= AIC_MAX_IN_FLIGHT) return fail(c, AIC_ERR_INVALID, "aic_render_wait: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    return wait_frame(c, slot, info);
}

int aic_render_wait_batch(aic_ctx *c, uint32_t slot, uint32_t n_frames, aic_frame_info *infos) {
    if (!c || slot >= AIC_MAX_IN_FLIGHT || (n_frames && !infos)) return fail(c, AIC_ERR_INVALID, "aic_render_wait_batch: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    return wait_frame(c, slot, nullptr, false, infos, n_frames);
}

int aic_assemble_strips_on(aic_ctx *c, const void *gathered_device, void *out_device, uint32_t width, uint32_t height,
                           uint32_t strip_rows, uint32_t n_parts, void *hip_stream) {
    if (!c || !gathered_device || !out_device || !strip_rows || !n_parts) return fail(c, AIC_ERR_INVALID, "aic_assemble_strips: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t max_rows = 0;
    for (uint32_t p = 0; p < n_parts; p++) {
        aic_partition pp = {strip_rows, n_parts, p, 0};
        uint32_t r = aic_partition_rows(height, &pp);
        if (r > max_rows) max_rows = r;
    }
    launch_assemble_strips((const uint32_t *)gathered_device, (uint32_t *)out_device, width, height, strip_rows, n_parts, max_rows, hip_stream ? (hipStream_t)hip_stream : c->stream);
    HIP_TRY(c, hipGetLastError());
    return AIC_OK;
}

int aic_assemble_strips_async(aic_ctx *c, const void *gathered_device, void *out_device, uint32_t width, uint32_t height,
                              uint32_t strip_rows, uint32_t n_parts) {
    return aic_assemble_strips_on(c, gathered_device, out_device, width, height, strip_rows, n_parts, nullptr);
}

int aic_assemble_strips(aic_ctx *c, const void *gathered_device, void *out_device, uint32_t width, uint32_t height,
                        uint32_t strip_rows, uint32_t n_parts) {
    const int rc = aic_assemble_strips_async(c, gathered_device, out_device, width, height, strip_rows, n_parts);
    if (rc != AIC_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return AIC_OK;
}

int aic_read_aux(aic_ctx *c, aic_pixel_aux *out, uint64_t n_records) {
    if (!c || !out) return fail(c, AIC_ERR_INVALID, "aic_read_aux: bad argument");
    if (n_records > c->aux_records) return fail(c, AIC_ERR_INVALID, "aic_read_aux: more records requested than the last frame produced");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(out, c->aux.p, n_records * sizeof(aic_pixel_aux), hipMemcpyDeviceToHost));
    return AIC_OK;
}

int aic_synchronize(aic_ctx *c) {
    if (!c) return AIC_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 1; i < AIC_MAX_IN_FLIGHT; i++)
        if (c->slots[i].stream) HIP_TRY(c, hipStreamSynchronize(c->slots[i].stream));
    return AIC_OK;
}

void *aic_stream(aic_ctx *c) { return c ? (void *)c->stream : nullptr; }

int aic_wait_event(aic_ctx *c, void *hip_event) {
    if (!c || !hip_event) return fail(c, AIC_ERR_INVALID, "aic_wait_event: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, (hipEvent_t)hip_event, 0));
    for (uint32_t i = 1; i < AIC_MAX_IN_FLIGHT; i++)
        if (c->slots[i].stream) HIP_TRY(c, hipStreamWaitEvent(c->slots[i].stream, (hipEvent_t)hip_event, 0));
    return AIC_OK;
}

int aic_stream_wait_frame(aic_ctx *c, uint32_t slot, void *hip_stream) {
    if (!c || slot >= AIC_MAX_IN_FLIGHT) return fail(c, AIC_ERR_INVALID, "aic_stream_wait_frame: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    aic_ctx::FrameSlot &fs = c->slots[slot];
    if (!fs.busy || !fs.npix) return AIC_OK;  // nothing in flight (or an empty partition: no event was recorded)
    HIP_TRY(c, hipStreamWaitEvent((hipStream_t)hip_stream, fs.ev1, 0));
    return AIC_OK;
}

int aic_probe_raycast(aic_ctx *c, const double origin[3], const double direction[3], int use_bounds, const int32_t lo[3],
                      const int32_t hi[3], int include_exit, uint32_t max_steps, aic_rc_step *out, uint32_t *n_out, int *ended) {
    if (!c || !origin || !direction || !out || !n_out || !ended) return fail(c, AIC_ERR_INVALID, "aic_probe_raycast: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = 6 * sizeof(double) + 8 * sizeof(int) + (size_t)max_steps * 8 * sizeof(double) + 64;
    hipError_t e = c->staging.ensure(bytes);
    if (e != hipSuccess) return hip_fail(c, "alloc staging", e);
    unsigned char *base = c->staging.p;
    double od[6] = {origin[0], origin[1], origin[2], direction[0], direction[1], direction[2]};
    int lohi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (use_bounds && lo && hi) {
        for (int a = 0; a < 3; a++) { lohi[a] = lo[a]; lohi[3 + a] = hi[a]; }
    }
    double *d_od = (double *)base;
    int *d_lohi = (int *)(base + 48);
    uint32_t *d_n = (uint32_t *)(base + 48 + 24);
    int *d_end = (int *)(base + 48 + 28);
    double *d_rec = (double *)(base + 48 + 32);
    HIP_TRY(c, hipMemcpyAsync(d_od, od, sizeof(od), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_lohi, lohi, 24, hipMemcpyHostToDevice, c->stream));
    launch_probe_raycast(d_od, use_bounds, d_lohi, include_exit, max_steps, d_rec, d_n, d_end, c->stream);
    HIP_TRY(c, hipGetLastError());
    std::vector<double> rec((size_t)max_steps * 8);
    uint32_t n = 0;
    int en = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&en, d_end, 4, hipMemcpyDeviceToHost, c->stream));
    if (max_steps) HIP_TRY(c, hipMemcpyAsync(rec.data(), d_rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n && i < max_steps; i++) {
        const double *r = rec.data() + 8 * (size_t)i;
        out[i].cube[0] = (int32_t)r[0]; out[i].cube[1] = (int32_t)r[1]; out[i].cube[2] = (int32_t)r[2];
        out[i].face = (int32_t)r[3];
        out[i].t_distance = r[4];
        out[i].intersection_point[0] = r[5]; out[i].intersection_point[1] = r[6]; out[i].intersection_point[2] = r[7];
    }
    *n_out = n;
    *ended = en;
    return AIC_OK;
}

int aic_probe_powf(aic_ctx *c, const float *x, const float *y, uint32_t n, float *out) {
    if (!c || (n && (!x || !y || !out))) return fail(c, AIC_ERR_INVALID, "aic_probe_powf: bad argument");
    if (!n) return AIC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipError_t e = c->staging.ensure((size_t)n * 12);
    if (e != hipSuccess) return hip_fail(c, "alloc staging", e);
    float *dx = (float *)c->staging.p, *dy = dx + n, *dout = dy + n;
    HIP_TRY(c, hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dy, y, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    launch_probe_powf(dx, dy, dout, n, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return AIC_OK;
}

int aic_probe_light_lut(aic_ctx *c, float out[256]) {
    if (!c || !out) return AIC_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(out, c->lut.p, 256 * sizeof(float), hipMemcpyDeviceToHost));
    return AIC_OK;
}

}  // extern "C"

#include "aic_light_host.inc"

// aic_multi.cpp -- one process, several MI355X: the multi-device context of the C ABI.
//
// `north_star` asks for the 8-GPU path BEHIND HeadlessRenderer: a Rust shim holds one object, not eight processes. An
// aic_multi owns one aic_ctx per device (device ids may repeat: two contexts on one GPU are how the path is tested on a
// single-GPU box), replicates every scene call on all of them (the scene is tiny next to 288 GB: SURVEY.md 8e), and
// renders a frame as the reference's row loop would be split (renderer.rs:537-555 treats rows as independent work
// items): the image is cut into 8-row strips dealt round-robin to the devices, every device traces its strips into a
// compact local buffer (aic_render_submit on its own stream -- the traces run concurrently), the compact buffers are
// copied to device 0 over xGMI (hipMemcpyPeerAsync: each peer uses its direct link to device 0, the same exchange the
// multi-process path does with an RCCL gather), and aic_assemble_strips de-interleaves them into the frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/aic_hip.h"

namespace {
constexpr uint32_t kStripRows = 8;  // (the kernel's work tile: 135 strips of a 1080-row frame deal evenly, 67.5 sixteen-row strips did not -- profiles/r06_experiments.txt T)
}

// A frame in flight (aic_multi_render_submit): its buffers and what aic_multi_render_wait needs to finish it
struct MultiSlot {
    std::vector<void *> local;       // per device: compact strips [local_rows][width] RGBA8
    std::vector<size_t> local_bytes;
    void *gathered = nullptr;        // on device 0: [n][max_rows][width]
    size_t gathered_bytes = 0;
    void *frame = nullptr;           // on device 0: the assembled frame when the caller wants a host copy
    size_t frame_bytes = 0;
    hipEvent_t done = nullptr;       // on device 0's transfer stream, behind the frame's assemble
    bool busy = false;
    size_t submitted = 0;            // devices whose share was submitted (they are waited for whatever happens)
    void *host_out = nullptr;        // where the finished frame goes when the caller asked for a host copy
    size_t frame_size = 0;           // bytes of the assembled frame
};

struct aic_multi {
    std::vector<aic_ctx *> ctx;
    std::vector<int> dev;
    MultiSlot slots[AIC_MULTI_MAX_IN_FLIGHT];
    hipStream_t xfer = nullptr;      // device 0: the peer copies and the de-interleave of every frame, in submission order, beside the traces
    size_t n_cubes[2] = {0, 0};      // per layer: cubes of the uploaded space (sizes the light volume hand-over)
    bool light_stale[2] = {false, false};  // device 0's light volume was changed (aic_multi_light_cubes_changed) and not yet handed to the others
    std::string err;
};

namespace {

int mfail(aic_multi *m, int code, const std::string &msg) {
    if (m) m->err = msg;
    return code;
}
int forward(aic_multi *m, size_t i, int rc) {
    if (rc != AIC_OK) m->err = "device " + std::to_string(m->dev[i]) + ": " + aic_last_error(m->ctx[i]);
    return rc;
}
int ensure(aic_multi *m, int device, void **p, size_t *have, size_t need) {
    if (need <= *have) return AIC_OK;
    if (hipSetDevice(device) != hipSuccess) return mfail(m, AIC_ERR_DEVICE, "hipSetDevice failed");
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    if (hipMalloc(p, need + need / 4) != hipSuccess) return mfail(m, AIC_ERR_DEVICE, "hipMalloc failed");
    *have = need + need / 4;
    return AIC_OK;
}

}  // namespace

extern "C" {

aic_multi *aic_create_multi(int n_devices, const int *device_ids, int *status) {
    int dummy;
    if (!status) status = &dummy;
    if (n_devices < 1 || n_devices > 64 || !device_ids) { *status = AIC_ERR_INVALID; return nullptr; }
    aic_multi *m = new aic_multi();
    for (int i = 0; i < n_devices; i++) {
        int st = AIC_OK;
        aic_ctx *c = aic_create(device_ids[i], &st);
        if (!c) {
            *status = st;
            aic_destroy_multi(m);
            return nullptr;
        }
        m->ctx.push_back(c);
        m->dev.push_back(device_ids[i]);
    }
    for (MultiSlot &sl : m->slots) {
        sl.local.assign((size_t)n_devices, nullptr);
        sl.local_bytes.assign((size_t)n_devices, 0);
    }
    if (hipSetDevice(device_ids[0]) != hipSuccess || hipStreamCreateWithFlags(&m->xfer, hipStreamNonBlocking) != hipSuccess) {
        *status = AIC_ERR_DEVICE;
        aic_destroy_multi(m);
        return nullptr;
    }
    // let device 0 be written by its peers directly (a failure only means the copies are staged by the runtime)
    for (int i = 1; i < n_devices; i++) {
        if (device_ids[i] == device_ids[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]) == hipSuccess && can && hipSetDevice(device_ids[i]) == hipSuccess)
            (void)hipDeviceEnablePeerAccess(device_ids[0], 0);
        (void)hipGetLastError();
    }
    *status = AIC_OK;
    return m;
}

void aic_destroy_multi(aic_multi *m) {
    if (!m) return;
    for (uint32_t s = 0; s < AIC_MULTI_MAX_IN_FLIGHT; s++)
        if (m->slots[s].busy) {
            m->slots[s].host_out = nullptr;  // (a frame nobody collected: its host target may be gone -- the devices are waited for, nothing is copied out)
            (void)aic_multi_render_wait(m, s, nullptr);
        }
    if (!m->dev.empty() && m->xfer) { (void)hipSetDevice(m->dev[0]); (void)hipStreamSynchronize(m->xfer); }
    for (MultiSlot &sl : m->slots) {
        for (size_t i = 0; i < m->ctx.size() && i < sl.local.size(); i++)
            if (sl.local[i]) { (void)hipSetDevice(m->dev[i]); (void)hipFree(sl.local[i]); }
        if (!m->dev.empty()) {
            (void)hipSetDevice(m->dev[0]);
            if (sl.gathered) (void)hipFree(sl.gathered);
            if (sl.frame) (void)hipFree(sl.frame);
            if (sl.done) (void)hipEventDestroy(sl.done);
        }
    }
    for (size_t i = 0; i < m->ctx.size(); i++) aic_destroy(m->ctx[i]);
    if (!m->dev.empty() && m->xfer) { (void)hipSetDevice(m->dev[0]); (void)hipStreamDestroy(m->xfer); }
    delete m;
}

int aic_multi_device_count(const aic_multi *m) { return m ? (int)m->ctx.size() : 0; }
aic_ctx *aic_multi_context(aic_multi *m, int i) { return (m && i >= 0 && (size_t)i < m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }
const char *aic_multi_last_error(const aic_multi *m) { return m ? m->err.c_str() : "no context"; }

// scene calls: replicated on every device
#define AIC_MULTI_FORWARD(call)                                            \
    if (!m) return AIC_ERR_INVALID;                                        \
    for (size_t i = 0; i < m->ctx.size(); i++) {                           \
        const int rc = forward(m, i, call);                                \
        if (rc != AIC_OK) return rc;                                       \
    }                                                                      \
    return AIC_OK;

int aic_multi_upload_space(aic_multi *m, int layer, const aic_space_desc *s) {
    if (!m) return AIC_ERR_INVALID;
    if (layer == 0 || layer == 1) m->n_cubes[layer] = 0;  // known only once every device has accepted the space
    for (size_t i = 0; i < m->ctx.size(); i++) {
        const int rc = forward(m, i, aic_upload_space(m->ctx[i], layer, s));
        if (rc != AIC_OK) return rc;
    }
    if (s && (layer == 0 || layer == 1)) m->n_cubes[layer] = (size_t)s->size[0] * (size_t)s->size[1] * (size_t)s->size[2];  // validated by aic_upload_space
    return AIC_OK;
}
int aic_multi_clear_space(aic_multi *m, int layer) {
    if (m && (layer == 0 || layer == 1)) m->n_cubes[layer] = 0;
    AIC_MULTI_FORWARD(aic_clear_space(m->ctx[i], layer))
}
int aic_multi_update_cubes(aic_multi *m, int layer, uint32_t n, const int32_t *xyz, const uint16_t *bi, const uint8_t *light) {
    AIC_MULTI_FORWARD(aic_update_cubes(m->ctx[i], layer, n, xyz, bi, light))
}
int aic_multi_update_light_volume(aic_multi *m, int layer, const uint8_t *light) { AIC_MULTI_FORWARD(aic_update_light_volume(m->ctx[i], layer, light)) }
int aic_multi_replace_blocks(aic_multi *m, int layer, uint32_t n, const uint32_t *indices, const aic_block_desc *descs, const uint16_t *const *voxels,
                             const float *const *palettes) {
    AIC_MULTI_FORWARD(aic_replace_blocks(m->ctx[i], layer, n, indices, descs, voxels, palettes))
}
int aic_multi_set_options(aic_multi *m, int layer, const aic_options *o) { AIC_MULTI_FORWARD(aic_set_options(m->ctx[i], layer, o)) }

// The light updater is a sequential relaxation (it does not shard): it runs on the first device, and the resulting volume
// is handed to the others, so that every device traces the same light.
namespace {
// device 0's light volume of a layer to every other device
int broadcast_light(aic_multi *m, int layer) {
    m->light_stale[layer] = false;
    if (m->ctx.size() == 1) return AIC_OK;
    std::vector<uint8_t> light(m->n_cubes[layer] * 4);
    if (light.empty()) return AIC_OK;
    int rc = forward(m, 0, aic_read_light_volume(m->ctx[0], layer, light.data()));
    for (size_t i = 1; rc == AIC_OK && i < m->ctx.size(); i++) rc = forward(m, i, aic_update_light_volume(m->ctx[i], layer, light.data()));
    if (rc != AIC_OK) m->light_stale[layer] = true;
    return rc;
}
}  // namespace

int aic_multi_evaluate_light(aic_multi *m, int layer, const aic_light_params *p, aic_light_info *info) {
    if (!m || m->ctx.empty() || (layer != 0 && layer != 1)) return mfail(m, AIC_ERR_INVALID, "aic_multi_evaluate_light: bad argument");
    const int rc = forward(m, 0, aic_evaluate_light(m->ctx[0], layer, p, info));
    if (rc != AIC_OK) return rc;
    return broadcast_light(m, layer);
}

// aic_light_cubes_changed on the device that runs the light updater. Its queue is device 0's alone; the texels it writes there
// (PackedLight::OPAQUE at the cubes that are now opaque, nothing else) are read back from device 0's host mirror and scattered
// into the other devices' volumes at once -- n texels, not the volume -- so every device always traces the same
// light. Should that hand-over fail, the layer is marked and aic_multi_render hands the whole volume over before it traces.
int aic_multi_light_cubes_changed(aic_multi *m, int layer, uint32_t n, const int32_t *xyz, int queue_order) {
    if (!m || m->ctx.empty() || (layer != 0 && layer != 1)) return mfail(m, AIC_ERR_INVALID, "aic_multi_light_cubes_changed: bad argument");
    int rc = forward(m, 0, aic_light_cubes_changed(m->ctx[0], layer, n, xyz, queue_order));
    if (rc != AIC_OK || !n || m->ctx.size() == 1) return rc;
    if (m->light_stale[layer]) return AIC_OK;  // the whole volume is owed already
    m->light_stale[layer] = true;
    std::vector<uint8_t> texels((size_t)n * 4);
    rc = forward(m, 0, aic_read_light_cubes(m->ctx[0], layer, n, xyz, texels.data()));
    for (size_t i = 1; rc == AIC_OK && i < m->ctx.size(); i++) rc = forward(m, i, aic_update_cubes(m->ctx[i], layer, n, xyz, nullptr, texels.data()));
    if (rc == AIC_OK) m->light_stale[layer] = false;
    else m->err.clear();  // (the call succeeds: the failed hand-over must not be left behind as the multi context's last error -- ADVICE r05)
    return AIC_OK;  // device 0 has taken the change; what the others lack is owed (light_stale) and paid by aic_multi_render
}

// A frame in flight: every device traces its strips on ITS slot `slot` (aic_render_submit: the submits return at once, the traces overlap); device 0's
// transfer stream waits for each share ON THE DEVICE (aic_stream_wait_frame), copies it over the peer's direct link and de-interleaves -- the host only enqueues.
// With several slots the next frames' traces run under this frame's copies, which is what a caller that renders frame after frame
// (record.rs:97-113) gets from a single context's aic_render_submit / aic_render_wait.
int aic_multi_render_submit(aic_multi *m, const aic_frame_desc *f, void *out_rgba8, int out_is_device, uint32_t slot) {
    if (!m || !f || !out_rgba8 || slot >= AIC_MULTI_MAX_IN_FLIGHT) return mfail(m, AIC_ERR_INVALID, "aic_multi_render_submit: bad argument");
    if (f->flags & (AIC_FRAME_AUX | AIC_FRAME_OUT_LINEAR | AIC_FRAME_OUT_COLORBUF))
        return mfail(m, AIC_ERR_INVALID, "aic_multi_render: RGBA8 frames only (use a single context for aux records / float output)");
    if (f->partition.n_parts > 1) return mfail(m, AIC_ERR_INVALID, "aic_multi_render partitions the frame itself");
    MultiSlot &sl = m->slots[slot];
    if (sl.busy) return mfail(m, AIC_ERR_INVALID, "aic_multi_render_submit: slot busy (aic_multi_render_wait it first)");
    const size_t n = m->ctx.size();
    const uint32_t w = f->width, h = f->height;
    for (int layer = 0; layer < 2; layer++)
        if (m->light_stale[layer]) {
            const int rc = broadcast_light(m, layer);
            if (rc != AIC_OK) return rc;
        }
    std::vector<uint32_t> rows(n);
    uint32_t max_rows = 0;
    for (size_t i = 0; i < n; i++) {
        const aic_partition p{kStripRows, (uint32_t)n, (uint32_t)i, 0};
        rows[i] = aic_partition_rows(h, &p);
        if (rows[i] > max_rows) max_rows = rows[i];
    }
    const size_t row_bytes = (size_t)w * 4;
    sl.submitted = 0;
    sl.host_out = nullptr;
    sl.frame_size = (size_t)h * row_bytes;
    if (!w || !h) { sl.busy = true; return AIC_OK; }  // (an empty frame: nothing to trace, the wait reports zeros)
    for (size_t i = 0; i < n; i++) {
        const int rc = ensure(m, m->dev[i], &sl.local[i], &sl.local_bytes[i], (size_t)(rows[i] ? rows[i] : 1) * row_bytes);
        if (rc != AIC_OK) return rc;
    }
    { const int rc = ensure(m, m->dev[0], &sl.gathered, &sl.gathered_bytes, n * (size_t)max_rows * row_bytes); if (rc != AIC_OK) return rc; }
    void *target = out_rgba8;
    if (!out_is_device) {
        const int rc = ensure(m, m->dev[0], &sl.frame, &sl.frame_bytes, sl.frame_size);
        if (rc != AIC_OK) return rc;
        target = sl.frame;
    }
    if (hipSetDevice(m->dev[0]) != hipSuccess) return mfail(m, AIC_ERR_DEVICE, "hipSetDevice failed");
    if (!sl.done && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) return mfail(m, AIC_ERR_DEVICE, "hipEventCreate failed");
    // A failure from here on must not leave the devices' slots busy (every later frame would fail with "slot busy"): whatever was
    // submitted is waited for before the error is returned.
    auto drain = [&](int rc, const std::string &msg) {
        for (size_t k = 0; k < sl.submitted; k++) (void)aic_render_wait(m->ctx[k], slot, nullptr);
        sl.submitted = 0;
        if (!msg.empty()) m->err = msg;
        return rc;
    };
    // 1. every device traces its strips
    for (size_t i = 0; i < n; i++) {
        aic_frame_desc fi = *f;
        fi.partition = n > 1 ? aic_partition{kStripRows, (uint32_t)n, (uint32_t)i, 0} : aic_partition{0, 1, 0, 0};
        const int rc = forward(m, i, aic_render_submit(m->ctx[i], &fi, n > 1 ? sl.local[i] : target, slot));
        if (rc != AIC_OK) return drain(rc, m->err);
        sl.submitted = i + 1;
    }
    // 2. behind each share, on device 0's transfer stream: its compact strips go to device 0 (peer copy over the direct link) ...
    for (size_t i = 0; i < n; i++) {
        const int rc = forward(m, i, aic_stream_wait_frame(m->ctx[i], slot, m->xfer));
        if (rc != AIC_OK) return drain(rc, m->err);
        if (n == 1 || !rows[i]) continue;
        char *dst = (char *)sl.gathered + i * (size_t)max_rows * row_bytes;
        if (hipSetDevice(m->dev[0]) != hipSuccess) return drain(AIC_ERR_DEVICE, "hipSetDevice failed");
        const hipError_t e = m->dev[i] == m->dev[0]
                                 ? hipMemcpyAsync(dst, sl.local[i], (size_t)rows[i] * row_bytes, hipMemcpyDeviceToDevice, m->xfer)
                                 : hipMemcpyPeerAsync(dst, m->dev[0], sl.local[i], m->dev[i], (size_t)rows[i] * row_bytes, m->xfer);
        if (e != hipSuccess) return drain(AIC_ERR_DEVICE, std::string("peer copy: ") + hipGetErrorString(e));
    }
    // 3. ... and are de-interleaved there
    if (n > 1) {
        const int rc = forward(m, 0, aic_assemble_strips_on(m->ctx[0], sl.gathered, target, w, h, kStripRows, (uint32_t)n, m->xfer));
        if (rc != AIC_OK) return drain(rc, m->err);
    }
    if (hipSetDevice(m->dev[0]) != hipSuccess || hipEventRecord(sl.done, m->xfer) != hipSuccess) return drain(AIC_ERR_DEVICE, "hipEventRecord failed");
    if (!out_is_device) sl.host_out = out_rgba8;
    sl.busy = true;
    return AIC_OK;
}

int aic_multi_render_wait(aic_multi *m, uint32_t slot, aic_frame_info *info) {
    if (!m || slot >= AIC_MULTI_MAX_IN_FLIGHT) return mfail(m, AIC_ERR_INVALID, "aic_multi_render_wait: bad argument");
    if (info) std::memset(info, 0, sizeof(*info));
    MultiSlot &sl = m->slots[slot];
    if (!sl.busy) return AIC_OK;
    sl.busy = false;
    int result = AIC_OK;
    for (size_t i = 0; i < sl.submitted; i++) {
        aic_frame_info fi;
        const int rc = forward(m, i, aic_render_wait(m->ctx[i], slot, &fi));  // (every share is waited for, whatever another returned)
        if (rc != AIC_OK) { if (result == AIC_OK) result = rc; continue; }
        if (info) {
            info->cubes_traced += fi.cubes_traced; info->n_outer += fi.n_outer; info->n_inner += fi.n_inner;
            info->n_hits += fi.n_hits; info->n_light += fi.n_light; info->flaws |= fi.flaws;
            info->rows_rendered += fi.rows_rendered;
            if (fi.kernel_ms > info->kernel_ms) info->kernel_ms = fi.kernel_ms;
            if (fi.total_ms > info->total_ms) info->total_ms = fi.total_ms;
            if (i == 0) { info->variant = fi.variant; info->tile_queues = fi.tile_queues; }  // (every device runs the same variant on a share of the same shape, bar the last strip)
        }
    }
    const size_t submitted = sl.submitted;
    sl.submitted = 0;
    if (!submitted) return result;  // an empty frame
    if (hipSetDevice(m->dev[0]) != hipSuccess || hipEventSynchronize(sl.done) != hipSuccess) return result != AIC_OK ? result : mfail(m, AIC_ERR_DEVICE, "waiting for the assembled frame failed");
    if (result == AIC_OK && sl.host_out) {
        const void *src = m->ctx.size() > 1 || sl.frame ? sl.frame : nullptr;
        if (!src || hipMemcpy(sl.host_out, src, sl.frame_size, hipMemcpyDeviceToHost) != hipSuccess) return mfail(m, AIC_ERR_DEVICE, "read-back failed");
    }
    sl.host_out = nullptr;
    return result;
}

// One frame, start to finish (= submit + wait on slot 0).
int aic_multi_render(aic_multi *m, const aic_frame_desc *f, void *out_rgba8, int out_is_device, aic_frame_info *info) {
    if (info) std::memset(info, 0, sizeof(*info));
    const int rc = aic_multi_render_submit(m, f, out_rgba8, out_is_device, 0);
    if (rc != AIC_OK) return rc;
    return aic_multi_render_wait(m, 0, info);
}

}  // extern "C"

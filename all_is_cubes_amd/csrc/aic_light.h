// aic_light.h -- device-side data layout of the light updater, shared by the host code
// (aic_light_host.inc) and the gather kernel (aic_light.hip).
//
// The reference's light updater (all-is-cubes/src/space/light/updater.rs) pops cubes off a priority queue,
// computes each cube's new light by walking a precomputed tree of ray bundles ("chart") through the space,
// and applies the result, re-queueing the cubes the result depended on. With feature "auto-threads" it computes
// batches of queue entries against one light state (updater.rs:231-268). Here the queue and the apply step stay on
// the host (they are sequential by definition), and the batch's compute_light calls -- all of the arithmetic --
// run on the device, one lane per cube, against the light volume that the trace kernel reads.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aic {

// chart/generator.rs FlatNode: the weight of the bundle per face and the index of the child bundle per step
// direction (0 = none; the root is node 0 and never a child). Order nx ny nz px py pz.
struct DevLightNode {
    float weight[6];
    uint32_t child[6];
};
static_assert(sizeof(DevLightNode) == 48, "chart node is 48 bytes");

// block/eval/derived.rs: what the light updater reads of an evaluated block.
struct DevDerived {
    float color[4];
    float face[6][4];   // nx ny nz px py pz
    float emission[3];
    uint32_t flags;     // bits 0-5: opaque per face; bit 6: visible
};
static_assert(sizeof(DevDerived) == 128, "derived record is 128 bytes");
static constexpr uint32_t kDerivedVisible = 1u << 6;

static constexpr uint32_t kLightFrameWords = 8;    // stack frame of the tree walk, dwords
static constexpr uint32_t kLightDepChunk = 64;     // dependency list chunk: word 0 = next chunk (or ~0), then 63 cube indices

// The effective ray-bundle tree for one maximum_distance, in pre-order (children in face order): what the wave-per-cube
// kernel walks. `end` = one past the last position of the node's subtree; nodes beyond the distance are leaves.
struct DevTreePos {
    float weight[6];
    uint32_t end;
    uint32_t parent;
    uint32_t offset;  // cube offset from the origin cube: (dx + 256) | (dy + 256) << 10 | (dz + 256) << 20
    uint32_t info;    // bits 0-2: face entered (0..5; 7 = Face7::Within, the root); bit 3: beyond maximum_distance; bits 8-23: depth
};
static_assert(sizeof(DevTreePos) == 40, "tree position record is 40 bytes");

struct LightJob {
    const uint16_t *grid;      // cube grid (DevLayer.pool)
    uint32_t index_mask;       // strips the class bits of the cube-grid entries (aic_device.h)
    const uint32_t *light;     // PackedLight texels
    const DevDerived *derived; // per block index
    uint32_t n_blocks;
    const DevLightNode *chart;
    const float *lut;          // PackedLight scalar -> f32 (256 entries)
    int32_t lo[3], size[3];
    uint32_t block_sky[7];     // texels nx ny nz px py pz mean (sky.rs:83-147)
    double max_dist_sq;        // LightPhysics::Rays { maximum_distance } squared
    const uint32_t *cubes;     // the batch: linear cube indices
    uint32_t n;
    uint32_t *out;             // [n][4]: texel, dependency count, first dependency chunk, cost
    uint32_t *dep_pool;        // chunk pool
    uint32_t dep_chunks;       // capacity in chunks
    uint32_t *dep_head;        // [0] next free chunk, [1] set when the pool ran out (the host grows it and reruns the batch)
    // When `out` and `dep_pool` are pinned host memory (small batches: no copy back), the counters stay on the device for the
    // allocator's atomics and the last block to finish copies them to `host_head` ([8], pinned); `done_count`: blocks finished.
    uint32_t *host_head;
    uint32_t *done_count;
    // Session mode (compute_light_session_kernel: one launch serves every small batch of an aic_evaluate_light call; see LightMailbox).
    struct LightMailbox *mailbox;  // pinned host memory; null outside a session
    uint32_t *session_word;        // device: [0] the sequence number workgroup 0 has published, [1] that batch's cube count (0 = leave)
    uint32_t *session_cubes;       // device: the batch's cubes, written by workgroup 0 (the same array `cubes` points to)
    uint32_t *light_rw;            // the light volume, writable: workgroup 0 scatters the texels the previous batch changed
    uint32_t session_seq;          // the first sequence number this launch waits for
    uint32_t coherent_light;       // light texels are read with agent-scope loads (another XCD's L2 may hold an older line)
    uint32_t *stack;           // [max_depth][kLightFrameWords][stack_stride]
    uint32_t stack_stride;     // lanes the stack was sized for (>= n)
    uint32_t max_depth;
    // wave-per-cube kernel only
    const DevTreePos *tree;
    uint32_t n_tree;
    const float *child_w;      // [n_tree][6 faces][6]: the weights of each position's children (zeros where there is none)
    // What the walk's expansion pass reads. A child entry is {position | (which of its six weights are > 0) << 22 | (info & 15) << 28,
    // cube offset}; position 0 = no child.
    const uint2 *child_ent;    // [n_tree][6 faces]: the child stepped to through that face
    const uint4 *node;         // [n_tree]: {its only child's entry (two words; unused unless it has exactly one), number of children, 0}
    uint32_t root_meta;        // the root's (info & 15) | weight mask << 4
    float4 *terms;             // [waves][4 * n_tree]: (incoming r, g, b; ray weight) by recursion-order number
    uint32_t *cands;           // [waves][2 * n_tree]: dependency candidates (cube offset | conditional << 30)
    // The walk's work queue: the root and the children of branching bundles, {tree position, alpha it is entered with, cube
    // offset, (info & 15) | weight mask << 4 | kLightQueueValid}. All zero between cubes: a nonzero last word publishes an entry.
    uint4 *front;              // [waves][n_front]
    uint32_t n_front;          // 1 + the number of positions whose parent has more than one child
    float *valpha;             // [waves][n_tree]: the alpha a visited bundle is entered with
};

void launch_compute_light(const LightJob &job, hipStream_t stream);
void launch_compute_light_waves(const LightJob &job, uint32_t n_blocks, uint32_t threads_per_cube, hipStream_t stream);
void launch_scatter_light(uint32_t *light, const uint32_t *index, const uint32_t *texel, uint32_t n, hipStream_t stream);

// What a small batch needs done on the device before its launch, passed in the kernel argument itself (no copies): the
// texels the previous batch changed, this batch's cubes, and the cleared counters (head[0..7] and the word behind them,
// which is LightJob::done_count when that is in use).
static constexpr uint32_t kLightQueueValid = 0x80000000u;
static constexpr uint32_t kLightPrepMax = 64;
// The hand-off page of a session, in pinned host memory. The host fills in a batch and then stores `seq` (release); workgroup 0
// polls `seq`, scatters the texels, copies the cubes to device memory and publishes the sequence number to the other
// workgroups through LightJob::session_word. The last workgroup out of a batch copies the counters to LightJob::host_head,
// clears them for the next batch, and stores `done` (system-scope release): results, dependency chunks and counters are in
// host memory by then. `n_cubes` 0 ends the session. A session whose host goes quiet for kLightSessionSpins polls ends
// itself and says so in `exited` (the sequence number it was waiting for): nothing the host does or fails to do can leave
// a kernel spinning for good.
struct LightMailbox {
    uint32_t seq, n_cubes, n_scatter, pad0;
    uint32_t cubes[64], scatter_index[64], scatter_texel[64];
    uint32_t pad1[12];
    // device to host (their own cache lines)
    uint32_t done, exited, error, pad2;
    uint64_t t_seen, t_done;  // wall_clock64() when workgroup 0 saw the batch / when the last workgroup finished it
    uint32_t pad3[8];
};
static_assert(sizeof(LightMailbox) % 64 == 0, "mailbox is whole cache lines");
static constexpr uint32_t kLightSessionSpins = 400000u;  // polls of ~1.5 us: about half a second of silence
void launch_compute_light_session(const LightJob &job, uint32_t n_blocks, uint32_t threads_per_cube, hipStream_t stream);

struct LightPrep {
    uint32_t *light, *cubes_out, *head;
    uint32_t n_scatter, n_cubes;
    uint32_t scatter_index[kLightPrepMax], scatter_texel[kLightPrepMax], cubes[kLightPrepMax];
};
void launch_prepare_light_batch(const LightPrep &prep, hipStream_t stream);
void launch_probe_log2f(const float *x, float *out, uint32_t n, hipStream_t stream);

}  // namespace aic

// aic_light.hip -- the light updater's gather step on the device: Space::compute_light
// (all-is-cubes/src/space/light/updater.rs:368-417) for a batch of cubes: compute_light_kernel, one lane per cube (the plain
// restatement), and compute_light_wave_kernel, one wave per cube (the production mapping; same bytes).
//
// Every f32 operation is done in the reference's order (built with -ffp-contract=off), so a texel computed here is
// the texel the reference computes: the walk of the ray-bundle tree is a depth-first recursion there
// (walk_ray_tree, updater.rs:427-530) and an explicit stack here, visiting the bundles in the same order and
// adding into the same accumulators. PackedLight::scalar_in's log2 (data.rs:214-218) is the correctly-rounded-table
// log2f of glibc/musl/Rust's std on Linux, restated below and pinned to the host's libm by a test.
//
// Bound: this is a pointer-chasing tree walk (48-byte chart nodes, 2-byte cube lookups, 128-byte block records),
// latency- and issue-bound like the trace kernel, not a bandwidth kernel: the chart (5.5 MB) and the scene's
// working set live in L2. Lanes of a wave walk different subtrees; the stack is laid out [level][word][lane] so
// that lanes at the same depth touch neighbouring addresses.

#include "aic_light.h"

namespace aic {
namespace {

// ---- PositiveSign / ZeroOne arithmetic (math/restricted_number.rs:240-326) ----
__device__ __forceinline__ float ps_new_clamped(float v) { return v > 0.f ? v : 0.f; }
__device__ __forceinline__ float ps_mul(float a, float b) {
    const float v = a * b;
    return (v != v) ? 0.f : v;
}

// ---- log2f: the table-driven routine of ARM optimized-routines as shipped in glibc >= 2.27 and musl
// (what f32::log2 calls on Linux): 16-entry table of 1/c and log2(c), degree-4 polynomial, double arithmetic.
__device__ const double kLog2Tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};

__device__ float log2f_exact(float x) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
        ix = __float_as_uint(x * 0x1p23f);  // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)tmp >> 23;
    const double invc = kLog2Tab[i][0], logc = kLog2Tab[i][1];
    const double z = (double)__uint_as_float(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = -0x1.712b6f70a7e4dp-2 * r2 + (0x1.ecabf496832ep-2 * r + -0x1.715479ffae3dep-1);
    const double p = 0x1.715475f35c8b8p0 * r + y0;
    y = y * r2 + p;
    return (float)y;
}

// data.rs:214-218 PackedLight::scalar_in: (log2(v) * 10 + 144).round() as u8 (saturating, NaN -> 0)
__device__ __forceinline__ uint32_t packed_scalar_in(float value) {
    const float x = roundf(log2f_exact(value) * 10.0f + 144.0f);
    if (x != x) return 0u;
    if (x <= 0.f) return 0u;
    if (x >= 255.f) return 255u;
    return (uint32_t)x;
}

__device__ __forceinline__ void normal_of(int f, int n[3]) {  // f: 0 nx 1 ny 2 nz 3 px 4 py 5 pz
    n[0] = n[1] = n[2] = 0;
    n[f >= 3 ? f - 3 : f] = f >= 3 ? 1 : -1;
}

struct Ctx {
    const LightJob &J;
    float incoming[3];
    float total_ray_weight;
    uint32_t cost;
    // dependency list
    int last_dep[3];
    bool has_dep;
    uint32_t n_deps, first_chunk, cur_chunk, fill;
    float sky_value[6][3];

    __device__ explicit Ctx(const LightJob &j) : J(j) {}

    __device__ bool index_of(const int c[3], uint32_t *out) const {  // vol.rs:988-1023
        const uint32_t dx = (uint32_t)c[0] - (uint32_t)J.lo[0], dy = (uint32_t)c[1] - (uint32_t)J.lo[1], dz = (uint32_t)c[2] - (uint32_t)J.lo[2];
        if ((dx >= (uint32_t)J.size[0]) | (dy >= (uint32_t)J.size[1]) | (dz >= (uint32_t)J.size[2])) return false;
        *out = (dx * (uint32_t)J.size[1] + dy) * (uint32_t)J.size[2] + dz;
        return true;
    }
    // sky.rs:113-147 BlockSky::light_outside
    __device__ uint32_t light_outside(const int c[3]) const {
        int n_less = 0, n_equal = 0, which = -1;
        for (int a = 0; a < 3; a++) {
            int lower;
            if (J.lo[a] == (int32_t)0x80000000) lower = -1;
            else {
                const int beyond = J.lo[a] - 1;
                lower = beyond < c[a] ? -1 : (beyond == c[a] ? 0 : 1);
            }
            if (lower == -1) n_less++;
            else if (lower == 0) { n_equal++; which = a; }
        }
        for (int a = 0; a < 3; a++) {
            const long long hi = (long long)J.lo[a] + J.size[a];
            const int upper = (long long)c[a] < hi ? -1 : ((long long)c[a] == hi ? 0 : 1);
            if (upper == -1) n_less++;
            else if (upper == 0) { n_equal++; which = 3 + a; }
        }
        if (n_less == 5 && n_equal == 1) return J.block_sky[which];
        if (n_less == 6) return 0u;      // PackedLight::UNINITIALIZED_AND_BLACK
        return 1u << 24;                 // PackedLight::NO_RAYS
    }
    __device__ uint32_t get_light(const int c[3]) const {  // updater.rs:572-582
        uint32_t i;
        if (index_of(c, &i)) return J.light[i];
        return light_outside(c);
    }
    __device__ void value_of(uint32_t texel, float v[3]) const {  // data.rs:137-143
        v[0] = J.lut[texel & 255u];
        v[1] = J.lut[(texel >> 8) & 255u];
        v[2] = J.lut[(texel >> 16) & 255u];
    }
    __device__ void add_weighted_light(const float color[3], float weight) {  // updater.rs:934-937
        const float w = ps_new_clamped(weight);
        for (int i = 0; i < 3; i++) incoming[i] += ps_mul(color[i], w);
        total_ray_weight += weight;
    }
    // updater.rs:899-927
    __device__ void end_of_ray(float alpha, float ray_bundle_weight, const float w[6]) {
        if (!(ray_bundle_weight > 0.f)) return;
        const float recip = ps_new_clamped(1.0f / ((w[0] + w[3]) + (w[1] + w[4]) + (w[2] + w[5])));
        float sky_light[3];
        for (int i = 0; i < 3; i++) {
            float pf[6];
            for (int f = 0; f < 6; f++) pf[f] = ps_mul(sky_value[f][i], ps_new_clamped(w[f]));
            const float s = (pf[0] + pf[3]) + (pf[1] + pf[4]) + (pf[2] + pf[5]);
            sky_light[i] = ps_mul(ps_mul(s, recip), ps_new_clamped(alpha));
        }
        add_weighted_light(sky_light, ray_bundle_weight);
    }
    __device__ void emit_dep(const int c[3]) {
        last_dep[0] = c[0]; last_dep[1] = c[1]; last_dep[2] = c[2];
        has_dep = true;
        uint32_t idx;
        if (!index_of(c, &idx)) return;  // light_needs_update ignores cubes outside the space (updater.rs:97-113)
        if (fill == kLightDepChunk) {
            const uint32_t next = atomicAdd(&J.dep_head[0], 1u);
            if (next >= J.dep_chunks) {
                J.dep_head[1] = 1u;
                cur_chunk = 0xffffffffu;
            } else {
                J.dep_pool[(size_t)next * kLightDepChunk] = 0xffffffffu;
                if (cur_chunk != 0xffffffffu) J.dep_pool[(size_t)cur_chunk * kLightDepChunk] = next;
                else if (n_deps == 0) first_chunk = next;
                cur_chunk = next;
            }
            fill = 1;
        }
        if (cur_chunk != 0xffffffffu) J.dep_pool[(size_t)cur_chunk * kLightDepChunk + fill] = idx;
        fill++;
        n_deps++;
    }
};

// updater.rs:770-895 LightBuffer::traverse. fe: face entered, 0..5, or -1 = Face7::Within.
__device__ void traverse(Ctx &b, float &alpha, uint32_t &dw_mask, const int hit_cube[3], int fe, const DevDerived *ev, bool &ahead_has,
                         uint32_t &ahead, bool behind_has, uint32_t behind, const float w[6]) {
    const uint32_t flags = ev->flags;
    if (!(flags & kDerivedVisible)) return;
    const bool hit_opaque_face = fe < 0 ? (flags & 63u) == 63u : ((flags >> fe) & 1u) != 0u;
    if (hit_opaque_face && fe < 0) {
        dw_mask = 0u;
        alpha = 0.f;
        return;
    }
    const float *scp = fe < 0 ? ev->color : ev->face[fe];
    float sc[4] = {scp[0], scp[1], scp[2], scp[3]};
    for (int i = 0; i < 3; i++) sc[i] = sc[i] > 1.f ? 1.f : sc[i];  // Rgba::clamp (color.rs:692-697)
    const float hit_alpha = sc[3];
    float wp[6];
    for (int f = 0; f < 6; f++) wp[f] = ((dw_mask >> f) & 1u) ? w[f] : 0.f;
    const float wsum = (wp[0] + wp[3]) + (wp[1] + wp[4]) + (wp[2] + wp[5]);
    const float em[3] = {ev->emission[0], ev->emission[1], ev->emission[2]};
    if (hit_alpha > 0.f && fe >= 0) {
        int n[3];
        normal_of(fe, n);
        const int light_cube[3] = {hit_cube[0] + n[0], hit_cube[1] + n[1], hit_cube[2] + n[2]};
        const uint32_t stored = behind_has ? behind : b.get_light(light_cube);
        float sv[3];
        b.value_of(stored, sv);
        const float a = ps_new_clamped(alpha), ww = ps_new_clamped(wsum);
        for (int i = 0; i < 3; i++) {
            const float from_face = em[i] + ps_mul(ps_mul(sc[i], sv[i]), sc[3]);
            b.incoming[i] += ps_mul(ps_mul(from_face, a), ww);
        }
        b.cost += 10u;
        if (!b.has_dep || !(b.last_dep[0] == light_cube[0] && b.last_dep[1] == light_cube[1] && b.last_dep[2] == light_cube[2])) b.emit_dep(light_cube);
        if (hit_opaque_face) alpha = 0.f;
        else alpha *= 1.0f - hit_alpha;
    }
    if (hit_alpha < 1.0f) {
        float sl[3] = {0.f, 0.f, 0.f};
        if (fe >= 0) {
            if (!ahead_has) { ahead = b.get_light(hit_cube); ahead_has = true; }
            b.value_of(ahead, sl);
        }
        const float a = ps_new_clamped(alpha), ww = ps_new_clamped(wsum);
        for (int i = 0; i < 3; i++) {
            const float from_block = em[i] + ps_mul(sl[i], hit_alpha);
            b.incoming[i] += ps_mul(ps_mul(from_block, a), ww);
        }
        b.cost += 10u;
        b.emit_dep(hit_cube);
        alpha *= 1.0f - hit_alpha;
    }
}

__global__ void __launch_bounds__(64) compute_light_kernel(const LightJob J) {
    const uint32_t tid = blockIdx.x * 64u + threadIdx.x;
    if (tid >= J.n) return;
    Ctx b(J);
    for (int i = 0; i < 3; i++) b.incoming[i] = 0.f;
    b.total_ray_weight = 0.f;
    b.cost = 0u;
    b.has_dep = false;
    b.last_dep[0] = b.last_dep[1] = b.last_dep[2] = 0;
    b.n_deps = 0u; b.first_chunk = 0xffffffffu; b.cur_chunk = 0xffffffffu; b.fill = kLightDepChunk;
    for (int f = 0; f < 6; f++) b.value_of(J.block_sky[f], b.sky_value[f]);

    const uint32_t ci = J.cubes[tid];
    const uint32_t sz = (uint32_t)J.size[2], sy = (uint32_t)J.size[1];
    const int origin[3] = {J.lo[0] + (int)(ci / (sy * sz)), J.lo[1] + (int)((ci / sz) % sy), J.lo[2] + (int)(ci % sz)};
    const DevDerived *ev_origin = &J.derived[J.grid[ci] & J.index_mask];
    const bool origin_is_opaque = (ev_origin->flags & 63u) == 63u;
    const bool origin_emits = !(ev_origin->emission[0] == 0.f && ev_origin->emission[1] == 0.f && ev_origin->emission[2] == 0.f);

    if (origin_is_opaque) {
        if (origin_emits) {  // !opaque_for_light_computation (updater.rs:1031-1033)
            const float e[3] = {ev_origin->emission[0], ev_origin->emission[1], ev_origin->emission[2]};
            b.add_weighted_light(e, 1.0f);
        }
    } else {
        // directions_to_seek_light (updater.rs:668-690)
        uint32_t dw_mask = 0u;
        if (ev_origin->flags & kDerivedVisible) dw_mask = 63u;
        else {
            bool nb_visible[6], nb_emits[6];
            for (int f = 0; f < 6; f++) {
                int n[3];
                normal_of(f, n);
                const int c[3] = {origin[0] + n[0], origin[1] + n[1], origin[2] + n[2]};
                uint32_t i;
                nb_visible[f] = false; nb_emits[f] = false;
                if (b.index_of(c, &i)) {
                    const DevDerived *d = &J.derived[J.grid[i] & J.index_mask];
                    nb_visible[f] = (d->flags & kDerivedVisible) != 0u;
                    nb_emits[f] = !(d->emission[0] == 0.f && d->emission[1] == 0.f && d->emission[2] == 0.f);
                }
            }
            for (int f = 0; f < 6; f++) {
                const int opp = f >= 3 ? f - 3 : f + 3;
                if (nb_visible[opp] || nb_emits[f]) dw_mask |= 1u << f;
            }
        }

        // walk_ray_tree (updater.rs:427-530), iteratively. The registers hold the call being executed; the frames of
        // its callers are in J.stack.
        uint32_t node = 0u;
        int f_next = 0;
        int cube[3] = {origin[0], origin[1], origin[2]};
        int fe = -1;
        float alpha = 1.0f;
        bool prev_has = false, ahead_has = false;
        uint32_t prev = 0u, ahead = 0u;
        float rbw = 0.f, cws = 0.f;
        uint32_t depth = 0u;
        uint32_t *const stack = J.stack + tid;
        const size_t sstride = J.stack_stride;

        bool starting = true;
        float ret = 0.f;
        for (;;) {
            const DevLightNode *nd = &J.chart[node];
            float w[6];
            for (int f = 0; f < 6; f++) w[f] = nd->weight[f];
            bool returned = false;
            if (starting) {
                float p[6];
                for (int f = 0; f < 6; f++) p[f] = ((dw_mask >> f) & 1u) ? w[f] : 0.f;
                rbw = (p[0] + p[3]) + (p[1] + p[4]) + (p[2] + p[5]);
                if (rbw <= 0.0f) {
                    ret = rbw;
                    returned = true;
                } else {
                    const double dx = ((double)cube[0] + 0.5) - ((double)origin[0] + 0.5), dy = ((double)cube[1] + 0.5) - ((double)origin[1] + 0.5),
                                 dz = ((double)cube[2] + 0.5) - ((double)origin[2] + 0.5);
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    uint32_t entered_index = 0u;
                    if (d2 > J.max_dist_sq) {
                        b.end_of_ray(alpha, rbw, w);
                        ret = rbw;
                        returned = true;
                    } else {
                        b.cost += 1u;
                        if (!b.index_of(cube, &entered_index)) {
                            b.end_of_ray(alpha, rbw, w);
                            ret = rbw;
                            returned = true;
                        } else {
                            ahead_has = false;
                            ahead = 0u;
                            traverse(b, alpha, dw_mask, cube, fe, &J.derived[J.grid[entered_index] & J.index_mask], ahead_has, ahead, prev_has, prev, w);
                            if (!(alpha > 0.0f)) {
                                b.end_of_ray(alpha, rbw, w);
                                ret = rbw;
                                returned = true;
                            } else {
                                cws = 0.f;
                                f_next = 0;
                                starting = false;
                            }
                        }
                    }
                }
            }
            if (!returned) {
                // the children of this call
                int f = f_next;
                uint32_t child = 0u;
                for (; f < 6; f++) {
                    child = nd->child[f];
                    if (child != 0u) break;
                }
                if (f < 6) {
                    // push this call's frame and start the child's
                    if (depth >= J.max_depth) { J.dep_head[1] = 2u; break; }  // cannot happen: max_depth is the chart's depth
                    uint32_t *fr = stack + (size_t)depth * kLightFrameWords * sstride;
                    fr[0 * sstride] = node | ((uint32_t)(f + 1) << 20) | (ahead_has ? 1u << 24 : 0u) | (dw_mask << 25);
                    fr[1 * sstride] = (uint32_t)cube[0];
                    fr[2 * sstride] = (uint32_t)cube[1];
                    fr[3 * sstride] = (uint32_t)cube[2];
                    fr[4 * sstride] = __float_as_uint(alpha);
                    fr[5 * sstride] = ahead;
                    fr[6 * sstride] = __float_as_uint(rbw);
                    fr[7 * sstride] = __float_as_uint(cws);
                    depth++;
                    int n[3];
                    normal_of(f, n);
                    cube[0] += n[0]; cube[1] += n[1]; cube[2] += n[2];
                    fe = f >= 3 ? f - 3 : f + 3;  // the child is entered through the opposite face
                    prev_has = ahead_has;
                    prev = ahead;
                    node = child;
                    starting = true;
                    continue;
                }
                const float rest = rbw - cws;
                b.end_of_ray(alpha, rest > 0.0f ? rest : 0.0f, w);
                ret = rbw;
            }
            // return to the caller
            if (depth == 0u) break;
            depth--;
            const uint32_t *fr = stack + (size_t)depth * kLightFrameWords * sstride;
            const uint32_t w0 = fr[0 * sstride];
            node = w0 & 0xfffffu;
            f_next = (int)((w0 >> 20) & 15u);
            ahead_has = ((w0 >> 24) & 1u) != 0u;
            dw_mask = w0 >> 25;
            cube[0] = (int)fr[1 * sstride];
            cube[1] = (int)fr[2 * sstride];
            cube[2] = (int)fr[3 * sstride];
            alpha = __uint_as_float(fr[4 * sstride]);
            ahead = fr[5 * sstride];
            rbw = __uint_as_float(fr[6 * sstride]);
            cws = __uint_as_float(fr[7 * sstride]) + ret;
            starting = false;
        }
    }

    // LightBuffer::finish (updater.rs:940-952)
    uint32_t texel;
    const float scale = ps_new_clamped(1.0f / fmaxf(b.total_ray_weight, 1.0f));
    if (b.total_ray_weight > 0.0f) {
        texel = packed_scalar_in(ps_mul(b.incoming[0], scale)) | (packed_scalar_in(ps_mul(b.incoming[1], scale)) << 8) |
                (packed_scalar_in(ps_mul(b.incoming[2], scale)) << 16) | (255u << 24);
    } else if (origin_is_opaque) {
        texel = 128u << 24;
    } else {
        texel = 1u << 24;
    }
    uint32_t *o = J.out + 4 * (size_t)tid;
    o[0] = texel;
    o[1] = b.n_deps;
    o[2] = b.first_chunk;
    o[3] = b.cost;
}

// ------------------------------------------------------------------------------------------------------------------
// One WAVE per cube. What a bundle adds to the accumulators depends only on the ray state along its own root path (alpha;
// the direction weights are fixed per cube once the walk has started), never on the accumulators and never on its
// siblings. So the tree is walked LEVEL BY LEVEL: the wave keeps the frontier of live bundles (at most one per ray: 602)
// in LDS, each lane visits frontier entries -- exactly the bundles the reference's recursion visits, no more -- and appends
// their children, with the alpha behind the parent, to the next frontier. Every contribution is written to the slot that
// its place in the reference's recursion order gives it -- a static number: three slots when a bundle is entered (face
// term, volume term, end of ray), one when it is left (the "rest" term, updater.rs:521-528) -- and a bit is set in an LDS
// bitmap. When the frontier is empty, the set bits in ascending order ARE the reference's order of additions: lane 0
// adds them up, so every f32 sum is the reference's sum. Dependencies likewise (two slots per bundle), with the drop of a
// face's light cube that repeats the previous entry (updater.rs:838-842) done in that ordered pass.

#ifndef AIC_LIGHT_CHUNK
// Bundles of a chain fetched at once by the small-batch builds of the walk (the large-batch build always takes them one by one).
// 2, 4 and 8 were measured (profiles/r04_experiments.txt J): no faster, 8 is slower -- a step of the walk is its own ~1 k
// dependent instructions on a SIMD that runs one wave, not the latency of its two fetches. So: 1.
#define AIC_LIGHT_CHUNK 1
#endif
constexpr uint32_t kLdsFlags = 1024u;   // DevDerived.flags of the first blocks, cached in LDS
constexpr uint32_t kOrderCap = 2048u;   // set bits gathered per pass of the ordered reduction
constexpr uint32_t kLightBlock = 256u;  // most threads a cube's block may have (1 or 4 waves)

// The walk's loop uses a dozen fields of the job; with ~100 scalar registers already spoken for, the compiler re-reads them from the kernel-argument
// segment inside the loop (four s_load + wait per step of a chain: ISA of round 4). A value passed through here lives in a vector register instead --
// the small-batch build has 370 of them to spare.
template <class T>
__device__ __forceinline__ T keep_in_vgpr(T x) {
    asm volatile("" : "+v"(x));
    return x;
}

struct WaveCtx {
    const LightJob &J;
    const DevDerived *derived_p;  // J.derived, held in vector registers (keep_in_vgpr)
    int origin[3];
    uint32_t m0;            // direction weights of the walk: bit per face (updater.rs:668-690)
    float sky_value[6][3];
    float4 *slots;          // [4 * n_tree] contributions by recursion-order number
    uint32_t *cslots;       // [2 * n_tree] dependency candidates
    uint32_t *term_bits, *cand_bits;  // LDS bitmaps
    uint32_t cost;
    const uint32_t *lds_flags;
    const float *lds_lut;

    __device__ explicit WaveCtx(const LightJob &j) : J(j), derived_p(j.derived) {}
    __device__ uint32_t flags_of(uint32_t block) const { return block < kLdsFlags ? lds_flags[block] : derived_p[block].flags; }
    __device__ bool index_of(const int c[3], uint32_t *out) const {
        const uint32_t dx = (uint32_t)c[0] - (uint32_t)J.lo[0], dy = (uint32_t)c[1] - (uint32_t)J.lo[1], dz = (uint32_t)c[2] - (uint32_t)J.lo[2];
        if ((dx >= (uint32_t)J.size[0]) | (dy >= (uint32_t)J.size[1]) | (dz >= (uint32_t)J.size[2])) return false;
        *out = (dx * (uint32_t)J.size[1] + dy) * (uint32_t)J.size[2] + dz;
        return true;
    }
    __device__ uint32_t light_outside(const int c[3]) const {  // sky.rs:113-147
        int n_less = 0, n_equal = 0, which = -1;
        for (int a = 0; a < 3; a++) {
            int lower;
            if (J.lo[a] == (int32_t)0x80000000) lower = -1;
            else {
                const int beyond = J.lo[a] - 1;
                lower = beyond < c[a] ? -1 : (beyond == c[a] ? 0 : 1);
            }
            if (lower == -1) n_less++;
            else if (lower == 0) { n_equal++; which = a; }
        }
        for (int a = 0; a < 3; a++) {
            const long long hi = (long long)J.lo[a] + J.size[a];
            const int upper = (long long)c[a] < hi ? -1 : ((long long)c[a] == hi ? 0 : 1);
            if (upper == -1) n_less++;
            else if (upper == 0) { n_equal++; which = 3 + a; }
        }
        if (n_less == 5 && n_equal == 1) return J.block_sky[which];
        if (n_less == 6) return 0u;
        return 1u << 24;
    }
    __device__ uint32_t get_light(const int c[3]) const {
        uint32_t i;
        if (index_of(c, &i)) return light_texel(i);
        return light_outside(c);
    }
    // In a session the volume changes between batches without a kernel boundary: the texels are read past the caches that are
    // not coherent across XCDs (agent scope), everything else (tables, grid, block records: constant during a call) stays cached.
    __device__ uint32_t light_texel(uint32_t i) const {
        return J.coherent_light ? __hip_atomic_load(const_cast<uint32_t *>(&J.light[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : J.light[i];
    }
    __device__ void value_of(uint32_t texel, float v[3]) const {
        v[0] = lds_lut[texel & 255u];
        v[1] = lds_lut[(texel >> 8) & 255u];
        v[2] = lds_lut[(texel >> 16) & 255u];
    }
    __device__ float bundle_weight(const float w[6]) const {  // (weight * direction_weights).sum(), face.rs:1046-1054
        float p[6];
        for (int f = 0; f < 6; f++) p[f] = ((m0 >> f) & 1u) ? w[f] : 0.f;
        return (p[0] + p[3]) + (p[1] + p[4]) + (p[2] + p[5]);
    }
    __device__ void emit_term(uint32_t seq, float x, float y, float z, float weight) {
        slots[seq] = make_float4(x, y, z, weight);
        atomicOr(&term_bits[seq >> 5], 1u << (seq & 31u));
    }
    __device__ void emit_cand(uint32_t seq, const int c[3], uint32_t conditional) {
        cslots[seq] = (uint32_t)(c[0] - origin[0] + 256) | ((uint32_t)(c[1] - origin[1] + 256) << 10) | ((uint32_t)(c[2] - origin[2] + 256) << 20) |
                      (conditional << 30);
        atomicOr(&cand_bits[seq >> 5], 1u << (seq & 31u));
    }
    // updater.rs:899-927 + add_weighted_light
    __device__ void end_of_ray(uint32_t seq, float alpha, float ray_bundle_weight, const float w[6]) {
        if (!(ray_bundle_weight > 0.f)) return;
        const float recip = ps_new_clamped(1.0f / ((w[0] + w[3]) + (w[1] + w[4]) + (w[2] + w[5])));
        const float ww = ps_new_clamped(ray_bundle_weight);
        float out[3];
        for (int i = 0; i < 3; i++) {
            float pf[6];
            for (int f = 0; f < 6; f++) pf[f] = ps_mul(sky_value[f][i], ps_new_clamped(w[f]));
            const float s = (pf[0] + pf[3]) + (pf[1] + pf[4]) + (pf[2] + pf[5]);
            const float sky_light = ps_mul(ps_mul(s, recip), ps_new_clamped(alpha));
            out[i] = ps_mul(sky_light, ww);
        }
        emit_term(seq, out[0], out[1], out[2], ray_bundle_weight);
    }
    // The part of walk_ray_tree that the bundle's CHILDREN depend on: is the bundle alive behind its cube, and with what
    // alpha (updater.rs:440-496; the alpha arithmetic of LightBuffer::traverse, updater.rs:800-893, without the light it
    // gathers). This is the critical path of the walk -- one level waits for the one before -- so it is kept apart from
    // `visit`, which computes what the bundle adds to the light and depends on nothing but {k, alpha_in}. `block` is the
    // block at the bundle's cube, `fi` the face the bundle enters it through (7: the origin cube).
    __device__ bool alpha_behind(uint32_t block, uint32_t fi, float alpha_in, float *alpha_out) const {
        const int fe = fi == 7u ? -1 : (int)fi;
        const uint32_t flags = flags_of(block);
        float alpha = alpha_in;
        if (flags & kDerivedVisible) {
            const bool hit_opaque_face = fe < 0 ? (flags & 63u) == 63u : ((flags >> fe) & 1u) != 0u;
            if (hit_opaque_face && fe < 0) {
                alpha = 0.f;
            } else {
                const DevDerived *ev = &derived_p[block];
                const float hit_alpha = fe < 0 ? ev->color[3] : ev->face[fe][3];
                if (hit_alpha > 0.f && fe >= 0) {
                    if (hit_opaque_face) alpha = 0.f;
                    else alpha *= 1.0f - hit_alpha;
                }
                if (hit_alpha < 1.0f) alpha *= 1.0f - hit_alpha;
            }
        }
        *alpha_out = alpha;
        return alpha > 0.0f;
    }
    // walk_ray_tree (updater.rs:427-530) for the bundle at tree position k, entered with alpha_in. Returns whether its
    // children are to be walked, then *alpha_out is the alpha behind the cube. Everything the bundle adds -- on entering
    // and, if it lives, after its children -- is emitted here, into its numbered slots.
    __device__ bool visit(uint32_t k, float alpha_in, float *alpha_out) {
        const DevTreePos *nd = &J.tree[k];
        float w[6];
        for (int f = 0; f < 6; f++) w[f] = nd->weight[f];
        const uint32_t info = nd->info, off = nd->offset, end = nd->end;
        // the children's chart weights, for the "rest" term below: fetched up front, beside the cube lookup
        const float *cw = J.child_w + (size_t)k * 36u;
        float cwv[36];
        for (int i = 0; i < 36; i++) cwv[i] = cw[i];
        const float rbw = bundle_weight(w);
        if (rbw <= 0.0f) return false;
        const uint32_t depth = (info >> 8) & 0xffffu;
        const uint32_t seq = 4u * k - depth;  // 3 per bundle entered before, 1 per bundle left before
        float alpha = alpha_in;
        // Every live bundle ends in exactly one end_of_ray: beyond the distance / outside the space (slot seq), the ray
        // used up inside the cube (slot seq + 2), or -- after its children -- with the weight they did not take (the slot of
        // the bundle's exit). The cases only select its arguments, so that the wave runs the arithmetic once.
        bool alive = false;
        uint32_t eseq = seq;
        float eweight = rbw;
        const int cube[3] = {origin[0] + (int)(off & 1023u) - 256, origin[1] + (int)((off >> 10) & 1023u) - 256, origin[2] + (int)((off >> 20) & 1023u) - 256};
        uint32_t idx = 0u;
        if (!(info & 8u)) {  // within maximum_distance
            cost += 1u;
            if (index_of(cube, &idx)) {
                const int fe = (info & 7u) == 7u ? -1 : (int)(info & 7u);
                // LightBuffer::traverse (updater.rs:770-895)
                const uint32_t block = J.grid[idx] & J.index_mask;
                const uint32_t flags = flags_of(block);
                if (flags & kDerivedVisible) {
                    const DevDerived *ev = &J.derived[block];
                    const bool hit_opaque_face = fe < 0 ? (flags & 63u) == 63u : ((flags >> fe) & 1u) != 0u;
                    if (hit_opaque_face && fe < 0) {
                        alpha = 0.f;
                    } else {
                        const float *scp = fe < 0 ? ev->color : ev->face[fe];
                        float sc[4] = {scp[0], scp[1], scp[2], scp[3]};
                        for (int i = 0; i < 3; i++) sc[i] = sc[i] > 1.f ? 1.f : sc[i];
                        const float hit_alpha = sc[3];
                        const float em[3] = {ev->emission[0], ev->emission[1], ev->emission[2]};
                        if (hit_alpha > 0.f && fe >= 0) {
                            int n[3];
                            normal_of(fe, n);
                            const int light_cube[3] = {cube[0] + n[0], cube[1] + n[1], cube[2] + n[2]};
                            float sv[3];
                            value_of(get_light(light_cube), sv);
                            const float a = ps_new_clamped(alpha), ww = ps_new_clamped(rbw);
                            float t[3];
                            for (int i = 0; i < 3; i++) t[i] = ps_mul(ps_mul(em[i] + ps_mul(ps_mul(sc[i], sv[i]), sc[3]), a), ww);
                            emit_term(seq, t[0], t[1], t[2], 0.f);
                            cost += 10u;
                            emit_cand(2u * k, light_cube, 1u);
                            if (hit_opaque_face) alpha = 0.f;
                            else alpha *= 1.0f - hit_alpha;
                        }
                        if (hit_alpha < 1.0f) {
                            float sl[3] = {0.f, 0.f, 0.f};
                            if (fe >= 0) value_of(light_texel(idx), sl);
                            const float a = ps_new_clamped(alpha), ww = ps_new_clamped(rbw);
                            float t[3];
                            for (int i = 0; i < 3; i++) t[i] = ps_mul(ps_mul(em[i] + ps_mul(sl[i], hit_alpha), a), ww);
                            emit_term(seq + 1u, t[0], t[1], t[2], 0.f);
                            cost += 10u;
                            emit_cand(2u * k + 1u, cube, 0u);
                            alpha *= 1.0f - hit_alpha;
                        }
                    }
                }
                if (alpha > 0.0f) {
                    // After the children: the weight they did not take (updater.rs:507-528). What a child returns is its own
                    // bundle weight (updater.rs:437-441), a function of its chart weights and this walk's direction weights only.
                    alive = true;
                    float cws = 0.0f;
                    for (int f = 0; f < 6; f++) cws += bundle_weight(cwv + 6 * f);  // no child on that face: zeros, and x + 0 = x
                    const float rest = rbw - cws;
                    eweight = rest > 0.0f ? rest : 0.0f;
                    eseq = 4u * end - depth - 1u;
                } else {
                    eseq = seq + 2u;
                }
            }
        }
        end_of_ray(eseq, alpha, eweight, w);
        *alpha_out = alpha;
        return alive && end > k + 1u;
    }
};

// Gathers set bits of bits[*word_io, n_words) into order[] in ascending order -- all of them if they fit kOrderCap, else
// the longest prefix of words that does -- and advances *word_io. All threads of the block take part (no divergence
// around the call): each counts and then writes out a contiguous stretch of words; `s_scan` (>= 4 words) carries the
// waves' counts. Returns the number gathered; 0 = nothing left; 0xffffffff = this span was empty but words remain.
__device__ uint32_t gather_set_bits(const uint32_t *bits, uint32_t n_words, uint32_t *word_io, uint32_t *order, uint32_t tid, uint32_t nt, uint32_t *s_scan) {
    const uint32_t w0 = *word_io;
    if (w0 >= n_words) return 0u;
    const uint32_t wl = tid & 63u, wid = tid >> 6, nw = nt >> 6;
    uint32_t span = n_words - w0, a, e, count, incl, total, base;
    for (;;) {
        const uint32_t per = (span + nt - 1u) / nt;
        a = min(w0 + tid * per, w0 + span);
        e = min(a + per, w0 + span);
        count = 0u;
        for (uint32_t w = a; w < e; w++) count += __popc(bits[w]);
        incl = count;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if ((int)wl >= d) incl += up;
        }
        if (wl == 63u) s_scan[wid] = incl;
        __syncthreads();
        base = 0u; total = 0u;
        for (uint32_t q = 0u; q < nw; q++) {
            const uint32_t t = s_scan[q];
            if (q < wid) base += t;
            total += t;
        }
        __syncthreads();
        if (total <= kOrderCap || span <= kOrderCap / 32u) break;
        span = max(kOrderCap / 32u, span / 2u);
    }
    uint32_t at = base + incl - count;
    for (uint32_t w = a; w < e; w++) {
        uint32_t v = bits[w];
        while (v) { const uint32_t bpos = __ffs(v) - 1u; v &= v - 1u; order[at++] = w * 32u + bpos; }
    }
    __syncthreads();
    *word_io = w0 + span;
    return total == 0u ? 0xffffffffu : total;
}

// SESSION: the same walk, but the launch stays on the device for a whole aic_evaluate_light call and is handed one small batch
// after another through a page of pinned host memory (LightMailbox, aic_light.h) instead of being launched once per batch:
// in the reference's order a scene is a chain of ~1800 dependent batches of 32 cubes, and a launch per batch costs its
// dispatch, the refill of every XCD's L2 (the caches are invalidated at a kernel boundary: 8.9 MB fetched per launch against
// 0.57 MB algorithmic) and the host's wake-up from hipStreamSynchronize on top of the walk itself. One workgroup per cube of
// the batch (gridDim.x >= the batch size), workgroup 0 talks to the host.
template <bool SESSION, int CHUNK, bool LDSQ>
__device__ __forceinline__ void compute_light_wave_body(const LightJob &J) {
    extern __shared__ uint32_t s_dyn[];  // term bitmap, candidate bitmap, visited bitmap
    __shared__ uint32_t s_flags[kLdsFlags];
    __shared__ float s_lut[256];
    __shared__ uint32_t s_count[3], s_scan[4], s_chunk[8], s_first;
    __shared__ float s_sum[4];
    __shared__ float4 s_stage[kLightBlock];
    __shared__ uint32_t s_order[kOrderCap];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x, nt = blockDim.x;  // `lane`: thread of the cube's block (1 or 4 waves)
    const uint32_t term_words = (4u * J.n_tree + 31u) / 32u, cand_words = (2u * J.n_tree + 31u) / 32u, vis_words = (J.n_tree + 31u) / 32u;
    uint32_t *const term_bits = s_dyn, *const cand_bits = s_dyn + term_words, *const vis_bits = cand_bits + cand_words;
    // LDSQ: the walk's work queue lives in LDS behind the bitmaps (16-byte aligned) instead of in global memory. A hand-off of a branching bundle's children
    // to other lanes -- up to nine on a root path -- then costs LDS round trips, not a store to L2, a poll of L2 and a fetch from L2 (~4 us each at the
    // small-batch build's one wave per SIMD). Used when the queue of this maximum_distance fits (light_wave_lds); all zero between cubes, like the global one.
    uint4 *const lds_front = reinterpret_cast<uint4 *>(s_dyn + ((term_words + cand_words + vis_words + 3u) & ~3u));
    if (LDSQ) {
        for (uint32_t i = lane; i < J.n_front; i += nt) lds_front[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    for (uint32_t i = lane; i < min(J.n_blocks, kLdsFlags); i += nt) s_flags[i] = J.derived[i].flags;
    for (uint32_t i = lane; i < 256u; i += nt) s_lut[i] = J.lut[i];
    __syncthreads();
    WaveCtx b(J);
    b.lds_flags = s_flags;
    b.lds_lut = s_lut;
    b.term_bits = term_bits;
    b.cand_bits = cand_bits;
    b.slots = J.terms + (size_t)wave * 4u * J.n_tree;
    b.cslots = J.cands + (size_t)wave * 2u * J.n_tree;
    for (int f = 0; f < 6; f++) b.value_of(J.block_sky[f], b.sky_value[f]);

    __shared__ uint32_t s_session[3];
    uint32_t session_seq = J.session_seq;
    for (uint32_t item = wave; SESSION || item < J.n; item += gridDim.x) {
        bool active = true;
        if (SESSION) {
            LightMailbox *const mb = J.mailbox;
            item = wave;
            if (wave == 0u) {
                if (lane == 0u) {
                    uint32_t spins = 0u, got;
                    while ((got = __hip_atomic_load(&mb->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) != session_seq && ++spins < kLightSessionSpins)
                        __builtin_amdgcn_s_sleep(4);
                    const bool seen = got == session_seq;
                    s_session[0] = seen ? __hip_atomic_load(&mb->n_cubes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                    s_session[1] = seen ? __hip_atomic_load(&mb->n_scatter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                    s_session[2] = seen ? 0u : 1u;
                    if (seen) __hip_atomic_store(&mb->t_seen, (uint64_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __syncthreads();
                const uint32_t n = min(s_session[0], 64u), ns = min(s_session[1], 64u);
                if (lane < 64u) {  // (one wave does the stores, so that its own fence below covers all of them)
                    if (lane < ns) {
                        const uint32_t idx = __hip_atomic_load(&mb->scatter_index[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const uint32_t texel = __hip_atomic_load(&mb->scatter_texel[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(&J.light_rw[idx], texel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane < n)
                        __hip_atomic_store(&J.session_cubes[lane], __hip_atomic_load(&mb->cubes[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    __threadfence();
                    if (lane == 0u) {
                        __hip_atomic_store(&J.session_word[1], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&J.session_word[0], session_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        if (s_session[2]) __hip_atomic_store(&mb->exited, session_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                __syncthreads();
            } else {
                if (lane == 0u) {
                    // (bounded far beyond workgroup 0's own patience: it always publishes something, a batch or the order to leave)
                    uint32_t spins = 0u, got;
                    while ((got = __hip_atomic_load(&J.session_word[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) != session_seq && ++spins < 16u * kLightSessionSpins)
                        __builtin_amdgcn_s_sleep(2);
                    const bool seen = got == session_seq;
                    s_session[0] = seen ? __hip_atomic_load(&J.session_word[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    if (!seen) __hip_atomic_store(&mb->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __syncthreads();
            }
            const uint32_t batch_n = min(s_session[0], 64u);
            if (batch_n == 0u) return;  // the session is over (or the host went quiet): the same for every thread of the block
            session_seq++;
            active = item < batch_n;
        }
        if (active) {
        const uint32_t ci = SESSION ? __hip_atomic_load(&J.session_cubes[item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : J.cubes[item];
        const uint32_t sz = (uint32_t)J.size[2], sy = (uint32_t)J.size[1];
        b.origin[0] = J.lo[0] + (int)(ci / (sy * sz));
        b.origin[1] = J.lo[1] + (int)((ci / sz) % sy);
        b.origin[2] = J.lo[2] + (int)(ci % sz);
        const DevDerived *ev_origin = &J.derived[J.grid[ci] & J.index_mask];
        const bool origin_is_opaque = (ev_origin->flags & 63u) == 63u;
        const bool origin_emits = !(ev_origin->emission[0] == 0.f && ev_origin->emission[1] == 0.f && ev_origin->emission[2] == 0.f);
        b.cost = 0u;
        uint32_t n_visits = 0u;  // bundles of the ray tree this thread visited (the unit of the light updater's roofline: bench.py)
#ifdef AIC_LIGHT_TIMING
        const long long t_begin = clock64();
        uint32_t n_rounds = 0u;
#endif

        if (!origin_is_opaque) {
            for (uint32_t i = lane; i < term_words + cand_words + vis_words; i += nt) s_dyn[i] = i == term_words + cand_words ? 1u : 0u;  // the root is visited
            // directions_to_seek_light (updater.rs:668-690)
            // (the six neighbours are looked at by six lanes at once: each is a chain of dependent fetches)
            bool nb_visible = false, nb_emits = false;
            const bool origin_visible = (ev_origin->flags & kDerivedVisible) != 0u;
            if (!origin_visible && lane < 6u) {
                int n[3];
                normal_of((int)lane, n);
                const int c[3] = {b.origin[0] + n[0], b.origin[1] + n[1], b.origin[2] + n[2]};
                uint32_t i;
                if (b.index_of(c, &i)) {
                    const DevDerived *d = &J.derived[J.grid[i] & J.index_mask];
                    nb_visible = (d->flags & kDerivedVisible) != 0u;
                    nb_emits = !(d->emission[0] == 0.f && d->emission[1] == 0.f && d->emission[2] == 0.f);
                }
            }
            const uint32_t vis = (uint32_t)__ballot(nb_visible), emi = (uint32_t)__ballot(nb_emits);
            if (lane == 0u) {
                uint32_t m0 = 0u;
                if (origin_visible) m0 = 63u;
                else
                    for (int f = 0; f < 6; f++) {
                        const int opp = f >= 3 ? f - 3 : f + 3;
                        if (((vis >> opp) & 1u) || ((emi >> f) & 1u)) m0 |= 1u << f;
                    }
                s_scan[0] = m0;
            }
            uint4 *const front = LDSQ ? lds_front : J.front + (size_t)wave * J.n_front;
            float *const valpha = keep_in_vgpr(J.valpha + (size_t)wave * J.n_tree);
            const uint4 *const node_p = keep_in_vgpr(J.node);
            const uint16_t *const grid_p = keep_in_vgpr(J.grid);
            const uint2 *const child_ent_p = keep_in_vgpr(J.child_ent);
            const uint32_t index_mask_v = keep_in_vgpr(J.index_mask), n_tree_v = keep_in_vgpr(J.n_tree);
            b.derived_p = keep_in_vgpr(J.derived);
            if (lane == 0u) {
                front[0] = make_uint4(0u, __float_as_uint(1.0f), J.tree[0].offset, J.root_meta | kLightQueueValid);
                valpha[0] = 1.0f;
                s_count[0] = 0u;  // entries claimed
                s_count[1] = 1u;  // entries appended
                s_count[2] = 0u;  // entries walked to their end
            }
            __syncthreads();
            b.m0 = __builtin_amdgcn_readfirstlane(s_scan[0]);  // the same in every lane: keep it in a scalar register
            // 1. Which bundles does the walk visit, and with what alpha? A bundle decides its children's alpha, so this is the
            //    serial part, a chain of dependent fetches and little else: its length decides what a small batch costs.
            //    Nearly all of the tree is chains (24 140 of the 25 183 positions of maximum_distance 30 have exactly one
            //    child; no path from the root branches more than nine times). So a lane walks: it follows a bundle's only child
            //    at once, and where a bundle branches it appends the children to a queue and looks for other work -- no
            //    barriers, so the walk lasts as long as its deepest path (about 25 to 50 steps) and not as long as the sum of
            //    the longest chains of barrier-separated rounds. Lanes never wait for one another (an idle lane looks at the
            //    queue once per step of its wave; the wave leaves when every appended entry has been walked to its end), so
            //    lanes of one wave cannot block each other. A visited bundle is a bit in `vis_bits` and its alpha in `valpha`:
            //    neither waits for anything. A step fetches two things side by side, 16 bytes of the bundle (its only
            //    child's entry, its number of children) and the block at its cube; where that cube is, the face it is entered
            //    through and whether the walk's directions give the bundle any weight came with the entry.
            {
                // The queue's counters are read as LDS atomics (`done` acquired before the appended count is read). Round 2 read
                // them through a volatile generic pointer, which the compiler turns into system-coherent flat loads that each
                // wait for all of the wave's outstanding memory operations (-DAIC_LIGHT_GENERIC_COUNTERS builds that again); on
                // hardware the difference is 1.4 % of the device time (profiles/r03_experiments.txt B), 300-seed fuzz green.
#ifndef AIC_LIGHT_GENERIC_COUNTERS
                struct {
                    uint32_t *c;
                    __device__ uint32_t operator[](int i) const {
                        return i == 2 ? __hip_atomic_load(&c[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
                                      : __hip_atomic_load(&c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } const q = {s_count};
#else
                volatile uint32_t *const q = s_count;
#endif
                bool active = false;
                uint32_t claim = 0xffffffffu, k = 0u, off = 0u, meta = 0u, guard = 0u;
                float alpha_in = 0.f;
                for (;;) {
                    if (!active) {
                        // the claim counter may run ahead of the appended count: such a claim is served when the entry arrives
                        if (claim == 0xffffffffu && q[0] < q[1]) claim = atomicAdd(&s_count[0], 1u);
                        if (claim != 0xffffffffu && claim < q[1]) {
                            const uint32_t w = __hip_atomic_load(&front[claim].w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (w != 0u) {
                                const uint4 e = front[claim];
                                k = e.x; alpha_in = __uint_as_float(e.y); off = e.z; meta = w & 0x3ffu;
                                active = true;
                                claim = 0xffffffffu;
                            }
                        }
                    }
                    if (active) {
                        // Along a chain the positions are consecutive (pre-order: a bundle's only child is the next position), and
                        // where a bundle's cube is does not depend on what the walk finds: so a lane fetches CHUNK bundles and
                        // their CHUNK cubes' blocks in two round trips and then takes up to CHUNK steps on what it has -- a step
                        // no longer costs a fetch's latency each. Whatever was fetched past the chain's end is not looked at.
                        uint4 nd[CHUNK];
_Pragma("unroll")
                        for (int i = 0; i < CHUNK; i++) nd[i] = node_p[min(k + (uint32_t)i, n_tree_v - 1u)];
                        uint32_t block[CHUNK];
                        bool inside[CHUNK];
                        {
                            uint32_t off_i = off;
_Pragma("unroll")
                            for (int i = 0; i < CHUNK; i++) {
                                if (i > 0) off_i = nd[i - 1].y;
                                const int cube[3] = {b.origin[0] + (int)(off_i & 1023u) - 256, b.origin[1] + (int)((off_i >> 10) & 1023u) - 256,
                                                     b.origin[2] + (int)((off_i >> 20) & 1023u) - 256};
                                uint32_t idx = 0u;
                                inside[i] = b.index_of(cube, &idx);
                                block[i] = grid_p[idx] & index_mask_v;  // cube 0's if outside: unused then
                            }
                        }
_Pragma("unroll")
                        for (int i = 0; i < CHUNK; i++) asm volatile("" : "+v"(nd[i].z), "+v"(block[i]));  // the fetches stay here, ahead of the decisions
_Pragma("unroll")
                        for (int i = 0; i < CHUNK; i++) {
                            float alpha;
                            // (weight * direction_weights).sum() > 0: some weight the walk's directions select is positive
                            if ((meta & 8u) || !inside[i] || !(b.m0 & (meta >> 4)) || !b.alpha_behind(block[i], meta & 7u, alpha_in, &alpha) || nd[i].z == 0u) {
                                atomicAdd(&s_count[2], 1u);
                                active = false;
                                break;
                            } else if (nd[i].z == 1u) {
                                const uint32_t k_next = nd[i].x & 0x3fffffu;
                                const bool consecutive = k_next == k + 1u;
                                k = k_next;
                                atomicOr(&vis_bits[k >> 5], 1u << (k & 31u));
#ifndef AIC_LIGHT_EXP_NOSTORE
                                valpha[k] = alpha;
#endif
                                off = nd[i].y;
                                meta = (nd[i].x >> 28) | (((nd[i].x >> 22) & 63u) << 4);
                                alpha_in = alpha;
                                if (!consecutive) break;  // (never, in a pre-order tree: what was fetched ahead is then not this child's)
                            } else {
                                const uint2 *ce = child_ent_p + (size_t)k * 6u;
                                uint2 ch[6];
                                for (int f = 0; f < 6; f++) ch[f] = ce[f];
                                const uint32_t at0 = atomicAdd(&s_count[1], nd[i].z);
                                uint32_t at = at0;
                                for (int f = 0; f < 6; f++)
                                    if (ch[f].x != 0u) {
                                        const uint32_t ck = ch[f].x & 0x3fffffu;
                                        atomicOr(&vis_bits[ck >> 5], 1u << (ck & 31u));
                                        valpha[ck] = alpha;
                                        uint32_t *const ent = reinterpret_cast<uint32_t *>(&front[at++]);
                                        ent[0] = ck; ent[1] = __float_as_uint(alpha); ent[2] = ch[f].y;
                                    }
                                at = at0;
                                for (int f = 0; f < 6; f++)  // published behind the rest of the entry
                                    if (ch[f].x != 0u)
                                        __hip_atomic_store(&front[at++].w, (ch[f].x >> 28) | (((ch[f].x >> 22) & 63u) << 4) | kLightQueueValid, __ATOMIC_RELEASE,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                                atomicAdd(&s_count[2], 1u);
                                active = false;
                                break;
                            }
                        }
                    }
                    if (__ballot(active) == 0ull) {  // nobody in this wave is walking
                        const uint32_t done = q[2];  // read before the appended count: equal means nothing is left anywhere
                        if (done == q[1]) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (++guard > (1u << 22)) {  // cannot happen; a GPU that spins forever is worse than an error
                        J.dep_head[1] = 3u;
                        break;
                    }
                }
            }
            __syncthreads();
            for (uint32_t i = lane, n = s_count[1]; i < n; i += nt) front[i].w = 0u;  // the queue is left as it was found
#ifdef AIC_LIGHT_TIMING
            n_rounds = s_count[1];
            if (lane == 0u) atomicAdd(&J.dep_head[6], (uint32_t)((clock64() - t_begin) >> 6));
#endif
            // 2. what every visited bundle adds: independent of one another, all lanes at once
            uint32_t vword = 0u;
            for (;;) {
                const uint32_t got = gather_set_bits(vis_bits, vis_words, &vword, s_order, lane, nt, s_scan);
                if (got == 0u) break;
                if (got == 0xffffffffu) continue;
                for (uint32_t i = lane; i < got; i += nt) {
                    const uint32_t k = s_order[i];
                    float alpha;
                    n_visits++;
                    (void)b.visit(k, valpha[k], &alpha);
                }
                __syncthreads();  // s_order is written again by the next gather
            }
        }
        uint32_t cost = b.cost;  // summed over the block: an integer, any order
        for (int d = 32; d > 0; d >>= 1) cost += __shfl_down(cost, d, 64);
        if ((lane & 63u) == 0u) s_scan[lane >> 6] = cost;
        if (lane == 0u) s_first = 0xffffffffu;
        __syncthreads();
        cost = 0u;
        for (uint32_t q = 0u; q < (nt >> 6); q++) cost += s_scan[q];
        __syncthreads();
        {   // visited bundles of this cube, summed over the block like `cost`: one atomic per wave
            uint32_t v = n_visits;
            for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
            if ((lane & 63u) == 0u && v != 0u) atomicAdd(&J.dep_head[5], v);
        }
#ifdef AIC_LIGHT_TIMING
        const long long t_walk = clock64();
        if (lane == 0u) { atomicAdd(&J.dep_head[2], (uint32_t)((t_walk - t_begin) >> 6)); atomicAdd(&J.dep_head[4], n_rounds); }
#endif

        // Ordered reduction. The set bits in ascending order are the reference's order of additions. They are gathered by
        // the whole block, the terms fetched a block's worth at a time (the next fetch in flight while this one is added),
        // and lanes 0..3 add them in that order: red, green, blue and the weight, one chain of f32 additions each, which is
        // what `LightBuffer` does to its four accumulators.
        float inc[3] = {0.f, 0.f, 0.f}, total = 0.f;
        uint32_t n_deps = 0u;
        if (origin_is_opaque) {
            if (origin_emits) {  // add_weighted_light(emission, 1.0)
                for (int i = 0; i < 3; i++) inc[i] += ps_mul(ev_origin->emission[i], ps_new_clamped(1.0f));
                total += 1.0f;
            }
        } else {
            float acc = 0.f;
            uint32_t word = 0u;
            for (;;) {
                const uint32_t got = gather_set_bits(term_bits, term_words, &word, s_order, lane, nt, s_scan);
                if (got == 0u) break;
                if (got == 0xffffffffu) continue;
                float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < got) nxt = b.slots[s_order[lane]];
                for (uint32_t c0 = 0u; c0 < got; c0 += nt) {
                    const uint32_t m = min(nt, got - c0);
                    s_stage[lane] = nxt;
                    __syncthreads();
                    if (c0 + nt + lane < got) nxt = b.slots[s_order[c0 + nt + lane]];
                    if (lane < 4u) {
                        const float *sf = reinterpret_cast<const float *>(s_stage) + lane;
#pragma unroll 16
                        for (uint32_t i = 0u; i < m; i++) acc += sf[4u * i];
                    }
                    __syncthreads();
                }
            }
            if (lane < 4u) s_sum[lane] = acc;
            // Will the host want this cube's dependency list? It re-queues the dependencies only when the texel moves by more than one unit
            // (apply_light_update, updater.rs:296-340: `difference_priority`), which most updates of a converging volume do not; then the list is
            // neither built nor brought back. Decided here exactly as the host will decide it: the texel the host compares with is the one in the
            // volume now unless it is Uninitialized (only those can be given a guess by a neighbour applied earlier in the same batch) -- for
            // those the list is always built.
            __syncthreads();
            bool emit_deps = true;
            {
                const float tot_ = s_sum[3];
                if (tot_ > 0.0f) {
                    const float scale_ = ps_new_clamped(1.0f / fmaxf(tot_, 1.0f));
                    const uint32_t new_ = packed_scalar_in(ps_mul(s_sum[0], scale_)) | (packed_scalar_in(ps_mul(s_sum[1], scale_)) << 8) |
                                          (packed_scalar_in(ps_mul(s_sum[2], scale_)) << 16) | (255u << 24);
                    const uint32_t old_ = b.light_texel(ci);
                    int diff_ = 0;
                    for (int sh = 0; sh < 24; sh += 8) {
                        const int x_ = (int)((new_ >> sh) & 255u), y_ = (int)((old_ >> sh) & 255u);
                        diff_ = max(diff_, x_ > y_ ? x_ - y_ : y_ - x_);
                    }
                    if ((new_ >> 24) != (old_ >> 24)) diff_ = min(255, diff_ + 255 / 4);  // data.rs:186-211
                    emit_deps = (old_ >> 24) == 0u || diff_ > 1;
                }
            }
            // Dependencies: a candidate is dropped if it is a face cube equal to the candidate before it (`if
            // dependencies.last() != Some(&light_cube)`, updater.rs:838-842 -- "the last pushed" and "the candidate before" are
            // the same cube whenever the test can succeed) or lies outside the space (light_needs_update ignores those); the
            // kept ones go to the cube's chunk list in order, a block's worth at a time.
            uint32_t prev_key = 0xffffffffu, cur_chunk = 0xffffffffu;
            uint32_t *const s_keys = reinterpret_cast<uint32_t *>(s_stage);
            const uint32_t wl = lane & 63u, wid = lane >> 6;
            word = 0u;
            while (emit_deps) {
                const uint32_t got = gather_set_bits(cand_bits, cand_words, &word, s_order, lane, nt, s_scan);
                if (got == 0u) break;
                if (got == 0xffffffffu) continue;
                uint32_t nxt = 0u;
                if (lane < got) nxt = b.cslots[s_order[lane]];
                for (uint32_t c0 = 0u; c0 < got; c0 += nt) {
                    const uint32_t m = min(nt, got - c0);
                    const uint32_t cnd = nxt, key = cnd & 0x3fffffffu;
                    s_keys[lane] = key;
                    __syncthreads();
                    if (c0 + nt + lane < got) nxt = b.cslots[s_order[c0 + nt + lane]];
                    const uint32_t before = lane ? s_keys[lane - 1u] : prev_key;
                    const uint32_t last_key = s_keys[m - 1u];
                    bool keep = lane < m && !((cnd >> 30) && key == before);
                    uint32_t idx = 0u;
                    if (keep) {
                        const int cube[3] = {b.origin[0] + (int)(key & 1023u) - 256, b.origin[1] + (int)((key >> 10) & 1023u) - 256,
                                             b.origin[2] + (int)((key >> 20) & 1023u) - 256};
                        keep = b.index_of(cube, &idx);
                    }
                    const unsigned long long kept = __ballot(keep);
                    if (wl == 0u) s_scan[wid] = (uint32_t)__popcll(kept);
                    __syncthreads();
                    uint32_t base = 0u, stage_total = 0u;
                    for (uint32_t q = 0u; q < (nt >> 6); q++) {
                        const uint32_t t = s_scan[q];
                        if (q < wid) base += t;
                        stage_total += t;
                    }
                    // the chunks this stretch of the list needs: 63 dependencies per chunk behind its link word
                    const uint32_t first_ord = n_deps / (kLightDepChunk - 1u);
                    const uint32_t last_ord = (n_deps + stage_total - 1u) / (kLightDepChunk - 1u);  // unused when stage_total == 0
                    if (lane == 0u && stage_total != 0u) {
                        uint32_t cc = cur_chunk;
                        s_chunk[0] = cc;
                        for (uint32_t ord = first_ord + (n_deps % (kLightDepChunk - 1u) != 0u ? 1u : 0u); ord <= last_ord; ord++) {
                            const uint32_t next = atomicAdd(&J.dep_head[0], 1u);
                            if (next >= J.dep_chunks) {
                                J.dep_head[1] = 1u;  // the host grows the pool and computes the batch again
                                cc = 0xffffffffu;
                            } else {
                                J.dep_pool[(size_t)next * kLightDepChunk] = 0xffffffffu;
                                if (cc != 0xffffffffu) J.dep_pool[(size_t)cc * kLightDepChunk] = next;
                                else if (ord == 0u) s_first = next;
                                cc = next;
                            }
                            s_chunk[ord - first_ord] = cc;
                        }
                    }
                    __syncthreads();
                    if (keep) {
                        const uint32_t d = n_deps + base + (uint32_t)__popcll(kept & ((1ull << wl) - 1ull));
                        const uint32_t chunk = s_chunk[d / (kLightDepChunk - 1u) - first_ord];
                        if (chunk != 0xffffffffu) J.dep_pool[(size_t)chunk * kLightDepChunk + 1u + d % (kLightDepChunk - 1u)] = idx;
                    }
                    if (stage_total != 0u) cur_chunk = s_chunk[last_ord - first_ord];
                    n_deps += stage_total;
                    prev_key = last_key;
                    __syncthreads();
                }
            }
            if (lane == 0u) { inc[0] = s_sum[0]; inc[1] = s_sum[1]; inc[2] = s_sum[2]; total = s_sum[3]; }
        }
        if (lane == 0u) {
            const uint32_t first_chunk = s_first;
            // LightBuffer::finish (updater.rs:940-952)
            uint32_t texel;
            const float scale = ps_new_clamped(1.0f / fmaxf(total, 1.0f));
            if (total > 0.0f) {
                texel = packed_scalar_in(ps_mul(inc[0], scale)) | (packed_scalar_in(ps_mul(inc[1], scale)) << 8) |
                        (packed_scalar_in(ps_mul(inc[2], scale)) << 16) | (255u << 24);
            } else if (origin_is_opaque) {
                texel = 128u << 24;
            } else {
                texel = 1u << 24;
            }
            uint32_t *o = J.out + 4 * (size_t)item;
            o[0] = texel;
            o[1] = n_deps;
            o[2] = first_chunk;
            o[3] = cost;
#ifdef AIC_LIGHT_TIMING
            atomicAdd(&J.dep_head[3], (uint32_t)((clock64() - t_walk) >> 6));
            atomicAdd(&J.dep_head[7], n_deps);
#endif
        }
        }  // active
        __syncthreads();  // LDS and the slots are reused by the wave's next cube
        if (SESSION) {
            // End of a batch: everything this workgroup wrote (results and dependency chunks, in host memory) is made visible to
            // the host before it is counted; the last one out hands over the counters, clears them for the next batch (which the
            // host posts only after it has seen `done`) and says so.
            __threadfence_system();
            __syncthreads();
            if (lane == 0u) {
                if (atomicAdd(J.done_count, 1u) == gridDim.x - 1u) {
                    __threadfence();
                    for (int i = 0; i < 8; i++) {
                        J.host_head[i] = __hip_atomic_load(&J.dep_head[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&J.dep_head[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __hip_atomic_store(J.done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&J.mailbox->t_done, (uint64_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __threadfence_system();
                    __hip_atomic_store(&J.mailbox->done, session_seq - 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    if (!SESSION && J.host_head) {
        // the results went straight to host memory; the counters the host wants with them are copied by the last block out
        __syncthreads();
        if (lane == 0u) {
            __threadfence();
            if (atomicAdd(J.done_count, 1u) == gridDim.x - 1u) {
                __threadfence();
                for (int i = 0; i < 8; i++) J.host_head[i] = __hip_atomic_load(&J.dep_head[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Two builds of the same body. A small batch (the reference's 32 cubes) is a latency problem: one block per CU at most, and
// the register allocator is left alone (131 VGPRs). A large batch is an occupancy problem: a CU's LDS holds four cubes'
// blocks, which needs four waves per SIMD, i.e. at most 128 VGPRs -- three fewer, at the price of a few stack slots.
__global__ void __launch_bounds__(kLightBlock) compute_light_wave_kernel(const LightJob J) { compute_light_wave_body<false, AIC_LIGHT_CHUNK, false>(J); }
__global__ void __launch_bounds__(kLightBlock) compute_light_wave_kernel_ldsq(const LightJob J) { compute_light_wave_body<false, AIC_LIGHT_CHUNK, true>(J); }
__global__ void __launch_bounds__(kLightBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) compute_light_wave_kernel_dense(const LightJob J) {
    compute_light_wave_body<false, 1, false>(J);
}
__global__ void __launch_bounds__(kLightBlock) compute_light_session_kernel(const LightJob J) { compute_light_wave_body<true, AIC_LIGHT_CHUNK, false>(J); }

__global__ void scatter_light_kernel(uint32_t *light, const uint32_t *index, const uint32_t *texel, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) light[index[i]] = texel[i];
}

__global__ void __launch_bounds__(64) prepare_light_batch_kernel(const LightPrep P) {
    const uint32_t i = threadIdx.x;
    if (i < P.n_scatter) P.light[P.scatter_index[i]] = P.scatter_texel[i];
    if (i < P.n_cubes) P.cubes_out[i] = P.cubes[i];
    if (i < 9u) P.head[i] = 0u;
}

__global__ void probe_log2f_kernel(const float *x, float *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log2f_exact(x[i]);
}

}  // namespace

void launch_compute_light(const LightJob &job, hipStream_t stream) {
    if (!job.n) return;
    hipLaunchKernelGGL(compute_light_kernel, dim3((job.n + 63u) / 64u), dim3(64), 0, stream, job);
}

namespace {
constexpr uint32_t kLightLdsQueueBudget = 96u << 10;  // dynamic LDS a small-batch block may take for bitmaps + queue (one block per CU; 17.5 KB are static)
uint32_t light_wave_bitmap_words(const LightJob &job) { return ((4u * job.n_tree + 31u) / 32u) + ((2u * job.n_tree + 31u) / 32u) + ((job.n_tree + 31u) / 32u); }
uint32_t light_wave_ldsq_bytes(const LightJob &job) { return ((light_wave_bitmap_words(job) + 3u) & ~3u) * 4u + job.n_front * 16u; }
uint32_t light_wave_lds(const LightJob &job) {
    const uint32_t lds = light_wave_bitmap_words(job) * 4u;
    // more dynamic LDS than the default limit needs an opt-in, per device
    static uint32_t lds_allowed[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint32_t &allowed = lds_allowed[dev & 63];
    const uint32_t want = std::max(lds, light_wave_ldsq_bytes(job) <= kLightLdsQueueBudget ? light_wave_ldsq_bytes(job) : 0u);
    if (want > allowed) {
        (void)hipFuncSetAttribute((const void *)compute_light_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)compute_light_wave_kernel_ldsq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        (void)hipFuncSetAttribute((const void *)compute_light_wave_kernel_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)compute_light_session_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        allowed = want;
    }
    return lds;
}
}  // namespace

void launch_compute_light_session(const LightJob &job, uint32_t n_blocks, uint32_t threads, hipStream_t stream) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(compute_light_session_kernel, dim3(n_blocks), dim3(threads == 256u ? 256u : 64u), light_wave_lds(job), stream, job);
}

void launch_compute_light_waves(const LightJob &job, uint32_t n_waves, uint32_t threads, hipStream_t stream) {
    if (!job.n || !n_waves) return;
    const uint32_t lds = light_wave_lds(job);
    static const bool no_ldsq = std::getenv("AIC_LIGHT_GLOBAL_QUEUE") != nullptr;  // (measurement switch: the walk's queue in global memory, as until round 4)
    if (n_waves > 256u) hipLaunchKernelGGL(compute_light_wave_kernel_dense, dim3(n_waves), dim3(threads == 256u ? 256u : 64u), lds, stream, job);
    else if (!no_ldsq && light_wave_ldsq_bytes(job) <= kLightLdsQueueBudget)
        hipLaunchKernelGGL(compute_light_wave_kernel_ldsq, dim3(n_waves), dim3(threads == 256u ? 256u : 64u), light_wave_ldsq_bytes(job), stream, job);
    else hipLaunchKernelGGL(compute_light_wave_kernel, dim3(n_waves), dim3(threads == 256u ? 256u : 64u), lds, stream, job);
}

void launch_scatter_light(uint32_t *light, const uint32_t *index, const uint32_t *texel, uint32_t n, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(scatter_light_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, light, index, texel, n);
}

void launch_prepare_light_batch(const LightPrep &prep, hipStream_t stream) {
    hipLaunchKernelGGL(prepare_light_batch_kernel, dim3(1), dim3(64), 0, stream, prep);
}

void launch_probe_log2f(const float *x, float *out, uint32_t n, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(probe_log2f_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, x, out, n);
}

}  // namespace aic

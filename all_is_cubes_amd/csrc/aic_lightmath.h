// aic_lightmath.h -- light interpolation of the trace kernel's SHADE event, as plain C++ that compiles for the device
// (aic_trace.hip) AND for the host (tests build it with g++ and compare it with the oracle, bit for bit, on random inputs:
// tests/test_lightmath_host.py), so the arithmetic is checked without a GPU.
//
// Replaces SpaceRaytracer::get_interpolated_light (all-is-cubes-render/src/raytracer/sr.rs:248-359),
// BlockSky::light_outside (all-is-cubes/src/space/sky.rs:113-147) and
// PackedLight::value_with_ambient_occlusion (all-is-cubes/src/space/light/data.rs:145-158).
//
// Shape (CDNA4-first): one straight-line common path -- every sample cube inside the space, which is every surface that is
// not on the outermost layer of cubes -- that issues its eight texel loads together and contains no per-lane branch; one
// single out-of-line block for the samples that fall outside the space (BlockSky) or outside i32 (BlockSky::mean). The
// compiler turns per-lane `if`s into exec-mask scaffolding (s_and_saveexec / s_cbranch_execz / s_or + hazard s_nops) that is
// issued whether or not a lane takes the branch, so the common path has none.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AIC_HD __host__ __device__ __forceinline__
#else
#define AIC_HD static inline
#endif
// "does any lane of the wave ...": a wave-uniform condition on the device (work that NO lane needs is skipped without per-lane exec-mask scaffolding),
// the condition itself on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define AIC_LM_ANY(x) (__ballot(x) != 0ull)
#else
#define AIC_LM_ANY(x) (x)
#endif

namespace aic {

// The light volume of a layer as the interpolation sees it (taken from DevLayer by the kernel: all of it stays in SGPRs).
struct LightGridView {
    const uint32_t *light;   // PackedLight texels r | g<<8 | b<<16 | status<<24, Z-major
    // Scalars, and every user reads them into locals BEFORE anything conditional: the compiler turns `c ? G.a : G.b` (a phi
    // of two loads) into one load with a selected address, which pins the whole struct in scratch memory on the device
    int lo_x, lo_y, lo_z;
    int size_x, size_y, size_z;
    uint32_t sky_nx, sky_ny, sky_nz, sky_px, sky_py, sky_pz, sky_mean;  // BlockSky texels
};

// f64::rem_euclid(1.0): fmod(x, 1) == x - trunc(x) exactly (sign of x kept, like fmod)
AIC_HD double lm_rem_euclid1(double x) {
    double r = x - trunc(x);
    r = copysign(r, x);
    return r < 0.0 ? r + 1.0 : r;
}
AIC_HD double lm_coarsestep(double x) {  // surface.rs:509-514
    double f = floor(x * 4.0);
    if (f < 0.0) f = 0.0;
    if (f > 3.0) f = 3.0;
    return (f + 0.5) / 4.0;
}
AIC_HD double lm_smoothstep(double x) {  // surface.rs:516-520
    if (x < 0.0) x = 0.0;
    if (x > 1.0) x = 1.0;
    return 3. * (x * x) - 2. * (x * x * x);
}

// BlockSky::light_outside (sky.rs:113-147). Per axis the cube is inside [lo, lo + size), exactly one cube past either end,
// or further away; the reference's six Ordering values reduce to: all three inside -> UNINITIALIZED_AND_BLACK, two inside and
// one adjacent -> that face's sky light, anything else -> NO_RAYS. (c - lo) mod 2^32 decides all of it: a cube below lo
// cannot alias into [0, size] because hi = lo + size fits an i32.
AIC_HD uint32_t lm_light_outside(const LightGridView &G, int cx, int cy, int cz) {
    // every field is read into a local before anything conditional (see LightGridView)
    const int glx = G.lo_x, gly = G.lo_y, glz = G.lo_z;
    const uint32_t k_nx = G.sky_nx, k_ny = G.sky_ny, k_nz = G.sky_nz, k_px = G.sky_px, k_py = G.sky_py, k_pz = G.sky_pz;
    const uint32_t dx = (uint32_t)cx - (uint32_t)glx, dy = (uint32_t)cy - (uint32_t)gly, dz = (uint32_t)cz - (uint32_t)glz;
    const uint32_t sx = (uint32_t)G.size_x, sy = (uint32_t)G.size_y, sz = (uint32_t)G.size_z;
    const int i32_min = -2147483647 - 1;
    const bool inx = dx < sx, iny = dy < sy, inz = dz < sz;
    const bool lox = (dx == 0xffffffffu) & (glx != i32_min), loy = (dy == 0xffffffffu) & (gly != i32_min), loz = (dz == 0xffffffffu) & (glz != i32_min);
    const bool hix = dx == sx, hiy = dy == sy, hiz = dz == sz;
    const int n_in = (inx ? 1 : 0) + (iny ? 1 : 0) + (inz ? 1 : 0);
    const int n_adj = ((lox | hix) ? 1 : 0) + ((loy | hiy) ? 1 : 0) + ((loz | hiz) ? 1 : 0);
    if (n_in == 3) return 0u;                       // UNINITIALIZED_AND_BLACK
    if (!(n_in == 2 && n_adj == 1)) return 1u << 24;  // NO_RAYS (status 1)
    uint32_t t = k_pz;                    // (which face: at most one of the six is set)
    t = hiy ? k_py : t;
    t = hix ? k_px : t;
    t = loz ? k_nz : t;
    t = loy ? k_ny : t;
    t = lox ? k_nx : t;
    return t;
}

// PackedLight::value_with_ambient_occlusion (data.rs:145-158): rgb from the 256-entry table, weight by status
AIC_HD void lm_texel_value_ao(uint32_t t, const float *lut, float out[4]) {
    out[0] = lut[t & 255u];
    out[1] = lut[(t >> 8) & 255u];
    out[2] = lut[(t >> 16) & 255u];
    const uint32_t status = t >> 24;
    out[3] = status == 255u ? 1.0f : (status == 128u ? 0.25f : 0.0f);
}
AIC_HD void lm_mix4(const float a[4], const float b[4], float amount, float out[4]) {  // sr.rs:491-497
    for (int i = 0; i < 4; i++) out[i] = a[i] + (b[i] - a[i]) * amount;
}

#if defined(__HIP_DEVICE_COMPILE__)
AIC_HD int lm_cvt_floor_i32(double v) {  // (int)floor(v), saturating
    int r;
    const double f = floor(v);
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
#endif
// Cube::containing on one coordinate (cube.rs:97-119): is there an i32 cube, and which
AIC_HD bool lm_cube_coord(double v, int *out) {
    const bool ok = (v >= -2147483648.0) && (v < 2147483648.0);
    *out = ok ? (int)floor(v) : 0;
    return ok;
}

// SpaceRaytracer::get_interpolated_light (sr.rs:248-359), up to the final division: returns the mixed (r, g, b, weight);
// the caller divides rgb by max(weight, 0.1) (sr.rs:355-358). `n_texels` (may be null) counts the get_packed_light calls
// the reference makes (the kernel's algorithmic-byte counter).
//
// Same arithmetic as the reference, organised around what is actually distinct: the tangent frame of a face is two signed
// coordinate axes (Face::rotation_from_nz, face.rs:395-404), so
//   * dot(surface_point, frame_axis) is +-surface_point[axis] (the +-0 terms of the reference's three-term dot product cannot
//     change the value that `- 0.5` is applied to);
//   * the four sample offsets dir_1*{-.5,+.5} + dir_2*{-.5,+.5} are exactly +-0.5 on the two tangent axes and 0 on the normal
//     axis, so the 4 (x2 planes) sample cubes are built from 2 + 2 + 2 floor() values;
//   * the light-grid index of a texel is a sum of three per-axis contributions.
// Texel decode, the light-leak rule, the bilinear / trilinear mix4 chain are unchanged, operation for operation.
AIC_HD void lm_interpolated_light(const LightGridView &G, const float *lut, int cx, int cy, int cz, double spx, double spy, double spz,
                                  int face, int mode, float fin[4], uint32_t *n_texels) {
    const double eps = 0.5 / 256.0;
    const uint32_t *const texels = G.light;
    const int glx = G.lo_x, gly = G.lo_y, glz = G.lo_z;
    const uint32_t gsx = (uint32_t)G.size_x, gsy = (uint32_t)G.size_y, gsz = (uint32_t)G.size_z;
    const uint32_t sky_mean = G.sky_mean;
    // face -> (normal axis, tangent axes) and the signs; NX NY NZ PX PY PZ = 1..6, Within = 0 uses the IDENTITY frame and a
    // zero normal. Two-bit fields indexed by the face.
    const uint32_t f2 = (uint32_t)face * 2u;
    const uint32_t an = (0x2492u >> f2) & 3u;   // 2,0,1,2,0,1,2   for face 0..6
    const uint32_t a1 = (0x0924u >> f2) & 3u;   // 0,1,2,0,1,2,0
    const uint32_t a2 = 3u - an - a1;           // 1,2,0,1,2,0,1
    const double ns = face >= 4 ? 1.0 : (face == 0 ? 0.0 : -1.0);
    const double s1 = face == 4 ? -1.0 : 1.0;                     // PX: -Y
    const double s2 = (face == 5 || face == 6) ? -1.0 : 1.0;      // PY: -X, PZ: -Y
    const bool n0 = an == 0u, n1 = an == 1u, t10 = a1 == 0u, t11 = a1 == 1u, t20 = a2 == 0u, t21 = a2 == 1u;
    const double spn = n0 ? spx : (n1 ? spy : spz), sp1 = t10 ? spx : (t11 ? spy : spz), sp2 = t20 ? spx : (t21 ? spy : spz);

    double mix_1 = lm_rem_euclid1(s1 * sp1 - 0.5);
    double mix_2 = lm_rem_euclid1(s2 * sp2 - 0.5);
    // dir_1 / dir_2 along their axes; past the middle of the cube the interpolation runs towards the other neighbour
    const bool flip1 = mix_1 > 0.5, flip2 = mix_2 > 0.5;
    mix_1 = flip1 ? 1.0 - mix_1 : mix_1;
    mix_2 = flip2 ? 1.0 - mix_2 : mix_2;
    const double g1 = flip1 ? -s1 : s1, g2 = flip2 ? -s2 : s2;
    if (mode == 2) { mix_1 = lm_coarsestep(mix_1); mix_2 = lm_coarsestep(mix_2); }
    else if (mode == 4) { mix_1 = lm_smoothstep(mix_1); mix_2 = lm_smoothstep(mix_2); }
    const float m1 = (float)mix_1, m2 = (float)mix_2;

    // height of the surface inside its cube: face.dot(sp) - face.dot(cube centre) + 0.5
    const int cn_i = n0 ? cx : (n1 ? cy : cz);
    const double cn = (double)cn_i + 0.5;
    const double height_in_cube = (face == 0) ? 0.5 : ((ns * spn) - (ns * cn) + 0.5);
    const bool one_plane = height_in_cube > (1.0 - eps);

    // the six sample coordinates: two along each role (normal: front / same plane; tangents: near / far)
    const double pnf = spn + ns * (1.0 - eps), pns = spn + ns * eps;
    const double p1n = sp1 + g1 * -0.5, p1f = sp1 + g1 * 0.5, p2n = sp2 + g2 * -0.5, p2f = sp2 + g2 * 0.5;
    // per-role grid parameters
    const int lo_n = n0 ? glx : (n1 ? gly : glz), lo_1 = t10 ? glx : (t11 ? gly : glz), lo_2 = t20 ? glx : (t21 ? gly : glz);
    const uint32_t sz_n = n0 ? gsx : (n1 ? gsy : gsz), sz_1 = t10 ? gsx : (t11 ? gsy : gsz), sz_2 = t20 ? gsx : (t21 ? gsy : gsz);
    // BYTE strides: a texel's place is a 32-bit byte offset from the volume's base (aic_upload_space keeps a light volume within 4 GiB), which the
    // device's loads take as scalar base + 32-bit vector offset -- no 64-bit address arithmetic per texel (round 6: eight v_lshl_add_u64 per SHADE event)
    const uint32_t stride_x = gsy * gsz * 4u, stride_y = gsz * 4u;
    const uint32_t st_n = n0 ? stride_x : (n1 ? stride_y : 4u), st_1 = t10 ? stride_x : (t11 ? stride_y : 4u), st_2 = t20 ? stride_x : (t21 ? stride_y : 4u);

    // Common path: all three surface-point coordinates far inside i32 (so every sample has a cube) and every sample cube
    // inside the space. Anything else is patched texel by texel below.
    const bool far_from_i32_edge = (fabs(spx) < 2147483646.0) && (fabs(spy) < 2147483646.0) && (fabs(spz) < 2147483646.0);
    int vnf = 0, vns = 0, v1n = 0, v1f = 0, v2n = 0, v2f = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // The device converts for every lane -- v_cvt_i32_f64 saturates, and `all_inside` below holds far_from_i32_edge, so what a lane outside the common path
    // gets here is never used -- instead of six default moves, a saved exec mask and six conditional moves (round 6). (The instruction itself: in C++ the
    // conversion of a value that does not fit is undefined, which the optimiser may act on.)
    vnf = lm_cvt_floor_i32(pnf); vns = lm_cvt_floor_i32(pns);
    v1n = lm_cvt_floor_i32(p1n); v1f = lm_cvt_floor_i32(p1f);
    v2n = lm_cvt_floor_i32(p2n); v2f = lm_cvt_floor_i32(p2f);
#else
    if (far_from_i32_edge) {
        vnf = (int)floor(pnf); vns = (int)floor(pns);
        v1n = (int)floor(p1n); v1f = (int)floor(p1f);
        v2n = (int)floor(p2n); v2f = (int)floor(p2f);
    }
#endif
    const uint32_t dnf = (uint32_t)vnf - (uint32_t)lo_n, dns = (uint32_t)vns - (uint32_t)lo_n;
    const uint32_t d1n = (uint32_t)v1n - (uint32_t)lo_1, d1f = (uint32_t)v1f - (uint32_t)lo_1;
    const uint32_t d2n = (uint32_t)v2n - (uint32_t)lo_2, d2f = (uint32_t)v2f - (uint32_t)lo_2;
    const bool all_inside = far_from_i32_edge & (dnf < sz_n) & (dns < sz_n) & (d1n < sz_1) & (d1f < sz_1) & (d2n < sz_2) & (d2f < sz_2);
    // (off the common path the three strides are zero, hence every offset: always a valid address -- three selects instead of one per texel)
    const uint32_t zt_n = all_inside ? st_n : 0u, zt_1 = all_inside ? st_1 : 0u, zt_2 = all_inside ? st_2 : 0u;
    const uint32_t bnf = dnf * zt_n, bns = dns * zt_n, b1n = d1n * zt_1, b1f = d1f * zt_1, b2n = d2n * zt_2, b2f = d2f * zt_2;
    const uint32_t q00 = b1n + b2n, q01 = b1n + b2f, q10 = b1f + b2n, q11 = b1f + b2f;
    const char *const tex_bytes = reinterpret_cast<const char *>(texels);
#define AIC_LM_TEXEL(off) (*reinterpret_cast<const uint32_t *>(tex_bytes + (uint32_t)(off)))
    // eight loads issued together
    uint32_t tf00 = AIC_LM_TEXEL(bnf + q00);  // front plane: near12, near1far2, near2far1, far12
    uint32_t tf01 = AIC_LM_TEXEL(bnf + q01);
    uint32_t tf10 = AIC_LM_TEXEL(bnf + q10);
    uint32_t tf11 = AIC_LM_TEXEL(bnf + q11);
    // same plane: a full-height surface -- every face of a whole-cube block, i.e. most surfaces of most all-is-cubes scenes -- interpolates in the front plane only
    // (sr.rs:339-354: `one_plane`), and the reference never fetches these four texels for it; a wave in which no lane needs them does not either (round 6)
    uint32_t ts00 = 0u, ts01 = 0u, ts10 = 0u, ts11 = 0u;
    if (AIC_LM_ANY(!one_plane)) {
        ts00 = AIC_LM_TEXEL(bns + q00);
        ts01 = AIC_LM_TEXEL(bns + q01);
        ts10 = AIC_LM_TEXEL(bns + q10);
        ts11 = AIC_LM_TEXEL(bns + q11);
    }
#undef AIC_LM_TEXEL
    uint32_t n_calls = one_plane ? 4u : 8u;
    if (!all_inside) {
        // Rare: a sample outside the space (BlockSky::light_outside on the reassembled cube) or without an i32 cube
        // (BlockSky::mean, sr.rs:307-311). One copy of the logic, a loop over the eight samples.
        n_calls = 0u;
        int wnf, wns, w1n, w1f, w2n, w2f;
        const bool knf = lm_cube_coord(pnf, &wnf), kns = lm_cube_coord(pns, &wns);
        const bool k1n = lm_cube_coord(p1n, &w1n), k1f = lm_cube_coord(p1f, &w1f);
        const bool k2n = lm_cube_coord(p2n, &w2n), k2f = lm_cube_coord(p2f, &w2f);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int k = 0; k < 8; k++) {
            const bool same_plane = (k & 4) != 0, far1 = (k & 2) != 0, far2 = (k & 1) != 0;  // plane, tangent-1 near/far, tangent-2 near/far
            const bool has_cube = (same_plane ? kns : knf) & (far1 ? k1f : k1n) & (far2 ? k2f : k2n);
            const int un = same_plane ? wns : wnf, u1 = far1 ? w1f : w1n, u2 = far2 ? w2f : w2n;
            const int c0 = n0 ? un : (t10 ? u1 : u2), c1 = n1 ? un : (t11 ? u1 : u2), c2 = (an == 2u) ? un : ((a1 == 2u) ? u1 : u2);
            const uint32_t ex = (uint32_t)c0 - (uint32_t)glx, ey = (uint32_t)c1 - (uint32_t)gly, ez = (uint32_t)c2 - (uint32_t)glz;
            const bool inside = has_cube & (ex < gsx) & (ey < gsy) & (ez < gsz);
            uint32_t t = texels[inside ? (ex * gsy + ey) * gsz + ez : 0u];
            const uint32_t t_out = lm_light_outside(G, c0, c1, c2);
            if (!inside) t = has_cube ? t_out : sky_mean;
            if (has_cube && !(one_plane && same_plane)) n_calls++;
            tf00 = k == 0 ? t : tf00; tf01 = k == 1 ? t : tf01; tf10 = k == 2 ? t : tf10; tf11 = k == 3 ? t : tf11;
            ts00 = k == 4 ? t : ts00; ts01 = k == 5 ? t : ts01; ts10 = k == 6 ? t : ts10; ts11 = k == 7 ? t : ts11;
        }
    }
    if (n_texels) *n_texels += n_calls;

    // one plane of four texels: light-leak fix (both side texels invalid => far corner := near corner), decode, bilinear mix
    float front[4];
    {
        if ((tf01 >> 24) != 255u && (tf10 >> 24) != 255u) tf11 = tf00;
        float a[4], b[4], c[4], d[4], ab[4], cd[4];
        lm_texel_value_ao(tf00, lut, a); lm_texel_value_ao(tf01, lut, b); lm_texel_value_ao(tf10, lut, c); lm_texel_value_ao(tf11, lut, d);
        lm_mix4(a, b, m2, ab); lm_mix4(c, d, m2, cd); lm_mix4(ab, cd, m1, front);
    }
    for (int i = 0; i < 4; i++) fin[i] = front[i];
    if (!one_plane) {  // a full-height surface (every face of a whole-cube block) interpolates in one plane only
        if ((ts01 >> 24) != 255u && (ts10 >> 24) != 255u) ts11 = ts00;
        float a[4], b[4], c[4], d[4], ab[4], cd[4], same[4];
        lm_texel_value_ao(ts00, lut, a); lm_texel_value_ao(ts01, lut, b); lm_texel_value_ao(ts10, lut, c); lm_texel_value_ao(ts11, lut, d);
        lm_mix4(a, b, m2, ab); lm_mix4(c, d, m2, cd); lm_mix4(ab, cd, m1, same);
        lm_mix4(same, front, (float)height_in_cube, fin);
    }
}

}  // namespace aic

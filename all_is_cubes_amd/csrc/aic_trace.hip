// aic_trace.hip -- hand-written CDNA4 (gfx950) kernels of the voxel raytracer.
//
// Replaces, behind the C ABI of include/aic_hip.h, the reference's per-image hot loop:
//   trace_scene_to_image_impl / RtScene::trace_patch / trace_ray_through_layers
//       (all-is-cubes-render/src/raytracer/renderer.rs:424-478, 516-556)
//   SpaceRaytracer::trace_ray_impl + TracingState (raytracer/sr.rs:135-238, 595-769)
//   SurfaceIter / VoxelSurfaceIter / DepthIter (raytracer/surface.rs:251-491)
//   Raycaster (all-is-cubes-base/src/raycast.rs:63-832)
//   ColorBuf / apply_transmittance (all-is-cubes/src/raytracer_components.rs:20-258)
//   Camera::project_ndc_into_world / post_process_color, Rgba::to_srgb8.
//
// Design (MI355X-first, not a translation of the reference's iterator stack):
//  * persistent waves: the grid is what is resident at the kernel's occupancy (2 waves per SIMD);
//    each wave pulls 8x8-pixel tiles (one wave-full) from a global counter, macro tile by macro
//    tile, costliest macro tiles of the previous frame first, and refills idle lanes one by one.
//  * every lane is a state machine. ONE predicated, straight-line Amanatides-Woo step serves both
//    DDA levels (cube grid and a block's voxels share the lookup pool, the registers and the
//    code); the expensive, divergent work -- shading a surface, entering a block, finishing /
//    starting a ray -- is parked per lane as an event and run kind by kind when enough lanes wait.
//  * DDA arithmetic is f64 in the reference's exact operation order (bit-exact hit
//    cubes/voxels/faces/t); built with -ffp-contract=off. Colour arithmetic is f32 in the
//    reference's order; powf follows the C library's algorithm, exp is evaluated in f64.
//  * no MFMA: the path is branchy integer/f64 traversal and gather loads, not a contraction.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstddef>
#include <cstdlib>

#include "aic_device.h"
#include "aic_lightmath.h"

namespace aic {

#define AIC_DEV __device__ __forceinline__

constexpr int FACE_WITHIN = 0;
constexpr int I32_MIN_ = (-2147483647 - 1);
constexpr int I32_MAX_ = 2147483647;

// first_last states (raycast.rs:153-165)
constexpr uint32_t FL_BEGINNING = 0, FL_INBOUNDS = 1, FL_ENDED = 2;

// ---------------------------------------------------------------------------------------
// f64 helpers with the reference's semantics

AIC_DEV int signum_101(double x) {  // raycast.rs:782-788
    if (x == 0.0) return 0;
    if (x != x) return 0;
    return (__double2hiint(x) < 0) ? -1 : 1;
}

// f64::rem_euclid(1.0): fmod(x,1) == x - trunc(x) exactly (sign of x kept, like fmod)
AIC_DEV double rem_euclid1(double x) {
    double r = x - trunc(x);
    r = copysign(r, x);
    return r < 0.0 ? r + 1.0 : r;
}

// ---- f64 divisions that cost less than the generic sequence, with the generic sequence's bits ----
// The compiler's a / b is v_div_scale x2, v_rcp_f64 (quarter rate), four fused multiply-adds that refine the reciprocal, a multiply, two more
// multiply-adds, v_div_fmas, v_div_fixup: 11 instructions, IEEE-correct for every input. Two cheaper forms, each used only where its precondition holds
// for the lane (checked on the operands' exponent fields) and replaced by the generic quotient, under the lanes' exec mask, where it does not:
//  * div_known_recip: the divisor's correctly rounded reciprocal y = RN(1 / b) is at hand (a ray's t_delta = 1 / |direction|, raycast.rs:766). Then
//    q0 = a * y is within 1.5 ulp of a / b, q1 = q0 + (a - b q0) y is a faithful quotient, and one more residual step q2 = q1 + (a - b q1) y is a / b
//    correctly rounded (Markstein's theorem: a faithful q, the exact residual r = a - b q, and y within half an ulp of 1 / b give RN(q + r y) = RN(a / b),
//    absent overflow / underflow) -- five multiply-adds, no reciprocal instruction (lvl_init). tests: the probe's step tables and 400 random rays (t bit-exact).
//  * three quotients by one divisor (the unprojection's x / w, y / w, z / w) share the reciprocal's refinement: with operands whose exponents are far from
//    the ends of the range v_div_scale scales nothing, v_div_fmas is a plain fused multiply-add and v_div_fixup passes the quotient through, so
//    the shared form performs the generic sequence's own operations on the same values.
// A block behind a wave-uniform branch that must STAY a branch: arithmetic without side effects is otherwise speculated -- the compiler computes the rare
// path for every wave and selects (seen with the generic division below: 11 instructions per quotient, executed always). An empty volatile asm
// statement cannot be speculated.
#define AIC_RARE_PATH() asm volatile("" ::: "memory")
// Exponent window of an operand, on the high dword: biased exponent in [768, 1280), i.e. 2^-255 <= |v| < 2^257; and the wider [512, 1536).
AIC_DEV bool f64_exp_in_768_1280(double v) { return (((uint32_t)__double2hiint(v) << 1) - (768u << 21)) < (512u << 21); }
AIC_DEV bool f64_exp_in_512_1536(double v) { return (((uint32_t)__double2hiint(v) << 1) - (512u << 21)) < (1024u << 21); }
AIC_DEV double div_known_recip(double a, double b, double y) {  // a / b for normal b > 0, y = RN(1 / b); a, b and a / b far from overflow and underflow; a is not -0
    const double q0 = a * y;
    const double r0 = fma(-b, q0, a);
    const double q1 = fma(r0, y, q0);
    const double r1 = fma(-b, q1, a);
    return fma(r1, y, q1);
}
// raycast.rs:797-819, split around its division: the dividend 1 - s.rem_euclid(1) (s and ds negated together for ds < 0: |ds| is the divisor either way) ...
AIC_DEV double scale_step_dividend(double s, double ds) { return 1.0 - rem_euclid1(ds < 0.0 ? -s : s); }
// ... and what becomes of the quotient q = dividend / |ds|
AIC_DEV double scale_step_result(double q, double s, double ds) {
    return (ds == 0.0 && !(s != s)) ? __longlong_as_double(0x7ff0000000000000LL) : q;
}

// cube.rs:97-119
AIC_DEV bool cube_containing(const double p[3], int out[3]) {
    const double MIN_INCLUSIVE = -2147483648.0;
    const double MAX_EXCLUSIVE = 2147483648.0;
    bool ok = (MIN_INCLUSIVE <= p[0]) & (MIN_INCLUSIVE <= p[1]) & (MIN_INCLUSIVE <= p[2]) & (p[0] < MAX_EXCLUSIVE) &
              (p[1] < MAX_EXCLUSIVE) & (p[2] < MAX_EXCLUSIVE);
    if (ok) {
        out[0] = (int)floor(p[0]);
        out[1] = (int)floor(p[1]);
        out[2] = (int)floor(p[2]);
    }
    return ok;
}

// Per-ray constants: Parameters::new (raycast.rs:749-771) minus the origin. Kept as scalars
// (never indexed dynamically) so they live in VGPRs.
struct RayDir {
    double dx, dy, dz;     // direction (zeroed if any |component| is not < 1e100)
    double tdx, tdy, tdz;  // t_delta = 1/|d|
    int sx, sy, sz;        // step = signum_101(d)
    bool fast;             // every component is zero or has its exponent in [768, 1280): divisions by it may use t_delta (div_by_dir)
};
AIC_DEV bool raydir_fast(double dx, double dy, double dz) {
    const int fx = f64_exp_in_768_1280(dx) | (dx == 0.0), fy = f64_exp_in_768_1280(dy) | (dy == 0.0), fz = f64_exp_in_768_1280(dz) | (dz == 0.0);
    return (fx & fy & fz) != 0;
}

AIC_DEV RayDir raydir_init(double dx, double dy, double dz) {
    RayDir r;
    const bool all_small = (fabs(dx) < 1e100) && (fabs(dy) < 1e100) && (fabs(dz) < 1e100);
    r.dx = all_small ? dx : 0.0;
    r.dy = all_small ? dy : 0.0;
    r.dz = all_small ? dz : 0.0;
    r.sx = signum_101(r.dx); r.sy = signum_101(r.dy); r.sz = signum_101(r.dz);
    r.tdx = 1.0 / fabs(r.dx); r.tdy = 1.0 / fabs(r.dy); r.tdz = 1.0 / fabs(r.dz);
    r.fast = raydir_fast(r.dx, r.dy, r.dz);
    return r;
}

// State of one DDA level (raycast.rs:99-121 State + FirstLast), with the step deferred: the
// reference emits `current()` and then advances; here the advance is performed at the start
// of the following `next`, which is observationally identical and lets `c*` double as the
// emitted cube.
struct Lvl {
    double tx, ty, tz;  // t_max
    double last_t;
    int cx, cy, cz;
    uint32_t st;        // bits 0-1 first_last | 2-4 last_face | 5-6 pick | 7 need_step | 8 include_exit
};
// While INBOUNDS: the coordinate value that means "left the bounds", per axis.
struct Lim {
    int x, y, z;
};
AIC_DEV uint32_t lvl_fl(const Lvl &s) { return s.st & 3u; }
AIC_DEV int lvl_face(const Lvl &s) { return (int)((s.st >> 2) & 7u); }

AIC_DEV int pick_axis(double tx, double ty, double tz) {  // raycast.rs:584-596
    if (tx < ty) return (tx < tz) ? 0 : 2;
    return (ty < tz) ? 1 : 2;
}

// Raycaster::new(origin, dir) [.within(lo,hi, include_exit)]  (raycast.rs:196-230, 513-545, 632-704)
struct LvlLim {
    Lvl s;
    Lim lim;
};
// Written without early exits (round 6): every lane computes everything and `valid` decides at the end -- a level that is State::EMPTY comes back as
// FL_ENDED with unspecified t_max / cube (nothing reads them: Raycaster::next returns None at once). With the exits, each one cost a saved exec mask, a
// branch and a dozen moves of default values on the path of every lane that did not take it.
AIC_DEV bool cube_containing_flat(double x, double y, double z, int out[3]) {  // cube.rs:97-119; `out` is unspecified when there is no cube
    const double MIN_INCLUSIVE = -2147483648.0;
    const double MAX_EXCLUSIVE = 2147483648.0;
    const int ok = (int)(MIN_INCLUSIVE <= x) & (int)(MIN_INCLUSIVE <= y) & (int)(MIN_INCLUSIVE <= z) & (int)(x < MAX_EXCLUSIVE) & (int)(y < MAX_EXCLUSIVE) & (int)(z < MAX_EXCLUSIVE);
    // (v_cvt_i32_f64 itself, which saturates: the C++ conversion of a value that does not fit is undefined, and the optimiser may act on that)
    const double fx = floor(x), fy = floor(y), fz = floor(z);
    asm("v_cvt_i32_f64 %0, %1" : "=v"(out[0]) : "v"(fx));
    asm("v_cvt_i32_f64 %0, %1" : "=v"(out[1]) : "v"(fy));
    asm("v_cvt_i32_f64 %0, %1" : "=v"(out[2]) : "v"(fz));
    return ok != 0;
}
AIC_DEV LvlLim lvl_init(double ox, double oy, double oz, const RayDir rd, bool bounded, int lox, int loy,
                        int loz, int hix, int hiy, int hiz, bool include_exit, double half_over_len) {
    LvlLim out;
    Lvl &s = out.s;
    Lim &lim = out.lim;
    int cube_o[3];
    int valid = cube_containing_flat(ox, oy, oz, cube_o);
    // MAXIMUM_BOUNDS.contains_cube (raycast.rs:485-499, 521-523); else State::EMPTY: produces nothing
    // (c in [MIN + 1, MAX - 2]  <=>  (unsigned)(c - (MIN + 1)) < 2^32 - 3)
    valid &= (int)((uint32_t)cube_o[0] - 0x80000001u < 0xfffffffdu) & (int)((uint32_t)cube_o[1] - 0x80000001u < 0xfffffffdu) & (int)((uint32_t)cube_o[2] - 0x80000001u < 0xfffffffdu);
    // bounds = MAXIMUM_BOUNDS ∩ given (empty => ORIGIN_EMPTY, which contains no cube)
    if (bounded) {
        lox = max(lox, I32_MIN_ + 1); loy = max(loy, I32_MIN_ + 1); loz = max(loz, I32_MIN_ + 1);
        hix = min(hix, I32_MAX_ - 1); hiy = min(hiy, I32_MAX_ - 1); hiz = min(hiz, I32_MAX_ - 1);
    } else {
        lox = loy = loz = I32_MIN_ + 1;
        hix = hiy = hiz = I32_MAX_ - 1;
    }
    valid &= (int)(hix > lox) & (int)(hiy > loy) & (int)(hiz > loz);
    // fast_forward (raycast.rs:632-704): plane_origin takes the upper bound on axes the ray descends, else the lower bound; one ray-plane
    // intersection per moving axis. If the largest t is positive the ray starts again half a cube short of it (`t_start`), else where it is
    // (t_start = +0: `ff` is the origin itself, and adding +0 to a t_max -- a quotient that is positive, +0 or infinite -- changes nothing).
    // One copy of the t_max arithmetic serves both (round 6; it was written twice, each copy behind its own per-lane branch).
    double t_start = 0.0;
    double ffx = ox, ffy = oy, ffz = oz;
    if (bounded) {
        const double pox = (double)((rd.sx < 0) ? hix : lox);
        const double poy = (double)((rd.sy < 0) ? hiy : loy);
        const double poz = (double)((rd.sz < 0) ? hiz : loz);
        const double relx = pox - ox, rely = poy - oy, relz = poz - oz;
        // ray_plane_intersection (raycast.rs:821-832) with an axis-aligned unit normal n = +-1:
        // (rel.n)/(dir.n) == rel_a / dir_a exactly (the +-1 factors and the +-0 terms cancel for the
        // finite values that reach this point). rel_a / dir_a = +-(rel_a / |dir_a|), the quotient by the reciprocal at hand (div_known_recip) for lanes
        // whose rel_a is neither zero nor tiny (the origin is inside i32, so it is not huge); computed for every lane, used for the moving axes.
        const int okx = f64_exp_in_512_1536(relx) | (rd.sx == 0), oky = f64_exp_in_512_1536(rely) | (rd.sy == 0), okz = f64_exp_in_512_1536(relz) | (rd.sz == 0);
        const bool ff_fast = ((int)rd.fast & okx & oky & okz) != 0;
        double qx = div_known_recip(relx, fabs(rd.dx), rd.tdx), qy = div_known_recip(rely, fabs(rd.dy), rd.tdy), qz = div_known_recip(relz, fabs(rd.dz), rd.tdz);
        if (__builtin_amdgcn_ballot_w64(!ff_fast) != 0ull) {  // (never, in an ordinary frame)
            AIC_RARE_PATH();
            if (!ff_fast) { qx = relx / fabs(rd.dx); qy = rely / fabs(rd.dy); qz = relz / fabs(rd.dz); }
        }
        double max_t = 0.0;
        max_t = rd.sx != 0 ? fmax(max_t, rd.sx < 0 ? -qx : qx) : max_t;
        max_t = rd.sy != 0 ? fmax(max_t, rd.sy < 0 ? -qy : qy) : max_t;
        max_t = rd.sz != 0 ? fmax(max_t, rd.sz < 0 ? -qz : qz) : max_t;
        const bool go = max_t > 0.0;  // last_t_distance == 0 at this point
        // 0.5 / direction.length() (raycast.rs:669) is a per-ray constant, computed once by the caller
        double ts = max_t - half_over_len;
        ts = isfinite(ts) ? ts : max_t;
        t_start = go ? ts : 0.0;
        ffx = go ? ox + rd.dx * ts : ox; ffy = go ? oy + rd.dy * ts : oy; ffz = go ? oz + rd.dz * ts : oz;
    }
    // the cube of the (fast-forwarded) origin; a fast-forwarded origin without one makes the level State::EMPTY
    int cube[3];
    valid &= (int)cube_containing_flat(ffx, ffy, ffz, cube);
    {
        // scale_to_integer_step on each axis (raycast.rs:797-819). The dividends are in [2^-53, 1] or +0: with a direction in the window (RayDir::fast)
        // div_known_recip's precondition holds
        const double ax = scale_step_dividend(ffx, rd.dx), ay = scale_step_dividend(ffy, rd.dy), az = scale_step_dividend(ffz, rd.dz);
        double qx = div_known_recip(ax, fabs(rd.dx), rd.tdx), qy = div_known_recip(ay, fabs(rd.dy), rd.tdy), qz = div_known_recip(az, fabs(rd.dz), rd.tdz);
        if (__builtin_amdgcn_ballot_w64(!rd.fast) != 0ull) {
            AIC_RARE_PATH();
            if (!rd.fast) { qx = ax / fabs(rd.dx); qy = ay / fabs(rd.dy); qz = az / fabs(rd.dz); }
        }
        s.tx = scale_step_result(qx, ffx, rd.dx) + t_start;
        s.ty = scale_step_result(qy, ffy, rd.dy) + t_start;
        s.tz = scale_step_result(qz, ffz, rd.dz) + t_start;
    }
    s.last_t = t_start;
    s.cx = cube[0]; s.cy = cube[1]; s.cz = cube[2];
    // exit coordinate once in bounds: moving up leaves at hi, moving down leaves at lo-1
    lim.x = rd.sx > 0 ? hix : lox - 1;
    lim.y = rd.sy > 0 ? hiy : loy - 1;
    lim.z = rd.sz > 0 ? hiz : loz - 1;
    s.st = valid ? (FL_BEGINNING | ((uint32_t)FACE_WITHIN << 2) | (include_exit ? 256u : 0u)) : FL_ENDED;
    return out;
}

// The deferred State::step (raycast.rs:577-626) along the axis recorded in `pick`.
AIC_DEV Lvl lvl_do_step(Lvl s, const RayDir rd) {
    const uint32_t axis = (s.st >> 5) & 3u;
    uint32_t face;
    if (axis == 0) {
        s.last_t = s.tx; s.tx += rd.tdx; s.cx += rd.sx; face = rd.sx > 0 ? 1u : 4u;
    } else if (axis == 1) {
        s.last_t = s.ty; s.ty += rd.tdy; s.cy += rd.sy; face = rd.sy > 0 ? 2u : 5u;
    } else {
        s.last_t = s.tz; s.tz += rd.tdz; s.cz += rd.sz; face = rd.sz > 0 ? 3u : 6u;
    }
    s.st = (s.st & ~(7u << 2) & ~128u) | (face << 2);  // FACE_TABLE; clears need_step
    return s;
}

// Raycaster::next (raycast.rs:239-284). lo*/hi* are only consulted before the ray has entered
// the bounds. Returns true if a step was produced: {c*, lvl_face, last_t, t*}; *is_exit tells
// whether it is the include_exit step (the only produced step whose cube is out of bounds).
struct NextResult {
    Lvl s;
    bool got, is_exit;
};
AIC_DEV NextResult lvl_next(Lvl s, const Lim lim, const RayDir rd, int lox, int loy, int loz, int hix, int hiy, int hiz) {
    NextResult R;
    R.got = false;
    R.is_exit = false;
    for (;;) {
        const uint32_t fl = lvl_fl(s);
        if (fl == FL_ENDED) { R.s = s; return R; }
        const bool stepped = (s.st & 128u) != 0;
        const uint32_t stepped_axis = (s.st >> 5) & 3u;
        if (stepped) s = lvl_do_step(s, rd);
        bool oob_enter = false, oob_exit = false;
        if (fl == FL_INBOUNDS) {
            // only the axis just stepped can have left; it can never be "not yet entered"
            const int c = stepped_axis == 0 ? s.cx : (stepped_axis == 1 ? s.cy : s.cz);
            const int l = stepped_axis == 0 ? lim.x : (stepped_axis == 1 ? lim.y : lim.z);
            oob_exit = stepped && (c == l);
        } else {
            // is_out_of_bounds_ahead (raycast.rs:711-728)
            {
                const bool low = s.cx < lox, high = s.cx >= hix;
                oob_enter |= rd.sx == 0 ? (low | high) : (rd.sx < 0 ? high : low);
                oob_exit |= rd.sx == 0 ? (low | high) : (rd.sx < 0 ? low : high);
            }
            {
                const bool low = s.cy < loy, high = s.cy >= hiy;
                oob_enter |= rd.sy == 0 ? (low | high) : (rd.sy < 0 ? high : low);
                oob_exit |= rd.sy == 0 ? (low | high) : (rd.sy < 0 ? low : high);
            }
            {
                const bool low = s.cz < loz, high = s.cz >= hiz;
                oob_enter |= rd.sz == 0 ? (low | high) : (rd.sz < 0 ? high : low);
                oob_exit |= rd.sz == 0 ? (low | high) : (rd.sz < 0 ? low : high);
            }
        }
        if (!oob_enter && !oob_exit) {
            const int pick = pick_axis(s.tx, s.ty, s.tz);
            const double tp = pick == 0 ? s.tx : (pick == 1 ? s.ty : s.tz);
            // valid_for_stepping (raycast.rs:563-570): with NaN-free t_max (guaranteed for a
            // non-EMPTY state) it is exactly "the smallest t_max is finite".
            if (!isfinite(tp)) {
                s.st = (s.st & ~3u) | FL_ENDED;
                R.got = lvl_face(s) == FACE_WITHIN;
                R.s = s;
                return R;
            }
            s.st = (s.st & ~3u & ~(3u << 5)) | FL_INBOUNDS | ((uint32_t)pick << 5) | 128u;
            R.got = true;
            R.s = s;
            return R;
        } else if (fl == FL_BEGINNING && oob_enter && !oob_exit) {
            const int pick = pick_axis(s.tx, s.ty, s.tz);
            const double tp = pick == 0 ? s.tx : (pick == 1 ? s.ty : s.tz);
            if (!isfinite(tp)) {
                s.st = (s.st & ~3u) | FL_ENDED;
                R.s = s;
                return R;
            }
            const int c = pick == 0 ? s.cx : (pick == 1 ? s.cy : s.cz);
            const int st = pick == 0 ? rd.sx : (pick == 1 ? rd.sy : rd.sz);
            if ((st > 0 && c == I32_MAX_) || (st < 0 && c == I32_MIN_)) {  // checked_add failed
                s.st = (s.st & ~3u) | FL_ENDED;
                R.s = s;
                return R;
            }
            s.st = (s.st & ~(3u << 5)) | ((uint32_t)pick << 5) | 128u;
            continue;
        } else if (fl == FL_INBOUNDS && !oob_enter && oob_exit) {
            s.st = (s.st & ~3u) | FL_ENDED;
            if (s.st & 256u) {
                R.is_exit = true;
                R.got = true;
            }
            R.s = s;
            return R;
        } else {
            s.st = (s.st & ~3u) | FL_ENDED;
            R.s = s;
            return R;
        }
    }
}

// RaycastStep::intersection_point (raycast.rs:409-439) for the step currently held in `s`.
//
// Same arithmetic as the reference, written without per-axis control flow. For an axis the ray moves along
// and that is not the face just crossed, the reference adds  1 - clamp((t_max - t) * d)  going up and
// clamp(-((t_max - t) * d))  going down; -(x * d) == x * (-d) exactly, so both clamp the one product
// (t_max - t) * |d|. That product is never NaN (t is finite, d finite and non-zero on this path), hence
// f64::clamp(0, 1) == min(max(c, 0), 1); its only other freedom, the sign of a zero, cannot reach the result
// (1 - +-0 == 1, and cube + +-0 == cube because an integer-valued cube coordinate is never -0).
AIC_DEV double ip_axis(bool is_face_axis, bool within, int cube, double o, double d, double t_max, double last_t) {
    const double cc = (double)cube;
    const bool neg = d < 0.0;                        // signum_101(d) < 0
    double c = (t_max - last_t) * fabs(d);
    c = fmin(fmax(c, 0.0), 1.0);
    const double moved = cc + (neg ? c : 1.0 - c);   // normal cube face hit
    const double plane = cc + (neg ? 1.0 : 0.0);     // the plane just crossed
    double v = is_face_axis ? plane : ((d == 0.0) ? o : moved);   // signum_101(d) == 0: the ray does not move from the origin
    return within ? o : v;
}
AIC_DEV void intersection_point(const Lvl s, double ox, double oy, double oz, double dx, double dy, double dz, double out[3]) {
    const int face = lvl_face(s);
    const bool within = face == FACE_WITHIN;
    const int face_axis = face > 3 ? face - 4 : face - 1;  // Face::axis(): NX NY NZ PX PY PZ = 1..6
    out[0] = ip_axis(face_axis == 0, within, s.cx, ox, dx, s.tx, s.last_t);
    out[1] = ip_axis(face_axis == 1, within, s.cy, oy, dy, s.ty, s.last_t);
    out[2] = ip_axis(face_axis == 2, within, s.cz, oz, dz, s.tz, s.last_t);
}

// ---------------------------------------------------------------------------------------
// colour helpers (f32, reference operation order)

AIC_DEV float ps_clamped(float v) { return v > 0.f ? v : 0.f; }          // restricted_number.rs:240-248
AIC_DEV float zo_clamped(float v) {                                        // restricted_number.rs:315-326
    if (v > 0.f && v <= 1.f) return v;
    if (v <= 0.f) return 0.f;
    return 1.f;
}
// PositiveSign::mul: 0 * inf => 0. Both factors are PositiveSign values (not NaN, sign bit clear: what the reference's type holds and aic_upload_* / the
// kernel's own clamps guarantee), so the product is >= +0 or the NaN of 0 * inf, and max(product, 0) -- one instruction: it returns the operand that
// is a number -- is the reference's "NaN becomes zero" (a compare and a select until round 6; fifteen of them in a SHADE event).
AIC_DEV float ps_mul(float a, float b) { return fmaxf(a * b, 0.0f); }

// f32::powf as the reference's libm computes it on x86-64 Linux. Rust's `f32::powf` is the C library's
// powf; glibc's (sysdeps/ieee754/flt-32/e_powf.c, from ARM's optimized-routines; not under
// /root/reference, restated from the published algorithm) is: log2(x) by a 16-entry table and a
// degree-4 polynomial, y*log2(x), exp2 by a 32-entry table and a cubic, all in f64, rounded to f32
// once. Table and coefficient values are the published __powf_log2_data / __exp2f_data. The
// multiply-adds are fused, as in the FMA build glibc selects on every current x86-64 CPU.
// Domain: 0 < x < 1 normal, y > 0 finite (everything apply_transmittance feeds it); the caller
// handles the rest of what can reach it (x == 0, x == 1, y == 0, y == +inf) itself. ~40 instructions instead of ~270; pinned against the host's
// powf on a million inputs (tests/test_gpu_encode.py).
__device__ const double kPowLog2Tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
__device__ const unsigned long long kPowExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
// s_pow: [0,32) the log2 table as (invc, logc) pairs, [32,64) the exp2 table bit patterns
AIC_DEV void pow_tables_to_lds(double *s_pow, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < 64u; i += nthreads)
        s_pow[i] = i < 32u ? kPowLog2Tab[i >> 1][i & 1u] : __longlong_as_double((long long)kPowExp2Tab[i - 32u]);
}
AIC_DEV bool powf_table_domain(float x, float y) {  // 0 < x < 1 normal; y > 0 finite
    const uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
    return ix >= 0x00800000u && ix < 0x3f800000u && iy > 0u && iy < 0x7f800000u;
}
// A 64-bit literal that is materialised where it is used (two s_mov). Left to itself the compiler hoists such constants out of
// the persistent loop into VGPR pairs, runs out of registers, spills them to scratch at kernel start (every lane of every wave
// storing the same 8 bytes: most of round 2's 46 MB of WRITE_SIZE per frame) and reloads them from memory in every SHADE event.
AIC_DEV double KC(double v) { asm volatile("" : "+s"(v)); return v; }
// population count of a wave mask as a 32-bit scalar (the compiler widens __popcll's result and then compares it on the VALU)
AIC_DEV uint32_t wave_popc(unsigned long long m) {
    uint32_t n;
    asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
    return n;
}
AIC_DEV float KF(float v) { asm volatile("" : "+v"(v)); return v; }
AIC_DEV float powf_table(float x, float y, const double *s_pow) {
    const uint32_t ix = __float_as_uint(x);
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int)top >> 23;
    const double invc = s_pow[2u * i], logc = s_pow[2u * i + 1u];
    const double z = (double)__uint_as_float(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = fma(KC(0x1.27616c9496e0bp-2), r, KC(-0x1.71969a075c67ap-2));
    const double pp = fma(KC(0x1.ec70a6ca7baddp-2), r, KC(-0x1.7154748bef6c8p-1));
    const double r4 = r2 * r2;
    double q = fma(KC(0x1.71547652ab82bp+0), r, y0);
    q = fma(pp, r2, q);
    yy = fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    // |y*log2(x)| >= 126: x < 1 and y > 0 make it negative -- underflow to 0 at <= -150, else the
    // general path rounds into the subnormals by itself
    if (ylogx <= -150.0) return 0.0f;
    // exp2_inline
    double kd = ylogx + KC(0x1.8p+47);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= KC(0x1.8p+47);
    const double rr = ylogx - kd;
    unsigned long long t = (unsigned long long)__double_as_longlong(s_pow[32u + (uint32_t)(ki & 31u)]);
    t += ki << 47;
    const double sc = __longlong_as_double((long long)t);
    const double zz = fma(KC(0x1.c6af84b912394p-5), rr, KC(0x1.ebfce50fac4f3p-3));
    const double rr2 = rr * rr;
    double e = fma(KC(0x1.62e42ff0c52d6p-1), rr, 1.0);
    e = fma(zz, rr2, e);
    e = e * sc;
    return (float)e;
}
// f32::exp as the reference's libm computes it (glibc sysdeps/ieee754/flt-32/e_expf.c, from ARM's optimized-routines; restated
// from the published algorithm like powf_table above): x * 32/ln2 split into an integer and a remainder, 2^(k/32) from the
// same 32-entry table as powf's exp2 step, a cubic in the remainder, all in f64, rounded to f32 once. Domain: |x| < 88 (the fog
// term feeds it [-1.6, 0]); no overflow / underflow handling.
AIC_DEV float expf_table(float x, const double *s_pow) {
    const double z = KC(0x1.71547652b82fep+5) * (double)x;  // InvLn2N = N / ln 2, N = 32
    double kd = z + KC(0x1.8p+52);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= KC(0x1.8p+52);
    const double r = z - kd;
    unsigned long long t = (unsigned long long)__double_as_longlong(s_pow[32u + (uint32_t)(ki & 31u)]);
    t += ki << 47;
    const double sc = __longlong_as_double((long long)t);
    const double zz = fma(KC(0x1.c6af84b912394p-20), r, KC(0x1.ebfce50fac4f3p-13));  // poly_scaled: C0 / N^3, C1 / N^2
    const double r2 = r * r;
    double y = fma(KC(0x1.62e42ff0c52d6p-6), r, 1.0);                                // C2 / N
    y = fma(zz, r2, y);
    y = y * sc;
    return (float)y;
}

struct ColorBuf {  // raytracer_components.rs:20-39
    float l0, l1, l2, t;
};
AIC_DEV void cb_add(ColorBuf &b, float s0, float s1, float s2, float st) {  // :87-92
    b.l0 += s0 * b.t;
    b.l1 += s1 * b.t;
    b.l2 += s2 * b.t;
    b.t *= st;
}
AIC_DEV bool cb_opaque(const ColorBuf &b) { return b.t < 1.0f / 256.0f; }  // :105-109
AIC_DEV float luminance(float r, float g, float b) { return g * 0.7152f + (r * 0.2126f + b * 0.0722f); }

// Rgba::from(ColorBuf) (raytracer_components.rs:122-147)
AIC_DEV void cb_to_rgba(const ColorBuf &b, float out[4]) {
    if (b.t >= 1.0f) {
        out[0] = out[1] = out[2] = out[3] = 0.f;
        return;
    }
    float alpha = 1.0f - b.t;
    float c0 = b.l0, c1 = b.l1, c2 = b.l2;
    if (__ballot(alpha != 1.0f) != 0ull) {  // x / 1.0f == x: fully opaque pixels (the usual case) need no division
        c0 = b.l0 / alpha; c1 = b.l1 / alpha; c2 = b.l2 / alpha;
    }
    bool ok = (c0 >= 0.f) & (c1 >= 0.f) & (c2 >= 0.f);  // false for negative or NaN
    out[0] = ok ? (c0 > 0.f ? c0 : 0.f) : 1.0f;
    out[1] = ok ? (c1 > 0.f ? c1 : 0.f) : 0.0f;
    out[2] = ok ? (c2 > 0.f ? c2 : 0.f) : 0.0f;
    out[3] = (alpha > 0.f && alpha <= 1.f) ? alpha : (alpha == 0.f ? 0.f : 1.0f);
}


AIC_DEV uint32_t round_sat_u8(float x) {  // `(x).round() as u8`
    float r = roundf(x);
    if (!(r > 0.f)) return 0u;  // NaN, negatives, zero
    if (r >= 255.f) return 255u;
    return (uint32_t)r;
}

// ---------------------------------------------------------------------------------------
// light (space/light/data.rs, space/sky.rs, sr.rs:241-359)

// The light volume of the layer as aic_lightmath.h sees it (every field a kernel argument: SGPRs)
// Each field goes through an empty asm so that it is an opaque scalar: a select between elements of a kernel-argument array
// (`axis == 0 ? L.lo[0] : ...`) is otherwise folded into ONE load with a selected address, and a kernel-argument array that is
// indexed per lane gets copied to scratch memory.
AIC_DEV int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
AIC_DEV uint32_t opaque_s(uint32_t v) { asm volatile("" : "+s"(v)); return v; }
AIC_DEV float opaque_s(float v) { asm volatile("" : "+s"(v)); return v; }
template <class LayerT>
AIC_DEV LightGridView light_view(const LayerT &L) {
    LightGridView G;
    G.light = L.light;
    G.lo_x = opaque_s(L.lo[0]); G.lo_y = opaque_s(L.lo[1]); G.lo_z = opaque_s(L.lo[2]);
    G.size_x = opaque_s(L.size[0]); G.size_y = opaque_s(L.size[1]); G.size_z = opaque_s(L.size[2]);
    G.sky_nx = opaque_s(L.block_sky[0]); G.sky_ny = opaque_s(L.block_sky[1]); G.sky_nz = opaque_s(L.block_sky[2]);
    G.sky_px = opaque_s(L.block_sky[3]); G.sky_py = opaque_s(L.block_sky[4]); G.sky_pz = opaque_s(L.block_sky[5]);
    G.sky_mean = opaque_s(L.block_sky[6]);
    return G;
}

template <bool DIAG, class LayerT>
AIC_DEV uint32_t get_packed_light(const LayerT &L, int cx, int cy, int cz, uint32_t &nlight) {  // sr.rs:241-246
    if (DIAG) nlight++;
    uint32_t dx = (uint32_t)cx - (uint32_t)L.lo[0];
    uint32_t dy = (uint32_t)cy - (uint32_t)L.lo[1];
    uint32_t dz = (uint32_t)cz - (uint32_t)L.lo[2];
    if ((dx >= (uint32_t)L.size[0]) | (dy >= (uint32_t)L.size[1]) | (dz >= (uint32_t)L.size[2]))
        return lm_light_outside(light_view(L), cx, cy, cz);
    size_t idx = ((size_t)dx * (size_t)L.size[1] + dy) * (size_t)L.size[2] + dz;
    return L.light[idx];
}


// ---------------------------------------------------------------------------------------
// LightingOption::Bounce (surface.rs:119-166): the secondary rays.
//
// A fully opaque surface lit with Bounce { samples } sends `samples` rays in Lambert-distributed directions and averages what they
// see; each of them is a whole SpaceRaytracer::trace_ray_impl(ray, accumulator = ColorBuf, include_sky = true,
// allow_ray_bounce = false) under the SAME GraphicsOptions -- transparency mode, fog (with the secondary ray's own length), the
// 1000-step cap, debug_pixel_cost -- whose surfaces are lit Flat (surface.rs:171-176: the bounce budget is one). It runs inside the
// SHADE event of the lane that found the surface, one lane at a time through plain loops (the iterator stack of the reference
// restated over lvl_init / lvl_next, as the oracle has it): Bounce is a quality option nobody streams frames with, so this path
// is written for exactness and small code, not speed, and only the <.., LMODE = 3, ..> instantiations contain it.
// The random directions: rand::rngs::SmallRng (xoshiro256++, seeded from the primary ray's direction bits through SplitMix64,
// sr.rs:165-178) and rand_distr::UnitSphere -- rand 0.10.1 / rand_distr 0.6.0, neither under /root/reference: restated from the
// published algorithms (the test oracle restates them separately), PARITY UNPINNED (the reference has no golden for Bounce).
struct BounceRng {
    unsigned long long s0, s1, s2, s3;
};
AIC_DEV unsigned long long rotl64(unsigned long long x, int k) { return (x << k) | (x >> (64 - k)); }
AIC_DEV BounceRng bounce_rng_seed(unsigned long long state) {  // SeedableRng::seed_from_u64 of Xoshiro256PlusPlus
    unsigned long long w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        state += 0x9e3779b97f4a7c15ull;
        unsigned long long z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        w[i] = z ^ (z >> 31);
    }
    return BounceRng{w[0], w[1], w[2], w[3]};
}
AIC_DEV unsigned long long bounce_rng_next(BounceRng &g) {
    const unsigned long long result = rotl64(g.s0 + g.s3, 23) + g.s0;
    const unsigned long long t = g.s1 << 17;
    g.s2 ^= g.s0;
    g.s3 ^= g.s1;
    g.s1 ^= g.s2;
    g.s0 ^= g.s3;
    g.s2 ^= t;
    g.s3 = rotl64(g.s3, 45);
    return result;
}
AIC_DEV double bounce_uniform_m1_1(BounceRng &g) {  // Uniform::<f64>::new(-1., 1.).sample
    const double value1_2 = __longlong_as_double((long long)((bounce_rng_next(g) >> 12) | 0x3ff0000000000000ull));
    return (value1_2 - 1.0) * 2.0 + -1.0;
}
AIC_DEV void bounce_unit_sphere(BounceRng &g, double out[3]) {  // rand_distr::UnitSphere (Marsaglia)
    for (;;) {
        const double x1 = bounce_uniform_m1_1(g), x2 = bounce_uniform_m1_1(g);
        const double sum = x1 * x1 + x2 * x2;
        if (sum >= 1.0) continue;
        const double factor = 2.0 * sqrt(1.0 - sum);
        out[0] = x1 * factor; out[1] = x2 * factor; out[2] = 1.0 - 2.0 * sum;
        return;
    }
}

struct SecSurface {  // what a secondary ray's Surface needs (Flat lighting: no intersection point)
    float r, g, b, a, e0, e1, e2;
    int cx, cy, cz, face;
    double t;
};

// trace_ray_impl(ray, ColorBuf, include_sky = true, allow_ray_bounce = false) -> Rgba::from(buf).to_rgb(); returns the ray's
// cubes_traced. `sky_mem`: the layer's sky[8][3] as memory (a per-lane index into a kernel-argument array would put it in scratch).
template <bool DIAG, class LayerT>
AIC_DEV uint32_t bounce_secondary_ray(const LayerT &L, const float *sky_mem, const float *lut, const double *s_pow, bool big,
                                      double ox, double oy, double oz, double dirx, double diry, double dirz, float out[3]) {
    const auto &opt = L.opt;
    const bool vol = opt.transparency == 1;
    const uint32_t idx_mask = big ? 0xffffu : kCubeIndexMask;
    const uint32_t oct = ((dirx >= 0.0) ? 4u : 0u) | ((diry >= 0.0) ? 2u : 0u) | ((dirz >= 0.0) ? 1u : 0u);  // Sky::sample (sky.rs:32-41)
    float sky[3];
    {
        const float *p = sky_mem + (L.sky_kind != 0 ? 3u * oct : 0u);
        sky[0] = p[0]; sky[1] = p[1]; sky[2] = p[2];
    }
    const double t_abs = sqrt(dirx * dirx + diry * diry + dirz * dirz);   // sr.rs:146
    const float t_view = (float)(t_abs / opt.view_distance);              // sr.rs:149-151
    const bool fog_on = opt.fog != 0;
    const float fog_blend = opt.fog == 1 ? 1.0f : (opt.fog == 2 ? 0.5f : 0.0f);
    const RayDir rd = raydir_init(dirx, diry, dirz);
    const double half_over_len = 0.5 / t_abs;
    const int olx = L.lo[0], oly = L.lo[1], olz = L.lo[2], osx = L.size[0], osy = L.size[1], osz = L.size[2];
    const int ohx = olx + osx, ohy = oly + osy, ohz = olz + osz;
    const LvlLim oi = lvl_init(ox, oy, oz, rd, true, olx, oly, olz, ohx, ohy, ohz, true, half_over_len);
    Lvl os = oi.s;
    const Lim ol = oi.lim;
    // the block the ray is inside (VoxelSurfaceIter, surface.rs:361-411)
    bool inb = false;
    Lvl is = oi.s;
    Lim il = oi.lim;
    int ilx = 0, ily = 0, ilz = 0, isx = 1, isy = 1, isz = 1, bcx = 0, bcy = 0, bcz = 0;
    uint32_t blk_res = 1u, vox_off = 0u, pal_off = 0u, n_inv = 0u;
    ColorBuf acc;
    acc.l0 = acc.l1 = acc.l2 = 0.f; acc.t = 1.0f;
    uint32_t count = 0;
    bool has_last = false;  // DepthIter.last_surface
    SecSurface last;
    last.r = last.g = last.b = last.a = last.e0 = last.e1 = last.e2 = 0.f; last.cx = last.cy = last.cz = last.face = 0; last.t = 0.0;

    auto count_step_should_stop = [&]() -> bool {  // sr.rs:625-656 (the exception hits are transparent: no effect on a ColorBuf)
        count++;
        if (count > 1000u) return true;
        return cb_opaque(acc);
    };
    // Surface::to_light with Flat illumination + trace_through_surface's accumulate (surface.rs:73-106, 171-176; sr.rs:697-717)
    auto through_surface = [&](const SecSurface &sf) {
        float r = sf.r, g = sf.g, b = sf.b, a = sf.a;
        if (opt.transparency == 2) {  // limit_alpha
            if (a > opt.threshold) a = 1.0f;
            else { r = g = b = a = 0.f; }
        }
        if (a == 0.f && sf.e0 == 0.f && sf.e1 == 0.f && sf.e2 == 0.f) return;
        int nx = 0, ny = 0, nz = 0;
        if (sf.face == 1) nx = -1; else if (sf.face == 2) ny = -1; else if (sf.face == 3) nz = -1;
        else if (sf.face == 4) nx = 1; else if (sf.face == 5) ny = 1; else if (sf.face == 6) nz = 1;
        uint32_t nl = 0;
        const uint32_t txl = get_packed_light<false>(L, sf.cx + nx, sf.cy + ny, sf.cz + nz, nl);
        const float i0 = lut[txl & 255u], i1 = lut[(txl >> 8) & 255u], i2 = lut[(txl >> 16) & 255u];
        float o0 = ps_mul(ps_mul(r, i0), a) + sf.e0, o1 = ps_mul(ps_mul(g, i1), a) + sf.e1, o2 = ps_mul(ps_mul(b, i2), a) + sf.e2;
        float tr = 1.0f - a;
        if (fog_on) {  // distance_fog (sr.rs:745-768), with THIS ray's t_to_view_distance and sky
            float rel = (float)sf.t * t_view;
            rel = rel < 0.0f ? 0.0f : (rel > 1.0f ? 1.0f : rel);
            const float sq = rel * rel;
            const float fog_exp = 1.0f - expf_table(-1.6f * rel, s_pow);
            const float fudged = fog_exp / 0.79810348f;
            const float amount = zo_clamped(fudged * (1.0f - fog_blend) + (sq * sq) * fog_blend);
            const float comp = 1.0f - amount;
            o0 = ps_mul(o0, comp) + ps_mul(sky[0], amount);
            o1 = ps_mul(o1, comp) + ps_mul(sky[1], amount);
            o2 = ps_mul(o2, comp) + ps_mul(sky[2], amount);
            tr *= comp;
        }
        cb_add(acc, o0, o1, o2, tr);
    };
    // trace_through_span (sr.rs:720-740) + apply_transmittance (raytracer_components.rs:215-258)
    auto through_span = [&](SecSurface sf, double exit_t) {
        float thickness = (float)((exit_t - sf.t) * t_abs);
        thickness = fmaxf(thickness, 0.0f);
        float coeff;
        if (thickness == 0.0f) {
            if (sf.a == 1.0f) coeff = 1.0f;
            else { sf.r = sf.g = sf.b = sf.a = 0.f; coeff = 0.0f; }
        } else {
            const float unit_t = 1.0f - sf.a;
            float depth_t;
            if (unit_t == 0.0f) depth_t = 0.0f;
            else if (unit_t == 1.0f) depth_t = 1.0f;
            else if (!(thickness < __uint_as_float(0x7f800000u))) depth_t = 0.0f;
            else depth_t = powf_table(unit_t, thickness, s_pow);
            sf.a = zo_clamped(1.0f - depth_t);
            const float ec = (unit_t == 1.0f) ? thickness : (depth_t - 1.f) / (unit_t - 1.f);
            coeff = fmaxf(ec, 0.0f);
        }
        const float c = ps_clamped(coeff);
        sf.e0 = ps_mul(sf.e0, c); sf.e1 = ps_mul(sf.e1, c); sf.e2 = ps_mul(sf.e2, c);
        through_surface(sf);
    };

    for (;;) {
        // ---- SurfaceIter::next (surface.rs:283-354): 1 Invisible, 2 EnterSurface, 3 EnterBlock, 0 the ray is over ----
        int kind = 0;
        double t = 0.0;
        SecSurface cur = last;
        if (inb) {
            const NextResult nr = lvl_next(is, il, rd, ilx, ily, ilz, ilx + isx, ily + isy, ilz + isz);
            is = nr.s;
            if (nr.got) {
                const double as = __hiloint2double((int)((1023u - (31u - (uint32_t)__clz((int)blk_res))) << 20), 0);  // 1 / resolution
                t = is.last_t * as;
                kind = 1;
                if (!nr.is_exit) {
                    const uint32_t vi = (uint32_t)(((uint32_t)(is.cx - ilx) * (uint32_t)isy + (uint32_t)(is.cy - ily)) * (uint32_t)isz + (uint32_t)(is.cz - ilz));
                    const uint32_t code = L.pool[(size_t)vox_off + vi];
                    if (code >= n_inv) {
                        const DevPaletteEntry *pe = &L.palette[pal_off + code];
                        kind = 2;
                        cur.r = pe->color[0]; cur.g = pe->color[1]; cur.b = pe->color[2]; cur.a = pe->color[3];
                        cur.e0 = pe->emission[0]; cur.e1 = pe->emission[1]; cur.e2 = pe->emission[2];
                        cur.cx = bcx; cur.cy = bcy; cur.cz = bcz; cur.face = lvl_face(is); cur.t = t;
                    }
                }
            } else {
                inb = false;
            }
        }
        if (kind == 0) {
            const NextResult nr = lvl_next(os, ol, rd, olx, oly, olz, ohx, ohy, ohz);
            os = nr.s;
            if (!nr.got) break;
            t = os.last_t;
            kind = 1;
            if (!nr.is_exit) {
                const size_t ci = ((size_t)(uint32_t)(os.cx - olx) * (size_t)osy + (size_t)(uint32_t)(os.cy - oly)) * (size_t)osz + (size_t)(uint32_t)(os.cz - olz);
                const uint32_t entry = L.pool[ci];
                const uint32_t bi = entry & idx_mask;
                const uint32_t cls = big ? ((L.cls[bi >> 4] >> ((bi & 15u) << 1)) & 3u) : (entry >> kCubeClassShift);
                const DevBlock *tb = &L.blocks[bi];
                if (cls == 1u) {
                    kind = 2;
                    cur.r = tb->color[0]; cur.g = tb->color[1]; cur.b = tb->color[2]; cur.a = tb->color[3];
                    cur.e0 = tb->emission[0]; cur.e1 = tb->emission[1]; cur.e2 = tb->emission[2];
                    cur.cx = os.cx; cur.cy = os.cy; cur.cz = os.cz; cur.face = lvl_face(os); cur.t = t;
                } else if (cls == 2u) {
                    // RaycastStep::recursive_raycast (raycast.rs:458-476): the sub-ray keeps the direction
                    kind = 3;
                    blk_res = tb->kind & 255u;
                    const uint32_t vlo = tb->vlo_packed, vsz = tb->vsize_packed;
                    ilx = (int)(vlo & 255u); ily = (int)((vlo >> 8) & 255u); ilz = (int)((vlo >> 16) & 255u);
                    isx = (int)(vsz & 255u); isy = (int)((vsz >> 8) & 255u); isz = (int)((vsz >> 16) & 255u);
                    vox_off = tb->vox_off; pal_off = tb->pal_off; n_inv = tb->n_invisible;
                    bcx = os.cx; bcy = os.cy; bcz = os.cz;
                    const double kd = (double)blk_res;
                    const LvlLim ii = lvl_init((ox - (double)bcx) * kd, (oy - (double)bcy) * kd, (oz - (double)bcz) * kd, rd, true, ilx, ily, ilz,
                                               ilx + isx, ily + isy, ilz + isz, true, half_over_len);
                    is = ii.s;
                    il = ii.lim;
                    inb = true;
                }
            }
        }
        // ---- the tracing loop's body (sr.rs:183-225), DepthIter (surface.rs:453-491) folded in for Volumetric ----
        if (vol) {
            if (count_step_should_stop()) break;
            if (kind == 2) {
                if (has_last) through_span(last, cur.t);
                last = cur;
                has_last = true;
            } else {
                if (has_last) { has_last = false; through_span(last, t); }
                if (kind == 3 && count_step_should_stop()) break;  // the buffered EnterBlock step
            }
        } else {
            if (count_step_should_stop()) break;
            if (kind == 2) through_surface(cur);
        }
    }
    // finish (sr.rs:658-693): the sky, then the optional cost visualisation
    cb_add(acc, sky[0] * 1.0f, sky[1] * 1.0f, sky[2] * 1.0f, 0.0f);
    if (opt.debug_pixel_cost) {
        const float n = ps_clamped((float)count);
        const float red = ps_clamped(ps_mul(0.02f, n) * 1.0f);
        const float green = ps_clamped(ps_mul(0.002f, n) * 1.0f);
        float cur_rgba[4];
        cb_to_rgba(acc, cur_rgba);
        const float blue = ps_clamped(luminance(cur_rgba[0], cur_rgba[1], cur_rgba[2]) * 0.2f);
        acc.l0 = red; acc.l1 = green; acc.l2 = blue; acc.t = 0.0f;
    }
    float c[4];
    cb_to_rgba(acc, c);
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
    (void)DIAG;
    return count;
}

// ---------------------------------------------------------------------------------------
// The image kernel.
//
// Reference semantics per ray = SpaceRaytracer::trace_ray_impl (sr.rs:135-238) driven by
// RtScene::trace_patch / trace_ray_through_layers and the draw_rgba encoder
// (renderer.rs:282-308, 424-478, 516-556). Execution model (CDNA4-first):
//
//  * A wave64 runs its pixels through its 64 lanes persistently: a lane that finishes
//    a ray is refilled with the next pixel of the wave's current 8x8 tile, and a new tile is pulled
//    from the frame's queue when that one is used up (wave-level __ballot + prefix popcount), so
//    lanes stay busy instead of idling behind the longest ray of a fixed packet.
//  * Every lane is a small state machine. The *stepping* state is a tight loop body: one
//    Amanatides-Woo step of whichever DDA level the lane is on (outer cube grid or inner block
//    voxels share the code and the registers), one 2-byte lookup, the step bookkeeping. Anything
//    expensive -- entering a block (a new bounded raycaster: divisions, sqrt), lighting a
//    surface (8 light texels + f64 geometry), compositing a span (powf/exp), finishing a ray
//    (sky, encode, store), starting a ray (unprojection) -- is an *event*: the lane parks, and
//    the wave runs the event code only when at least half of its live lanes are parked (or none
//    can step). Rare heavy paths therefore execute with well-filled waves instead of taxing
//    every step of every lane.
//  * The per-lane order of operations is exactly the reference's, so results are bit-identical
//    to the sequential formulation (hit cubes, t values, step counts).

struct Diag {
    uint32_t n_outer, n_inner, n_hits, n_light;
    uint32_t layer;  // layer of that first hit: 0 world, 1 UI (its block_index is an index into THAT layer's block table)
    int hit;  // first Hit carrying a Position
    int cube[3], voxel[3], res, face, block;
    double t;
};
struct SurfDiag {  // DIAG only: identity of a pending surface
    uint32_t nlight;
    int cube[3], voxel[3], res, face, block;
};

// camera (camera_struct.rs:238-257; euclid Transform3D::transform_point3d)
template <class MatP>
AIC_DEV void unproject(MatP m, double x, double y, double z, double out[3]) {
    const double px = x * m[0] + y * m[4] + z * m[8] + m[12];
    const double py = x * m[1] + y * m[5] + z * m[9] + m[13];
    const double pz = x * m[2] + y * m[6] + z * m[10] + m[14];
    const double pw = x * m[3] + y * m[7] + z * m[11] + m[15];
    if (pw > 0.0) {
        // Three quotients by one divisor: the generic division's reciprocal (v_rcp_f64 and its two refinement steps) made once, then each quotient's
        // multiply and two residual steps -- the generic sequence's own operations when nothing needs scaling, which the exponent window guarantees
        // (see div_known_recip above); 21 instructions for the three instead of 33. Lanes outside the window divide generically.
        const bool fast = ((int)f64_exp_in_768_1280(pw) & (int)f64_exp_in_768_1280(px) & (int)f64_exp_in_768_1280(py) & (int)f64_exp_in_768_1280(pz)) != 0;
        double r = __builtin_amdgcn_rcp(pw);
        double e = fma(-pw, r, 1.0);
        r = fma(r, e, r);
        e = fma(-pw, r, 1.0);
        r = fma(r, e, r);
        const double qx = px * r, qy = py * r, qz = pz * r;
        out[0] = fma(fma(-pw, qx, px), r, qx); out[1] = fma(fma(-pw, qy, py), r, qy); out[2] = fma(fma(-pw, qz, pz), r, qz);
        if (__builtin_amdgcn_ballot_w64(!fast) != 0ull) {
            AIC_RARE_PATH();
            if (!fast) { out[0] = px / pw; out[1] = py / pw; out[2] = pz / pw; }
        }
    } else {
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        out[0] = out[1] = out[2] = nan;
    }
}

constexpr float NO_WORLD_TO_SHOW = 0.5028865f;  // palette.rs:76 #BCBCBC decoded to linear

// viewport.rs:104-113
AIC_DEV double fb_x_edge(uint32_t w, uint32_t x) { return ((double)x) / (double)w * 2.0 - 1.0; }
AIC_DEV double fb_y_edge(uint32_t h, uint32_t y) { return -(((double)y) / (double)h * 2.0 - 1.0); }

// Rgba::to_srgb8 colour channels (color.rs:1038-1054) without powf: `thr[k]` (k = 1..255) is the smallest f32 whose reference encoding is >= k (built on
// the host with the reference formula), so the encoding of c is the number of thresholds <= c. A fast estimate k seeds the count, the thresholds make
// it exact. The kernel holds them as a WINDOW table `w` of kSrgbWindowWords floats -- w[j] = thr[j - 1], with -inf below thr[1] and NaN above
// thr[255] -- so that the four thresholds around an estimate, thr[k - 1 .. k + 2], are w[k .. k + 3] for every k in 0..255: c >= -inf always holds,
// c >= NaN never. The three channels' eight reads are issued together and compared without a branch; the count of thresholds <= c among the four
// settles the encoding unless it is 0 or 4 (the estimate was off by two or more: v_log_f32 / v_exp_f32 are good to about an ulp, so never seen), and
// then the thresholds are searched as before round 6 (two dependent LDS reads per step of two loops per channel, for every pixel).
constexpr uint32_t kSrgbWindowWords = 260u;
AIC_DEV void srgb_window_to_lds(float *w, const float *thr, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < kSrgbWindowWords; i += nthreads)
        w[i] = i < 2u ? __uint_as_float(0xff800000u) : (i <= 256u ? thr[i - 1u] : __uint_as_float(0x7fc00000u));
}
AIC_DEV int srgb8_estimate(float c) {  // 0..255 (c > 0)
    const float cc = fminf(c, 1.0f);
    const float e = cc <= 0.0031308f ? cc * 12.92f : 1.055f * __builtin_amdgcn_exp2f(0.41666666f * __builtin_amdgcn_logf(cc)) - 0.055f;
    const int k = (int)(e * 255.f + 0.5f);
    return k < 0 ? 0 : (k > 255 ? 255 : k);
}
AIC_DEV uint32_t srgb8_search(float c, int k, const float *__restrict__ w) {  // c > 0; thr[j] = w[j + 1]
    while (k < 255 && c >= w[k + 2]) k++;
    while (k > 0 && c < w[k + 1]) k--;
    return (uint32_t)k;
}
AIC_DEV void srgb8_rgb(float r, float g, float b, const float *__restrict__ w, uint32_t &R, uint32_t &G, uint32_t &B) {
    // 0, negatives (cannot occur) and NaN encode to 0
    const bool pr = r > 0.f, pg = g > 0.f, pb = b > 0.f;
    const int kr = pr ? srgb8_estimate(r) : 0, kg = pg ? srgb8_estimate(g) : 0, kb = pb ? srgb8_estimate(b) : 0;
    const float r0 = w[kr], r1 = w[kr + 1], r2 = w[kr + 2], r3 = w[kr + 3];
    const float g0 = w[kg], g1 = w[kg + 1], g2 = w[kg + 2], g3 = w[kg + 3];
    const float b0 = w[kb], b1 = w[kb + 1], b2 = w[kb + 2], b3 = w[kb + 3];
    const int nr = (int)(r >= r0) + (int)(r >= r1) + (int)(r >= r2) + (int)(r >= r3);
    const int ng = (int)(g >= g0) + (int)(g >= g1) + (int)(g >= g2) + (int)(g >= g3);
    const int nb = (int)(b >= b0) + (int)(b >= b1) + (int)(b >= b2) + (int)(b >= b3);
    R = pr ? (uint32_t)(kr - 2 + nr) : 0u;
    G = pg ? (uint32_t)(kg - 2 + ng) : 0u;
    B = pb ? (uint32_t)(kb - 2 + nb) : 0u;
    // (n - 1) > 2 unsigned <=> n is 0 or 4
    const bool open_r = pr & ((uint32_t)(nr - 1) > 2u), open_g = pg & ((uint32_t)(ng - 1) > 2u), open_b = pb & ((uint32_t)(nb - 1) > 2u);
    if (__builtin_amdgcn_ballot_w64(open_r | open_g | open_b) != 0ull) {
        AIC_RARE_PATH();
        if (open_r) R = srgb8_search(r, kr, w);
        if (open_g) G = srgb8_search(g, kg, w);
        if (open_b) B = srgb8_search(b, kb, w);
    }
}

// lane event word (ev): what the lane needs next. Values below 4 mean "stepping" (the wave scheduler's test), with
//   1  EV_FRESH   the first cube of a level was emitted by the event that set the level up: look it up without stepping
//   2  EV_DEAD    the current level's Raycaster has ended: it cannot step any further
// and the parked kinds (EV_DEAD may ride along with SHADE / ENTER: the event still needs the ended level's state)
constexpr uint32_t EV_FRESH = 1u, EV_DEAD = 2u, EV_SHADE = 4u, EV_ENTER = 8u, EV_FINISH = 16u, EV_NEWRAY = 32u, EV_DONE = 64u,
                   EV_TAKE = 128u;  // with NEWRAY: take a new pixel first
// lane state bits (st):
//   9-13   flags below;  14-15 the antialiasing sample being traced
//   16-18  the suspended outer level's Face while inside a block; 21 ST_OUTER_ALIVE: that level can go on stepping
//   24-26  sign bits of the ray direction (x: 26, y: 25, z: 24; set = component >= 0) = the sky octant
//   22     ST_DIR_FAST: RayDir::fast of the ray's direction (divisions by it may use t_delta: div_known_recip)
constexpr uint32_t ST_IN_BLOCK = 1u << 9, ST_HAS_LAST = 1u << 10, ST_OPAQUE = 1u << 12, ST_TRACED = 1u << 13, ST_OUTER_ALIVE = 1u << 21, ST_DIR_FAST = 1u << 22;

#ifndef AIC_MIN_WAVES
#define AIC_MIN_WAVES 4  // waves per SIMD the production variants are built for (128 VGPRs; cold lane state lives in LDS)
#endif
#ifndef AIC_T_BATCH
#define AIC_T_BATCH 32  // run a kind of parked work once this many lanes wait on it
#endif
#ifndef AIC_N_FEW
#define AIC_N_FEW 24    // ... or once at most this many lanes can still step (32 until the fast steps made stepping cheaper: r03 E)
#endif
#ifndef AIC_WG_THREADS
#define AIC_WG_THREADS 256  // threads per persistent workgroup (a multiple of 64)
#endif
#ifndef AIC_STEP_REPS
#define AIC_STEP_REPS 2  // full stepping passes per scheduler trip (3 until round 4; swept again with AIC_FAST_STEPS once no pending span kept lanes
                         // out of the fast steps: profiles/r04_experiments.txt G)
#endif
#ifndef AIC_FAST_MIN
#define AIC_FAST_MIN 16  // ... while at least this many lanes of the wave can take one
#endif
#ifndef AIC_FAST_STEPS
#define AIC_FAST_STEPS 16  // bookkeeping-free steps a lane may take ahead of each full pass (0: none; 8 until round 4) ...
#endif
// ---- Lane exchange between the waves of a workgroup ("regime-sorted waves"; DESIGN.md 4.2, tools/wave_sim) ----
// A wave runs ONE kind of work per scheduler round (a stepping trip, or one kind of event) and the lanes of the other kinds idle: every phase
// ran at 25-39 of 64 lanes (profiles/r04_phase_cycles.txt) on a kernel bound by instruction issue. The production variants therefore share a POOL
// of parked rays per workgroup, in LDS: before a round the wave trades lanes that would idle for parked rays of the kind it is about to run
// (compare-and-swap on a slot's tag, then a plain exchange of the ray's 40 hot dwords; the cold state stays in the ray's LDS column, whose index
// travels with the ray), and while the pool has free slots it parks such lanes and starts new pixels in them, so that the workgroup holds more rays
// than lanes (the reservoir the sorting needs). Nothing ever waits for another wave: a claim that fails is simply not made. A ray is a pure
// function of its own state, which moves as a whole, so frames are bit-identical for any pool size or policy.
#ifndef AIC_EXCHANGE
#define AIC_EXCHANGE 1
#endif
#ifndef AIC_XWG_THREADS
#define AIC_XWG_THREADS 256  // threads of a workgroup of the exchanging variants (four per CU, a pool of 72 each: measured ahead of two workgroups of 512 with a pool of 160,
                             // whose eight waves lose more claims to one another and fill the frame's tail worse -- profiles/r05_experiments.txt B)
#endif
#ifndef AIC_POOL
#define AIC_POOL (AIC_XWG_THREADS >= 512 ? 160 : 64)  // parked rays per workgroup (<= 192: up to three tags per lane are scanned); what the CU's 160 KB leave room for beside 80-byte
                                                       // columns (round 5: 72 slots beside 72-byte columns, the ray's origin and direction in global memory)
#endif
#ifndef AIC_XCHG_MIN_GAIN
#define AIC_XCHG_MIN_GAIN 8  // a wave that has lanes of the chosen kind tops up only if the pool adds at least this many
#endif
#ifndef AIC_COLD_SCOPE
#define AIC_COLD_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP  // (experiment: __HIP_MEMORY_SCOPE_WAVEFRONT = plain loads the CU's L1 may serve stale -- timing only)
#endif
#ifndef AIC_XCHG_FULL
#define AIC_XCHG_FULL 48     // a wave with this many lanes of one kind runs it as it is
#endif
#ifndef AIC_XCHG_PARK_MIN
#define AIC_XCHG_PARK_MIN 8  // an exchange that only parks (nothing to take) is made for at least this many lanes
#endif
#ifndef AIC_XCHG_DEPOSIT
#define AIC_XCHG_DEPOSIT 3   // while slots are free: 0 park nothing, 2 park event lanes that do not run now, 3 stepping lanes too (tools/wave_sim: 3 > 2 > 1)
#endif
constexpr uint32_t TAG_FREE = 0u, TAG_STEP = 1u, TAG_SHADE = 2u, TAG_ENTER = 3u, TAG_RAY = 4u, TAG_BUSY = 7u;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The first cube of a freshly initialised level, for the ENTER / RAY events: Raycaster::next run until it yields its first step or ends, so that the
// stepping loop only ever sees levels that are already inside their bounds. Written like the stepping trip (round 6): wave masks for every decision, the per-lane state updated in place by
// one exec-masked block -- lvl_next above, inlined into ENTER and NEWRAY, was compiled into ~600 instructions of nested exec-mask scaffolding with some
// thirty register copies per turn of its loop. Raycaster::next from FirstLast::Beginning (raycast.rs:239-284) is: while the cube is outside the bounds on
// the side the ray comes from and not past them (is_out_of_bounds_ahead, raycast.rs:711-728: "not yet entered"), step (State::step, raycast.rs:577-626:
// along the axis of the smallest t_max, ties to the later axis; a level whose smallest t_max is not finite cannot step and ends); a cube inside the
// bounds is emitted -- also by a level that cannot step, if it has not stepped yet (Face7::Within); anything else ends the level with nothing emitted.
// (`checked_add` of the stepped coordinate cannot fail: a coordinate at i32::MAX with the ray going up, or at MIN going down, is past the bounds.)
// Returns the emitted cube in the kernel's conventions: `lax` = the axis stepped along last, or 8 | Face7::Within for a cube emitted without a step.
struct FirstCube {
    double tx, ty, tz, last_t;
    int cx, cy, cz;
    uint32_t lax;
    bool got, inbounds;  // emitted a cube; the level can go on (FirstLast::InBounds)
};
AIC_DEV FirstCube lvl_first_masks(const Lvl s0, const RayDir rd, int lox, int loy, int loz, int hix, int hiy, int hiz) {
    typedef unsigned long long mask_t;
    FirstCube f;
    f.tx = s0.tx; f.ty = s0.ty; f.tz = s0.tz; f.last_t = s0.last_t;
    f.cx = s0.cx; f.cy = s0.cy; f.cz = s0.cz;
    f.lax = 8u | (uint32_t)FACE_WITHIN;
    const mask_t negx = __builtin_amdgcn_ballot_w64(rd.sx < 0), posx = __builtin_amdgcn_ballot_w64(rd.sx > 0);
    const mask_t negy = __builtin_amdgcn_ballot_w64(rd.sy < 0), posy = __builtin_amdgcn_ballot_w64(rd.sy > 0);
    const mask_t negz = __builtin_amdgcn_ballot_w64(rd.sz < 0), posz = __builtin_amdgcn_ballot_w64(rd.sz > 0);
    mask_t active = __builtin_amdgcn_ballot_w64(lvl_fl(s0) != FL_ENDED);
    mask_t m_got = 0ull, m_inb = 0ull, m_stepped = 0ull;
    const uint32_t finite_classes = 0x1f8u;  // v_cmp_class: -normal, -denormal, -0, +0, +denormal, +normal
    while (active != 0ull) {
        // is_out_of_bounds_ahead: per axis "not yet entered" is below the bounds going up, above going down, either for an axis the ray does not move
        // along; "left" the other way round
        const mask_t lowx = __builtin_amdgcn_ballot_w64(f.cx < lox), highx = __builtin_amdgcn_ballot_w64(f.cx >= hix);
        const mask_t lowy = __builtin_amdgcn_ballot_w64(f.cy < loy), highy = __builtin_amdgcn_ballot_w64(f.cy >= hiy);
        const mask_t lowz = __builtin_amdgcn_ballot_w64(f.cz < loz), highz = __builtin_amdgcn_ballot_w64(f.cz >= hiz);
        const mask_t enter = (lowx & ~negx) | (highx & ~posx) | (lowy & ~negy) | (highy & ~posy) | (lowz & ~negz) | (highz & ~posz);
        const mask_t exit_ = (lowx & ~posx) | (highx & ~negx) | (lowy & ~posy) | (highy & ~negy) | (lowz & ~posz) | (highz & ~negz);
        const mask_t m_in = active & ~(enter | exit_), m_go = active & enter & ~exit_;
        const mask_t m_any = m_in | m_go;
        mask_t m_fin, sv, mx, m_stp;
        double mn;
        asm volatile(
            "s_and_saveexec_b64 %[sv], %[any]\n\t"
            "v_min_f64 %[mn], %[tx], %[ty]\n\t"
            "v_min_f64 %[mn], %[mn], %[tz]\n\t"
            "v_cmp_class_f64 %[fin], %[mn], %[cls]\n\t"   // valid_for_stepping (raycast.rs:563-570): the smallest t_max is finite
            "s_and_b64 %[stp], %[fin], %[go]\n\t"          // the lanes that step
            "s_mov_b64 exec, %[stp]\n\t"
            "v_mov_b64 %[lt], %[mn]\n\t"
            "v_cmp_eq_f64 %[mx], %[tz], %[mn]\n\t"         // Z
            "v_cmp_eq_f64 vcc, %[ty], %[mn]\n\t"
            "s_andn2_b64 vcc, vcc, %[mx]\n\t"              // Y
            "s_mov_b64 exec, %[mx]\n\t"
            "v_add_f64 %[tz], %[tz], %[tdz]\n\t"
            "v_add_u32 %[cz], %[cz], %[sz]\n\t"
            "v_mov_b32 %[lax], 2\n\t"
            "s_or_b64 %[mx], %[mx], vcc\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_add_f64 %[ty], %[ty], %[tdy]\n\t"
            "v_add_u32 %[cy], %[cy], %[sy]\n\t"
            "v_mov_b32 %[lax], 1\n\t"
            "s_andn2_b64 exec, %[stp], %[mx]\n\t"          // X = stepping lanes that took neither
            "v_add_f64 %[tx], %[tx], %[tdx]\n\t"
            "v_add_u32 %[cx], %[cx], %[sx]\n\t"
            "v_mov_b32 %[lax], 0\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            : [tx] "+v"(f.tx), [ty] "+v"(f.ty), [tz] "+v"(f.tz), [lt] "+v"(f.last_t), [cx] "+v"(f.cx), [cy] "+v"(f.cy), [cz] "+v"(f.cz), [lax] "+v"(f.lax),
              [mn] "=&v"(mn), [sv] "=&s"(sv), [mx] "=&s"(mx), [fin] "=&s"(m_fin), [stp] "=&s"(m_stp)
            : [tdx] "v"(rd.tdx), [tdy] "v"(rd.tdy), [tdz] "v"(rd.tdz), [sx] "v"(rd.sx), [sy] "v"(rd.sy), [sz] "v"(rd.sz), [any] "s"(m_any), [go] "s"(m_go),
              [cls] "s"(finite_classes)
            : "vcc", "scc");
        m_got |= m_in & (m_fin | ~m_stepped);
        m_inb |= m_in & m_fin;
        m_stepped |= m_stp;
        active = m_stp;
    }
    f.got = __builtin_amdgcn_inverse_ballot_w64(m_got);
    f.inbounds = __builtin_amdgcn_inverse_ballot_w64(m_inb);
    return f;
}

// How a DDA level is held in registers by the image kernel (both levels -- the cube grid and a block's
// voxel volume -- use the same registers and the same stepping code):
//   t[3]   t_max, exactly the reference's (raycast.rs:99-121)
//   td[3]  t_delta = 1/|d|, a per-ray constant shared by both levels (a sub-ray keeps its direction)
//   r[3]   steps the ray can still take along each axis before it leaves the level's bounds, MINUS ONE: for a
//          coordinate c relative to the level's lower corner, r = size - 1 - c going up, c going down
//          (direction sign = the octant bit). Stepping decrements it; the decrement's borrow (r was 0) <=> the
//          include_exit step. The coordinate is recovered when an event needs it: c = size - 1 - r or r.
//   boff   BYTE offset of the current cube's / voxel's u16 in the pool; ss[3] the signed byte strides.
//          Stepping adds ss[axis]: no index arithmetic in the loop, and the lookup is a
//          scalar-base + 32-bit-offset load (the pool is at most 4 GiB: aic_upload_space checks).
//   thr    a looked-up code >= thr is a visible surface (voxels: the block's first visible palette code;
//          cubes with class bits: 0x4000, i.e. class >= 1)
// The axis to step along is recomputed from t[] at every step (two v_min_f64, two compares) instead of being
// carried in the state: it is a pure function of t[], which nothing modifies between steps.

template <bool VOL, int LMODE, bool DIAG, bool BIG, bool XC>
__global__ __launch_bounds__((AIC_EXCHANGE && XC && !DIAG && LMODE != 3) ? AIC_XWG_THREADS : AIC_WG_THREADS, (DIAG || LMODE == 3) ? 2 : AIC_MIN_WAVES) void trace_image_kernel(const DevFrame F) {
    // XCHG: the production variants -- lanes are exchanged between the workgroup's waves through a pool of parked rays, and the part of a ray's
    // cold state that only ENTER / SHADE / FINISH touch (origin, direction, antialiasing sums) lives in global memory to make room for it.
    // The aux-recording and Bounce variants (more per-lane state, built for 2 waves per SIMD) keep everything in LDS and exchange nothing.
    // (XC: the production variants exist with and without the exchange; the launcher picks by the frame's size -- DevFrame::exchange)
    constexpr bool XCHG = AIC_EXCHANGE && XC && !DIAG && LMODE != 3;
    constexpr uint32_t WGT = XCHG ? (uint32_t)AIC_XWG_THREADS : (uint32_t)AIC_WG_THREADS;  // threads per workgroup
    constexpr uint32_t NPOOL = XCHG ? (uint32_t)AIC_POOL : 0u;                            // pool slots
    constexpr uint32_t NCOL = WGT + NPOOL;                                                // LDS columns: one per lane and one per slot
    static_assert(NPOOL <= 192u, "three tags per lane are scanned");
    // ---- persistent waves: each wave pulls 8x8-pixel tiles from a global counter until the
    // image is exhausted, so cheap (sky) and expensive (geometry) tiles balance dynamically ----
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t tile_x0 = 0, tile_y0 = 0;  // wave-uniform: pixel origin of the tile the refill is drawing from
    // small decode tables live in LDS for the life of the persistent workgroup
    __shared__ float s_lut[256];     // PackedLight scalar decode (light/data.rs:301-354)
    __shared__ float s_thr[kSrgbWindowWords];  // sRGB8 encode thresholds, as a window table (srgb8_rgb)
    __shared__ double s_pow[64];     // powf tables (powf_table)
    pow_tables_to_lds(s_pow, threadIdx.x, WGT);
    for (uint32_t i = threadIdx.x; i < 256u; i += WGT) s_lut[i] = F.light_lut[i];
    srgb_window_to_lds(s_thr, F.srgb_thr, threadIdx.x, WGT);
    __syncthreads();
    const float *lut = s_lut;
    // Kernel arguments and the persistent loop. The stepping phase needs three scalars of them (the pool pointer and the cube
    // grid's two byte strides, below); everything else is wanted by events only. Left as plain uses of `F`, the compiler
    // fetches all ~250 dwords of DevFrame before the loop, keeps what fits in SGPRs and parks the rest in VGPR lanes
    // (v_writelane / v_readlane + hazard nops: round 2's 92 spilled SGPRs and two VGPRs lost to them). Instead every event
    // phase reads the arguments it needs through an OPAQUE pointer to the kernel-argument segment (see the event phase):
    // scalar loads that hit the constant cache, issued where they are used, holding no register across the loop.
    const int ostx = 2 * F.layer.size[1] * F.layer.size[2], osty = 2 * F.layer.size[2];  // byte strides of the cube grid (z: 2)
    const bool ui_pass_s = F.pass == 1;  // (DIAG builds: the stepping phase labels first hits with their layer)
    // cube grid at offset 0, then every block's voxel volume. The pointer is laundered through an
    // s_mov so that it is a computed SGPR pair rather than a re-loadable kernel argument: under SGPR
    // pressure the compiler would otherwise re-fetch it (s_load + wait) in front of every lookup.
    unsigned long long pool_bits = (unsigned long long)F.layer.pool;
    asm volatile("s_mov_b64 %0, %1" : "=s"(pool_bits) : "s"(pool_bits));
    // BIG: block tables past 16384 entries -- plain 16-bit indices in the grid, classes from L.cls
    const uint32_t idx_mask = BIG ? 0xffffu : kCubeIndexMask;
    const uint32_t outer_thr = BIG ? 0x10000u : (1u << kCubeClassShift);

    // ---- per-lane state, hot: lives in registers across the stepping loop ----
    double tx = 0, ty = 0, tz = 0, last_t = 0;   // t_max of the current level, t of the step that entered the current cube
    double tdx = 0, tdy = 0, tdz = 0;            // t_delta
    uint32_t rx = 0, ry = 0, rz = 0;             // steps left before leaving the bounds, minus one, per axis
    uint32_t boff = 0;                           // byte offset of the current cube / voxel in the pool
    int ssx = 0, ssy = 0, ssz = 0;               // signed byte strides of the current level
    uint32_t thr = outer_thr;                    // codes >= thr are visible surfaces
    uint32_t raw = 0;                            // the code looked up last (read by the event it raised)
    uint32_t lax = 8u;                           // axis last stepped along (0..2), or 8 | Face set by an event (FACE_TABLE applied late)
    uint32_t st = 0u;
    uint32_t count = 0;
    ColorBuf acc;
    acc.l0 = acc.l1 = acc.l2 = 0.f; acc.t = 1.0f;
    // DepthIter.last_surface, already shaded: its premultiplied light and transmittance
    float pend0 = 0.f, pend1 = 0.f, pend2 = 0.f, pend_tr = 1.f;
    // the block the lane is inside: palette offset; log2(resolution) << 24 | stored-volume lower corner; stored-volume size
    uint32_t blk_pal_off = 0, blk_geo = 0, blk_vsz = 0;
    // LightingOption::Bounce (LMODE 3 only; dead code elsewhere): the ray's SmallRng (sr.rs:165-178) and what its secondary rays traced
    BounceRng brng{0ull, 0ull, 0ull, 0ull};
    uint32_t sec_steps = 0;
    // ---- per-lane state, cold: only events touch it, so it lives in LDS (one column per thread: conflict-free
    //      ds_read/ds_write), not in registers -- that is what lets the kernel run at 3-4 waves per SIMD ----
    //      A column belongs to a RAY, not to a lane: it travels with the ray when lanes are exchanged (`col`), and every pool slot holds a spare one.
    //      The exchanging variants hold the rows up to C_DZ / K_PXY only (80 bytes per column): they re-derive a ray's origin from its pixel where ENTER
    //      needs it (`ray_of_pixel`, the code NEWRAY runs: the same bits) and keep the antialiasing sums in global memory (DevFrame::ray_cold; antialiased frames only).
    //      (The suspended level's last t is not kept: a level that is resumed steps -- and sets it -- before anything reads it, or the ray is over.)
    enum { C_STX, C_STY, C_STZ,                     // the suspended outer level while inside a block: t_max
           C_TABS,                                  // |direction| (sr.rs:146)
           C_DX, C_DY, C_DZ,                        // ray direction (sanitised: Parameters::new, raycast.rs:749-771)
           C_OX, C_OY, C_OZ, N_C64_ALL };           // ray origin
    enum { K_SRX, K_SRY, K_SRZ, K_SBOFF,            // the suspended outer level: steps left, byte offset
           K_TVIEW, K_PXY,                          // |direction| / view distance; pixel x | row << 16
           K_BLK, K_S0, K_S1, K_S2, K_ST, N_C32_ALL };  // block index (aux records); ColorBuf::mean accumulators (antialiasing)
    constexpr int N_C64 = XCHG ? (int)C_OX : (int)N_C64_ALL, N_C32 = XCHG ? (int)K_BLK : (int)N_C32_ALL;
    __shared__ double c64[N_C64][NCOL];
    __shared__ uint32_t c32[N_C32][NCOL];
    __shared__ uint32_t s_steps[WGT];               // per thread: steps of the rays it finished (RaytraceInfo)
    // the pool of parked rays (XCHG): per slot the ray's 40 hot dwords (or, free, just the spare column in word 38), and its tag
    __shared__ u32x4 s_pool[NPOOL ? NPOOL : 1u][10];
    constexpr uint32_t NTAG = NPOOL > 128u ? 192u : (NPOOL > 64u ? 128u : 64u);
    __shared__ uint32_t s_tag[NTAG];                // (padded to a multiple of 64: one to three tags per lane are read; the entries past the pool stay BUSY)
    __shared__ u32x4 s_census;                      // parked rays by kind: STEP, SHADE, ENTER, RAY (advisory: read without a claim)
    __shared__ uint8_t s_pick[WGT / 64u][2][64];    // per wave: the slots a round's givers are paired with (wanted kind; free)
    const uint32_t tid = threadIdx.x;
    uint32_t col = tid;  // the LDS column holding this lane's ray
    s_steps[tid] = 0u;
    if (!XCHG) c32[XCHG ? 0 : (int)K_BLK][tid] = 0u;
    c32[K_PXY][tid] = 0u;
    if (XCHG) {
        for (uint32_t i = tid; i < NTAG; i += WGT) s_tag[i] = i < NPOOL ? TAG_FREE : TAG_BUSY;
        for (uint32_t i = tid; i < NPOOL; i += WGT) {
            s_pool[i][9] = u32x4{0u, 0u, WGT + i, 0u};  // the slot's spare column
            c32[K_PXY][WGT + i] = 0u;
        }
        if (tid == 0u) s_census = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    uint32_t dry = 0u;                              // wave-uniform, 0 / 1: this wave has seen the tile queue exhausted (a word, not a bool: a bool that the compiler
                                                    // cannot prove uniform is kept as a lane mask, and every read of it costs six instructions)
    // The tile queue this wave takes from (DevFrame::n_queues > 1): its XCD's own to begin with, the next one's when that is empty. How many queues it has
    // seen empty is a word of LDS per wave, read once per tile -- kept in a scalar register for the life of the wave it cost the production variant
    // four spilled VGPRs (12 bytes of scratch per lane, 1.2 GB of scratch written back per C3 frame: profiles/r04_experiments.txt K).
    __shared__ uint32_t s_queues_tried[WGT / 64u];
    __shared__ uint32_t s_tile_state[WGT / 64u][4];  // (XCHG) the wave's tile_x0, tile_y0, next_idx between its RAY phases (registers elsewhere)
    __shared__ uint32_t s_idle_spins[WGT / 64u];  // (XCHG) rounds in which the wave found only rays in transit: bounded, so that nothing can hang
    if (lane == 0u) { s_queues_tried[threadIdx.x >> 6] = 0u; s_idle_spins[threadIdx.x >> 6] = 0u; }
    SurfDiag pend_d;
    double pend_t = 0.0;
    bool pend_visible = false;
    if (DIAG) {
        pend_d.nlight = 0; pend_d.res = pend_d.face = pend_d.block = 0;
        for (int a = 0; a < 3; a++) pend_d.cube[a] = pend_d.voxel[a] = 0;
    }
    // pixel bookkeeping: in LDS (K_PXY, K_S*); the antialiasing sample being traced rides in st bits 14-15
    uint32_t px_steps = 0, px_steps_prev = 0;          // DIAG: steps of this pixel
    Diag dg;
    if (DIAG) {
        dg.n_outer = dg.n_inner = dg.n_hits = dg.n_light = 0; dg.layer = 0; dg.hit = 0; dg.res = dg.face = dg.block = 0; dg.t = 0.0;
        for (int a = 0; a < 3; a++) dg.cube[a] = dg.voxel[a] = 0;
    }
    uint32_t tot_outer = 0, tot_inner = 0, tot_hits = 0, tot_light = 0;

    uint32_t ev = EV_NEWRAY | EV_TAKE;  // every lane starts by taking a pixel
#ifdef AIC_PROFILE
    // the counters live in the little LDS the kernel leaves free: as registers they would spill the stepping loop
    __shared__ uint32_t s_prof[WGT / 64u][40];
    __shared__ uint32_t s_ray_t0[NCOL];  // per ray: the clock when it started
    uint32_t ray_dur_max = 0u, ray_dur_steps = 0u; // per lane: the longest ray's duration and its step count
    uint32_t tail_trips = 0u, tail_events = 0u, tail_lanes = 0u;  // -DAIC_TAIL_PROF: after the wave saw the queue dry: trips, event phases, lanes stepping at trip start
    uint32_t *const prof = s_prof[tid >> 6];
    if (lane < 40u) prof[lane] = 0u;
    uint32_t prof_tm = (uint32_t)__builtin_readcyclecounter();
    const uint32_t prof_t0 = prof_tm;
#define AIC_PROF(i, v) { const uint32_t v_ = (uint32_t)(v); if (lane == 0u) prof[i] += v_; }
#ifdef AIC_TAIL_PROF
// phase clocks after the wave saw the queue dry go to slots 24.. instead (12 step, 13+18 shade, 14 enter, 15-17 ray, 19 scheduler)
#define AIC_TICK(i) { const uint32_t now_ = (uint32_t)__builtin_readcyclecounter(); const int j_ = (i) == 12 ? 24 : (i) == 13 || (i) == 18 ? 25 : (i) == 14 ? 26 : (i) == 19 ? 28 : 27; \
                      if (lane == 0u) prof[__builtin_amdgcn_readfirstlane((int)dry) ? j_ : (i)] += now_ - prof_tm; prof_tm = now_; }
#else
#define AIC_TICK(i) { const uint32_t now_ = (uint32_t)__builtin_readcyclecounter(); if (lane == 0u) prof[i] += now_ - prof_tm; prof_tm = now_; }
#endif
#else
#define AIC_PROF(i, v)
#define AIC_TICK(i)
#endif
// -DAIC_SECTION_MARKS: comments in the listing that tools/listing_lines.py --sections splits the instruction stream by (static counts per section; the
// markers are volatile asm statements and pin a little of the schedule, so this is a reading aid, not the production build)
#ifdef AIC_SECTION_MARKS
#define AIC_SECTION(name) asm volatile("; AIC_SECTION " #name ::: "memory")
#else
#define AIC_SECTION(name)
#endif
    uint32_t next_idx = F.tile * F.tile;  // wave-uniform: next unassigned pixel of tile_cur (tile_px = tile exhausted)
    if (XCHG && lane == 0u) { s_tile_state[tid >> 6][0] = 0u; s_tile_state[tid >> 6][1] = 0u; s_tile_state[tid >> 6][2] = next_idx; }

    // FACE_TABLE (raycast.rs:618-623) applied late: the Face of the cube the level is in
    auto face_now = [&]() -> uint32_t {
        if (lax & 8u) return lax & 7u;
        // the sign of the ray along a stepped axis is its octant bit (a stepped axis has a non-zero direction)
        const uint32_t positive = (st >> (26u - lax)) & 1u;
        return (positive ? 1u : 4u) + lax;
    };
    // per-ray constants of the Raycaster, rebuilt from the stored direction when an event needs them
    auto make_rd = [&](double dx, double dy, double dz) {
        RayDir r;
        r.dx = dx; r.dy = dy; r.dz = dz;
        r.tdx = tdx; r.tdy = tdy; r.tdz = tdz;
        r.sx = signum_101(dx); r.sy = signum_101(dy); r.sz = signum_101(dz);
        r.fast = (st & ST_DIR_FAST) != 0u;
        return r;
    };
    // coordinate (relative to the level's lower corner) from the steps left along an axis
    // (the counters hold steps left MINUS ONE, so that running out is the borrow of the decrement: see dda_step)
    auto coord = [](uint32_t positive, int size, uint32_t r) -> int { return positive ? size - 1 - (int)r : (int)r; };

    // -- State::step (raycast.rs:577-626) along the axis of the smallest t_max (strict <, ties to the
    //    later axis: raycast.rs:584-596): X iff tx<ty && tx<tz, Y iff !(tx<ty) && ty<tz, else Z.
    //    One exec-masked run per axis: last_t = t; t += t_delta; steps_left -= 1; offset += stride. --
    auto dda_step = [&](const unsigned long long m_who) -> unsigned long long {
        unsigned long long sv, mx, bz, by, bx;
        // last_t = the smallest t_max, whichever axis holds it: Z iff tz is that minimum (ties go to the later axis),
        // Y iff ty is and tz is not, X otherwise -- two v_min and two compares instead of three compares and three copies.
        // The steps-left counters are biased by one, so "ran out" is the borrow of the decrement (the carry-out of an
        // exec-masked v_sub_co is zero for the lanes it does not run on). Keeping the stepped axis in wave masks instead
        // of `lax` (3 vector instructions less, 6 scalar more) was measured and is slower: profiles/r03_experiments.txt I.
        asm volatile(
            "s_and_saveexec_b64 %[sv], %[m]\n\t"
            "v_min_f64 %[lt], %[tx], %[ty]\n\t"
            "v_min_f64 %[lt], %[lt], %[tz]\n\t"
            "v_cmp_eq_f64 %[mx], %[tz], %[lt]\n\t"        // Z
            "v_cmp_eq_f64 vcc, %[ty], %[lt]\n\t"
            "s_andn2_b64 vcc, vcc, %[mx]\n\t"             // Y
            "s_mov_b64 exec, %[mx]\n\t"
            "v_add_f64 %[tz], %[tz], %[tdz]\n\t"
            "v_sub_co_u32 %[rz], %[bz], %[rz], 1\n\t"
            "v_add_u32 %[bo], %[bo], %[ssz]\n\t"
            "v_mov_b32 %[lax], 2\n\t"
            "s_or_b64 %[mx], %[mx], vcc\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_add_f64 %[ty], %[ty], %[tdy]\n\t"
            "v_sub_co_u32 %[ry], %[by], %[ry], 1\n\t"
            "v_add_u32 %[bo], %[bo], %[ssy]\n\t"
            "v_mov_b32 %[lax], 1\n\t"
            "s_andn2_b64 exec, %[m], %[mx]\n\t"           // X = stepping lanes that took neither
            "v_add_f64 %[tx], %[tx], %[tdx]\n\t"
            "v_sub_co_u32 %[rx], %[bx], %[rx], 1\n\t"
            "v_add_u32 %[bo], %[bo], %[ssx]\n\t"
            "v_mov_b32 %[lax], 0\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_or_b64 %[bz], %[bz], %[by]\n\t"
            "s_or_b64 %[bz], %[bz], %[bx]\n\t"
            : [tx] "+v"(tx), [ty] "+v"(ty), [tz] "+v"(tz), [lt] "+v"(last_t), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz),
              [bo] "+v"(boff), [lax] "+v"(lax), [sv] "=&s"(sv), [mx] "=&s"(mx),
              [bz] "=&s"(bz), [by] "=&s"(by), [bx] "=&s"(bx)
            : [tdx] "v"(tdx), [tdy] "v"(tdy), [tdz] "v"(tdz), [ssx] "v"(ssx), [ssy] "v"(ssy), [ssz] "v"(ssz), [m] "s"(m_who)
            : "vcc", "scc");  // (the scalar mask operations write SCC)
        return bz;  // lanes whose level ran out of steps: it left its bounds
    };
    // (XCHG) counts a round that could do nothing; true once the wave has had 2^22 of them (a bug's hang would cost the GPU box). The count is cumulative over the
    // wave's life, not reset by progress (a reset would be a masked LDS store in every round): an idle round sleeps >= 256 cycles, so the bound is >= 10^9 cycles --
    // half a second -- of ONE wave's life spent waiting for rays in transit, against frames of milliseconds with a few such rounds per wave
    // (profiles/r05_phase_cycles.txt: 649 / 4415 in a whole C2 / C3 frame). A wave that does give up says so: DevCounters::bailed makes the frame fail on the host.
    auto spun_out = [&]() -> bool {
        const uint32_t n_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_idle_spins[threadIdx.x >> 6]) + 1u;
        if (lane == 0u) s_idle_spins[threadIdx.x >> 6] = n_;
        if (n_ > (1u << 22)) {
            // (the argument through the opaque kernel-argument pointer, like the event phase's: a plain use of `F` here would hold the pointer in registers for the life of the wave)
            typedef const __attribute__((address_space(4))) DevFrame KFrameB;
            KFrameB *Fb = (KFrameB *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(Fb));
            if (lane == 0u) atomicAdd(&Fb->sub[Fb->n_sub > 1u ? (blockIdx.x & (Fb->n_sub - 1u)) : 0u].counters->bailed, 1ull);
            return true;
        }
        return false;
    };
    for (;;) {
        AIC_SECTION(scheduler);
        // ---- wave scheduler: step, or run ONE kind of parked work for all lanes waiting on it ----
        // Kinds: SHADE (light + composite a surface), ENTER (a block), RAY (finish / start a ray). A
        // kind is run when enough lanes wait on it to fill the wave reasonably (AIC_T_BATCH), or when
        // so few lanes can still step (AIC_N_FEW) that waiting longer only idles the wave.
        unsigned long long m_st = __ballot(ev < 4u);
        unsigned long long b_shade = __ballot((ev & EV_SHADE) != 0u);
        unsigned long long b_enter = __ballot((ev & EV_ENTER) != 0u);
        unsigned long long b_ray = __ballot((ev & (EV_FINISH | EV_NEWRAY)) != 0u);
        int n_step = (int)wave_popc(m_st);
        int c_shade = (int)wave_popc(b_shade), c_enter = (int)wave_popc(b_enter), c_ray = (int)wave_popc(b_ray);
        uint32_t run = 0u;  // kind to run this trip (an EV_* bit), 0 = step
        if constexpr (XCHG) {
            // ---- regime-sorted waves: run the kind that fills the wave best, own lanes plus what the workgroup's pool can add ----
            // what this round's exchange (if any) decided: the lanes whose claim succeeded, those of them that took a spare column, the slot's address and
            // index, the tag the slot is released under, the kind taken (the swap itself sits on the round's common path, below)
            unsigned long long x_got = 0ull, x_fresh = 0ull;
            // (left undefined in a round without an exchange -- nothing reads them under an empty mask -- rather than zeroed: four vector moves in every round)
            // (two vector registers and a scalar carried to the swap: each one zeroed in a round without an exchange is an instruction in every round)
            uint32_t x_paddr = 0u, x_slot_tag = 0u;  // the slot's address; its index | the released tag << 8
            uint32_t x_want = 0u;                    // (wave-uniform)
            // a wave that is full of one kind as it stands (AIC_XCHG_FULL lanes: at most one kind can be) runs it without looking at the pool
            static_assert(AIC_XCHG_FULL > 32, "two kinds could both be full");
            static_assert(EV_SHADE == (2u << 1) && EV_ENTER == (2u << 2) && EV_FINISH == (2u << 3), "the kinds' event bits");
            bool settled = true;  // the kind's own lanes are known to be there (no look at the counts after the exchange)
            {
                // (the partial maxima through opaque_s: fused, the three become a v_max3 with two copies in front and a v_readfirstlane behind)
                const int m01 = opaque_s(n_step > c_shade ? n_step : c_shade), m23 = opaque_s(c_enter > c_ray ? c_enter : c_ray);
                if ((m01 > m23 ? m01 : m23) >= AIC_XCHG_FULL) run = c_ray >= AIC_XCHG_FULL ? EV_FINISH : (c_enter >= AIC_XCHG_FULL ? EV_ENTER : (c_shade >= AIC_XCHG_FULL ? EV_SHADE : 0u));
                else run = 0xffffffffu;
            }
            if (run == 0xffffffffu) {
            // parked rays by kind, read at the top of the round. A count is never BEHIND the tags (a parked ray is counted before its slot is released, and taken
            // off after it is claimed), so it is never negative and "every count zero" means that no claimable ray is parked; it can run AHEAD of them (a ray
            // claimed by another wave and not yet taken off): then the kind chosen for it may find nothing, which the end of the round handles.
            // (A plain LDS load behind a compiler barrier, so that it is made again in every round: through a `volatile` pointer it was compiled as a FLAT load --
            //  address-space inference leaves volatile accesses alone -- followed by s_waitcnt vmcnt(0), i.e. every scheduler round waited for every store the
            //  wave still had in flight. Made here, not at the top of the round: a wave that is full of one kind does not wait for counts it does not read.)
            asm volatile("" ::: "memory");
            const u32x4 cs = s_census;
            const int pk_step = __builtin_amdgcn_readfirstlane((int)cs.x), pk_shade = __builtin_amdgcn_readfirstlane((int)cs.y),
                      pk_enter = __builtin_amdgcn_readfirstlane((int)cs.z), pk_ray = __builtin_amdgcn_readfirstlane((int)cs.w);
            // the kind with the most lanes -- its own + the parked rays that fit into the wave's other lanes; ties go to the events, the later kind first (a
            // parked event lane blocks its ray, a stepping lane can wait): the four totals as keys  total << 9 | kind << 7 | own lanes  and one maximum
            // (kind: 0 stepping, 1 SHADE, 2 ENTER, 3 RAY; EV_SHADE / EV_ENTER / EV_FINISH = 2 << kind)
            auto key_of = [](int own, int pk, int kind) -> int {
                const int all = own + pk;
                return ((all < 64 ? all : 64) << 9) | (kind << 7) | own;
            };
            const int key0 = key_of(n_step, pk_step, 0), key1 = key_of(c_shade, pk_shade, 1), key2 = key_of(c_enter, pk_enter, 2), key3 = key_of(c_ray, pk_ray, 3);
            const int ka = opaque_s(key0 > key1 ? key0 : key1), kb = opaque_s(key2 > key3 ? key2 : key3), kmax = ka > kb ? ka : kb;
            if (kmax < 512) break;  // nothing of its own and nothing parked: whoever parks a ray later is running and serves it
            const int best = kmax >> 9, kind = (kmax >> 7) & 3, mine = kmax & 127;
            run = kind == 0 ? 0u : (2u << kind);
            settled = mine != 0;
            AIC_TICK(19);
            const int dry_i = __builtin_amdgcn_readfirstlane((int)dry);
            const int alive = n_step + c_shade + c_enter + c_ray, parked = pk_step + pk_shade + pk_enter + pk_ray;
            const int n_others = alive - mine - ((AIC_XCHG_DEPOSIT < 3 && run != 0u) ? n_step : 0);  // lanes holding a ray that will not run now (and may be parked)
            // An exchange costs a few hundred instructions: it is made for a top-up of at least AIC_XCHG_MIN_GAIN lanes, for a kind the wave has none of, or -- while
            // the image has pixels left and the pool a free slot -- to park at least AIC_XCHG_PARK_MIN lanes:
            //     best - mine >= MIN_GAIN  ||  mine == 0  ||  (!dry && parked < NPOOL && n_others >= PARK_MIN)
            // as one comparison of integers that are >= 0 where their term holds (a dozen scalar instructions; as booleans every term is a compare, a 64-bit
            // select and a 64-bit and / or)
            static_assert(AIC_XCHG_PARK_MIN > 0 && AIC_XCHG_DEPOSIT != 0, "parking wants lanes to park");
            const int park_room = (int)NPOOL - 1 - parked, not_dry = -dry_i;
            const int park_ok_i = park_room < not_dry ? park_room : not_dry;                             // >= 0: parking is possible
            const int t_park = n_others - AIC_XCHG_PARK_MIN, t_gain = best - mine - AIC_XCHG_MIN_GAIN, t_none = -opaque_s(mine);  // (opaque: else the negation is made on the vector unit, for the carry that says mine != 0)
            const int t_p = park_ok_i < t_park ? park_ok_i : t_park;
            const int t_a = opaque_s(t_gain > t_none ? t_gain : t_none);  // (kept apart: fused, the two maxima become a v_max3 with copies in front and a vector compare behind)
            if ((t_a > t_p ? t_a : t_p) >= 0) {
                AIC_SECTION(exchange);
                const bool may_park = park_ok_i >= 0 && n_others > 0;
                // ---- the exchange: lanes that would idle ("givers": empty lanes first, then rays of other kinds) are paired with parked rays of the wanted
                // kind, and -- while the image has pixels left -- what remains of them with free slots. A pairing is a claim (compare-and-swap of the slot's
                // tag), a plain swap of the 40 hot dwords and the column index, and the release of the slot under the tag of what it now holds. ----
                // (the lane number through an empty asm: what is derived from it -- LDS addresses of the tags and lists, comparisons with the pool size -- is
                //  otherwise computed before the persistent loop and kept in registers the event code needs)
                uint32_t ln = lane;
                asm volatile("" : "+v"(ln));
                // (written as straight-line selects and exec-masked stores: per-lane `if`s and short-circuit conditions cost a saved / restored exec mask each, and
                //  assigning the hot variables inside a branch makes the compiler copy all of them on the path that does not take it -- in every round)
                const uint32_t want = run == 0u ? TAG_STEP : (run == EV_SHADE ? TAG_SHADE : (run == EV_ENTER ? TAG_ENTER : TAG_RAY));
                const bool e_done = ev == EV_DONE, e_take = ev == (EV_NEWRAY | EV_TAKE);
                const bool empty = e_done | e_take;
                uint32_t my_tag = (ev & (EV_FINISH | EV_NEWRAY)) != 0u ? TAG_RAY : TAG_STEP;
                my_tag = (ev & EV_ENTER) != 0u ? TAG_ENTER : my_tag;
                my_tag = (ev & EV_SHADE) != 0u ? TAG_SHADE : my_tag;
                my_tag = empty ? TAG_FREE : my_tag;
                const bool giver = (my_tag != want) & (!e_take | (want != TAG_RAY));  // (an empty lane about to take a pixel IS the RAY kind)
                const unsigned long long m_giv = __builtin_amdgcn_ballot_w64(giver);
                auto below = [](unsigned long long m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
                const uint32_t n_giv = wave_popc(m_giv);
                const uint32_t rank = below(m_giv);  // (givers in lane order)
                // the slots' tags, three per lane (the array is padded to 192 with BUSY); the first 64 slots of the wanted kind and the first 64 free ones go
                // into the wave's pairing lists
                const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)((tid - ln) >> 6));
                const uint32_t t0 = __hip_atomic_load(&s_tag[ln], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t t1 = NPOOL > 64u ? __hip_atomic_load(&s_tag[ln + 64u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : TAG_BUSY;
                const uint32_t t2 = NPOOL > 128u ? __hip_atomic_load(&s_tag[ln + 128u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : TAG_BUSY;
                const unsigned long long w0 = __builtin_amdgcn_ballot_w64(t0 == want), w1 = __builtin_amdgcn_ballot_w64(t1 == want), w2 = __builtin_amdgcn_ballot_w64(t2 == want);
                // (every wave would otherwise list the same lowest slots first, and two waves exchanging at the same moment would fight over them: odd waves
                //  list from the top down)
                const uint32_t nw0 = wave_popc(w0), nw1 = wave_popc(w1), nw2 = wave_popc(w2), n_w = nw0 + nw1 + nw2;
                const bool down = (wv & 1u) != 0u;
                uint8_t *const pick_w = s_pick[wv][0], *const pick_f = s_pick[wv][1];
                { const uint32_t u_ = below(w0), i_ = down ? n_w - 1u - u_ : u_; if ((t0 == want) & (i_ < 64u)) pick_w[i_] = (uint8_t)ln; }
                if (NPOOL > 64u) { const uint32_t u_ = nw0 + below(w1), i_ = down ? n_w - 1u - u_ : u_; if ((t1 == want) & (i_ < 64u)) pick_w[i_] = (uint8_t)(ln + 64u); }
                if (NPOOL > 128u) { const uint32_t u_ = nw0 + nw1 + below(w2), i_ = down ? n_w - 1u - u_ : u_; if ((t2 == want) & (i_ < 64u)) pick_w[i_] = (uint8_t)(ln + 128u); }
                uint32_t n_f = 0u;
                if (may_park) {  // (uniform) the free slots, for the lanes that park
                    const unsigned long long f0 = __builtin_amdgcn_ballot_w64(t0 == TAG_FREE), f1 = __builtin_amdgcn_ballot_w64(t1 == TAG_FREE), f2 = __builtin_amdgcn_ballot_w64(t2 == TAG_FREE);
                    const uint32_t nf0 = wave_popc(f0), nf1 = wave_popc(f1);
                    n_f = nf0 + nf1 + wave_popc(f2);
                    { const uint32_t u_ = below(f0), i_ = down ? n_f - 1u - u_ : u_; if ((t0 == TAG_FREE) & (i_ < 64u)) pick_f[i_] = (uint8_t)ln; }
                    if (NPOOL > 64u) { const uint32_t u_ = nf0 + below(f1), i_ = down ? n_f - 1u - u_ : u_; if ((t1 == TAG_FREE) & (i_ < 64u)) pick_f[i_] = (uint8_t)(ln + 64u); }
                    if (NPOOL > 128u) { const uint32_t u_ = nf0 + nf1 + below(f2), i_ = down ? n_f - 1u - u_ : u_; if ((t2 == TAG_FREE) & (i_ < 64u)) pick_f[i_] = (uint8_t)(ln + 128u); }
                }
                __builtin_amdgcn_wave_barrier();  // (the lists are written by some lanes and read by others of this wave: LDS operations of a wave are performed in order; this keeps the compiler from reordering them)
                const uint32_t n_take = n_w < n_giv ? n_w : n_giv;  // (at most 64)
                const uint32_t n_free = n_f < 64u ? n_f : 64u;
                // a giver of rank < n_take takes a parked ray; the rays of the ranks after that are parked while slots are free
                const bool takes = giver & (rank < n_take);
                const uint32_t j_ = rank - n_take;
                const bool parks = giver & !empty & (AIC_XCHG_DEPOSIT >= 3 || my_tag != TAG_STEP) & (rank >= n_take) & (j_ < n_free);  // (n_free is 0 unless may_park)
                const uint32_t expect = takes ? want : TAG_FREE;
                const uint32_t slot = (takes | parks) ? (uint32_t)(takes ? pick_w : pick_f)[takes ? rank : (j_ & 63u)] : 0u;
                bool got = false;
                if (takes | parks) {
                    uint32_t seen = expect;
                    got = __hip_atomic_compare_exchange_strong(&s_tag[slot], &seen, TAG_BUSY, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // (nothing of a ray lives in global memory except the antialiasing sums, and whoever stores those waits for the store at once: cold_sums_store)
                x_got = __builtin_amdgcn_ballot_w64(got);
                x_fresh = x_got & ~__builtin_amdgcn_ballot_w64(takes);
                x_paddr = (uint32_t)(uintptr_t)&s_pool[0][0] + slot * 160u;
                x_slot_tag = slot | (my_tag << 8); x_want = want;
                AIC_PROF(37, wave_popc(__builtin_amdgcn_ballot_w64(takes | parks)) - wave_popc(x_got));  // claims lost to another wave
                if (x_got == 0ull) { AIC_PROF(38, 1); }  // an exchange that moved nothing
            // (the two masks through v_readfirstlane: they are wave-uniform by construction, but a build in which the compiler's divergence analysis loses that --
            //  the -DAIC_PROFILE one did -- would hand vector registers to the scalar operands below, and to every mask of the stepping phase after them)
            {
                auto uniform64 = [](unsigned long long m) -> unsigned long long {
                    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
                };
                x_got = uniform64(x_got);
                x_fresh = uniform64(x_fresh);
            }
            // ---- The swap proper, under exec = the lanes whose claim succeeded: one LDS exchange (`ds_wrxchg_rtn`: write the register, return what was there)
            // per hot variable, IN PLACE -- like the stepping code's asm blocks. (Assigned from loaded values in C++, the compiler renames the hot variables
            // and copies all 38 registers around each block. Earlier builds kept these blocks on the round's common path, skipped by an empty mask, because
            // inside the exchange's branch the same copies appeared at the branch; with the in-place asm they do not, and a round without an exchange -- four
            // in five -- no longer passes through them; profiles/r05_experiments.txt L has the listings' instruction counts.) ----
            if (x_got != 0ull) {
                unsigned long long sv;
                asm volatile(
                    "s_and_saveexec_b64 %[sv], %[m]\n\t"
                    "s_cbranch_execz .Lxa%=\n\t"
                    "ds_wrxchg_rtn_b64 %[x0], %[p], %[x0]\n\t"
                    "ds_wrxchg_rtn_b64 %[x1], %[p], %[x1] offset:8\n\t"
                    "ds_wrxchg_rtn_b64 %[x2], %[p], %[x2] offset:16\n\t"
                    "ds_wrxchg_rtn_b64 %[x3], %[p], %[x3] offset:24\n\t"
                    "ds_wrxchg_rtn_b64 %[x4], %[p], %[x4] offset:32\n\t"
                    "ds_wrxchg_rtn_b64 %[x5], %[p], %[x5] offset:40\n\t"
                    "ds_wrxchg_rtn_b64 %[x6], %[p], %[x6] offset:48\n\t"
                    "s_waitcnt lgkmcnt(0)\n"
                    ".Lxa%=:\n\t"
                    "s_mov_b64 exec, %[sv]\n\t"
                    : [x0] "+v"(tx), [x1] "+v"(ty), [x2] "+v"(tz), [x3] "+v"(last_t), [x4] "+v"(tdx), [x5] "+v"(tdy), [x6] "+v"(tdz), [sv] "=&s"(sv)
                    : [p] "v"(x_paddr), [m] "s"(x_got)
                    : "memory", "scc");
                asm volatile(
                    "s_and_saveexec_b64 %[sv], %[m]\n\t"
                    "s_cbranch_execz .Lxb%=\n\t"
                    "ds_wrxchg_rtn_b32 %[w0], %[p], %[w0] offset:56\n\t"
                    "ds_wrxchg_rtn_b32 %[w1], %[p], %[w1] offset:60\n\t"
                    "ds_wrxchg_rtn_b32 %[w2], %[p], %[w2] offset:64\n\t"
                    "ds_wrxchg_rtn_b32 %[w3], %[p], %[w3] offset:68\n\t"
                    "ds_wrxchg_rtn_b32 %[w4], %[p], %[w4] offset:72\n\t"
                    "ds_wrxchg_rtn_b32 %[w5], %[p], %[w5] offset:76\n\t"
                    "ds_wrxchg_rtn_b32 %[w6], %[p], %[w6] offset:80\n\t"
                    "ds_wrxchg_rtn_b32 %[w7], %[p], %[w7] offset:84\n\t"
                    "ds_wrxchg_rtn_b32 %[w8], %[p], %[w8] offset:88\n\t"
                    "ds_wrxchg_rtn_b32 %[w9], %[p], %[w9] offset:92\n\t"
                    "ds_wrxchg_rtn_b32 %[wa], %[p], %[wa] offset:96\n\t"
                    "ds_wrxchg_rtn_b32 %[wb], %[p], %[wb] offset:100\n\t"
                    "s_waitcnt lgkmcnt(0)\n"
                    ".Lxb%=:\n\t"
                    "s_mov_b64 exec, %[sv]\n\t"
                    : [w0] "+v"(rx), [w1] "+v"(ry), [w2] "+v"(rz), [w3] "+v"(boff), [w4] "+v"(ssx), [w5] "+v"(ssy), [w6] "+v"(ssz), [w7] "+v"(thr), [w8] "+v"(raw),
                      [w9] "+v"(lax), [wa] "+v"(st), [wb] "+v"(count), [sv] "=&s"(sv)
                    : [p] "v"(x_paddr), [m] "s"(x_got)
                    : "memory", "scc");
                asm volatile(
                    "s_and_saveexec_b64 %[sv], %[m]\n\t"
                    "s_cbranch_execz .Lxc%=\n\t"
                    "ds_wrxchg_rtn_b32 %[w0], %[p], %[w0] offset:104\n\t"
                    "ds_wrxchg_rtn_b32 %[w1], %[p], %[w1] offset:108\n\t"
                    "ds_wrxchg_rtn_b32 %[w2], %[p], %[w2] offset:112\n\t"
                    "ds_wrxchg_rtn_b32 %[w3], %[p], %[w3] offset:116\n\t"
                    "ds_wrxchg_rtn_b32 %[w4], %[p], %[w4] offset:120\n\t"
                    "ds_wrxchg_rtn_b32 %[w5], %[p], %[w5] offset:124\n\t"
                    "ds_wrxchg_rtn_b32 %[w6], %[p], %[w6] offset:128\n\t"
                    "ds_wrxchg_rtn_b32 %[w7], %[p], %[w7] offset:132\n\t"
                    "ds_wrxchg_rtn_b32 %[w8], %[p], %[w8] offset:136\n\t"
                    "ds_wrxchg_rtn_b32 %[w9], %[p], %[w9] offset:140\n\t"
                    "ds_wrxchg_rtn_b32 %[wa], %[p], %[wa] offset:144\n\t"
                    "ds_wrxchg_rtn_b32 %[wb], %[p], %[wb] offset:148\n\t"
                    "ds_wrxchg_rtn_b32 %[wc], %[p], %[wc] offset:152\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_mov_b64 exec, %[f]\n\t"            // the lanes that took a spare column: an empty lane about to take a pixel
                    "v_mov_b32 %[w0], 0xa0\n"             // ev = EV_NEWRAY | EV_TAKE
                    ".Lxc%=:\n\t"
                    "s_mov_b64 exec, %[sv]\n\t"
                    : [w0] "+v"(ev), [w1] "+v"(blk_pal_off), [w2] "+v"(blk_geo), [w3] "+v"(blk_vsz), [w4] "+v"(acc.l0), [w5] "+v"(acc.l1), [w6] "+v"(acc.l2), [w7] "+v"(acc.t),
                      [w8] "+v"(pend0), [w9] "+v"(pend1), [wa] "+v"(pend2), [wb] "+v"(pend_tr), [wc] "+v"(col), [sv] "=&s"(sv)
                    : [p] "v"(x_paddr), [m] "s"(x_got), [f] "s"(x_fresh)
                    : "memory", "scc");
                static_assert((EV_NEWRAY | EV_TAKE) == 0xa0u, "the literal in the asm above");
                const bool got = __builtin_amdgcn_inverse_ballot_w64(x_got), fresh = __builtin_amdgcn_inverse_ballot_w64(x_fresh);
                // the counts: every lane that parked a ray adds it to its kind (LDS atomics on one word each: a few cycles per lane, two instructions) BEFORE
                // its slot is released under the tag of what it now holds -- LDS operations of a wave are performed in order, so no wave can claim the ray, and
                // take it off the count, before it is on it: a count is never negative; one lane takes off what was taken
                const uint32_t n_picked = wave_popc(x_got & ~x_fresh);
                const uint32_t x_slot = x_slot_tag & 255u, x_tag = x_slot_tag >> 8;
                if (got && x_tag != TAG_FREE) atomicAdd(reinterpret_cast<uint32_t *>(&s_census) + (x_tag - 1u), 1u);
                if (got) __hip_atomic_store(&s_tag[x_slot], x_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0u && n_picked != 0u) atomicSub(reinterpret_cast<uint32_t *>(&s_census) + (x_want - 1u), n_picked);
                (void)fresh;
                AIC_PROF(31, 1);
                AIC_PROF(32, wave_popc(x_got));
                AIC_PROF(34, n_picked);
                AIC_PROF(35, wave_popc(__builtin_amdgcn_ballot_w64(got && x_tag != TAG_FREE)));
                m_st = __ballot(ev < 4u);
                b_shade = __ballot((ev & EV_SHADE) != 0u);
                b_enter = __ballot((ev & EV_ENTER) != 0u);
                b_ray = __ballot((ev & (EV_FINISH | EV_NEWRAY)) != 0u);
                n_step = (int)wave_popc(m_st);
                c_shade = (int)wave_popc(b_shade); c_enter = (int)wave_popc(b_enter); c_ray = (int)wave_popc(b_ray);
                AIC_TICK(33);
            }
            }
            }
            else { AIC_TICK(19); }
            AIC_SECTION(scheduler_tail);
            if (!settled && (run == 0u ? n_step : (run == EV_SHADE ? c_shade : (run == EV_ENTER ? c_enter : c_ray))) == 0) {
                // (the kind was chosen for what the pool holds, and every claim was lost to another wave -- or only rays in transit were found)
                AIC_PROF(36, 1);
                if (spun_out()) break;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
        } else {
        if ((m_st | b_shade | b_enter | b_ray) == 0ull) break;
        {
            int best = c_shade;
            uint32_t kind = EV_SHADE;
            if (c_enter > best) { best = c_enter; kind = EV_ENTER; }
            if (c_ray > best) { best = c_ray; kind = EV_FINISH; }
            // thresholds scale with the lanes still alive, so that a wave that is running out of rays
            // (the frame's tail) keeps batching instead of running every event for a lane or two
            const int alive = n_step + c_shade + c_enter + c_ray;
#ifndef AIC_FRAC_T
#define AIC_FRAC_T 4  // eighths of the lanes alive
#endif
#ifndef AIC_FRAC_N
#define AIC_FRAC_N 3
#endif
            const int part_t = (alive * AIC_FRAC_T) >> 3, part_n = (alive * AIC_FRAC_N) >> 3;
            const int t_lo = opaque_s(part_t > 0 ? part_t : 1);  // (kept apart from the min: fused, the pair becomes a v_med3 and a v_readfirstlane)
            const int t_batch = t_lo < AIC_T_BATCH ? t_lo : AIC_T_BATCH;
            const int n_few = part_n < AIC_N_FEW ? part_n : AIC_N_FEW;
            if (best > 0 && (best >= t_batch || n_step <= n_few)) run = kind;
        }
        AIC_TICK(19);
        }
        if (run != 0u) {
            AIC_SECTION(event_prologue);
            // ============================ event phase ======================================
            // the kernel arguments, through an opaque pointer (see the top of the kernel): `F`, `L`, `opt` below shadow the
            // by-value parameter for the whole phase
            typedef const __attribute__((address_space(4))) DevFrame KFrame;
            KFrame *Fq = (KFrame *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(Fq));
            KFrame &F = *Fq;
            const auto &L = F.layer;
            const auto &opt = L.opt;
            // the frame this workgroup belongs to (DevSub: a launch may trace several; everything per frame is a scalar load at a uniform offset)
            typedef const __attribute__((address_space(4))) DevSub KSub;
            KSub *Sq = reinterpret_cast<KSub *>(reinterpret_cast<const __attribute__((address_space(4))) char *>(Fq) + offsetof(DevFrame, sub)) +
                       (F.n_sub > 1u ? (blockIdx.x & (F.n_sub - 1u)) : 0u);
            asm volatile("" : "+s"(Sq));
            KSub &S = *Sq;
            const bool ui_pass = F.pass == 1;
            const bool include_sky = !ui_pass;
            const bool fog_on = (opt.fog != 0) && include_sky;
            const size_t npix = (size_t)F.width * F.local_rows;
            const int n_samples = F.antialias ? 4 : 1;
            const int olx = L.lo[0], oly = L.lo[1], olz = L.lo[2];
            const int osx_i = L.size[0], osy_i = L.size[1], osz_i = L.size[2];
            const uint32_t tile_px = F.tile * F.tile;  // pixels per tile: 256 or 64
            const uint32_t macro_shift = (uint32_t)__ffs((int)F.macro) - 1u;
            const uint32_t macro_px_shift = macro_shift + (uint32_t)__ffs((int)F.tile) - 1u;  // log2 of a macro tile's edge in pixels
            const uint32_t n_virtual = (F.macros_x * F.macros_y) << (macro_shift * 2u);  // work items incl. the macro tiles' overhang
            // sky colour seen along this ray (Sky::sample, sky.rs:32-41); the octant was fixed at ray start. An octant sky is
            // read per lane from the kernel-argument segment as ordinary memory: indexing the array per lane in registers
            // would make the compiler keep a copy of it in scratch.
            auto sky_now = [&](float out[3]) {
                if (L.sky_kind != 0) {
                    const float *p = (const float *)((const char *)(const void *)Fq + offsetof(DevFrame, layer) + offsetof(DevLayer, sky)) + 3u * ((st >> 24) & 7u);
                    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
                } else {
                    out[0] = L.sky[0][0]; out[1] = L.sky[0][1]; out[2] = L.sky[0][2];
                }
            };
            AIC_PROF(0, 1);
            AIC_PROF(1, c_shade + c_enter + c_ray);
#ifdef AIC_TAIL_PROF
            if (__builtin_amdgcn_readfirstlane((int)dry)) { tail_events++; AIC_PROF(30, 1); }
#endif
            AIC_PROF(4, run == EV_SHADE ? 1 : 0); AIC_PROF(5, run == EV_SHADE ? c_shade : 0);
            AIC_PROF(6, run == EV_ENTER ? 1 : 0); AIC_PROF(7, run == EV_ENTER ? c_enter : 0);
            AIC_PROF(8, run == EV_FINISH ? 1 : 0); AIC_PROF(9, run == EV_FINISH ? c_ray : 0);
            // -- The ray of a pixel: RtScene::trace_patch's sample point (renderer.rs:424-451) through Camera::project_ndc_into_world
            //    (camera_struct.rs:238-257), or MultiOrthoCamera::project_pixel_into_world for orthographic views. `o` = the ray's origin and, if asked
            //    for, `dir` = its direction as given (not yet sanitised). NEWRAY runs this when the ray starts; the exchanging variants run it AGAIN
            //    (origin only) wherever a later event needs the origin, instead of keeping 24 bytes per ray for the whole of its life: same code, same
            //    inputs (the pixel in K_PXY, the sample in st, the kernel arguments), hence the same bits. Returns false when there is no ray. --
            auto ray_of_pixel = [&](const uint32_t x, const uint32_t lrow, const int sample, const bool want_dir, double o[3], double dir[3]) -> bool {
                // The common cameras first, behind ONE scalar fetch (DevFrame::ray_mode, made by the launcher): a pixel grid with the host-made edge tables and a layer
                // that holds a space. (The general form below asks five kernel arguments one after the other -- strips? pixel centres? patches? tables? orthographic? --
                // each a scalar load and a wait in front of its branch: ~600 cycles of NEWRAY's and ENTER's few thousand.) Same operations on the same values as below.
                const uint32_t mode = F.ray_mode;
                if (mode <= 1u) {
                    const double *const ex = F.edge_x, *const ey = F.edge_y;
                    uint32_t y = lrow;
                    if (mode == 1u) {
                        const uint32_t srows = opaque_s(F.strip_rows);
                        const uint32_t strip = lrow / srows;
                        y = (F.part + strip * F.n_parts) * srows + (lrow - strip * srows);
                    }
                    const double x0 = ex[x], x1 = ex[x + 1u], y0 = ey[y], y1 = ey[y + 1u];
                    double px, py;
                    if (n_samples == 4) {
                        const double ux = (sample == 0) ? 1. / 8. : (sample == 1) ? 3. / 8. : (sample == 2) ? 5. / 8. : 7. / 8.;
                        const double uy = (sample == 0) ? 5. / 8. : (sample == 1) ? 1. / 8. : (sample == 2) ? 7. / 8. : 3. / 8.;
                        px = x0 + (x1 - x0) * ux;
                        py = y0 + (y1 - y0) * uy;
                    } else {
                        px = (x0 + x1) / 2.0;
                        py = (y0 + y1) / 2.0;
                    }
                    unproject(S.inv, px, py, 0.0, o);
                    if (want_dir) {
                        double f[3];
                        unproject(S.inv, px, py, 1.0, f);
                        dir[0] = f[0] - o[0]; dir[1] = f[1] - o[1]; dir[2] = f[2] - o[2];
                    }
                    return true;
                }
                const size_t pix = (size_t)lrow * F.width + x;
                // global row of this local row under the strip partition
                uint32_t y = lrow;
                if (F.n_parts > 1u) {
                    const uint32_t srows = opaque_s(F.strip_rows);
                    const uint32_t strip = lrow / srows;
                    y = (F.part + strip * F.n_parts) * srows + (lrow - strip * srows);
                }
                double px, py;  // renderer.rs:428-433 sample points, else the patch centre
                if (F.pixel_centers) {  // Viewport::normalize_fb_x / _y (viewport.rs:89-99): the text renderer's rays
                    const uint32_t fw = opaque_s(F.width), fh = opaque_s(F.height);
                    px = ((double)x + 0.5) / (double)fw * 2.0 - 1.0;
                    py = -(((double)y + 0.5) / (double)fh * 2.0 - 1.0);
                } else {
                    double x0, x1, y0, y1;  // the pixel's NdcRect {min: (x0, y0), max: (x1, y1)} (renderer.rs:537-550)
                    if (F.patches) {
                        const double *r = F.patches + 4u * pix;
                        x0 = r[0]; y0 = r[1]; x1 = r[2]; y1 = r[3];
                    } else if (F.edge_x) {
                        // fb_x_edge / fb_y_edge of the pixel's two edges each, from the frame's tables (DevFrame::edge_x: made by the same operations)
                        x0 = F.edge_x[x]; x1 = F.edge_x[x + 1u]; y0 = F.edge_y[y]; y1 = F.edge_y[y + 1u];
                    } else {
                        const uint32_t fw = opaque_s(F.width), fh = opaque_s(F.height);
                        x0 = fb_x_edge(fw, x); x1 = fb_x_edge(fw, x + 1);
                        y0 = fb_y_edge(fh, y); y1 = fb_y_edge(fh, y + 1);
                    }
                    if (n_samples == 4) {
                        const double ux = (sample == 0) ? 1. / 8. : (sample == 1) ? 3. / 8. : (sample == 2) ? 5. / 8. : 7. / 8.;
                        const double uy = (sample == 0) ? 5. / 8. : (sample == 1) ? 1. / 8. : (sample == 2) ? 7. / 8. : 3. / 8.;
                        px = x0 + (x1 - x0) * ux;
                        py = y0 + (y1 - y0) * uy;
                    } else {
                        px = (x0 + x1) / 2.0;
                        py = (y0 + y1) / 2.0;
                    }
                }
                bool have_ray = L.present != 0;
                o[0] = o[1] = o[2] = 0.0;
                if (want_dir) dir[0] = dir[1] = dir[2] = 0.0;
                if (have_ray && F.ortho_n) {
                    // MultiOrthoCamera::project_pixel_into_world (ortho.rs:186-199, 284-294): the view whose rectangle holds
                    // the pixel, its transform applied to the pixel, then TryFrom<Ray> for AaRay / From<AaRay> for Ray
                    // (ray.rs:305-358: the origin becomes cube + f32 offset, the direction the unit axis vector).
                    // trace_axis_aligned_ray produces what trace_ray produces on that Ray (axis_aligned.rs:8-9).
                    int vsel = -1;
                    for (int v = 0; v < F.ortho_n; v++) {
                        const uint32_t vx = F.ortho[v].x0, vy = F.ortho[v].y0, vw = F.ortho[v].w, vh = F.ortho[v].h;
                        if (vsel < 0 && x >= vx && y >= vy && x - vx < vw && y - vy < vh) vsel = v;
                    }
                    have_ray = false;
                    if (vsel >= 0) {
                        const DevOrthoView *V = &F.ortho[vsel];
                        double p[3];
                        unproject(V->m, (double)(x - V->x0), (double)(y - V->y0), 0.0, p);
                        int cube[3];
                        if (cube_containing(p, cube)) {
                            have_ray = true;
                            for (int a = 0; a < 3; a++) {
                                o[a] = (double)cube[a] + (double)(float)(p[a] - (double)cube[a]);
                                if (want_dir) dir[a] = V->dir[a];
                            }
                        }
                    }
                } else if (have_ray) {
                    unproject(S.inv, px, py, 0.0, o);
                    if (want_dir) {
                        double f[3];
                        unproject(S.inv, px, py, 1.0, f);
                        dir[0] = f[0] - o[0]; dir[1] = f[1] - o[1]; dir[2] = f[2] - o[2];
                    }
                }
                return have_ray;
            };
            // The exchanging variants' global part of the cold state: the four antialiasing sums, 16 bytes per column of this workgroup, in frames traced with
            // antialiasing only. Loads are workgroup-scope atomics (`sc0`: past the CU's vector L1, which a store made by another wave of the workgroup does
            // not update -- the sums may last have been written by a lane of another wave -- and served by the L2; agent scope, `sc1`, goes past the XCD's L2
            // as well). A store is waited for at once: the ray may be parked and claimed by another wave in the very next round.
            // Layout: per workgroup four arrays of NCOL floats: a wave's lanes read neighbouring words (columns stay mostly in lane order).
            char *const cold_wg = XCHG ? reinterpret_cast<char *>(F.ray_cold) + (size_t)blockIdx.x * (size_t)(NCOL * 16u) : nullptr;
            auto off32 = [&](uint32_t k) -> uint32_t { return (k * NCOL + col) * 4u; };
            auto cold_origin = [&](double &ox, double &oy, double &oz) {
                if constexpr (XCHG) {
                    const uint32_t pxy_ = c32[K_PXY][col];
                    double o_[3];
                    (void)ray_of_pixel(pxy_ & 0xffffu, pxy_ >> 16, (int)((st >> 14) & 3u), false, o_, nullptr);
                    ox = o_[0]; oy = o_[1]; oz = o_[2];
                } else {
                    ox = c64[XCHG ? 0 : (int)C_OX][col]; oy = c64[XCHG ? 0 : (int)C_OY][col]; oz = c64[XCHG ? 0 : (int)C_OZ][col];
                }
            };
            auto cold_direction = [&](double &dx, double &dy, double &dz) { dx = c64[C_DX][col]; dy = c64[C_DY][col]; dz = c64[C_DZ][col]; };
            auto cold_sums_load = [&](float v[4]) {
                if constexpr (XCHG) {
                    for (uint32_t k = 0; k < 4u; k++) v[k] = __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t *>(cold_wg + off32(k)), __ATOMIC_RELAXED, AIC_COLD_SCOPE));
                } else {
                    v[0] = __uint_as_float(c32[XCHG ? 0 : (int)K_S0][col]); v[1] = __uint_as_float(c32[XCHG ? 0 : (int)K_S1][col]);
                    v[2] = __uint_as_float(c32[XCHG ? 0 : (int)K_S2][col]); v[3] = __uint_as_float(c32[XCHG ? 0 : (int)K_ST][col]);
                }
            };
            auto cold_sums_store = [&](float v0, float v1, float v2, float v3) {
                if constexpr (XCHG) {
                    *reinterpret_cast<float *>(cold_wg + off32(0u)) = v0; *reinterpret_cast<float *>(cold_wg + off32(1u)) = v1;
                    *reinterpret_cast<float *>(cold_wg + off32(2u)) = v2; *reinterpret_cast<float *>(cold_wg + off32(3u)) = v3;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (in L2 before another wave can be handed the ray)
                } else {
                    c32[XCHG ? 0 : (int)K_S0][col] = __float_as_uint(v0); c32[XCHG ? 0 : (int)K_S1][col] = __float_as_uint(v1);
                    c32[XCHG ? 0 : (int)K_S2][col] = __float_as_uint(v2); c32[XCHG ? 0 : (int)K_ST][col] = __float_as_uint(v3);
                }
            };
            const uint32_t posx = (st >> 26) & 1u, posy = (st >> 25) & 1u, posz = (st >> 24) & 1u;
            // -- shading a discovered surface: compute_illumination + trace_through_span +
            //    Surface::to_light (surface.rs:73-206; sr.rs:697-740). For Volumetric transparency the
            //    span's exit distance is the t of the ray's NEXT TraceStep, which is already fixed when
            //    the surface is discovered: it is the t_max of the step the level takes next. The
            //    contribution is therefore computed here in full and merely *applied* by the stepping code
            //    when that next step is counted (so the order count -> stop-check -> accumulate is kept). --
            // (one kind runs per phase: each kind's section sits behind a UNIFORM branch on `run`, the per-lane test inside it -- written as one combined condition
            //  the copies that carry a section's conditionally updated variables to the merge behind it are made in every phase, whichever kind runs)
            if (run == EV_SHADE) {
            AIC_SECTION(shade_geometry);
            if (ev & EV_SHADE) {
                const bool inb = (st & ST_IN_BLOCK) != 0;
                const uint32_t blk_res = 1u << (blk_geo >> 24), blk_vlo = blk_geo & 0xffffffu;
                const double as = inb ? __hiloint2double((int)((1023u - (blk_geo >> 24)) << 20), 0) : 1.0;  // 1 / resolution
                const double t_enter = last_t * as;  // surface.rs:385-386
                const uint32_t s_rx = c32[K_SRX][col], s_ry = c32[K_SRY][col], s_rz = c32[K_SRZ][col];
                // the Space cube, and the current level's cube in absolute coordinates
                const int ocx = coord(posx, osx_i, inb ? s_rx : rx) + olx, ocy = coord(posy, osy_i, inb ? s_ry : ry) + oly,
                          ocz = coord(posz, osz_i, inb ? s_rz : rz) + olz;
                Lvl ca;
                ca.tx = tx; ca.ty = ty; ca.tz = tz; ca.last_t = last_t; ca.st = face_now() << 2;
                if (inb) {
                    ca.cx = coord(posx, (int)(blk_vsz & 255u), rx) + (int)(blk_vlo & 255u);
                    ca.cy = coord(posy, (int)((blk_vsz >> 8) & 255u), ry) + (int)((blk_vlo >> 8) & 255u);
                    ca.cz = coord(posz, (int)((blk_vsz >> 16) & 255u), rz) + (int)((blk_vlo >> 16) & 255u);
                } else {
                    ca.cx = ocx; ca.cy = ocy; ca.cz = ocz;
                }
                // colour record of the surface: the block's single voxel, or the voxel's palette entry
                const uint32_t shade_ref = inb ? (blk_pal_off + raw) : (0x80000000u | (raw & idx_mask));
                // ... fetched HERE, ahead of the light: both records begin with rgba and emission (DevBlock, DevPaletteEntry: 32 bytes), so it is one address and
                // two loads without a branch, and they are in flight beside the light's texel loads. (Written where the record is used, the compiler issued them
                // there -- behind the wait for the texels: two memory round trips in a row in every SHADE phase; round 6.)
                static_assert(offsetof(DevBlock, color) == 0 && offsetof(DevBlock, emission) == 16 && offsetof(DevPaletteEntry, color) == 0 && offsetof(DevPaletteEntry, emission) == 16, "one record layout");
                const char *const rec_p = (shade_ref & 0x80000000u) ? reinterpret_cast<const char *>(&L.blocks[shade_ref & 0xffffu]) : reinterpret_cast<const char *>(&L.palette[shade_ref]);
                const float4 rec_col = *reinterpret_cast<const float4 *>(rec_p);
                const float4 rec_em = *reinterpret_cast<const float4 *>(rec_p + 16);  // (.w: DevBlock::kind / the palette entry's pad -- unused)
                // illumination (surface.rs:113-206)
                float i0 = 1.0f, i1 = 1.0f, i2 = 1.0f;
                uint32_t nl = 0;
                double ip[3] = {0.0, 0.0, 0.0};  // the surface point in space coordinates (LMODE 2: light interpolation; 3: the bounce rays' origin)
                AIC_SECTION(shade_light);
                if (LMODE != 0) {
                    const int face = lvl_face(ca);
                    if (LMODE == 1 || LMODE == 3) {  // Flat; Bounce falls back to it for surfaces that are not fully opaque (surface.rs:171-176)
                        int nx = 0, ny = 0, nz = 0;
                        if (face == 1) nx = -1; else if (face == 2) ny = -1; else if (face == 3) nz = -1;
                        else if (face == 4) nx = 1; else if (face == 5) ny = 1; else if (face == 6) nz = 1;
                        const uint32_t txl = get_packed_light<DIAG>(L, ocx + nx, ocy + ny, ocz + nz, nl);
                        i0 = lut[txl & 255u]; i1 = lut[(txl >> 8) & 255u]; i2 = lut[(txl >> 16) & 255u];
                    }
                    if (LMODE >= 2) {
                        // (the origin enters intersection_point only for a lane whose step is Face7::Within or whose ray does not move along some axis --
                        //  raycast.rs:409-439 -- : it is fetched for those lanes, i.e. hardly ever; half of the SHADE event's cold-state traffic)
                        double ox = 0.0, oy = 0.0, oz = 0.0, dx, dy, dz;  // (the origin as the lane's LEVEL sees it: the block's own coordinates inside a block)
                        cold_direction(dx, dy, dz);
                        const bool need_o = (face == FACE_WITHIN) | (dx == 0.0) | (dy == 0.0) | (dz == 0.0);
                        if (__ballot(need_o) != 0ull) {
                            if (need_o) {
                                cold_origin(ox, oy, oz);
                                if (inb) {
                                    const double kd = (double)blk_res;
                                    ox = (ox - (double)ocx) * kd; oy = (oy - (double)ocy) * kd; oz = (oz - (double)ocz) * kd;
                                }
                            }
                        }
                        // one copy of intersection_point for both levels (round 6: it was written per level, and a wave shading cube faces and voxel faces
                        // together ran both); a voxel's point goes back to space coordinates (surface.rs:406-407)
                        double vp[3];
                        intersection_point(ca, ox, oy, oz, dx, dy, dz, vp);
                        ip[0] = inb ? vp[0] * as + (double)ocx : vp[0];
                        ip[1] = inb ? vp[1] * as + (double)ocy : vp[1];
                        ip[2] = inb ? vp[2] * as + (double)ocz : vp[2];
                    }
                    if (LMODE == 2) {
                        // get_interpolated_light (sr.rs:248-359; aic_lightmath.h), then rgb / max(weight, 0.1)
                        float fin[4];
                        lm_interpolated_light(light_view(L), lut, ocx, ocy, ocz, ip[0], ip[1], ip[2], face, opt.lighting, fin, DIAG ? &nl : nullptr);
                        const float w = fmaxf(fin[3], 0.1f);
                        if (__ballot(w != 1.0f) == 0ull) {  // x / 1.0f == x: fully lit neighbourhoods (the usual case) need no division
                            i0 = fin[0]; i1 = fin[1]; i2 = fin[2];
                        } else {
                            // Three quotients by the same w in [0.1, 1]: the compiler's own correctly rounded f32 division (v_rcp_f32, one Newton step on the
                            // reciprocal, q = x r, two residual corrections, a last fused one) with the reciprocal and its refinement made ONCE and without the
                            // operand scaling and the special-case fix-up, which do nothing here: w is a normal number near one, x is a light value -- zero, or
                            // between 2^-15 times an interpolation weight and 2^12 --, so v_div_scale_f32 leaves both alone and no operand is infinite or NaN
                            // (18 instructions instead of 33, one quarter-rate reciprocal instead of three; round 6).
                            const float r0 = __builtin_amdgcn_rcpf(w);
                            const float rr = __builtin_fmaf(__builtin_fmaf(-w, r0, 1.0f), r0, r0);
                            auto quot = [&](const float x) -> float {
                                float q = x * rr;
                                q = __builtin_fmaf(__builtin_fmaf(-w, q, x), rr, q);
                                return __builtin_fmaf(__builtin_fmaf(-w, q, x), rr, q);
                            };
                            i0 = quot(fin[0]); i1 = quot(fin[1]); i2 = quot(fin[2]);
                        }
                    }
                }
                AIC_TICK(18);
                AIC_SECTION(shade_colour_span);
                // colour record
                float r = rec_col.x, g = rec_col.y, b = rec_col.z, a = rec_col.w, e0 = rec_em.x, e1 = rec_em.y, e2 = rec_em.z;
                bool will_flush = true;
                if (VOL) {
                    // exit distance = t of the next TraceStep: the next step of this level, or -- if this
                    // level cannot step any more -- of the enclosing cube grid; none => the span is never emitted
                    double t_exit = 0.0;
                    if (!(ev & EV_DEAD)) {
                        const int pk = pick_axis(tx, ty, tz);
                        t_exit = (pk == 0 ? tx : (pk == 1 ? ty : tz)) * as;
                    } else if (inb && (st & ST_OUTER_ALIVE)) {
                        const double s_tx = c64[C_STX][col], s_ty = c64[C_STY][col], s_tz = c64[C_STZ][col];
                        const int pk = pick_axis(s_tx, s_ty, s_tz);  // the suspended outer level's next step
                        t_exit = pk == 0 ? s_tx : (pk == 1 ? s_ty : s_tz);
                    } else {
                        will_flush = false;
                    }
                    // trace_through_span (sr.rs:720-740) + apply_transmittance (raytracer_components.rs:215-258)
                    float thickness = (float)((t_exit - t_enter) * c64[C_TABS][col]);
                    thickness = fmaxf(thickness, 0.0f);
                    float coeff;
                    if (thickness == 0.0f) {
                        if (a == 1.0f) coeff = 1.0f;
                        else { r = g = b = a = 0.f; coeff = 0.0f; }
                    } else {
                        const float unit_t = 1.0f - a;
                        // powf(0, y>0) == 0 and powf(1, y) == 1 exactly: skip the general evaluation
                        float depth_t;
                        if (unit_t == 0.0f) depth_t = 0.0f;
                        else if (unit_t == 1.0f) depth_t = 1.0f;
                        else if (!(thickness < __uint_as_float(0x7f800000u))) depth_t = 0.0f;  // powf(0 < x < 1, +inf) == 0
                        else depth_t = powf_table(unit_t, thickness, s_pow);  // 0 < unit_t < 1 normal (aic_upload_* reject alpha outside [0, 1]), thickness > 0 finite
                        a = zo_clamped(1.0f - depth_t);
                        const float ec = (unit_t == 1.0f) ? thickness : (depth_t - 1.f) / (unit_t - 1.f);
                        coeff = fmaxf(ec, 0.0f);
                    }
                    const float c = ps_clamped(coeff);
                    e0 = ps_mul(e0, c); e1 = ps_mul(e1, c); e2 = ps_mul(e2, c);
                }
                AIC_SECTION(shade_to_light_fog);
                // Surface::to_light (surface.rs:73-106)
                if (opt.transparency == 2) {  // limit_alpha (graphics_options.rs:496-507)
                    if (a > opt.threshold) a = 1.0f;
                    else { r = g = b = a = 0.f; }
                }
                // (Volumetric: only for a span that WILL be emitted -- the reference lights a surface when its span comes out of
                //  DepthIter and passes the stop check, i.e. at the ray's next counted step: sr.rs:183-199)
                if (LMODE == 3 && a == 1.0f && (!VOL || (will_flush && count <= 999u))) {
                    // compute_illumination with the RNG and a fully opaque diffuse colour (surface.rs:85-88, 119-166): `samples` secondary
                    // rays from just above the surface, directions normal + UnitSphere sample; their mean replaces the Flat light
                    const int face = lvl_face(ca);
                    double nvx = 0.0, nvy = 0.0, nvz = 0.0;
                    if (face == 1) nvx = -1.0; else if (face == 2) nvy = -1.0; else if (face == 3) nvz = -1.0;
                    else if (face == 4) nvx = 1.0; else if (face == 5) nvy = 1.0; else if (face == 6) nvz = 1.0;
                    const float *sky_mem = (const float *)((const char *)(const void *)Fq + offsetof(DevFrame, layer) + offsetof(DevLayer, sky));
                    const uint32_t n_samples_b = (uint32_t)opt.bounce_samples & 255u;  // `samples: u8`
                    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
                    for (uint32_t k = 0; k < n_samples_b; k++) {
                        double u[3];
                        bounce_unit_sphere(brng, u);
                        float c3[3];
                        sec_steps += bounce_secondary_ray<DIAG>(L, sky_mem, lut, s_pow, BIG, ip[0] + nvx * 0.0001, ip[1] + nvy * 0.0001, ip[2] + nvz * 0.0001,
                                                                nvx + u[0], nvy + u[1], nvz + u[2], c3);
                        m0 += c3[0]; m1 += c3[1]; m2 += c3[2];
                    }
                    const float kk = ps_clamped(1.0f / (float)n_samples_b);  // Rgb * f32::from(samples).recip() (color.rs:912-925)
                    i0 = ps_mul(m0, kk); i1 = ps_mul(m1, kk); i2 = ps_mul(m2, kk);
                    nl = 0;  // (the Flat texel read above is not a get_packed_light call of the reference's for this surface)
                }
                float o0 = 0.f, o1 = 0.f, o2 = 0.f, tr = 1.0f;
                const bool visible = !(a == 0.f && e0 == 0.f && e1 == 0.f && e2 == 0.f);
                if (visible) {
                    o0 = ps_mul(ps_mul(r, i0), a) + e0;
                    o1 = ps_mul(ps_mul(g, i1), a) + e1;
                    o2 = ps_mul(ps_mul(b, i2), a) + e2;
                    tr = 1.0f - a;
                    if (fog_on) {  // distance_fog (sr.rs:745-768)
                        float sky[3];
                        sky_now(sky);
                        const float fog_blend = opt.fog == 1 ? 1.0f : (opt.fog == 2 ? 0.5f : 0.0f);
                        float rel = (float)t_enter * __uint_as_float(c32[K_TVIEW][col]);
                        rel = rel < 0.0f ? 0.0f : (rel > 1.0f ? 1.0f : rel);
                        const float sq = rel * rel;
                        float amount;
                        if (opt.fog == 1) {
                            // Abrupt: blend == 1, so the exponential term is (finite, >= 0) * 0.0 == +0.0 and
                            // +0.0 + rel^4 * 1.0 == rel^4 exactly -- no exp needed
                            amount = zo_clamped((sq * sq) * 1.0f);
                        } else {
                            const float fog_exp = 1.0f - expf_table(-1.6f * rel, s_pow);
                            const float fudged = fog_exp / 0.79810348f;
                            amount = zo_clamped(fudged * (1.0f - fog_blend) + (sq * sq) * fog_blend);
                        }
                        const float comp = 1.0f - amount;
                        o0 = ps_mul(o0, comp) + ps_mul(sky[0], amount);
                        o1 = ps_mul(o1, comp) + ps_mul(sky[1], amount);
                        o2 = ps_mul(o2, comp) + ps_mul(sky[2], amount);
                        tr *= comp;
                    }
                }
                AIC_SECTION(shade_apply);
                SurfDiag sd;
                if (DIAG) {
                    sd.nlight = nl;
                    sd.cube[0] = ocx; sd.cube[1] = ocy; sd.cube[2] = ocz;
                    if (inb) {
                        sd.voxel[0] = ca.cx; sd.voxel[1] = ca.cy; sd.voxel[2] = ca.cz;
                        sd.res = (int)blk_res; sd.block = (int)c32[XCHG ? 0 : (int)K_BLK][col];
                    } else {
                        sd.voxel[0] = sd.voxel[1] = sd.voxel[2] = 0;
                        sd.res = 1; sd.block = (int)(shade_ref & 0xffffu);
                    }
                    sd.face = lvl_face(ca);
                }
                if (VOL) {
                    // DepthIter.last_surface: applied when the next TraceStep is counted
                    if (will_flush) {
#ifndef AIC_EARLY_APPLY
#define AIC_EARLY_APPLY 1
#endif
                        // The span is applied when the ray's next TraceStep is counted and passes the stop check (sr.rs:625-656).
                        // For a lane whose level can step and that is below the step cap that is CERTAIN: its next step (a lookup or the
                        // exit step) is produced, counted (count + 1 <= 1000) and not stopped (the ray is not opaque, or it would not be
                        // shading). Nothing reads the accumulator in between, so the same additions can be made now -- unless they make
                        // the ray opaque: the stop check of that next step still has to see it transparent, so then the span waits as
                        // before. A lane without a pending span takes the bookkeeping-free fast steps at once.
                        bool early = AIC_EARLY_APPLY && !DIAG && !(ev & EV_DEAD) && count <= 999u;
                        const float n0 = acc.l0 + o0 * acc.t, n1 = acc.l1 + o1 * acc.t, n2 = acc.l2 + o2 * acc.t, nt = acc.t * tr;
                        early = early && !(nt < 1.0f / 256.0f);
#ifndef AIC_OPAQUE_SHORTCUT
#define AIC_OPAQUE_SHORTCUT 1
#endif
                        // A span that makes the ray opaque decides the rest of the ray, whatever lies behind it: the ray's next step is
                        // counted and applies the span (the accumulator is opaque from there), the step after that -- if the iterators
                        // still yield one -- is counted and stops the ray (sr.rs:183-189, 625-656). What those two steps find is
                        // irrelevant (a surface would never be lit, an EnterBlock's second item is that stopping step), only whether they
                        // exist: a level that is not over always yields a next step (a cell or its exit step), and a second one unless that
                        // next step is the exit step of the cube grid, or of a block whose enclosing grid level has ended. So the lane
                        // finishes here: two lookups, two full stepping passes and a scheduler round trip less for every ray that ends
                        // on a solid. (Not the aux-recording variant: its per-level lookup counters follow the reference's iterators.)
                        const bool finish_now = AIC_OPAQUE_SHORTCUT && !DIAG && !(ev & EV_DEAD) && count <= 998u && (nt < 1.0f / 256.0f);
                        if (finish_now) {
                            const int pk = pick_axis(tx, ty, tz);
                            const uint32_t r_next = pk == 0 ? rx : (pk == 1 ? ry : rz);  // steps left on that axis minus one: 0 = the exit step is next
                            const bool second = (r_next != 0u) || (inb && (st & ST_OUTER_ALIVE));
                            acc.l0 = n0; acc.l1 = n1; acc.l2 = n2; acc.t = nt;
                            count += second ? 2u : 1u;
                            st |= ST_OPAQUE;
                            ev = (ev & EV_SHADE) | EV_FINISH;  // (EV_SHADE is cleared below)
                        } else if (early) {
                            acc.l0 = n0; acc.l1 = n1; acc.l2 = n2; acc.t = nt;
                        } else {
                            pend0 = o0; pend1 = o1; pend2 = o2; pend_tr = tr;
                            st |= ST_HAS_LAST;
                        }
                        if (DIAG) { pend_d = sd; pend_t = t_enter; pend_visible = visible; }
                    }
                } else if (visible) {
                    cb_add(acc, o0, o1, o2, tr);  // trace_through_surface (sr.rs:697-717)
                    if (cb_opaque(acc)) {
                        st |= ST_OPAQUE;
                        // (as above, Surface transparency: the ray's next step is counted and stops it; a level that is not over yields one)
                        if (AIC_OPAQUE_SHORTCUT && !DIAG && !(ev & EV_DEAD) && count <= 999u) {
                            count += 1u;
                            ev = (ev & EV_SHADE) | EV_FINISH;
                        }
                    }
                    if (DIAG) {
                        dg.n_hits++;
                        dg.n_light += sd.nlight;
                        if (!dg.hit) {
                            dg.hit = 1;
                            dg.layer = ui_pass ? 1u : 0u;
                            for (int a2 = 0; a2 < 3; a2++) { dg.cube[a2] = sd.cube[a2]; dg.voxel[a2] = sd.voxel[a2]; }
                            dg.res = sd.res; dg.face = sd.face; dg.block = sd.block; dg.t = t_enter;
                        }
                    }
                }
                ev &= ~EV_SHADE;
            }
            AIC_SECTION(shade_end);
            } else if (run == EV_ENTER) {
            AIC_SECTION(enter);
            // -- entering a recursive block: RaycastStep::recursive_raycast (raycast.rs:458-476),
            //    advanced to its first in-bounds voxel (or to its end) --
            if (ev & EV_ENTER) {
                const uint32_t blk_index = raw & idx_mask;
                if (DIAG) c32[XCHG ? 0 : (int)K_BLK][col] = blk_index;
                const DevBlock *tb = &L.blocks[blk_index];
                const uint32_t blk_res = tb->kind & 255u;
                const uint32_t blk_vlo = tb->vlo_packed;
                blk_geo = ((31u - (uint32_t)__clz((int)blk_res)) << 24) | blk_vlo;
                blk_vsz = tb->vsize_packed;
                blk_pal_off = tb->pal_off;
                double ox, oy, oz, edx, edy, edz;
                cold_origin(ox, oy, oz);
                cold_direction(edx, edy, edz);
                const uint32_t n_invisible = tb->n_invisible;
                const uint32_t vox_off = tb->vox_off;
                const double kd = (double)blk_res;
                const int acx = coord(posx, osx_i, rx) + olx, acy = coord(posy, osy_i, ry) + oly, acz = coord(posz, osz_i, rz) + olz;
                const double sx_ = (ox - (double)acx) * kd, sy_ = (oy - (double)acy) * kd, sz_ = (oz - (double)acz) * kd;
                // suspend the outer level; its Face goes to st[16..18]
                c64[C_STX][col] = tx; c64[C_STY][col] = ty; c64[C_STZ][col] = tz;
                c32[K_SRX][col] = rx; c32[K_SRY][col] = ry; c32[K_SRZ][col] = rz; c32[K_SBOFF][col] = boff;
                st = (st & ~((7u << 16) | ST_OUTER_ALIVE)) | (face_now() << 16) | ((ev & EV_DEAD) ? 0u : ST_OUTER_ALIVE);
                const int ilx = (int)(blk_vlo & 255u), ily = (int)((blk_vlo >> 8) & 255u), ilz = (int)((blk_vlo >> 16) & 255u);
                const int isx = (int)(blk_vsz & 255u), isy = (int)((blk_vsz >> 8) & 255u), isz = (int)((blk_vsz >> 16) & 255u);
                const RayDir rd = make_rd(edx, edy, edz);
                const LvlLim ll = lvl_init(sx_, sy_, sz_, rd, true, ilx, ily, ilz, ilx + isx, ily + isy, ilz + isz, true, 0.5 / c64[C_TABS][col]);  // (0.5 / direction.length(), raycast.rs:669: the quotient NEWRAY's fast-forward used)
                const FirstCube f = lvl_first_masks(ll.s, rd, ilx, ily, ilz, ilx + isx, ily + isy, ilz + isz);
                tx = f.tx; ty = f.ty; tz = f.tz; last_t = f.last_t;
                const int vcx = f.cx - ilx, vcy = f.cy - ily, vcz = f.cz - ilz;
                rx = posx ? (uint32_t)(isx - 1 - vcx) : (uint32_t)vcx;
                ry = posy ? (uint32_t)(isy - 1 - vcy) : (uint32_t)vcy;
                rz = posz ? (uint32_t)(isz - 1 - vcz) : (uint32_t)vcz;
                boff = 2u * (vox_off + (uint32_t)(((uint32_t)vcx * (uint32_t)isy + (uint32_t)vcy) * (uint32_t)isz + (uint32_t)vcz));
                ssx = posx ? 2 * isy * isz : -2 * isy * isz; ssy = posy ? 2 * isz : -2 * isz; ssz = posz ? 2 : -2;
                thr = n_invisible;
                // a produced first voxel still needs its lookup (FRESH); a level that produced nothing, or ended with it, is DEAD
                lax = f.lax;
                st |= ST_IN_BLOCK;
                ev = (f.got ? EV_FRESH : 0u) | (f.inbounds ? 0u : EV_DEAD);
            }
            AIC_SECTION(enter_end);
            } else {
            AIC_SECTION(finish);
            // -- finishing a ray: TracingState::finish + layer tail + (last sample) encode & store --
            uint32_t pxy = 0;
            int sample = 0;
            bool want = false;
            if (run == EV_FINISH) {
                pxy = c32[K_PXY][col];
                sample = (ev & EV_TAKE) ? 0 : (int)((st >> 14) & 3u);  // (a lane about to take a pixel starts at sample 0 whatever its state word holds: a lane
                                                                         //  that parked its ray keeps the word of the slot's previous content)
            }
            if (run == EV_FINISH && (ev & EV_FINISH)) {
                if (st & ST_TRACED) {
                    // finish (sr.rs:658-693): the sky hit, then the optional cost visualisation
                    if (include_sky) {
                        float sky[3];
                        sky_now(sky);
                        cb_add(acc, sky[0] * 1.0f, sky[1] * 1.0f, sky[2] * 1.0f, 0.0f);
                    } else {
                        cb_add(acc, 0.f, 0.f, 0.f, 1.0f);
                    }
                    if (opt.debug_pixel_cost) {  // accum.rs:228-234
                        const float n = ps_clamped((float)count);
                        const float red = ps_clamped(ps_mul(0.02f, n) * 1.0f);
                        const float green = ps_clamped(ps_mul(0.002f, n) * 1.0f);
                        float cur_rgba[4];
                        cb_to_rgba(acc, cur_rgba);
                        const float blue = ps_clamped(luminance(cur_rgba[0], cur_rgba[1], cur_rgba[2]) * 0.2f);
                        acc.l0 = red; acc.l1 = green; acc.l2 = blue; acc.t = 0.0f;
                    }
                    { uint32_t t_ = threadIdx.x; asm volatile("" : "+v"(t_)); s_steps[t_] += count + (LMODE == 3 ? sec_steps : 0u); }  // (the address made here, not held across the loop)  // RaytraceInfo + secondary_info (sr.rs:690-692)
#ifdef AIC_PROFILE
                    { const uint32_t dur_ = (uint32_t)__builtin_readcyclecounter() - s_ray_t0[col];
                      if (dur_ > ray_dur_max) { ray_dur_max = dur_; ray_dur_steps = count; } }
#endif
                    if (DIAG) px_steps += count + (LMODE == 3 ? sec_steps : 0u);
                }
                const uint32_t x = pxy & 0xffffu, lrow = pxy >> 16;
                float aa_sum[4] = {0.f, 0.f, 0.f, 0.f};
                if (S.tile_cost && count > 48u) {
                    // longest ray of the macro tile so far. Only rays long enough to matter for the frame's
                    // tail are recorded (the rest leave their tile at cost 0: handed out last, in index
                    // order), and the plain read filters out almost every atomic.
                    uint32_t *tc = &S.tile_cost[(lrow >> macro_px_shift) * F.macros_x + (x >> macro_px_shift)];  // tile edge and macro are powers of two
                    if (count > *tc) atomicMax(tc, count);
                }
                const size_t pix = (size_t)lrow * F.width + x;
                if (ui_pass) {
                    S.acc_buf[(size_t)sample * npix + pix] = make_float4(acc.l0, acc.l1, acc.l2, acc.t);
                } else {
                    if (!F.ortho_n && !cb_opaque(acc)) {  // renderer.rs:474-477: P::paint(NO_WORLD_TO_SHOW) replaces the accumulator
                        // (render_orthographic has no such layer tail: ortho.rs:103-131)
                        acc.l0 = 0.f + (NO_WORLD_TO_SHOW * 1.0f) * 1.0f;
                        acc.l1 = acc.l0;
                        acc.l2 = acc.l0;
                        acc.t = 1.0f * (1.0f - 1.0f);
                    }
                    if (n_samples == 4) {
                        cold_sums_load(aa_sum);
                        aa_sum[0] += acc.l0; aa_sum[1] += acc.l1; aa_sum[2] += acc.l2; aa_sum[3] += acc.t;
                        if (sample < 3) cold_sums_store(aa_sum[0], aa_sum[1], aa_sum[2], aa_sum[3]);  // (the last sample's sums are used below and dropped)
                    }
                }
                sample++;
                if (sample < n_samples) {
                    ev = EV_NEWRAY;  // next antialiasing sample of the same pixel
                } else {
                    if (!ui_pass) {
                        ColorBuf pixel;
                        if (n_samples == 4) {  // ColorBuf::mean (raytracer_components.rs:97-102)
                            pixel.l0 = aa_sum[0] / 4.0f; pixel.l1 = aa_sum[1] / 4.0f; pixel.l2 = aa_sum[2] / 4.0f; pixel.t = aa_sum[3] / 4.0f;
                        } else {
                            pixel = acc;
                        }
                        // encoder: Camera::post_process_color(Rgba::from(buf)).to_srgb8()
                        float c[4];
                        cb_to_rgba(pixel, c);
                        if (F.out_mode != 0) {  // float outputs: the linear Rgba, or the ColorBuf as it is
                            reinterpret_cast<float4 *>(S.out)[pix] =
                                F.out_mode == 1 ? make_float4(c[0], c[1], c[2], c[3]) : make_float4(pixel.l0, pixel.l1, pixel.l2, pixel.t);
                        } else {
                        const float ex = S.exposure;
                        float r = ps_mul(c[0], ex), g = ps_mul(c[1], ex), bl = ps_mul(c[2], ex);
                        const float m = F.maximum_intensity;
                        if (isfinite(m)) {  // ToneMappingOperator::apply (graphics_options.rs:352-368)
                            if (F.tone_mapping == 0) {
                                r = r < 0.f ? 0.f : (r > m ? m : r);
                                g = g < 0.f ? 0.f : (g > m ? m : g);
                                bl = bl < 0.f ? 0.f : (bl > m ? m : bl);
                            } else {
                                const float scale = ps_clamped(1.0f / (1.0f + luminance(r, g, bl) / m));
                                r = ps_mul(r, scale); g = ps_mul(g, scale); bl = ps_mul(bl, scale);
                            }
                        }
                        uint32_t R, G, B;
                        srgb8_rgb(r, g, bl, s_thr, R, G, B);
                        const uint32_t A = round_sat_u8(c[3] * 255.0f);
                        S.out[pix] = R | (G << 8) | (B << 16) | (A << 24);
                        }
                    }
                    if (DIAG) {
                        if (F.aux) {
                            DevAux &a = F.aux[pix];
                            a.hit = dg.hit & 1;
                            for (int k = 0; k < 3; k++) { a.cube[k] = dg.cube[k]; a.voxel[k] = dg.voxel[k]; }
                            a.resolution = dg.res; a.face = dg.face; a.block_index = dg.block;
                            a.cubes_traced = px_steps_prev + px_steps; a.layer = dg.layer; a.t_distance = dg.t;
                        }
                        tot_outer += dg.n_outer; tot_inner += dg.n_inner; tot_hits += dg.n_hits; tot_light += dg.n_light;
                    }
                    sample = 0;
                    ev = EV_NEWRAY | EV_TAKE;
                }
            }
            AIC_SECTION(refill);
            if (run == EV_FINISH) { AIC_TICK(16) }
            // -- starting a ray: lane refill + Camera::project_ndc_into_world + Raycaster::within --
            if (run == EV_FINISH) {
                // wave-level refill (uniform control flow): hand the next unassigned pixels to the
                // lanes that finished a pixel -- ballot + prefix popcount -- pulling a fresh tile
                // from the global counter whenever the current one is used up.
                want = (ev & (EV_NEWRAY | EV_TAKE)) == (EV_NEWRAY | EV_TAKE);
                uint32_t wv_ = 0u;  // (XCHG) the wave's number, made here: derived before the loop, the address of its LDS words would be one more register held across it
                if constexpr (XCHG) {
                    wv_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                    asm volatile("" : "+s"(wv_));
                    tile_x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tile_state[wv_][0]);
                    tile_y0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tile_state[wv_][1]);
                    next_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tile_state[wv_][2]);
                }
                for (;;) {
                    const unsigned long long need = __ballot(want);
                    if (need == 0ull) break;
                    if (next_idx >= tile_px) {
                        // next work item: tiles are numbered macro tile by macro tile (macro x macro tiles
                        // each, row-major inside), and the macro tiles are taken in `tile_order` -- the
                        // previous frame's costliest first -- so that long rays start early while
                        // neighbouring tiles still run together and share their cache lines
                        uint32_t t = n_virtual;
                        if (!dry) {
                            const int leader = __ffsll((long long)need) - 1;
                            if (F.n_queues > 1u) {
                                // the XCD's own queue; when it is empty, the next XCD's (for good: `my_queue` moves on), until all have been seen empty
                                const uint32_t nq = F.n_queues;
                                uint32_t xcc;
                                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                                uint32_t queues_tried = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_queues_tried[threadIdx.x >> 6]);
                                uint32_t my_queue = ((xcc & 15u) + queues_tried) % nq;
                                const uint32_t tried_before = queues_tried;
                                while (queues_tried < nq) {
                                    uint32_t u = 0u;
                                    if ((int)lane == leader) u = atomicAdd(&S.counters->tile_next_q[my_queue][0], 1u);
                                    u = (uint32_t)__builtin_amdgcn_readlane((int)u, leader);
                                    const uint32_t q0 = S.queue_start[my_queue], q1 = S.queue_start[my_queue + 1u];
                                    const uint32_t j = u >> (macro_shift * 2u);
                                    if (j < q1 - q0) {
                                        t = ((q0 + j) << (macro_shift * 2u)) | (u & ((1u << (macro_shift * 2u)) - 1u));
                                        break;
                                    }
                                    my_queue = my_queue + 1u == nq ? 0u : my_queue + 1u;
                                    queues_tried++;
                                }
                                if (queues_tried != tried_before && lane == 0u) s_queues_tried[threadIdx.x >> 6] = queues_tried;
                            } else {
                                if ((int)lane == leader) t = atomicAdd(&S.counters->tile_next, 1u);
                                t = (uint32_t)__builtin_amdgcn_readlane((int)t, leader);  // (a scalar: what depends on it -- `dry` -- stays wave-uniform for the compiler)
                            }
                        }
                        if (t >= n_virtual) {  // image exhausted
#ifdef AIC_PROFILE
                            if (lane == 0u && prof[2] == 0u) prof[2] = (uint32_t)__builtin_readcyclecounter() - prof_t0;  // wave saw the queue run dry
#endif
                            dry = 1u;
                            next_idx = tile_px;
                            if (want) ev = EV_DONE;  // these lanes are done
                            break;
                        }
                        const uint32_t m_shift = macro_shift * 2u;
                        uint32_t mt = t >> m_shift;
                        const uint32_t inner = t & ((1u << m_shift) - 1u);
                        if (S.tile_order) mt = S.tile_order[mt];
                        // (divisors taken through opaque_s: a division by a loop-invariant value is otherwise expanded into a
                        //  reciprocal that is computed before the persistent loop and then lives in -- or is spilled from -- a VGPR)
                        const uint32_t mx_ = opaque_s(F.macros_x);
                        const uint32_t mrow = mt / mx_;
                        const uint32_t tx_ = ((mt - mrow * mx_) << macro_shift) + (inner & ((1u << macro_shift) - 1u));
                        const uint32_t ty_ = (mrow << macro_shift) + (inner >> macro_shift);
                        tile_x0 = tx_ * F.tile;
                        tile_y0 = ty_ * F.tile;
                        if (tile_x0 >= F.width || tile_y0 >= F.local_rows) continue;  // a macro tile's overhang past the image edge
                        next_idx = 0;
                    }
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));  // lanes of `need` below this one (v_mbcnt: no mask of the lanes below to keep in registers)
                    const uint32_t avail = tile_px - next_idx;
                    if (want && rank < avail) {
                        const uint32_t pidx = next_idx + rank;
                        // pixel order inside a tile: four 8x8 quadrants, row-major inside each
                        const uint32_t x = tile_x0 + (pidx & 7u) + (((pidx >> 6) & 1u) << 3);
                        const uint32_t lrow = tile_y0 + ((pidx >> 3) & 7u) + (((pidx >> 7) & 1u) << 3);
                        if (x < F.width && lrow < F.local_rows && (!F.patches || lrow * F.width + x < F.n_patches)) {  // pixels of partial tiles outside the image are skipped
                            pxy = x | (lrow << 16);
                            want = false;
                        }
                    }
                    const uint32_t n_need = wave_popc(need);
                    next_idx += n_need < avail ? n_need : avail;
                }
                if constexpr (XCHG) {
                    if (lane == 0u) { s_tile_state[wv_][0] = tile_x0; s_tile_state[wv_][1] = tile_y0; s_tile_state[wv_][2] = next_idx; }
                }
            }
            AIC_SECTION(newray);
            if (run == EV_FINISH) { AIC_TICK(17) }
            if (run == EV_FINISH && (ev & EV_NEWRAY) && ev != EV_DONE && !want) {
                const uint32_t x = pxy & 0xffffu, lrow = pxy >> 16;
                const size_t pix = (size_t)lrow * F.width + x;
                if (ev & EV_TAKE) {
                    if (n_samples == 4) { const float z_ = KF(0.f); cold_sums_store(z_, z_, z_, z_); }
                    if (DIAG) {
                        dg.n_outer = dg.n_inner = dg.n_hits = dg.n_light = 0; dg.layer = 0; dg.hit = 0; dg.res = dg.face = dg.block = 0; dg.t = 0.0;
                        for (int a = 0; a < 3; a++) dg.cube[a] = dg.voxel[a] = 0;
                        px_steps = 0; px_steps_prev = 0;
                        if (F.use_init && F.aux) {  // continue the UI pre-pass's per-pixel record
                            const DevAux &a = F.aux[pix];
                            dg.hit = a.hit;
                            for (int k = 0; k < 3; k++) { dg.cube[k] = a.cube[k]; dg.voxel[k] = a.voxel[k]; }
                            dg.res = a.resolution; dg.face = a.face; dg.block = a.block_index; dg.t = a.t_distance;
                            dg.layer = a.layer;
                            px_steps_prev = a.cubes_traced;
                        }
                    }
                }
                if (F.use_init) {
                    const float4 v = S.acc_buf[(size_t)sample * npix + pix];
                    acc.l0 = v.x; acc.l1 = v.y; acc.l2 = v.z; acc.t = v.w;
                } else {
                    acc.l0 = acc.l1 = acc.l2 = 0.f;
                    acc.t = KF(1.0f);  // made here: as a literal it is hoisted into a register that lives across the whole loop
                }
                if (DIAG && sample > 0) dg.hit = dg.hit | 2;  // only the first sample's position is reported
                if (!ui_pass && S.has_backdrop) {  // Exception::Backdrop hit: ColorBuf::from(Rgba)
                    const float a = opaque_s(S.backdrop[3]);
                    cb_add(acc, S.backdrop[0] * a, S.backdrop[1] * a, S.backdrop[2] * a, 1.0f - a);
                }
                count = 0;
                st = (uint32_t)sample << 14;
                double o[3], dir[3];
                const bool have_ray = ray_of_pixel(x, lrow, sample, true, o, dir);
                if (have_ray) {
                    const double ox = o[0], oy = o[1], oz = o[2];
                    const double dirx = dir[0], diry = dir[1], dirz = dir[2];
                    if (LMODE == 3) {  // SmallRng::seed_from_u64(bits(dx) + bits(dy) + bits(dz)) of the ray as given (sr.rs:165-178)
                        brng = bounce_rng_seed((unsigned long long)__double_as_longlong(dirx) + (unsigned long long)__double_as_longlong(diry) +
                                               (unsigned long long)__double_as_longlong(dirz));
                        sec_steps = 0;
                    }
                    const double t_abs = sqrt(dirx * dirx + diry * diry + dirz * dirz);  // sr.rs:146
                    c64[C_TABS][col] = t_abs;
                    c32[K_TVIEW][col] = __float_as_uint((float)(t_abs / opt.view_distance));  // sr.rs:149-151
                    const RayDir rd = raydir_init(dirx, diry, dirz);
                    c64[C_DX][col] = rd.dx; c64[C_DY][col] = rd.dy; c64[C_DZ][col] = rd.dz;
                    if constexpr (!XCHG) { c64[XCHG ? 0 : (int)C_OX][col] = ox; c64[XCHG ? 0 : (int)C_OY][col] = oy; c64[XCHG ? 0 : (int)C_OZ][col] = oz; }
                    tdx = rd.tdx; tdy = rd.tdy; tdz = rd.tdz;
                    const uint32_t qx = dirx >= 0.0 ? 1u : 0u, qy = diry >= 0.0 ? 1u : 0u, qz = dirz >= 0.0 ? 1u : 0u;
                    const uint32_t octant = (qx << 2) + (qy << 1) + qz;
                    const int ohx = olx + osx_i, ohy = oly + osy_i, ohz = olz + osz_i;
                    // the sanitised direction equals the original unless it was zeroed, in which case no fast-forward happens
                    const double half_over_len = 0.5 / t_abs;
                    const LvlLim ll = lvl_init(ox, oy, oz, rd, true, olx, oly, olz, ohx, ohy, ohz, true, half_over_len);
                    const FirstCube fs = lvl_first_masks(ll.s, rd, olx, oly, olz, ohx, ohy, ohz);
                    tx = fs.tx; ty = fs.ty; tz = fs.tz; last_t = fs.last_t;
                    const int ccx = fs.cx - olx, ccy = fs.cy - oly, ccz = fs.cz - olz;
                    rx = qx ? (uint32_t)(osx_i - 1 - ccx) : (uint32_t)ccx;
                    ry = qy ? (uint32_t)(osy_i - 1 - ccy) : (uint32_t)ccy;
                    rz = qz ? (uint32_t)(osz_i - 1 - ccz) : (uint32_t)ccz;
                    boff = 2u * (uint32_t)(((uint32_t)ccx * (uint32_t)osy_i + (uint32_t)ccy) * (uint32_t)osz_i + (uint32_t)ccz);
                    ssx = qx ? ostx : -ostx; ssy = qy ? osty : -osty; ssz = qz ? 2 : -2;
                    thr = outer_thr;
                    lax = fs.lax;
                    st = ST_TRACED | (octant << 24) | ((uint32_t)sample << 14) | (rd.fast ? ST_DIR_FAST : 0u);
#ifdef AIC_PROFILE
                    s_ray_t0[col] = (uint32_t)__builtin_readcyclecounter();
#endif
                    if (cb_opaque(acc)) st |= ST_OPAQUE;
                    ev = (fs.got ? EV_FRESH : 0u) | (fs.inbounds ? 0u : EV_DEAD);
                } else {
                    ev = EV_FINISH;
                }
            }
            if (run == EV_FINISH && !want) c32[K_PXY][col] = pxy;
            AIC_SECTION(newray_end);
            }
            if (run == EV_SHADE) { AIC_TICK(13) } else if (run == EV_ENTER) { AIC_TICK(14) } else { AIC_TICK(15) }
            continue;
        }

        AIC_SECTION(stepping);
        // ============================ stepping phase ======================================
        // One Amanatides-Woo step of the lane's current level -- the cube grid or a block's voxel
        // volume: same registers, same code, one 2-byte lookup in the shared pool
        // (SurfaceIter::next + Raycaster::next + State::step).
        //
        // The chip issues about one instruction per 4 cycles per SIMD, vector or scalar (tools/ubench/issue_rate; 2.2 for the simplest),
        // so the trip is written for instruction count: every decision is a 64-bit wave mask in SGPRs (a lane flag
        // costs nothing to test), per-lane state is updated IN PLACE by short exec-masked runs in inline assembly
        // (left to the compiler, the divergent branches of this loop become chains of Flow blocks with ~150 register
        // copies per trip), and the rare paths -- leaving a block, applying a pending span -- sit behind uniform
        // branches.
        {
        typedef unsigned long long mask_t;
        mask_t m_act = m_st;                                                           // lanes stepping
        mask_t m_fresh = __builtin_amdgcn_ballot_w64((ev & EV_FRESH) != 0u);           // (only ever set on stepping lanes)
        mask_t m_dead = __builtin_amdgcn_ballot_w64((ev & EV_DEAD) != 0u) & m_act;
        mask_t m_inb = __builtin_amdgcn_ballot_w64((st & ST_IN_BLOCK) != 0u);
        mask_t m_opq = __builtin_amdgcn_ballot_w64((st & ST_OPAQUE) != 0u);
        mask_t m_hl = VOL ? __builtin_amdgcn_ballot_w64((st & ST_HAS_LAST) != 0u) : 0ull;
        // a trip adds at most AIC_STEP_REPS * (AIC_FAST_STEPS + 2) to a lane's step count: lanes this far below the 1000-step cap
        // (count_step_should_stop, sr.rs:639-651) cannot reach it during the trip
        const mask_t m_far_from_cap = __builtin_amdgcn_ballot_w64(count < 1000u - (uint32_t)(AIC_STEP_REPS * (AIC_FAST_STEPS + 2)));
        // what the trip decides for each lane is collected in masks and written to the event words once, after the loop
        mask_t t_shade = 0ull, t_enter = 0ull, t_fin = 0ull, t_deadpark = 0ull;
        // Fast steps need AIC_FAST_MIN takers while the frame is in full swing (other waves want the issue slots); once this wave has
        // seen the pixel queue dry it is draining its last rays and what counts is how soon the longest of them ends: a lone ray
        // then takes its fast steps too (38 instructions a step instead of a full pass's ~180).
        const uint32_t fast_min = (uint32_t)AIC_FAST_MIN - (uint32_t)__builtin_amdgcn_readfirstlane((int)dry) * (uint32_t)(AIC_FAST_MIN - 1);  // (`dry` is wave-uniform; the compiler cannot tell)
        AIC_PROF(22, 1);
        AIC_PROF(23, __popcll(m_act));
#ifdef AIC_TAIL_PROF
        if (__builtin_amdgcn_readfirstlane((int)dry)) { tail_trips++; tail_lanes += (uint32_t)__popcll(m_act); AIC_PROF(29, 1); }
#endif
#define AIC_LANE(m) __builtin_amdgcn_inverse_ballot_w64(m)
// Waits for a lookup issued into `raw` by an earlier asm statement. The loaded register goes in as a plain INPUT and the code comes
// out in a fresh register, copied AFTER the wait: every later use depends on this statement. (Until round 4 this was
// `asm("s_waitcnt vmcnt(0)" : "+v"(raw))`: a tied operand, for which the compiler may place a register copy in front of the
// statement -- i.e. read the destination of a load still in flight. Register allocation happened never to need one; the first
// change that gave `raw` another live range -- the speculative lookups -- made it appear, in front of the full pass's wait.)
#define AIC_WAIT_RAW() { const uint32_t raw_in_flight_ = raw; asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %1" : "=&v"(raw) : "v"(raw_in_flight_)); }
#pragma unroll 1
        for (int rep = 0; rep < AIC_STEP_REPS && m_act != 0ull; rep++) {
            AIC_PROF(10, 1);
            AIC_PROF(11, __popcll(m_act));
            mask_t m_step = m_act & ~(m_fresh | m_dead);  // the level takes its next step
            // (the step itself: dda_step, defined above the persistent loop -- the SHADE event's fused step uses it too)
            // -- Steps that cannot mean anything, taken ahead of the bookkeeping below. Four steps in five find an invisible cube
            //    or voxel inside the bounds while the ray has no span pending (DepthIter), is not opaque yet and is far from the
            //    1000-step cap: such a TraceStep is counted and has no other effect (sr.rs:625-656, surface.rs:453-491). A lane
            //    in that state takes up to AIC_FAST_STEPS of them here -- step, look up, count: a third of the instructions of a
            //    full pass -- and goes on into the full pass below with a further step; a lane whose fast step found something
            //    or left the bounds has taken its step of this pass and joins the bookkeeping with that lookup. --
#ifndef AIC_TAIL_PROF
            AIC_PROF(24, __popcll(m_step & m_hl));
            AIC_PROF(25, __popcll(m_step & ~(m_hl | m_opq) & m_far_from_cap));
#endif
            mask_t m_pre_exit = 0ull, m_pre_look = 0ull;  // lanes whose step of this pass was taken here: left the bounds / looked something up
            if (!BIG && AIC_FAST_STEPS > 0) {
                mask_t m_f = m_step & ~(m_hl | m_opq) & m_far_from_cap;
#pragma unroll
                for (int f = 0; f < AIC_FAST_STEPS; f++) {
#if AIC_FAST_MIN > 0
                    // a fast step costs the same however few lanes take it: too few, and their steps are cheaper taken by
                    // the full passes that run anyway
                    if (wave_popc(m_f) < fast_min) break;
#else
                    if (m_f == 0ull) break;
#endif
                    mask_t m_fx, m_fe, m_fb;
                    if (DIAG) {
                        m_fx = dda_step(m_f);
                        const mask_t m_fl = m_f & ~m_fx;
                        mask_t sv;
                        asm volatile(
                            "s_mov_b64 %[sv], exec\n\t"
                            "s_mov_b64 exec, %[m]\n\t"
                            "global_load_ushort %[raw], %[bo], %[pool]\n\t"
                            "s_mov_b64 exec, %[sv]\n\t"
                            : [raw] "+v"(raw), [sv] "=&s"(sv)
                            : [bo] "v"(boff), [pool] "s"(pool_bits), [m] "s"(m_fl)
                            : "memory");
                        dg.n_inner += AIC_LANE(m_fl & m_inb) ? 1u : 0u; dg.n_outer += AIC_LANE(m_fl & ~m_inb) ? 1u : 0u;
                        AIC_WAIT_RAW();
                        // a code >= thr is a visible voxel, or a cube whose block is visible or recursive (class bits): the full pass decides
                        m_fe = __builtin_amdgcn_ballot_w64(raw >= thr) & m_fl;
                        m_fb = m_fl & ~m_fe;  // an Invisible TraceStep: counted, nothing else
                        asm volatile("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(count) : "s"(m_fb) : "vcc");
                    } else {
                        // The whole fast step as one block: the DDA step of dda_step above, then -- exec narrowing as it goes instead of
                        // being restored and set again -- the lookup under the lanes still inside their bounds, the comparison with the
                        // level's threshold under the same mask, and the count under the lanes that found nothing. (Left to the compiler,
                        // the pieces come with hazard nops between them, a vector compare for the loop's 64-bit population count,
                        // and six more mask operations.)
                        mask_t sv, mx, by, bx;
                        asm volatile(
                            "s_and_saveexec_b64 %[sv], %[m]\n\t"
                            "v_min_f64 %[lt], %[tx], %[ty]\n\t"
                            "v_min_f64 %[lt], %[lt], %[tz]\n\t"
                            "v_cmp_eq_f64 %[mx], %[tz], %[lt]\n\t"        // Z
                            "v_cmp_eq_f64 vcc, %[ty], %[lt]\n\t"
                            "s_andn2_b64 vcc, vcc, %[mx]\n\t"             // Y
                            "s_mov_b64 exec, %[mx]\n\t"
                            "v_add_f64 %[tz], %[tz], %[tdz]\n\t"
                            "v_sub_co_u32 %[rz], %[fx], %[rz], 1\n\t"
                            "v_add_u32 %[bo], %[bo], %[ssz]\n\t"
                            "v_mov_b32 %[lax], 2\n\t"
                            "s_or_b64 %[mx], %[mx], vcc\n\t"
                            "s_mov_b64 exec, vcc\n\t"
                            "v_add_f64 %[ty], %[ty], %[tdy]\n\t"
                            "v_sub_co_u32 %[ry], %[by], %[ry], 1\n\t"
                            "v_add_u32 %[bo], %[bo], %[ssy]\n\t"
                            "v_mov_b32 %[lax], 1\n\t"
                            "s_andn2_b64 exec, %[m], %[mx]\n\t"           // X = stepping lanes that took neither
                            "v_add_f64 %[tx], %[tx], %[tdx]\n\t"
                            "v_sub_co_u32 %[rx], %[bx], %[rx], 1\n\t"
                            "v_add_u32 %[bo], %[bo], %[ssx]\n\t"
                            "v_mov_b32 %[lax], 0\n\t"
                            "s_or_b64 %[fx], %[fx], %[by]\n\t"
                            "s_or_b64 %[fx], %[fx], %[bx]\n\t"            // left the bounds
                            "s_andn2_b64 exec, %[m], %[fx]\n\t"           // still inside: look up
                            "global_load_ushort %[raw], %[bo], %[pool]\n\t"
                            "s_waitcnt vmcnt(0)\n\t"
                            "v_cmp_ge_u32 vcc, %[raw], %[thr]\n\t"        // found something (a visible voxel, a visible or recursive block)
                            "s_mov_b64 %[fe], vcc\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"             // an Invisible TraceStep: counted, nothing else
                            "s_mov_b64 %[fb], exec\n\t"
                            "v_add_u32 %[cnt], 1, %[cnt]\n\t"
                            "s_mov_b64 exec, %[sv]\n\t"
                            : [tx] "+v"(tx), [ty] "+v"(ty), [tz] "+v"(tz), [lt] "+v"(last_t), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz),
                              [bo] "+v"(boff), [lax] "+v"(lax), [raw] "+v"(raw), [cnt] "+v"(count), [sv] "=&s"(sv), [mx] "=&s"(mx),
                              [fx] "=&s"(m_fx), [by] "=&s"(by), [bx] "=&s"(bx), [fe] "=&s"(m_fe), [fb] "=&s"(m_fb)
                            : [tdx] "v"(tdx), [tdy] "v"(tdy), [tdz] "v"(tdz), [ssx] "v"(ssx), [ssy] "v"(ssy), [ssz] "v"(ssz), [m] "s"(m_f),
                              [pool] "s"(pool_bits), [thr] "v"(thr)
                            : "memory", "vcc", "scc");
                    }
                    AIC_PROF(20, 1);
                    AIC_PROF(21, __popcll(m_f));
                    m_pre_exit |= m_fx;
                    m_pre_look |= m_fe;
                    m_f = m_fb;
                }
                m_step &= ~(m_pre_exit | m_pre_look);
            }
            // -- left the bounds? (raycast.rs:265-274) only the axis just stepped can have run out of steps --
            const mask_t m_exit = dda_step(m_step) | m_pre_exit;
            // -- can the level step again? valid_for_stepping (raycast.rs:563-570): "the smallest t_max is finite" held when
            //    the level was set up (lvl_first marks a level that cannot step as dead), and a step only adds the finite
            //    t_delta of an axis whose t_max was finite, so it holds for every level this loop sees: no per-step check --
            // a cube is produced by a fresh level, or by a step that stays in bounds
            const mask_t m_load = m_fresh | (m_step & ~m_exit);  // lookups still to be made; the fast steps' hold theirs in `raw` already
            const mask_t m_lookup = m_load | m_pre_look;
            // the level is over: it left its bounds, cannot step again, or had ended before
            const mask_t m_over = m_exit | m_dead;
            // -- the lookup: one u16 from the pool, for whichever level this is (scalar base + 32-bit byte offset) --
            {
                mask_t sv;
                asm volatile(
                    "s_mov_b64 %[sv], exec\n\t"
                    "s_mov_b64 exec, %[m]\n\t"
                    "global_load_ushort %[raw], %[bo], %[pool]\n\t"
                    "s_mov_b64 exec, %[sv]\n\t"
                    : [raw] "+v"(raw), [sv] "=&s"(sv)
                    : [bo] "v"(boff), [pool] "s"(pool_bits), [m] "s"(m_load)
                    : "memory");
            }
            if (DIAG) { dg.n_inner += AIC_LANE(m_load & m_inb) ? 1u : 0u; dg.n_outer += AIC_LANE(m_load & ~m_inb) ? 1u : 0u; }
            const mask_t m_produced = m_lookup | m_exit;  // the include_exit step is an Invisible TraceStep
            // ---- TracingState::count_step_should_stop (sr.rs:625-656) ----
            asm volatile("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(count) : "s"(m_produced) : "vcc");
            const mask_t m_stop = m_produced & (__builtin_amdgcn_ballot_w64(count > 1000u) | m_opq);
            const mask_t m_go = m_produced & ~m_stop;
            AIC_WAIT_RAW();
            // TraceStep of a looked-up code: voxel -- visible iff its (re-ordered) palette code is past the invisible
            // ones; cube -- the class of its block rides in the top two bits of the grid entry (aic_device.h)
            mask_t m_blk, m_surf;
            if (BIG) {
                // block tables past 16384 entries: plain 16-bit indices, classes from the table in global memory
                uint32_t cls = 0u;
                if (AIC_LANE(m_lookup & ~m_inb)) cls = (F.layer.cls[raw >> 4] >> ((raw & 15u) << 1)) & 3u;
                m_blk = __builtin_amdgcn_ballot_w64(cls == 2u);
                m_surf = (__builtin_amdgcn_ballot_w64(raw >= thr) & m_lookup & m_inb) | __builtin_amdgcn_ballot_w64(cls == 1u);
            } else {
                m_blk = __builtin_amdgcn_ballot_w64(raw >= (2u << kCubeClassShift)) & m_lookup & ~m_inb;
                m_surf = __builtin_amdgcn_ballot_w64(raw >= thr) & m_lookup & ~m_blk;
            }
            const mask_t m_some = m_blk | m_surf;
#ifndef AIC_TAIL_PROF
            AIC_PROF(30, __popcll(m_some | m_exit | m_dead));  // lanes of this pass that needed its bookkeeping
#endif
            // -- the level is finished: resume the cube grid, or the ray is complete --
            // (degenerate rays only) a surface / block produced by a level that is over still needs this level's
            // state for its event: the level stays, marked dead, and is left on a later trip
            const mask_t m_defer = m_over & m_some;
            const mask_t m_leave = m_over & ~m_some & m_inb;
            const mask_t m_rayover = m_over & ~m_some & ~m_inb;
            mask_t m_newdead = 0ull;
            if (m_leave != 0ull) {
#ifndef AIC_TAIL_PROF
                AIC_PROF(26, 1);
                AIC_PROF(27, __popcll(m_leave));
#endif
                // Everything under exec = the leaving lanes, in one block: the suspended level comes back from its LDS columns; the
                // cube grid's strides are +-stride by the octant bits of st (bit set = the ray goes up that axis:
                // m - (stride ^ m) with m = the bit sign-extended is +stride for m = -1, -stride for m = 0); the outer level's Face
                // from st[16..18]; it goes on stepping if it was alive when the block was entered.
                mask_t sv;
                uint32_t t_;
                // LDS byte addresses of the lane's columns (the low half of a generic LDS pointer is the LDS offset)
                const uint32_t lds64 = (uint32_t)(uintptr_t)&c64[0][col], lds32 = (uint32_t)(uintptr_t)&c32[0][col];
                asm volatile(
                    "s_mov_b64 %[sv], exec\n\t"
                    "s_mov_b64 exec, %[m]\n\t"
                    "ds_read_b64 %[tx], %[a64] offset:%[o0]\n\t"
                    "ds_read_b64 %[ty], %[a64] offset:%[o1]\n\t"
                    "ds_read_b64 %[tz], %[a64] offset:%[o2]\n\t"
                    "ds_read_b32 %[rx], %[a32] offset:%[p0]\n\t"
                    "ds_read_b32 %[ry], %[a32] offset:%[p1]\n\t"
                    "ds_read_b32 %[rz], %[a32] offset:%[p2]\n\t"
                    "ds_read_b32 %[bo], %[a32] offset:%[p3]\n\t"
                    "v_bfe_i32 %[t], %[st], 26, 1\n\t"
                    "v_xor_b32 %[ssx], %[osx], %[t]\n\t"
                    "v_sub_u32 %[ssx], %[t], %[ssx]\n\t"
                    "v_bfe_i32 %[t], %[st], 25, 1\n\t"
                    "v_xor_b32 %[ssy], %[osy], %[t]\n\t"
                    "v_sub_u32 %[ssy], %[t], %[ssy]\n\t"
                    "v_bfe_i32 %[t], %[st], 24, 1\n\t"
                    "v_xor_b32 %[ssz], 2, %[t]\n\t"
                    "v_sub_u32 %[ssz], %[t], %[ssz]\n\t"
                    "v_mov_b32 %[thr], %[othr]\n\t"
                    "v_bfe_u32 %[t], %[st], 16, 3\n\t"
                    "v_or_b32 %[lax], 8, %[t]\n\t"
                    "v_and_b32 %[t], %[alive], %[st]\n\t"
                    "v_cmp_eq_u32 %[nd], 0, %[t]\n\t"               // the outer level had ended already (inactive lanes: 0)
                    "v_and_b32 %[st], %[notinb], %[st]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_mov_b64 exec, %[sv]\n\t"
                    : [tx] "+v"(tx), [ty] "+v"(ty), [tz] "+v"(tz), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz),
                      [bo] "+v"(boff), [ssx] "+v"(ssx), [ssy] "+v"(ssy), [ssz] "+v"(ssz), [thr] "+v"(thr), [lax] "+v"(lax), [st] "+v"(st),
                      [t] "=&v"(t_), [sv] "=&s"(sv), [nd] "=&s"(m_newdead)
                    : [a64] "v"(lds64), [a32] "v"(lds32), [m] "s"(m_leave), [osx] "s"(ostx), [osy] "s"(osty), [othr] "n"(BIG ? 0x10000u : (1u << kCubeClassShift)),
                      [alive] "n"(ST_OUTER_ALIVE), [notinb] "n"(~ST_IN_BLOCK),
                      [o0] "n"(C_STX * NCOL * 8), [o1] "n"(C_STY * NCOL * 8), [o2] "n"(C_STZ * NCOL * 8),
                      [p0] "n"(K_SRX * NCOL * 4), [p1] "n"(K_SRY * NCOL * 4),
                      [p2] "n"(K_SRZ * NCOL * 4), [p3] "n"(K_SBOFF * NCOL * 4)
                    : "memory");
                m_inb &= ~m_leave;
            }
            // ---- DepthIter::next (surface.rs:453-491): a pending surface's span ends at this step;
            // its contribution was computed when it was shaded, apply it now ----
            if (VOL) {
                const mask_t m_apply = m_go & m_hl;
                if (m_apply != 0ull) {
#ifndef AIC_TAIL_PROF
                    AIC_PROF(28, 1);
                    AIC_PROF(29, __popcll(m_apply));
#endif
                    const bool apply = AIC_LANE(m_apply);
                    acc.l0 = apply ? acc.l0 + pend0 * acc.t : acc.l0;
                    acc.l1 = apply ? acc.l1 + pend1 * acc.t : acc.l1;
                    acc.l2 = apply ? acc.l2 + pend2 * acc.t : acc.l2;
                    acc.t = apply ? acc.t * pend_tr : acc.t;
                    const mask_t m_now_opq = __builtin_amdgcn_ballot_w64(acc.t < 1.0f / 256.0f) & m_apply;  // cb_opaque
                    st = apply ? ((st & ~ST_HAS_LAST) | (AIC_LANE(m_now_opq) ? ST_OPAQUE : 0u)) : st;
                    m_hl &= ~m_apply;
                    m_opq |= m_now_opq;
                    if (DIAG && apply && pend_visible) {
                        dg.n_hits++;
                        dg.n_light += pend_d.nlight;
                        if (!dg.hit) {
                            dg.hit = 1;
                            dg.layer = ui_pass_s ? 1u : 0u;
                            for (int a2 = 0; a2 < 3; a2++) { dg.cube[a2] = pend_d.cube[a2]; dg.voxel[a2] = pend_d.voxel[a2]; }
                            dg.res = pend_d.res; dg.face = pend_d.face; dg.block = pend_d.block; dg.t = pend_t;
                        }
                    }
                }
            }
            // DepthIter emits a second, buffered item for EnterBlock (surface.rs:478-488): count it too
            mask_t m_stop2 = 0ull;
            if (VOL) {
                const mask_t m_second = m_go & m_blk;
                asm volatile("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(count) : "s"(m_second) : "vcc");
                m_stop2 = m_second & (__builtin_amdgcn_ballot_w64(count > 1000u) | m_opq);
            }
            // A surface discovered by the very step whose pending span made the ray opaque is never lit: DepthIter has emitted the
            // span and keeps the new surface as last_surface, but the ray's next step fails count_step_should_stop and that
            // surface's span is never produced (surface.rs:453-491, sr.rs:183-189). Until round 4 such a lane was sent to SHADE all
            // the same -- exact, its result was never applied, and wasted: every ray that ends on a solid with something behind it
            // paid one whole SHADE event for nothing (and, with Bounce lighting, traced secondary rays the reference never traces).
            const mask_t m_shade = m_go & m_surf & ~m_opq;
            const mask_t m_enter = m_go & m_blk & ~m_stop2;
            const mask_t m_fin = m_stop | m_stop2 | m_rayover;
            // a lane that parks leaves the trip with its event; one whose level ended while it still owes an event keeps DEAD
            t_shade |= m_shade;
            t_enter |= m_enter;
            t_fin |= m_fin;
            t_deadpark |= (m_defer & ~m_fin) | (m_newdead & m_fin);
            m_act &= ~(m_shade | m_enter | m_fin);
            m_dead = m_newdead & m_act;
            m_fresh = 0ull;
        }
        // new event words: the lanes that took part drop FRESH / DEAD, then take what the trip decided; a lane still
        // stepping whose level ended on the last step carries DEAD into the next trip
        ev = AIC_LANE(m_st) ? 0u : ev;
        ev = AIC_LANE(t_shade) ? EV_SHADE : ev;
        ev = AIC_LANE(t_enter) ? EV_ENTER : ev;
        ev = AIC_LANE(t_fin) ? EV_FINISH : ev;
        if ((t_deadpark | m_dead) != 0ull) ev = AIC_LANE(t_deadpark | m_dead) ? (ev | EV_DEAD) : ev;
#undef AIC_LANE
        }
        AIC_TICK(12);
    }

    AIC_SECTION(epilogue);
    // the frame this workgroup belongs to (as in the event phase: through the opaque kernel-argument pointer)
    typedef const __attribute__((address_space(4))) DevFrame KFrameE;
    typedef const __attribute__((address_space(4))) DevSub KSubE;
    KFrameE *Fe = (KFrameE *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(Fe));
    const uint32_t n_sub_e = Fe->n_sub > 1u ? Fe->n_sub : 1u;
    KSubE &SE = Fe->sub[blockIdx.x & (n_sub_e - 1u)];
    DevCounters *const counters_e = SE.counters;
#ifdef AIC_PROFILE
    if (lane == 0) {
        prof[3] = (uint32_t)__builtin_readcyclecounter() - prof_t0;  // wave lifetime
        const uint32_t wid = blockIdx.x * (WGT / 64u) + (threadIdx.x >> 6);
        if (wid < 2048u) {
            counters_e->wave_prof[wid][0] = prof_t0;
            counters_e->wave_prof[wid][1] = prof_t0 + prof[2];
            counters_e->wave_prof[wid][2] = prof_t0 + prof[3];
            counters_e->wave_prof[wid][3] = prof[9];
        }
        atomicMax(&counters_e->prof[0], (unsigned long long)prof[3]);
        atomicMax(&counters_e->prof[1], (unsigned long long)prof[2]);
    }
#ifdef AIC_TAIL_PROF
    {   // trips (12 bits), event phases (10 bits) and mean lanes stepping per trip (x16, 10 bits) after the wave saw the queue dry
        const uint32_t wid = blockIdx.x * (WGT / 64u) + (threadIdx.x >> 6);
        const uint32_t ml = tail_trips ? (tail_lanes * 16u) / tail_trips : 0u;
        if (lane == 0 && wid < 2048u) counters_e->wave_prof[wid][3] = (tail_trips > 4095u ? 4095u : tail_trips) | ((tail_events > 1023u ? 1023u : tail_events) << 12) | ((ml > 1023u ? 1023u : ml) << 22);
    }
#endif
#ifdef AIC_RAY_PROF
    {   // the wave's longest ray: duration in the upper bits, its step count in the lower 10
        uint32_t best_ = (ray_dur_max & ~1023u) | (ray_dur_steps > 1023u ? 1023u : ray_dur_steps);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o_ = (uint32_t)__shfl_down((int)best_, off, 64); best_ = o_ > best_ ? o_ : best_; }
        const uint32_t wid = blockIdx.x * (WGT / 64u) + (threadIdx.x >> 6);
        if (lane == 0 && wid < 2048u) counters_e->wave_prof[wid][3] = best_;
    }
#endif
    if (lane == 0) for (int i = 2; i < 40; i++) atomicAdd(&counters_e->prof[i], (unsigned long long)prof[i]);
#endif
    // ---- RaytraceInfo sum (renderer.rs:555): wave reduction then one atomic per wave ----
    uint32_t t_end = threadIdx.x;
    asm volatile("" : "+v"(t_end));  // (as above: the word's address is not a value to keep for the life of the wave)
    unsigned long long s = s_steps[t_end];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0 && s) atomicAdd(&counters_e->cubes_traced, s);
    if (DIAG) {
        unsigned long long v[4] = {tot_outer, tot_inner, tot_hits, tot_light};
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        }
        if (lane == 0) {
            if (v[0]) atomicAdd(&counters_e->n_outer, v[0]);
            if (v[1]) atomicAdd(&counters_e->n_inner, v[1]);
            if (v[2]) atomicAdd(&counters_e->n_hits, v[2]);
            if (v[3]) atomicAdd(&counters_e->n_light, v[3]);
        }
    }
    // the last wave out hands the frame's sums to the host (pinned memory; the end of the kernel makes the stores visible)
    // (No fences: an agent-scope release fence writes the XCD's whole L2 back -- the frame's pixels -- and 4096 waves doing that cost 10 % of a C2
    //  frame. The sums are agent-scope atomics, performed at the memory side; the wave waits for its own to be acknowledged before it is counted, and
    //  the last wave reads them with agent-scope loads.)
    unsigned long long *const host_counters_e = SE.host_counters;
    if (lane == 0 && host_counters_e) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the waves of THIS frame: the grid is a multiple of n_sub workgroups, dealt to the frames round-robin)
        if (atomicAdd(&counters_e->waves_done, 1u) == (gridDim.x / n_sub_e) * (blockDim.x >> 6) - 1u) {
            unsigned long long *const src = &counters_e->cubes_traced;  // five consecutive sums and `bailed` (DevCounters)
#pragma unroll
            for (int i = 0; i < 6; i++) host_counters_e[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------
// small kernels

// aic_update_cubes: scatter of SpaceChange::{CubeBlock,CubeLight} (updating.rs:146-166)
__global__ void scatter_cubes_kernel(uint16_t *grid, uint32_t *light, const int32_t *xyz, const uint16_t *bi,
                                     const uint32_t *lt, uint32_t n, int lx, int ly, int lz, int sx, int sy, int sz,
                                     const uint32_t *cls) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t dx = (uint32_t)xyz[3 * i + 0] - (uint32_t)lx;
    uint32_t dy = (uint32_t)xyz[3 * i + 1] - (uint32_t)ly;
    uint32_t dz = (uint32_t)xyz[3 * i + 2] - (uint32_t)lz;
    if ((dx >= (uint32_t)sx) | (dy >= (uint32_t)sy) | (dz >= (uint32_t)sz)) return;
    size_t idx = ((size_t)dx * sy + dy) * sz + dz;
    if (bi) {
        uint32_t b = bi[i];
        if (cls) b |= ((cls[b >> 4] >> ((b & 15u) << 1)) & 3u) << kCubeClassShift;  // cls != null: tagged grid
        grid[idx] = (uint16_t)b;
    }
    if (lt) light[idx] = lt[i];
}

// (re)writes the class bits of every cube-grid entry from the class table (aic_device.h)
__global__ void tag_cubes_kernel(uint16_t *grid, size_t n, const uint32_t *cls, int from_tagged, int to_tagged) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t b = grid[i];
    if (from_tagged) b &= kCubeIndexMask;
    if (to_tagged) b |= ((cls[b >> 4] >> ((b & 15u) << 1)) & 3u) << kCubeClassShift;
    grid[i] = (uint16_t)b;
}

// Orders the tiles of the next frame by the cost the previous frame measured for them (its longest
// ray, in steps), costliest first: the rays most likely to be long start early instead of landing
// in the frame's tail, where a wave with two live lanes still pays a whole event phase for each.
// One workgroup: histogram over 1024 cost buckets per queue, prefix sum, scatter. Order inside a bucket is
// whatever the atomics give -- every pixel is traced exactly once either way.
//
// Queues (round 4): each XCD has its own L2, and with one dispenser for the chip the 4 waves' worth of rays of a macro tile and of
// its neighbours run on all eight at once -- every L2 fetches the same lines. With n_queues > 1 the macro tiles are dealt to queues
// by the super-block (2^sb_shift macro tiles on a side) they lie in, a workgroup serves the queue of the XCD it runs on (and helps
// the others when its own is empty), and `order` comes out as n_queues segments, each costliest first; queue_start[q] is where
// segment q begins. cost == nullptr: no record to go by (index order inside a queue, as far as the atomics keep it).
__device__ __forceinline__ uint32_t tile_queue_of(uint32_t mt, uint32_t macros_x, uint32_t sb_shift, uint32_t n_queues) {
    // mt / macros_x without the ~40-instruction integer division (this runs twice per macro tile on one workgroup, ahead of every frame):
    // a float quotient is within one of the truth for these sizes (mt < 2^24), corrected exactly
    uint32_t my = (uint32_t)((float)mt * __builtin_amdgcn_rcpf((float)macros_x));
    if (my * macros_x > mt) my--;
    else if ((my + 1u) * macros_x <= mt) my++;
    const uint32_t mx = mt - my * macros_x;
    const uint32_t v = (mx >> sb_shift) + 3u * (my >> sb_shift);
    return (n_queues & (n_queues - 1u)) == 0u ? (v & (n_queues - 1u)) : v % n_queues;
}
// One workgroup of kOrderThreads = 256 threads (four waves, 33 KB of LDS): what ONE retiring workgroup of a trace kernel leaves free on a CU. With 1024
// threads it needed a whole CU to drain, and while frames are streamed every CU is full of persistent trace workgroups: rocprofv3 showed it
// waiting 0.14 ms (C2) / 1.6 ms (C3) for a place to run (profiles/r04_experiments.txt L).
constexpr uint32_t kOrderThreads = 256;
// (a workgroup per job -- OrderJobs, aic_device.h: the frames of a batch, aic_render_submit_batch, are ordered by one launch)
__global__ __launch_bounds__(kOrderThreads) void order_tiles_kernel(const OrderJobs jobs, uint32_t n_tiles, uint32_t macros_x, uint32_t sb_shift, uint32_t n_queues,
                                                                    uint32_t clear_the_cost, uint32_t n_clear_words) {
    const uint32_t *__restrict__ const cost = jobs.cost[blockIdx.x];
    uint32_t *__restrict__ const order = jobs.order[blockIdx.x];
    uint32_t *__restrict__ const queue_start = jobs.queue_start[blockIdx.x];
    uint32_t *const clear_cost = (clear_the_cost && cost) ? const_cast<uint32_t *>(cost) : nullptr;
    uint32_t *const clear_words = jobs.clear_words[blockIdx.x];
    __shared__ uint32_t hist[kMaxTileQueues * 1024];
    __shared__ uint32_t scan[kOrderThreads];
    // (behind a frame this launch also does the slot's clearing -- the frame's counters, and below the cost record once it has been read -- instead of two
    //  fill launches that would each wait for room on a CU)
    if (clear_words) for (uint32_t i = threadIdx.x; i < n_clear_words; i += kOrderThreads) clear_words[i] = 0u;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_bins = n_queues * 1024u;
    const uint32_t per_thread = n_bins / kOrderThreads;  // 4 * n_queues consecutive buckets each
    const uint32_t *const cp = cost ? cost : order;  // no record: any readable words, masked away
    const uint32_t use = cost ? ~0u : 0u;
    for (uint32_t b = tid; b < n_bins; b += kOrderThreads) hist[b] = 0;
    __syncthreads();
    // (eight tiles per thread at a time: the eight cost fetches are issued together, not one ahead of each atomic)
    for (uint32_t base = tid; base < n_tiles; base += 8u * kOrderThreads) {
        uint32_t c[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) c[k] = cp[min(base + k * kOrderThreads, n_tiles - 1u)] & use;  // (unconditional: nothing keeps the eight fetches apart)
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            const uint32_t t = base + k * kOrderThreads;
            if (t < n_tiles) atomicAdd(&hist[tile_queue_of(t, macros_x, sb_shift, n_queues) * 1024u + 1023u - (c[k] < 1023u ? c[k] : 1023u)], 1u);
        }
    }
    __syncthreads();
    // exclusive prefix sum over the n_queues * 1024 buckets: thread `tid` owns `per_thread` consecutive buckets
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per_thread; k++) mine += hist[tid * per_thread + k];
    scan[tid] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < kOrderThreads; off <<= 1) {  // Hillis-Steele over the partial sums
        const uint32_t v = tid >= off ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    uint32_t run = scan[tid] - mine;
    for (uint32_t k = 0; k < per_thread; k++) {
        const uint32_t h = hist[tid * per_thread + k];
        hist[tid * per_thread + k] = run;  // start of each bucket
        run += h;
    }
    __syncthreads();
    if (queue_start && tid <= n_queues) queue_start[tid] = tid < n_queues ? hist[tid * 1024u] : n_tiles;
    __syncthreads();
    for (uint32_t base = tid; base < n_tiles; base += 8u * kOrderThreads) {
        uint32_t c[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) c[k] = cp[min(base + k * kOrderThreads, n_tiles - 1u)] & use;
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            const uint32_t t = base + k * kOrderThreads;
            if (t < n_tiles) {
                const uint32_t pos = atomicAdd(&hist[tile_queue_of(t, macros_x, sb_shift, n_queues) * 1024u + 1023u - (c[k] < 1023u ? c[k] : 1023u)], 1u);
                order[pos] = t;
                if (clear_cost) clear_cost[t] = 0u;
            }
        }
    }
}

// aic_assemble_strips: [n_parts][max_rows][w] compacted strips -> [h][w]
__global__ void assemble_strips_kernel(const uint32_t *gathered, uint32_t *out, uint32_t w, uint32_t h, uint32_t strip_rows,
                                       uint32_t n_parts, uint32_t max_rows) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)w * h) return;
    uint32_t y = (uint32_t)(i / w), x = (uint32_t)(i % w);
    uint32_t strip = y / strip_rows;
    uint32_t part = strip % n_parts;
    uint32_t lrow = (strip / n_parts) * strip_rows + (y % strip_rows);
    out[i] = gathered[((size_t)part * max_rows + lrow) * w + x];
}

// aic_probe_raycast: the device Raycaster, one ray
__global__ void probe_raycast_kernel(const double *od, int use_bounds, const int *lohi, int include_exit, uint32_t max_steps,
                                     double *out_rec, uint32_t *n_out, int *ended) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double ox = od[0], oy = od[1], oz = od[2], dx = od[3], dy = od[4], dz = od[5];
    int lo[3] = {lohi[0], lohi[1], lohi[2]}, hi[3] = {lohi[3], lohi[4], lohi[5]};
    if (!use_bounds) {
        lo[0] = lo[1] = lo[2] = I32_MIN_ + 1;
        hi[0] = hi[1] = hi[2] = I32_MAX_ - 1;
    }
    const RayDir rd = raydir_init(dx, dy, dz);
    const LvlLim ll = lvl_init(ox, oy, oz, rd, use_bounds != 0, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], include_exit != 0,
                               0.5 / sqrt(rd.dx * rd.dx + rd.dy * rd.dy + rd.dz * rd.dz));
    Lvl s = ll.s;
    const Lim lim = ll.lim;
    uint32_t n = 0;
    *ended = 0;
    // A bounded raycaster with its exit step -- what the image kernel's events set up -- takes its first step through the events' own code
    // (lvl_first_masks), so that the reference's step tables pin that too; lvl_next goes on from the state it leaves.
    bool first_by_masks = use_bounds != 0 && include_exit != 0;
    while (n < max_steps) {
        NextResult nr;
        if (first_by_masks) {
            first_by_masks = false;
            const FirstCube f = lvl_first_masks(s, rd, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
            nr.got = f.got; nr.is_exit = false;
            nr.s.tx = f.tx; nr.s.ty = f.ty; nr.s.tz = f.tz; nr.s.last_t = f.last_t;
            nr.s.cx = f.cx; nr.s.cy = f.cy; nr.s.cz = f.cz;
            const uint32_t face = (f.lax & 8u) ? (f.lax & 7u) : (((f.lax == 0u ? rd.sx : (f.lax == 1u ? rd.sy : rd.sz)) > 0 ? 1u : 4u) + f.lax);
            // "emitted, step scheduled" as lvl_next leaves it: InBounds | pick | need_step, or Ended
            nr.s.st = (face << 2) | 256u | (f.inbounds ? (FL_INBOUNDS | ((uint32_t)pick_axis(f.tx, f.ty, f.tz) << 5) | 128u) : FL_ENDED);
        } else {
            nr = lvl_next(s, lim, rd, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
        }
        s = nr.s;
        if (!nr.got) {
            *ended = 1;
            break;
        }
        double ip[3];
        intersection_point(s, ox, oy, oz, dx, dy, dz, ip);
        double *r = out_rec + 8 * (size_t)n;
        // record: cube[3] as doubles, face, t, ip[3]
        r[0] = (double)s.cx; r[1] = (double)s.cy; r[2] = (double)s.cz;
        r[3] = (double)lvl_face(s); r[4] = s.last_t; r[5] = ip[0]; r[6] = ip[1]; r[7] = ip[2];
        n++;
    }
    *n_out = n;
}

// f32::powf evaluated in f64 and rounded once: only for the probe below, outside powf_table's domain (the trace kernel never
// leaves that domain)
AIC_DEV float powf_exact(float x, float y) { return (float)pow((double)x, (double)y); }

// aic_probe_powf: the device's powf (table path where its domain allows, as the trace kernel chooses)
__global__ void probe_powf_kernel(const float *x, const float *y, float *out, uint32_t n) {
    __shared__ double s_pow[64];
    pow_tables_to_lds(s_pow, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = powf_table_domain(x[i], y[i]) ? powf_table(x[i], y[i], s_pow) : powf_exact(x[i], y[i]);
}

// ---------------------------------------------------------------------------------------
// host-callable launchers (used by aic_abi.cpp)

template <bool VOL, int LMODE, bool DIAG, bool BIG, bool XC>
static void launch_trace_x(const DevFrame &F, hipStream_t stream) {
    // persistent waves: enough workgroups to fill the chip at the kernel's occupancy, never more
    // waves than tiles (each wave pulls 8x8-pixel tiles from counters->tile_next)
    const uint32_t n_tiles = F.tiles_x * F.tiles_y;
    constexpr bool XCHG = AIC_EXCHANGE && XC && !DIAG && LMODE != 3;
    constexpr uint32_t WGT = XCHG ? (uint32_t)AIC_XWG_THREADS : (uint32_t)AIC_WG_THREADS;
    const uint32_t wg_waves = WGT / 64u;
    const uint32_t n_sub = F.n_sub > 1u ? F.n_sub : 1u;  // frames of this launch (DevSub): each gets an equal share of the resident grid
    const uint32_t resident_groups = F.n_cus * 4u * (uint32_t)((DIAG || LMODE == 3) ? 2 : AIC_MIN_WAVES) / wg_waves / n_sub;  // 4 SIMDs per CU, that many waves on each
    // A frame smaller than the chip that is STREAMED (aic_render_submit: a rank's strips of a multi-GPU frame, several in
    // flight) gets a grid in proportion to its tiles -- four tiles per wave, so that lanes are refilled instead of waves ending
    // after one tile, and the kernels of the frames in flight are resident side by side (an eighth of a 1080p frame, 8 in
    // flight: 0.105 ms per frame against 0.138 with one tile per wave). A synchronous frame (aic_render) is a matter of
    // latency: one tile per wave, as many waves as there are tiles.
    const uint32_t tiles_per_wave = F.tiles_per_wave ? F.tiles_per_wave : 1u;  // (the host's choice; AIC_TILES_PER_WAVE, read when the context is made, overrides it)
    const uint32_t by_tiles = (n_tiles + wg_waves - 1u) / wg_waves;
    uint32_t grid = (n_tiles + wg_waves * tiles_per_wave - 1u) / (wg_waves * tiles_per_wave);
    const uint32_t floor_groups = by_tiles < 128u ? by_tiles : 128u;
    if (grid < floor_groups) grid = floor_groups;
    if (grid > resident_groups) grid = resident_groups;
    if (XCHG && F.antialias && grid > F.ray_cold_groups / n_sub) grid = F.ray_cold_groups / n_sub;  // (the host sizes the antialiasing sums' buffer for the resident grid: trace_ray_cold_bytes)
    grid *= n_sub;
    if (grid == 0) return;
    DevFrame G = F;
    G.ray_mode = (F.layer.present && !F.pixel_centers && !F.patches && !F.ortho_n && F.edge_x && F.edge_y) ? (F.n_parts > 1u ? 1u : 0u) : 2u;  // (DevFrame::ray_mode)
    hipLaunchKernelGGL((trace_image_kernel<VOL, LMODE, DIAG, BIG, XC>), dim3(grid), dim3(WGT), 0, stream, G);
}

// The production variants (no per-pixel records, not Bounce) exist twice: with the lane exchange (a frame of many tiles per persistent wave) and without
// (DevFrame::exchange == 0: a frame of a tile or two per wave -- a rank's share of a multi-GPU frame, the test images -- which the pool only costs).
template <bool VOL, int LMODE, bool DIAG, bool BIG>
static void launch_trace(const DevFrame &F, hipStream_t stream) {
    if constexpr (AIC_EXCHANGE && !DIAG && LMODE != 3) {
        if (F.exchange && (F.ray_cold || !F.antialias)) { launch_trace_x<VOL, LMODE, DIAG, BIG, true>(F, stream); return; }
    }
    launch_trace_x<VOL, LMODE, DIAG, BIG, false>(F, stream);
}

template <bool DIAG, bool BIG>
static void launch_trace_diag(const DevFrame &F, bool vol, int lmode, hipStream_t stream) {
    if (vol) {
        if (lmode == 0) launch_trace<true, 0, DIAG, BIG>(F, stream);
        else if (lmode == 1) launch_trace<true, 1, DIAG, BIG>(F, stream);
        else if (lmode == 3) launch_trace<true, 3, DIAG, BIG>(F, stream);
        else launch_trace<true, 2, DIAG, BIG>(F, stream);
    } else {
        if (lmode == 0) launch_trace<false, 0, DIAG, BIG>(F, stream);
        else if (lmode == 1) launch_trace<false, 1, DIAG, BIG>(F, stream);
        else if (lmode == 3) launch_trace<false, 3, DIAG, BIG>(F, stream);
        else launch_trace<false, 2, DIAG, BIG>(F, stream);
    }
}

// DevFrame::ray_cold of the exchanging variants (antialiased frames): 16 bytes per LDS column (a lane's or a pool slot's) of every workgroup of the resident grid
size_t trace_ray_cold_bytes(uint32_t n_cus, uint32_t *groups) {
    if (!AIC_EXCHANGE) { *groups = 0; return 0; }
    const uint32_t g = n_cus * 4u * (uint32_t)AIC_MIN_WAVES / ((uint32_t)AIC_XWG_THREADS / 64u);
    *groups = g;
    return (size_t)g * ((size_t)AIC_XWG_THREADS + (size_t)AIC_POOL) * 16u;
}

void launch_trace_image(const DevFrame &F, bool diag, hipStream_t stream) {
    const bool vol = F.layer_transparency == 1;
    const int l = F.layer_lighting;
    const int lmode = l == 0 ? 0 : (l == 1 ? 1 : (l == 5 ? 3 : 2));  // None, Flat, Bounce, the interpolated three
    const bool big = F.layer.cls_in_code == 0u;  // block table past 16384 entries: untagged cube grid
    if (diag) {
        if (big) launch_trace_diag<true, true>(F, vol, lmode, stream);
        else launch_trace_diag<true, false>(F, vol, lmode, stream);
    } else {
        if (big) launch_trace_diag<false, true>(F, vol, lmode, stream);
        else launch_trace_diag<false, false>(F, vol, lmode, stream);
    }
}

void launch_tag_cubes(uint16_t *grid, size_t n, const uint32_t *cls, int from_tagged, int to_tagged, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(tag_cubes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, grid, n, cls, from_tagged, to_tagged);
}

void launch_scatter_cubes(uint16_t *grid, uint32_t *light, const int32_t *xyz, const uint16_t *bi, const uint32_t *lt,
                          uint32_t n, const int lo[3], const int size[3], const uint32_t *cls, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(scatter_cubes_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, grid, light, xyz, bi, lt, n, lo[0],
                       lo[1], lo[2], size[0], size[1], size[2], cls);
}

void launch_probe_powf(const float *x, const float *y, float *out, uint32_t n, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(probe_powf_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, x, y, out, n);
}

void launch_order_tiles_jobs(const OrderJobs &jobs, uint32_t n_jobs, uint32_t n_tiles, uint32_t macros_x, uint32_t sb_shift, uint32_t n_queues, hipStream_t stream,
                             bool clear_cost, uint32_t n_clear_words) {
    if (!n_tiles || !n_jobs) return;
    if (n_queues < 1u) n_queues = 1u;
    if (n_queues > kMaxTileQueues) n_queues = kMaxTileQueues;
    hipLaunchKernelGGL(order_tiles_kernel, dim3(n_jobs), dim3(kOrderThreads), 0, stream, jobs, n_tiles, macros_x ? macros_x : 1u, sb_shift, n_queues, clear_cost ? 1u : 0u,
                       n_clear_words);
}
void launch_order_tiles(const uint32_t *cost, uint32_t *order, uint32_t n_tiles, uint32_t macros_x, uint32_t sb_shift, uint32_t n_queues, uint32_t *queue_start,
                        hipStream_t stream, bool clear_cost, uint32_t *clear_words, uint32_t n_clear_words) {
    OrderJobs jobs{};
    jobs.cost[0] = cost; jobs.order[0] = order; jobs.queue_start[0] = queue_start; jobs.clear_words[0] = clear_words;
    launch_order_tiles_jobs(jobs, 1u, n_tiles, macros_x, sb_shift, n_queues, stream, clear_cost, n_clear_words);
}

void launch_assemble_strips(const uint32_t *gathered, uint32_t *out, uint32_t w, uint32_t h, uint32_t strip_rows,
                            uint32_t n_parts, uint32_t max_rows, hipStream_t stream) {
    size_t n = (size_t)w * h;
    if (!n) return;
    hipLaunchKernelGGL(assemble_strips_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, gathered, out, w, h,
                       strip_rows, n_parts, max_rows);
}

void launch_probe_raycast(const double *od, int use_bounds, const int *lohi, int include_exit, uint32_t max_steps,
                          double *out_rec, uint32_t *n_out, int *ended, hipStream_t stream) {
    hipLaunchKernelGGL(probe_raycast_kernel, dim3(1), dim3(64), 0, stream, od, use_bounds, lohi, include_exit, max_steps, out_rec,
                       n_out, ended);
}

}  // namespace aic

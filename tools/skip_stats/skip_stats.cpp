// Analysis tool: how many traversal steps could exact macro-stepping through empty regions save?
// A plain f64 two-level DDA over a workload's rays (statistics only: not bit-exact with the reference and not part of the
// product or the oracle). Two questions, per level (cube grid / block voxels):
//  * skip_stats: with a per-cell "the (m+1)^3 cube ahead of this cell is empty" field (isotropic Chebyshev, or one field per ray
//    octant), how many single steps and how many macro steps of each size n = m+1 does a ray take, and how many cells does a
//    macro step cover;
//  * brick_runs: with aligned B^3 bricks marked "all invisible", how long are the runs of consecutive cells a ray spends inside
//    one empty brick (the histogram the verdict asks for).
// Driver: tools/skip_stats/run.py; results: profiles/r04_experiments.txt (A).
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
#include <algorithm>
#include <functional>
struct Vol { const uint8_t *cls; const uint8_t *skip; int sx, sy, sz; };
static int g_dir = 0;
struct Blk { Vol v; int res; int lo[3]; };
static inline size_t idx(const Vol &v, int x, int y, int z) { return ((size_t)x * v.sy + y) * v.sz + z; }
struct Stats { double cells=0, lookups_single=0, macro[16]={0}, macro_cells[16]={0}; double hits=0, enters=0; };
// walk a volume from entry point; returns: 0 left volume, 1 stopped (opaque)
// p: origin in volume coords (continuous), d: direction. lo-bounded [0,s)
struct Ray { double o[3], d[3]; };
template <class OnCell>
static int walk(const Vol &v, const Ray &r, double t0, Stats &S, int nq[], int nqn, OnCell on) {
    // find entry
    double tmin = t0, tmax = 1e300;
    int s[3] = {v.sx, v.sy, v.sz};
    for (int a = 0; a < 3; a++) {
        if (r.d[a] == 0) { if (r.o[a] < 0 || r.o[a] >= s[a]) return 0; continue; }
        double ta = (0 - r.o[a]) / r.d[a], tb = (s[a] - r.o[a]) / r.d[a];
        if (ta > tb) std::swap(ta, tb);
        tmin = std::max(tmin, ta); tmax = std::min(tmax, tb);
    }
    if (tmin >= tmax) return 0;
    double te = tmin + 1e-9;
    int c[3]; double t[3], td[3]; int st[3];
    for (int a = 0; a < 3; a++) {
        double p = r.o[a] + r.d[a] * te;
        c[a] = (int)std::floor(p); if (c[a] < 0) c[a] = 0; if (c[a] >= s[a]) c[a] = s[a] - 1;
        st[a] = r.d[a] > 0 ? 1 : (r.d[a] < 0 ? -1 : 0);
        td[a] = st[a] ? 1.0 / std::fabs(r.d[a]) : 1e300;
        t[a] = st[a] ? ((st[a] > 0 ? (c[a] + 1 - r.o[a]) : (c[a] - r.o[a])) / r.d[a]) : 1e300;
    }
    for (;;) {
        // lookup at c
        S.cells++;
        size_t i = idx(v, c[0], c[1], c[2]);
        int cl = v.cls[i];
        int res = on(cl, c, std::min(std::min(t[0], t[1]), t[2]));
        if (res == 1) return 1;
        int oct = (r.d[0] > 0 ? 4 : 0) | (r.d[1] > 0 ? 2 : 0) | (r.d[2] > 0 ? 1 : 0);
        int m = cl == 0 ? v.skip[(g_dir ? (size_t)oct * v.sx * v.sy * v.sz : 0) + i] : 0;
        // quantise
        int n = 1;
        for (int k = 0; k < nqn; k++) if (m + 1 >= nq[k]) n = std::max(n, nq[k]);
        if (n == 1) S.lookups_single++; else S.macro[n]++;
        int moved[3] = {0, 0, 0};
        int steps = 0;
        for (;;) {
            int a = (t[0] < t[1]) ? ((t[0] < t[2]) ? 0 : 2) : ((t[1] < t[2]) ? 1 : 2);
            t[a] += td[a]; c[a] += st[a]; moved[a]++; steps++;
            if (c[a] < 0 || c[a] >= s[a]) { S.cells++; /* exit step counted */ if (n > 1) S.macro_cells[n] += steps; return 0; }
            if (moved[a] >= n) break;
            S.cells++;  // intermediate empty cell
        }
        if (n > 1) S.macro_cells[n] += steps;
    }
}
// ---- aligned empty bricks: run lengths ----
struct Runs { double hist[40] = {0}; double cells = 0, in_empty = 0; };
template <class OnCell>
static int walk_cells(const Vol &v, const Ray &r, OnCell on) {  // visits every cell (single steps); on() returns 1 to stop
    double tmin = 0, tmax = 1e300;
    int s[3] = {v.sx, v.sy, v.sz};
    for (int a = 0; a < 3; a++) {
        if (r.d[a] == 0) { if (r.o[a] < 0 || r.o[a] >= s[a]) return 0; continue; }
        double ta = (0 - r.o[a]) / r.d[a], tb = (s[a] - r.o[a]) / r.d[a];
        if (ta > tb) std::swap(ta, tb);
        tmin = std::max(tmin, ta); tmax = std::min(tmax, tb);
    }
    if (tmin >= tmax) return 0;
    double te = tmin + 1e-9;
    int c[3]; double t[3], td[3]; int st[3];
    for (int a = 0; a < 3; a++) {
        double p = r.o[a] + r.d[a] * te;
        c[a] = (int)std::floor(p); if (c[a] < 0) c[a] = 0; if (c[a] >= s[a]) c[a] = s[a] - 1;
        st[a] = r.d[a] > 0 ? 1 : (r.d[a] < 0 ? -1 : 0);
        td[a] = st[a] ? 1.0 / std::fabs(r.d[a]) : 1e300;
        t[a] = st[a] ? ((st[a] > 0 ? (c[a] + 1 - r.o[a]) : (c[a] - r.o[a])) / r.d[a]) : 1e300;
    }
    for (;;) {
        if (on(c) == 1) return 1;
        int a = (t[0] < t[1]) ? ((t[0] < t[2]) ? 0 : 2) : ((t[1] < t[2]) ? 1 : 2);
        t[a] += td[a]; c[a] += st[a];
        if (c[a] < 0 || c[a] >= s[a]) return 0;
    }
}
static void run_level(const Vol &v, const Ray &r, int B, Runs &R, const uint8_t *brick_empty, int bx, int by, int bz,
                      const std::function<int(int, const int *)> &inner) {
    long cur = -1; int len = 0;
    auto flush = [&]() { if (len > 0) { R.hist[len < 39 ? len : 39]++; R.in_empty += len; } len = 0; cur = -1; };
    walk_cells(v, r, [&](const int c[3]) -> int {
        R.cells++;
        long b = ((long)(c[0] / B) * by + (c[1] / B)) * bz + (c[2] / B);
        if (brick_empty[b]) { if (b != cur) { flush(); cur = b; } len++; }
        else flush();
        return inner(v.cls[idx(v, c[0], c[1], c[2])], c);
    });
    flush();
}
extern "C" void brick_runs(const uint8_t *gcls, const uint16_t *gblk, const int gs[3], const int glo[3], int nblk, const uint8_t **bcls,
                           const int *bres, const int *blo, const int *bsz, const uint8_t *g_empty, const uint8_t **b_empty, int B,
                           const double *rays, long nrays, double *out) {
    Runs Ro, Ri;
    Vol g{gcls, nullptr, gs[0], gs[1], gs[2]};
    auto nb = [&](int n) { return (n + B - 1) / B; };
    for (long k = 0; k < nrays; k++) {
        Ray r; for (int a = 0; a < 3; a++) { r.o[a] = rays[6 * k + a] - glo[a]; r.d[a] = rays[6 * k + 3 + a]; }
        run_level(g, r, B, Ro, g_empty, nb(gs[0]), nb(gs[1]), nb(gs[2]), [&](int cl, const int *c) -> int {
            if (cl == 0 || cl == 2) return 0;
            if (cl == 1) return 1;
            int b = gblk[idx(g, c[0], c[1], c[2])];
            Vol bv{bcls[b], nullptr, bsz[3 * b], bsz[3 * b + 1], bsz[3 * b + 2]};
            Ray q; double Rr = bres[b];
            for (int a = 0; a < 3; a++) { q.o[a] = (r.o[a] - c[a]) * Rr - blo[3 * b + a]; q.d[a] = r.d[a] * Rr; }
            int stop = 0;
            run_level(bv, q, B, Ri, b_empty[b], nb(bv.sx), nb(bv.sy), nb(bv.sz), [&](int cl2, const int *) -> int { if (cl2 == 1) { stop = 1; return 1; } return 0; });
            return stop;
        });
    }
    double *o = out;
    for (Runs *R : {&Ro, &Ri}) { *o++ = R->cells; *o++ = R->in_empty; for (int i = 0; i < 40; i++) *o++ = R->hist[i]; }
}
extern "C" void set_dir(int d) { g_dir = d; }
extern "C" void skip_stats(const uint8_t *gcls, const uint8_t *gskip, const uint16_t *gblk, const int gs[3], const int glo[3],
                           int nblk, const uint8_t **bcls, const uint8_t **bskip, const int *bres, const int *blo, const int *bsz,
                           const double *rays, long nrays, int *nq, int nqn, double *out) {
    Stats So, Si;
    Vol g{gcls, gskip, gs[0], gs[1], gs[2]};
    for (long k = 0; k < nrays; k++) {
        Ray r; for (int a = 0; a < 3; a++) { r.o[a] = rays[6 * k + a] - glo[a]; r.d[a] = rays[6 * k + 3 + a]; }
        walk(g, r, 0.0, So, nq, nqn, [&](int cl, const int c[3], double texit) -> int {
            if (cl == 0) return 0;
            if (cl == 1) { So.hits++; return 1; }
            if (cl == 2) { So.hits++; return 0; }
            // recursive
            int b = gblk[idx(g, c[0], c[1], c[2])];
            So.enters++;
            Vol bv{bcls[b], bskip[b], bsz[3 * b], bsz[3 * b + 1], bsz[3 * b + 2]};
            Ray q; double R = bres[b];
            for (int a = 0; a < 3; a++) { q.o[a] = (r.o[a] - c[a]) * R - blo[3 * b + a]; q.d[a] = r.d[a] * R; }
            // sub-ray: t scaled by 1/R: same t parameter since d scaled by R
            int res = walk(bv, q, 0.0, Si, nq, nqn, [&](int cl2, const int *, double) -> int {
                if (cl2 == 0) return 0;
                Si.hits++;
                return cl2 == 1 ? 1 : 0;
            });
            return res;
        });
    }
    double *o = out;
    for (Stats *S : {&So, &Si}) {
        *o++ = S->cells; *o++ = S->lookups_single; *o++ = S->hits; *o++ = S->enters;
        for (int n = 0; n < 16; n++) { *o++ = S->macro[n]; *o++ = S->macro_cells[n]; }
    }
}

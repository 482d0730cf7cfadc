// Analysis tool: a discrete-event model of trace_image_kernel's wave scheduler, to compare lane-exchange
// policies between the waves of a workgroup ("regime-sorted waves") before -- and beside -- building them on the GPU.
// Statistics only: a plain f64 two-level DDA turns every pixel's ray into the sequence of things the kernel does for it (fast
// steps, full-pass steps, SHADE / ENTER / RAY events); the model then runs the kernel's scheduler over those sequences, wave by
// wave, with the instruction counts of each phase as its cost. Not bit-exact with the reference, not part of the product or the
// oracle. Driver: tools/wave_sim/run.py; results: profiles/r05_experiments.txt (A).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <queue>
#include <deque>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------------------------------
// part 1: rays -> token strings
//   'f'  a step that finds an invisible cube / voxel inside the bounds (fast-step material)
//   'l'  the first lookup of a level (FRESH): no step, needs a full pass; finds nothing
//   'S'  a step (or first lookup) that finds a surface: full pass, then a SHADE event; the ray goes on
//   'O'  the same, and the SHADE event ends the ray (the span makes it opaque: finish_now)
//   'E'  finds a recursive block: full pass, then an ENTER event
//   'L'  leaves a block (full pass with the leave path), the cube grid goes on
//   'X'  leaves the cube grid / cannot go on: full pass, then the ray is finished
// A ray's string always ends in 'O' or 'X'; "" = the ray misses the space (NEWRAY -> FINISH at once).
struct Vol { const float *alpha; int sx, sy, sz; };  // alpha < 0: invisible; else the voxel's alpha (emission-only voxels: 0)
static inline size_t vidx(const Vol &v, int x, int y, int z) { return ((size_t)x * v.sy + y) * v.sz + z; }
struct Ray { double o[3], d[3]; };
struct Dda {
    int c[3], st[3], s[3]; double t[3], td[3]; double t_in; bool ok;
    Dda(const Vol &v, const Ray &r) {
        s[0] = v.sx; s[1] = v.sy; s[2] = v.sz;
        double tmin = 0, tmax = 1e300; ok = true;
        for (int a = 0; a < 3; a++) {
            if (r.d[a] == 0) { if (r.o[a] < 0 || r.o[a] >= s[a]) ok = false; continue; }
            double ta = (0 - r.o[a]) / r.d[a], tb = (s[a] - r.o[a]) / r.d[a];
            if (ta > tb) std::swap(ta, tb);
            tmin = std::max(tmin, ta); tmax = std::min(tmax, tb);
        }
        if (tmin >= tmax) ok = false;
        if (!ok) return;
        t_in = tmin;
        const double te = tmin + 1e-9;
        for (int a = 0; a < 3; a++) {
            double p = r.o[a] + r.d[a] * te;
            c[a] = (int)std::floor(p); if (c[a] < 0) c[a] = 0; if (c[a] >= s[a]) c[a] = s[a] - 1;
            st[a] = r.d[a] > 0 ? 1 : (r.d[a] < 0 ? -1 : 0);
            td[a] = st[a] ? 1.0 / std::fabs(r.d[a]) : 1e300;
            t[a] = st[a] ? ((st[a] > 0 ? (c[a] + 1 - r.o[a]) : (c[a] - r.o[a])) / r.d[a]) : 1e300;
        }
    }
    double t_next() const { return std::min(std::min(t[0], t[1]), t[2]); }
    bool step() {  // false: left the volume
        int a = (t[0] < t[1]) ? ((t[0] < t[2]) ? 0 : 2) : ((t[1] < t[2]) ? 1 : 2);
        t[a] += td[a]; c[a] += st[a];
        return c[a] >= 0 && c[a] < s[a];
    }
};

struct Scene {
    Vol grid; const uint16_t *gblk; const int8_t *gcls;  // gcls: 0 invisible, 1 single visible voxel, 3 recursive
    const float *galpha;                                  // alpha of a single-voxel block's voxel
    int nblk; const float **balpha; const int *bres, *blo, *bsz; int glo[3];
};

static void ray_tokens(const Scene &S, const Ray &r0, std::string &out) {
    out.clear();
    Ray r = r0; for (int a = 0; a < 3; a++) r.o[a] -= S.glo[a];
    const double dlen = std::sqrt(r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2]);
    Dda g(S.grid, r);
    if (!g.ok) return;
    float T = 1.0f; int count = 0;
    bool first = true;
    auto surface = [&](float a, double t_enter, double t_exit) -> bool {  // true: the span makes the ray opaque (it ends here)
        float th = (float)((t_exit - t_enter) * dlen); if (th < 0) th = 0;
        float tr = a >= 1.0f ? 0.0f : (a <= 0.0f ? 1.0f : std::pow(1.0f - a, th));
        T *= tr;
        return T < 1.0f / 256.0f;
    };
    double te = g.t_in;  // t at which the current cube was entered
    for (;;) {
        if (!first) { te = g.t_next(); if (!g.step()) { out += 'X'; return; } }
        if (++count > 1000) { out += 'X'; return; }
        const size_t gi = vidx(S.grid, g.c[0], g.c[1], g.c[2]);
        const int cl = S.gcls[gi];
        if (cl == 0) { out += first ? 'l' : 'f'; first = false; continue; }
        first = false;
        if (cl == 1) {
            if (surface(S.galpha[gi], te, g.t_next())) { out += 'O'; return; }
            out += 'S';
            continue;
        }
        // recursive block
        out += 'E';
        const int b = S.gblk[gi];
        Vol bv{S.balpha[b], S.bsz[3 * b], S.bsz[3 * b + 1], S.bsz[3 * b + 2]};
        Ray q; const double R = S.bres[b];
        for (int a = 0; a < 3; a++) { q.o[a] = (r.o[a] - g.c[a]) * R - S.blo[3 * b + a]; q.d[a] = r.d[a] * R; }
        Dda in(bv, q);
        if (!in.ok) { out += 'L'; continue; }  // the ray misses the stored volume: the level is dead, left at the next pass
        bool ifirst = true;
        double ite = std::max(in.t_in, te);  // (the sub-ray's t is the ray's t: positions and direction are both scaled by R)
        for (;;) {
            if (!ifirst) { ite = in.t_next(); if (!in.step()) { out += 'L'; break; } }
            if (++count > 1000) { out += 'X'; return; }
            const float a = bv.alpha[vidx(bv, in.c[0], in.c[1], in.c[2])];
            if (a < 0.0f) { out += ifirst ? 'l' : 'f'; ifirst = false; continue; }
            ifirst = false;
            if (surface(a, ite, in.t_next())) { out += 'O'; return; }
            out += 'S';
        }
    }
}

extern "C" long make_tokens(const float *galpha_vol, const int8_t *gcls, const uint16_t *gblk, const int gs[3], const int glo[3], int nblk,
                            const float **balpha, const int *bres, const int *blo, const int *bsz, const double *rays, long nrays,
                            char *out, long out_cap, long *offsets) {
    Scene S{Vol{galpha_vol, gs[0], gs[1], gs[2]}, gblk, gcls, galpha_vol, nblk, balpha, bres, blo, bsz, {glo[0], glo[1], glo[2]}};
    std::string tok; long pos = 0;
    for (long k = 0; k < nrays; k++) {
        Ray r; for (int a = 0; a < 3; a++) { r.o[a] = rays[6 * k + a]; r.d[a] = rays[6 * k + 3 + a]; }
        ray_tokens(S, r, tok);
        offsets[k] = pos;
        if (pos + (long)tok.size() > out_cap) return -1;
        memcpy(out + pos, tok.data(), tok.size()); pos += (long)tok.size();
    }
    offsets[nrays] = pos;
    return pos;
}

// ---------------------------------------------------------------------------------------------------------------------
// part 2: the scheduler model
struct Params {
    int width, height;
    int n_cus, wg_per_cu, waves_per_wg;      // 256, 4, 4
    int t_batch, n_few, frac_t, frac_n;      // 32, 24, 4, 3
    int step_reps, fast_steps, fast_min;     // 2, 16, 16
    // instruction counts
    int c_sched, c_fast, c_pass, c_leave, c_shade, c_enter, c_finish, c_refill, c_newray;
    double cyc_per_inst;     // SIMD cycles per instruction with the SIMD's waves all busy
    double cyc_lone;         // cycles per instruction of a wave alone on its SIMD
    double lat_step;         // exposed memory latency of a dependent lookup (cycles)
    // exchange
    int pool;                // slots per workgroup (0: no exchange)
    int reservoir;           // 1: a lane that deposits into a free slot takes a new pixel (extra columns); 0: it idles until it picks a ray up
    int c_xchg_base, c_xchg_move;  // instructions of a scheduler round's pool scan; of moving rays (once per round that moves any)
    int min_gain;            // exchange only if it adds at least this many lanes to the phase
    int policy;              // 0: top up the kind the wave would run anyway; 1: choose the kind by own + pool lanes
    int deposit_free;        // deposit minority lanes into free slots (0 / 1)
    int keep_free;           // free slots the deposits leave alone
    int cold_order;          // 1: tiles in index order (cold frame); 0: costliest first
    int hot;                 // > 0 (with cold_order 1): a ray of more than this many tokens that finishes puts the 8 macro tiles around its own ahead of the queue
    int retire;              // > 0: once the queue is dry, a wave with at most this many rays parks all of them (if the slots are free and another wave of
                             // the workgroup lives) and ends: the frame's tail is run by fewer, fuller waves
};
struct Lane { int ray; int pos; int kind; };  // kind: 0 stepping, 1 SHADE, 2 ENTER, 3 RAY (finish/newray), 4 idle (bubble), 5 done
enum { K_STEP = 0, K_SHADE = 1, K_ENTER = 2, K_RAY = 3, K_IDLE = 4, K_DONE = 5 };
struct Slot { int ray, pos, kind; };  // kind -1: free
struct Wave { Lane l[64]; double clock; int wg; int simd; bool done; int tile; int next_idx; bool dry; int queues_tried; };

struct Out {
    double makespan, busy_inst, phases[4], lanes[4], fast_iters, fast_lanes, pass_iters, pass_lanes, trips, trip_lanes, xchg_rounds, xchg_moved,
        sched_rounds, inst_kind[6], dry_time_median, inst_dry, retired;
};

extern "C" int simulate(const char *tok, const long *off, const Params *Pp, Out *O) {
    const Params &P = *Pp;
    const int W = P.width, H = P.height;
    const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
    const int macros_x = (tiles_x + 1) / 2, macros_y = (tiles_y + 1) / 2;
    const int n_macro = macros_x * macros_y;
    // macro tile cost = its longest ray (tokens ~ steps)
    std::vector<int> cost(n_macro, 0);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        long k = (long)y * W + x; int len = (int)(off[k + 1] - off[k]);
        int m = (y / 16) * macros_x + (x / 16);
        cost[m] = std::max(cost[m], len > 48 ? len : 0);
    }
    // (SIM_COST=<file of n_macro int32>: another cost record -- e.g. an estimate made before the frame -- in place of the true one; a policy experiment)
    if (const char *cf = getenv("SIM_COST")) { if (FILE *f = fopen(cf, "rb")) { size_t got = fread(cost.data(), 4, (size_t)n_macro, f); fclose(f); if (got != (size_t)n_macro) return 7; } }
    // queues: super-blocks of 128 px (8 macro tiles), (bx + 3 by) mod 8
    const int NQ = 8;
    int sb_px = 1; while (sb_px * 2 <= H / 8) sb_px *= 2;
    const int sb = std::max(1, sb_px / 16);
    std::vector<std::vector<int>> queue(NQ);
    for (int m = 0; m < n_macro; m++) { int mx = m % macros_x, my = m / macros_x; queue[((mx / sb) + 3 * (my / sb)) % NQ].push_back(m); }
    if (!P.cold_order) for (auto &q : queue) std::stable_sort(q.begin(), q.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    if (P.cold_order == 2) for (auto &q : queue) {  // no cost record, but not index order either: a stride permutation of the queue (golden-ratio stride, made coprime)
        const size_t n = q.size(); if (n < 3) continue;
        size_t st = (size_t)(n * 0.6180339887); if (st < 1) st = 1;
        auto gcd = [](size_t a, size_t b) { while (b) { size_t t = a % b; a = b; b = t; } return a; };
        while (gcd(st, n) != 1) st++;
        std::vector<int> r(n); for (size_t i = 0; i < n; i++) r[i] = q[(i * st) % n];
        q.swap(r);
    }
    if (P.cold_order == 3) for (auto &q : queue) {  // bottom-up / reversed index order
        std::reverse(q.begin(), q.end());
    }
    std::vector<size_t> qpos(NQ, 0);  // in tiles: 4 per macro
    std::vector<char> mstate(n_macro, 0);  // (hot) 0 untouched, 1 started from the queue, 2 claimed by the hot list
    std::deque<int> hotq;                  // (hot) tiles handed out ahead of the queue
    long hot_pushed = 0;
    auto take_tile = [&](Wave &w) -> bool {  // sets w.tile (tile index) ; false when everything is handed out
        if (P.hot != 0 && !hotq.empty()) { w.tile = hotq.front(); hotq.pop_front(); w.next_idx = 0; return true; }
        int xcd = (w.wg % NQ);
        while (w.queues_tried < NQ) {
            int q = (xcd + w.queues_tried) % NQ;
            if (qpos[q] < queue[q].size() * 4) {
                size_t u = qpos[q]++;
                int m = queue[q][u / 4], inner = (int)(u % 4);
                int tx = (m % macros_x) * 2 + (inner & 1), ty = (m / macros_x) * 2 + (inner >> 1);
                if (tx >= tiles_x || ty >= tiles_y) continue;
                if (P.hot != 0) { if (mstate[m] == 2) continue; mstate[m] = 1; }
                w.tile = ty * tiles_x + tx; w.next_idx = 0;
                return true;
            }
            w.queues_tried++;
        }
        return false;
    };
    std::vector<char> flagged(n_macro, 0);
    auto push_hot = [&](int ray) {
        const int rx = (ray % W) / 16, ry = (ray / W) / 16;
        if (flagged[ry * macros_x + rx]) return;
        flagged[ry * macros_x + rx] = 1;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            const int nx = rx + dx, ny = ry + dy;
            if ((dx == 0 && dy == 0) || nx < 0 || ny < 0 || nx >= macros_x || ny >= macros_y) continue;
            const int nm = ny * macros_x + nx;
            if (mstate[nm] != 0) continue;
            mstate[nm] = 2; hot_pushed++;
            for (int inner = 0; inner < 4; inner++) {
                const int tx = nx * 2 + (inner & 1), ty = ny * 2 + (inner >> 1);
                if (tx < tiles_x && ty < tiles_y) hotq.push_back(ty * tiles_x + tx);
            }
        }
    };
    const int n_wg = P.n_cus * P.wg_per_cu, n_waves = n_wg * P.waves_per_wg;
    std::vector<Wave> waves(n_waves);
    std::vector<std::vector<Slot>> pools(n_wg, std::vector<Slot>(P.pool, Slot{-1, 0, -1}));
    const int simds = P.n_cus * 4;
    std::vector<int> simd_active(simds, 0);
    std::vector<int> live_in_wg(n_wg, P.waves_per_wg);
    for (int i = 0; i < n_waves; i++) {
        Wave &w = waves[i];
        w.wg = i / P.waves_per_wg; w.clock = 0; w.done = false; w.tile = -1; w.next_idx = 64; w.dry = false; w.queues_tried = 0;
        int cu = w.wg / P.wg_per_cu;
        w.simd = cu * 4 + (i % 4);
        simd_active[w.simd]++;
        for (auto &l : w.l) { l.ray = -1; l.pos = 0; l.kind = K_RAY; }
    }
    memset(O, 0, sizeof(*O));
    typedef std::pair<double, int> Ev;
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> pq;
    for (int i = 0; i < n_waves; i++) pq.push(Ev(0.0, i));
    std::vector<double> dry_times;
    auto tk = [&](const Lane &l) -> char { return tok[off[l.ray] + l.pos]; };
    auto raylen = [&](int ray) -> int { return (int)(off[ray + 1] - off[ray]); };

    long guard = 0;
    while (!pq.empty()) {
        if (++guard > 200000000L) return -2;
        int wi = pq.top().second; pq.pop();
        Wave &w = waves[wi];
        int cnt[6] = {0, 0, 0, 0, 0, 0};
        for (auto &l : w.l) cnt[l.kind]++;
        if (P.hot < 0) for (auto &l : w.l) if (l.kind <= K_RAY && l.ray >= 0 && l.pos > -P.hot) push_hot(l.ray);  // (hot < 0: a ray IN FLIGHT past -hot tokens flags its tile)
        std::vector<Slot> &pool = pools[w.wg];
        int pc[5] = {0, 0, 0, 0, 0};  // pool census by kind; [4] free
        for (auto &s : pool) pc[s.kind < 0 ? 4 : s.kind]++;
        const int alive_own = cnt[K_STEP] + cnt[K_SHADE] + cnt[K_ENTER] + cnt[K_RAY];
        if (alive_own == 0 && (P.pool == 0 || pc[4] == P.pool)) {
            // nothing left (idle lanes with an empty pool and a dry queue are done)
            bool any_idle_can_take = false;
            if (!w.dry && cnt[K_IDLE]) any_idle_can_take = true;
            if (!any_idle_can_take) { w.done = true; simd_active[w.simd]--; live_in_wg[w.wg]--; O->makespan = std::max(O->makespan, w.clock); continue; }
        }
        double inst = P.c_sched;
        int n_dep = 0;  // dependent lookups in this round (latency floor)
        if (P.retire != 0 && P.pool > 0 && w.dry && alive_own > 0 && alive_own <= std::abs(P.retire) && pc[4] >= alive_own && live_in_wg[w.wg] > 1) {
            for (auto &l : w.l) if (l.kind <= K_RAY) {
                if (l.ray >= 0 || l.kind != K_RAY) { for (auto &sl : pool) if (sl.kind < 0) { sl = Slot{l.ray, l.pos, l.kind}; break; } }
                l.ray = -1; l.kind = K_DONE;
            }
            inst += P.c_xchg_base + P.c_xchg_move;
            O->busy_inst += inst; O->inst_dry += inst; O->retired++;
            const int na_ = std::max(1, simd_active[w.simd]);
            w.clock += inst * std::max(P.cyc_lone, P.cyc_per_inst * na_);
            w.done = true; simd_active[w.simd]--; live_in_wg[w.wg]--; O->makespan = std::max(O->makespan, w.clock);
            continue;
        }
        O->sched_rounds++;
        // ---- choose the kind to run (the kernel's rule) ----
        auto choose = [&](const int c[4], int n_step_lanes) -> int {
            int best = c[K_SHADE], kind = K_SHADE;
            if (c[K_ENTER] > best) { best = c[K_ENTER]; kind = K_ENTER; }
            if (c[K_RAY] > best) { best = c[K_RAY]; kind = K_RAY; }
            int alive = n_step_lanes + c[K_SHADE] + c[K_ENTER] + c[K_RAY];
            int part_t = (alive * P.frac_t) >> 3, part_n = (alive * P.frac_n) >> 3;
            int t_lo = part_t > 0 ? part_t : 1;
            int t_batch = std::min(t_lo, P.t_batch), n_few = std::min(part_n, P.n_few);
            if (best > 0 && (best >= t_batch || n_step_lanes <= n_few)) return kind;
            return K_STEP;
        };
        int run = K_STEP;
        int own[4] = {cnt[K_STEP], cnt[K_SHADE], cnt[K_ENTER], cnt[K_RAY] + ((!w.dry) ? cnt[K_IDLE] * P.reservoir : 0)};
        if (P.pool > 0) inst += P.c_xchg_base;
        // what a kind would run with if chosen: own lanes + what the pool adds (a top-up below min_gain is not made, unless the wave has none of its own)
        int total[4], add[4];
        for (int k = 0; k < 4; k++) {
            const int mine = cnt[k] + ((k == K_RAY && P.reservoir && !w.dry) ? cnt[K_IDLE] : 0);
            int a = P.pool > 0 ? std::min(pc[k], 64 - cnt[k] - cnt[K_DONE]) : 0;
            if (a < ((w.dry && P.retire < 0) ? 1 : P.min_gain) && mine > 0) a = 0;  // (retire < 0: as -retire, and a dry wave tops up by any number)
            add[k] = a; total[k] = mine + a;
        }
        if (P.pool > 0 && P.policy >= 1 && P.policy != 3) {
            // the kind that fills the wave best; events ahead of stepping on ties and once they fill 3/4 of the wave
            run = K_STEP; int best = total[K_STEP];
            for (int k = 1; k < 4; k++) if (total[k] > best || (total[k] == best && best > 0) || (P.policy == 2 && total[k] >= 48)) { best = total[k]; run = k; if (P.policy == 2 && total[k] >= 48) break; }
        } else if (P.policy != 3) {
            run = choose(own, cnt[K_STEP]);
            if (total[run] == 0) { int best = 0; for (int k = 0; k < 4; k++) if (total[k] > best) { best = total[k]; run = k; } }
        }
        // (policy 3: the kernel's rule as built -- a wave with 48 lanes of one kind runs it as it is; otherwise the kind with the most own + parked lanes, and an
        //  exchange only for a top-up of min_gain lanes, for a kind the wave has none of, or to park at least 8 lanes)
        bool do_exchange = P.pool > 0;
        if (P.pool > 0 && P.policy == 3) {
            int cm = own[K_STEP]; run = K_STEP;
            for (int k = 1; k < 4; k++) if (own[k] >= cm) { cm = own[k]; run = k; }
            if (cm >= 48) do_exchange = false;
            else {
                int best = -1;
                for (int k = 0; k < 4; k++) { const int t = std::min(64, own[k] + pc[k]); if (t >= best) { best = t; run = k; } }
                const int mine = own[run];
                const int alive = own[0] + own[1] + own[2] + own[3];
                const int n_others = alive - mine;
                const bool may_park = !w.dry && pc[4] > 0 && n_others > 0;
                do_exchange = best - mine >= P.min_gain || (mine == 0 && best > 0) || (may_park && n_others >= 8);
                for (int k = 0; k < 4; k++) add[k] = std::min(pc[k], 64 - cnt[k] - cnt[K_DONE]);
            }
        }
        // ---- exchange: top the chosen kind up from the pool; deposit minority lanes ----
        if (do_exchange) {
            int moved = 0;
            // lanes that can give: not of kind `run`, not done. Order: idle first, then the kinds with the fewest lanes in this wave
            std::vector<int> givers;
            for (int i = 0; i < 64; i++) if (w.l[i].kind == K_IDLE) givers.push_back(i);
            int order[4] = {K_STEP, K_SHADE, K_ENTER, K_RAY};
            std::sort(order, order + 4, [&](int a, int b) { return cnt[a] < cnt[b]; });
            for (int kk = 0; kk < 4; kk++) { int k = order[kk]; if (k == run) continue; for (int i = 0; i < 64; i++) if (w.l[i].kind == k) givers.push_back(i); }
            int want = 0; for (auto &s : pool) if (s.kind == run) want++;
            int n_pair = std::min((int)givers.size(), want);
            n_pair = std::min(n_pair, add[run]);
            if (n_pair > 0) {
                size_t gi = 0;
                for (auto &s : pool) {
                    if (gi >= (size_t)n_pair) break;
                    if (s.kind != run) continue;
                    Lane &l = w.l[givers[gi++]];
                    Slot got = s;
                    if (l.kind == K_IDLE) s = Slot{-1, 0, -1}; else s = Slot{l.ray, l.pos, l.kind};
                    l.ray = got.ray; l.pos = got.pos; l.kind = got.kind;
                    moved++;
                }
                givers.erase(givers.begin(), givers.begin() + n_pair);
            }
            // deposits into free slots: lanes of kinds this wave holds few of (they would wait long here)
            if (P.deposit_free && !w.dry) {
                int free_n = 0; for (auto &s : pool) if (s.kind < 0) free_n++;
                int can = free_n - P.keep_free;
                for (size_t g = 0; g < givers.size() && can > 0; g++) {
                    Lane &l = w.l[givers[g]];
                    if (l.kind == K_IDLE) continue;
                    if (P.deposit_free == 1 && (l.kind == K_STEP || cnt[l.kind] >= 16)) continue;  // 1: only event lanes this wave holds few of
                    if (P.deposit_free == 2 && l.kind == K_STEP) continue;                          // 2: every event lane that does not run now; 3: stepping lanes too
                    for (auto &s : pool) if (s.kind < 0) { s = Slot{l.ray, l.pos, l.kind}; break; }
                    cnt[l.kind]--;
                    l.ray = -1; l.pos = 0; l.kind = P.reservoir ? K_RAY : K_IDLE;  // reservoir: the lane takes a new pixel at the next RAY phase
                    can--; moved++;
                }
            }
            if (moved) { inst += P.c_xchg_move; O->xchg_rounds++; O->xchg_moved += moved; }
        }
        // ---- run the phase ----
        int served = 0;
        if (run == K_SHADE) {
            for (auto &l : w.l) if (l.kind == K_SHADE) {
                served++;
                char c = tk(l);
                if (c == 'O') { l.kind = K_RAY; }  // finish_now
                else { l.pos++; l.kind = K_STEP; }
            }
            inst += P.c_shade; O->inst_kind[K_SHADE] += P.c_shade;
        } else if (run == K_ENTER) {
            for (auto &l : w.l) if (l.kind == K_ENTER) { served++; l.pos++; l.kind = K_STEP; }
            inst += P.c_enter; O->inst_kind[K_ENTER] += P.c_enter;
        } else if (run == K_RAY) {
            bool refilled = false;
            for (auto &l : w.l) if (l.kind == K_RAY || (l.kind == K_IDLE && P.reservoir && !w.dry)) {
                served++;
                if (P.hot > 0 && l.ray >= 0 && raylen(l.ray) > P.hot) push_hot(l.ray);  // a costly ray ends: its macro tile's neighbours are likely costly too
                // finish (if it holds a ray), then take a pixel
                for (;;) {
                    if (w.next_idx >= 64) { if (w.dry || !take_tile(w)) { if (!w.dry) { w.dry = true; dry_times.push_back(w.clock); } break; } }
                    if (w.next_idx < 64) break;
                }
                if (w.dry && w.next_idx >= 64) { l.ray = -1; l.kind = (P.pool > 0) ? K_IDLE : K_DONE; continue; }
                int pidx = w.next_idx++;
                int tx = w.tile % tiles_x, ty = w.tile / tiles_x;
                int x = tx * 8 + (pidx & 7), y = ty * 8 + (pidx >> 3);
                refilled = true;
                if (x >= W || y >= H) { l.ray = -1; l.kind = K_RAY; continue; }  // skipped pixel: asks again
                l.ray = y * W + x; l.pos = 0;
                l.kind = raylen(l.ray) == 0 ? K_RAY : K_STEP;
                if (raylen(l.ray) == 0) l.ray = -1;  // missed the space: finishes at the next RAY phase (counted as a refill only)
            }
            (void)refilled;
            inst += P.c_finish + P.c_refill + P.c_newray; O->inst_kind[K_RAY] += P.c_finish + P.c_refill + P.c_newray;
        } else {
            // a stepping trip: step_reps x (up to fast_steps fast steps + one full pass)
            O->trips++; O->trip_lanes += cnt[K_STEP];
            std::vector<int> act;
            for (int i = 0; i < 64; i++) if (w.l[i].kind == K_STEP) act.push_back(i);
            served = (int)act.size();
            for (int rep = 0; rep < P.step_reps && !act.empty(); rep++) {
                // fast steps: every stepping lane whose level is not fresh takes them until one finds something
                std::vector<int> fast;
                for (int i : act) if (tk(w.l[i]) != 'l') fast.push_back(i);
                const int fmin = w.dry ? 1 : P.fast_min;
                for (int f = 0; f < P.fast_steps; f++) {
                    if ((int)fast.size() < fmin || fast.empty()) break;
                    O->fast_iters++; O->fast_lanes += fast.size();
                    inst += P.c_fast; O->inst_kind[K_STEP] += P.c_fast; n_dep++;
                    std::vector<int> still;
                    // a lane whose step finds nothing goes on; one that finds something has TAKEN its step: the full pass handles that token
                    for (int i : fast) { Lane &l = w.l[i]; if (tk(l) == 'f') { l.pos++; still.push_back(i); } }
                    fast.swap(still);
                }
                // the full pass: every active lane handles one token
                O->pass_iters++; O->pass_lanes += act.size();
                bool any_leave = false;
                std::vector<int> next;
                for (int i : act) {
                    Lane &l = w.l[i];
                    char c = tk(l);
                    switch (c) {
                        case 'f': case 'l': l.pos++; next.push_back(i); break;
                        case 'L': l.pos++; any_leave = true; next.push_back(i); break;
                        case 'S': case 'O': l.kind = K_SHADE; break;
                        case 'E': l.kind = K_ENTER; break;
                        default: l.kind = K_RAY; break;  // 'X'
                    }
                }
                inst += P.c_pass + (any_leave ? P.c_leave : 0); O->inst_kind[K_STEP] += P.c_pass + (any_leave ? P.c_leave : 0); n_dep++;
                act.swap(next);
            }
        }
        if (run != K_STEP) { O->phases[run]++; O->lanes[run] += served; }
        O->busy_inst += inst;
        if (w.dry) O->inst_dry += inst;
        const int na = std::max(1, simd_active[w.simd]);
        const double dur = std::max(inst * std::max(P.cyc_lone, P.cyc_per_inst * na), inst * P.cyc_lone + n_dep * P.lat_step);
        w.clock += dur;
        pq.push(Ev(w.clock, wi));
    }
    if (!dry_times.empty()) { std::sort(dry_times.begin(), dry_times.end()); O->dry_time_median = dry_times[dry_times.size() / 2]; }
    return 0;
}

// Analysis tool ("first, on the CPU"): how much of the light updater's dependent chain of 32-cube batches could cross-batch
// speculation remove WITHOUT leaving the reference's update order?
//
// The proposal: besides the batch at hand (k), compute speculatively the cubes the queue would pop next, against the volume as it stands BEFORE batch k is
// applied; after applying k on the host, every cube of the ACTUAL batch k+1 whose speculative result exists and whose read set (the cubes whose texels
// compute_light read: ComputedLight::dependencies, updater.rs:820-881) contains no texel changed since that result was computed can take the result as it is --
// same inputs, same texel, same dependency list: byte-identical by construction. A batch whose 32 cubes are ALL served this way needs no launch of its own,
// which is the only way the chain gets shorter (a launch costs its ~0.09 ms whether it computes 32 cubes or 3).
//
// This tool runs the reference-order relaxation on the CPU (the test oracle's restatement, included as source: an analysis instrument, not part of the product
// and not a measurement of it) and counts, for a speculation window of 32 x (depth - 1) queue entries: per-cube hits, batches served whole, and the speculative
// computations spent per update. Driver: tools/light_spec/run.py; table: profiles/r05_experiments.txt (D).
#include "../../oracle/aic_oracle.cpp"

#include <unordered_map>

extern "C" void light_spec_stats(const orc_space *space, int32_t maximum_distance, int32_t epsilon, int32_t batch, int32_t hb_width, int32_t depth, double *out) {
    LightSpace ls(space, maximum_distance, hb_width);
    fast_evaluate_light(ls);
    const int eps = priority_from_difference(epsilon);
    struct Spec { ComputedLight c; uint64_t stamp; };
    std::unordered_map<uint32_t, Spec> cache;          // speculative results by cube
    std::vector<uint64_t> changed_at(ls.light.size(), 0);  // the step at which a texel last changed
    uint64_t step = 1;
    double batches = 0, cubes_total = 0, hits = 0, in_window = 0, whole = 0, spec_computed = 0, nearly = 0;
    std::vector<uint32_t> cubes, changed;
    std::vector<ComputedLight> results;
    for (;;) {
        if (ls.queue.peek_priority() <= eps) break;
        cubes.clear();
        for (int k = 0; k < batch; k++) { uint32_t idx; if (!ls.queue.pop(&idx)) break; cubes.push_back(idx); }
        if (cubes.empty()) break;
        batches++;
        // which of this batch's cubes have a valid speculative result?
        int served = 0;
        for (uint32_t idx : cubes) {
            cubes_total++;
            auto it = cache.find(idx);
            if (it == cache.end()) continue;
            in_window++;
            bool clean = true;
            for (const I3 &d : it->second.c.dependencies) {
                size_t di;
                if (ls.index(d, &di) && changed_at[di] > it->second.stamp) { clean = false; break; }
            }
            if (clean) { hits++; served++; }
        }
        if (served == (int)cubes.size()) whole++;
        if (served + 2 >= (int)cubes.size()) nearly++;
        // the speculation made now: the next (depth - 1) * batch queue entries, against the volume before this batch is applied
        if (depth > 1) {
            LightQueue peek = ls.queue;  // (a copy: peeking must not disturb the order)
            std::vector<uint32_t> next;
            for (int k = 0; k < (depth - 1) * batch; k++) { uint32_t idx; if (!peek.pop(&idx)) break; next.push_back(idx); }
            for (uint32_t idx : next) {
                auto it = cache.find(idx);
                if (it != cache.end()) {
                    bool clean = true;
                    for (const I3 &d : it->second.c.dependencies) { size_t di; if (ls.index(d, &di) && changed_at[di] > it->second.stamp) { clean = false; break; } }
                    if (clean) continue;  // still good: not computed again
                }
                Spec s{compute_light(ls, ls.geom.cube_of(idx)), step};
                cache[idx] = std::move(s);
                spec_computed++;
            }
        }
        // the batch itself, as the reference does it
        compute_batch(ls, cubes, results);
        for (const ComputedLight &c : results) {
            changed.clear();
            apply_light_update(ls, c, &changed);
            step++;
            for (uint32_t ci : changed) changed_at[ci] = step;
        }
        for (uint32_t idx : cubes) cache.erase(idx);
        if (cache.size() > 100000) cache.clear();
    }
    out[0] = batches; out[1] = cubes_total; out[2] = in_window; out[3] = hits; out[4] = whole; out[5] = spec_computed; out[6] = nearly;
}

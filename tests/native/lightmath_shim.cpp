// Test shim: the trace kernel's light interpolation (all_is_cubes_amd/csrc/aic_lightmath.h, the very source the HIP kernel
// inlines) compiled for the HOST, behind a C entry point, so that tests/test_lightmath_host.py can compare it with the oracle
// bit for bit without a GPU. Built by the test with g++ -ffp-contract=off (the kernel's own floating-point contract).
#include <stdint.h>

#include "aic_lightmath.h"

extern "C" uint32_t shim_interpolated_light(const uint32_t *light, const int32_t lo[3], const int32_t size[3], const uint32_t block_sky[7],
                                            const float *lut, const int32_t cube[3], const double sp[3], int32_t face, int32_t mode,
                                            float out_rgb[3]) {
    aic::LightGridView G;
    G.light = light;
    G.lo_x = lo[0]; G.lo_y = lo[1]; G.lo_z = lo[2];
    G.size_x = size[0]; G.size_y = size[1]; G.size_z = size[2];
    G.sky_nx = block_sky[0]; G.sky_ny = block_sky[1]; G.sky_nz = block_sky[2];
    G.sky_px = block_sky[3]; G.sky_py = block_sky[4]; G.sky_pz = block_sky[5]; G.sky_mean = block_sky[6];
    float fin[4];
    uint32_t n = 0;
    aic::lm_interpolated_light(G, lut, cube[0], cube[1], cube[2], sp[0], sp[1], sp[2], face, mode, fin, &n);
    const float w = fin[3] > 0.1f ? fin[3] : 0.1f;  // sr.rs:355-358 (the kernel does this part itself: fmaxf, then the division)
    out_rgb[0] = fin[0] / w; out_rgb[1] = fin[1] / w; out_rgb[2] = fin[2] / w;
    return n;
}

extern "C" uint32_t shim_light_outside(const int32_t lo[3], const int32_t size[3], const uint32_t block_sky[7], const int32_t cube[3]) {
    aic::LightGridView G;
    G.light = nullptr;
    G.lo_x = lo[0]; G.lo_y = lo[1]; G.lo_z = lo[2];
    G.size_x = size[0]; G.size_y = size[1]; G.size_z = size[2];
    G.sky_nx = block_sky[0]; G.sky_ny = block_sky[1]; G.sky_nz = block_sky[2];
    G.sky_px = block_sky[3]; G.sky_py = block_sky[4]; G.sky_pz = block_sky[5]; G.sky_mean = block_sky[6];
    return aic::lm_light_outside(G, cube[0], cube[1], cube[2]);
}

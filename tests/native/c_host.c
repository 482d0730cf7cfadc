/* c_host.c -- a plain C99 host of the drop-in boundary: include/aic_hip.h and -laic_hip, nothing else (no HIP headers, no C++,
 * no Python). It is what a binding in any language does underneath (INTEGRATION.md shows the Rust extern "C" block): make a
 * context, hand over a Space snapshot and the options, draw a frame into host memory, change two cubes, draw again.
 *
 *   c_host <scene file> <output file>
 *
 * The scene file is written by tests/test_c_host.py (the arrays of aic_space_desc, then aic_options and aic_frame_desc as the
 * header lays them out, then a cube update); the output file receives both frames' RGBA8 rows and aic_frame_info records, which
 * the test compares with the same calls made through ctypes and with the oracle. Without a usable device aic_create must fail
 * with AIC_ERR_NO_DEVICE: exit code 2 (the CPU suite runs exactly that).
 *
 * Exit codes: 0 ok, 2 no device (reported, not an error of this program), 1 anything else. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aic_hip.h"

static void *read_block(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) {
        fprintf(stderr, "c_host: short scene file\n");
        exit(1);
    }
    return p;
}

static void check(aic_ctx *ctx, int rc, const char *what) {
    if (rc != AIC_OK) {
        fprintf(stderr, "c_host: %s failed with status %d: %s\n", what, rc, aic_last_error(ctx));
        exit(1);
    }
}

int main(int argc, char **argv) {
    /* libaic_hip.so's load-time default for the HIP runtime's hardware queues is in place before main() in a host that links it */
    const char *hwq = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", hwq ? hwq : "(unset)");
    if (aic_abi_version() != AIC_ABI_VERSION) {
        fprintf(stderr, "c_host: header is ABI %d, library is ABI %d\n", AIC_ABI_VERSION, aic_abi_version());
        return 1;
    }
    int status = -1;
    aic_ctx *ctx = aic_create(0, &status);
    if (!ctx) {
        printf("aic_create: status %d%s\n", status, status == AIC_ERR_NO_DEVICE ? " (AIC_ERR_NO_DEVICE)" : "");
        return status == AIC_ERR_NO_DEVICE ? 2 : 1;
    }
    if (argc < 3) {
        fprintf(stderr, "usage: c_host <scene file> <output file>\n");
        return 1;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 1;
    }
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "AICSCENE", 8) != 0) {
        fprintf(stderr, "c_host: %s is not a scene file\n", argv[1]);
        return 1;
    }
    aic_space_desc sp;
    memset(&sp, 0, sizeof sp);
    struct { int32_t lo[3], size[3]; uint32_t n_blocks; int32_t sky_kind; uint64_t n_voxels, n_palette; float sky[8][3]; uint8_t block_sky[7][4]; uint32_t n_update; } *h =
        read_block(f, 3 * 4 + 3 * 4 + 4 + 4 + 8 + 8 + 8 * 3 * 4 + 7 * 4 + 4);
    memcpy(sp.lo, h->lo, sizeof sp.lo);
    memcpy(sp.size, h->size, sizeof sp.size);
    sp.n_blocks = h->n_blocks;
    sp.sky_kind = h->sky_kind;
    sp.n_voxels = h->n_voxels;
    sp.n_palette = h->n_palette;
    memcpy(sp.sky, h->sky, sizeof sp.sky);
    memcpy(sp.block_sky, h->block_sky, sizeof sp.block_sky);
    const size_t n_cubes = (size_t)sp.size[0] * (size_t)sp.size[1] * (size_t)sp.size[2];
    sp.block_index = read_block(f, n_cubes * 2);
    sp.light = read_block(f, n_cubes * 4);
    sp.blocks = read_block(f, (size_t)sp.n_blocks * sizeof(aic_block_desc));
    sp.voxels = read_block(f, (size_t)sp.n_voxels * 2);
    sp.palette = read_block(f, (size_t)sp.n_palette * 8 * sizeof(float));
    aic_options *opt = read_block(f, sizeof(aic_options));
    aic_frame_desc *frame = read_block(f, sizeof(aic_frame_desc));
    const uint32_t n_update = h->n_update;
    int32_t *up_xyz = read_block(f, (size_t)n_update * 3 * 4);
    uint16_t *up_block = read_block(f, (size_t)n_update * 2);
    uint8_t *up_light = read_block(f, (size_t)n_update * 4);
    fclose(f);

    char name[128];
    check(ctx, aic_device_name(ctx, name, sizeof name), "aic_device_name");
    printf("device: %s\n", name);
    check(ctx, aic_upload_space(ctx, AIC_LAYER_WORLD, &sp), "aic_upload_space");
    check(ctx, aic_set_options(ctx, AIC_LAYER_WORLD, opt), "aic_set_options");

    const size_t rows = aic_partition_rows(frame->height, &frame->partition);
    const size_t bytes = rows * frame->width * 4;
    uint8_t *image = malloc(bytes ? bytes : 1);
    aic_frame_info info[2];
    FILE *out = fopen(argv[2], "wb");
    if (!image || !out) {
        perror(argv[2]);
        return 1;
    }
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) check(ctx, aic_update_cubes(ctx, AIC_LAYER_WORLD, n_update, up_xyz, up_block, up_light), "aic_update_cubes");
        memset(&info[pass], 0, sizeof info[pass]);
        check(ctx, aic_render(ctx, frame, image, /*out_is_device=*/0, &info[pass]), "aic_render");
        printf("frame %d: %ux%u, %llu cubes traced, kernel %.3f ms, flaws %u\n", pass, frame->width, frame->height, (unsigned long long)info[pass].cubes_traced,
               info[pass].kernel_ms, info[pass].flaws);
        if (fwrite(image, 1, bytes, out) != bytes || fwrite(&info[pass], sizeof info[pass], 1, out) != 1) {
            perror(argv[2]);
            return 1;
        }
    }
    fclose(out);
    /* an invalid call is reported, not swallowed: layer 7 does not exist */
    if (aic_clear_space(ctx, 7) != AIC_ERR_INVALID) {
        fprintf(stderr, "c_host: aic_clear_space(layer 7) did not return AIC_ERR_INVALID\n");
        return 1;
    }
    aic_destroy(ctx);
    return 0;
}

/*
 * aic_oracle.h -- C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a CPU restatement of the reference raytracer hot path
 * (kpreid/all-is-cubes v0.10.0: all-is-cubes-base/src/raycast.rs,
 * all-is-cubes-render/src/raytracer/{sr,surface,accum,renderer}.rs,
 * all-is-cubes/src/raytracer_components.rs, camera/{camera_struct,viewport}.rs, ...).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker. Nothing under all_is_cubes_amd/ may include, link or
 * call it.
 *
 * Flat scene layout (shared *logical* layout with the product's C ABI so the same numpy
 * arrays can feed both; the two libraries have independent struct definitions and code).
 */
#ifndef AIC_ORACLE_H
#define AIC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One entry of the Space's block palette: the `Evoxels` of one evaluated block
 * (all-is-cubes/src/block/eval/voxel_storage.rs:199-227). */
typedef struct orc_block {
    int32_t resolution;  /* 1,2,4,...,128 (resolution.rs:18-31) */
    int32_t vlo[3];      /* lower corner of the stored voxel volume */
    int32_t vsize[3];    /* size of the stored voxel volume (may be < resolution^3) */
    uint32_t vox_off;    /* offset (in u16 elements) into orc_space.voxels, Z-major */
    uint32_t pal_off;    /* offset (in entries) into orc_space.palette */
    uint32_t pal_len;
    uint32_t is_one;     /* 1 = EvoxelsInner::One(voxel): palette[pal_off] is the voxel */
    int32_t name_char;   /* first character of the display name (text.rs:30-38), or '#' */
} orc_block;

typedef struct orc_space {
    int32_t lo[3];
    int32_t size[3];
    const uint16_t *block_index;      /* [size.x*size.y*size.z] Z-major (vol.rs:988-1023) */
    const uint8_t *light;             /* [n][4] PackedLight::as_texel r,g,b,status */
    const uint8_t *always_invisible;  /* [n] TracingCubeData.always_invisible (sr.rs:553-564) */
    uint32_t n_blocks;
    const orc_block *blocks;
    const uint16_t *voxels;
    const float *palette;             /* [m][8]: rgba, emission rgb, pad */
    int32_t sky_kind;                 /* 0 = Sky::Uniform (sky[0]), 1 = Sky::Octants */
    float sky[8][3];
} orc_space;

typedef struct orc_options {
    int32_t fog;              /* 0 None, 1 Abrupt, 2 Compromise, 3 Physical */
    int32_t transparency;     /* 0 Surface, 1 Volumetric, 2 Threshold */
    float threshold;          /* TransparencyOption::Threshold value */
    int32_t lighting;         /* 0 None, 1 Flat, 2 Coarse, 3 Linear, 4 Smoothstep, 5 Bounce */
    int32_t bounce_samples;
    int32_t antialiasing;     /* 0 None, 1 IfCheap, 2 Always */
    int32_t debug_pixel_cost;
    int32_t tone_mapping;     /* 0 Clamp, 1 Reinhard */
    float maximum_intensity;  /* may be +inf */
    float exposure;           /* Camera::exposure() */
    double view_distance;
} orc_options;

typedef struct orc_camera {
    double inverse_projection_view[16]; /* euclid order m11,m12,...,m44 (row vectors) */
    uint32_t width, height;             /* framebuffer size */
} orc_camera;

/* One step as reported by Raycaster::next (raycast.rs:301-310). */
typedef struct orc_rc_step {
    int32_t cube[3];
    int32_t face;        /* Face7 discriminant: 0 Within,1 NX,2 NY,3 NZ,4 PX,5 PY,6 PZ */
    double t_distance;
    double t_max[3];
    double intersection_point[3];
} orc_rc_step;

/* TraceStep / DepthStep records (surface.rs:220-247, 493-504). */
typedef struct orc_trace_step {
    int32_t kind;        /* TraceStep: 0 EnterSurface 1 Invisible 2 EnterBlock;
                            DepthStep: 10 Invisible 11 Span 12 EnterBlock */
    int32_t block_index;
    double t_distance;   /* surface / invisible / enter-block t */
    double exit_t_distance; /* Span only */
    float color[4];
    float emission[3];
    int32_t cube[3];
    int32_t resolution;
    int32_t voxel[3];
    double intersection_point[3];
    int32_t normal;
} orc_trace_step;

/* Per-pixel auxiliary record: the first Hit with a Position that reached the accumulator
 * (hit.rs:92-123), plus the step count of the pixel's rays. */
typedef struct orc_pixel_aux {
    int32_t hit;          /* 0 none, 1 surface hit */
    int32_t cube[3];
    int32_t voxel[3];
    int32_t resolution;
    int32_t face;
    int32_t block_index;
    uint32_t cubes_traced;
    double t_distance;
} orc_pixel_aux;

typedef struct orc_info {
    uint64_t cubes_traced;
    uint64_t n_outer;   /* outer grid lookups in bounds */
    uint64_t n_inner;   /* voxel lookups in bounds */
    uint64_t n_hits;    /* visible surfaces converted to light */
    uint64_t n_light;   /* light texel fetches */
} orc_info;

double orc_scale_to_integer_step(double s, double ds);

/* Raycaster::new(origin,dir) [.within(bounds, include_exit) if use_bounds]; writes up to
 * max_steps steps, returns the number produced; *ended = 1 if the iterator returned None. */
int32_t orc_raycast(const double origin[3], const double direction[3], int32_t use_bounds,
                    const int32_t lo[3], const int32_t hi[3], int32_t include_exit,
                    int32_t max_steps, orc_rc_step *out, int32_t *ended);

/* step `outer_index` of ray.cast() -> recursive_raycast(ray, resolution, bounds) steps. */
int32_t orc_recursive_raycast(const double origin[3], const double direction[3],
                              int32_t outer_index, int32_t resolution, const int32_t lo[3],
                              const int32_t hi[3], double sub_origin[3], int32_t max_steps,
                              orc_rc_step *out, int32_t *ended);

int32_t orc_surface_iter(const orc_space *space, const double origin[3],
                         const double direction[3], int32_t max_steps, orc_trace_step *out);
int32_t orc_depth_iter(const orc_space *space, const double origin[3], const double direction[3],
                       int32_t max_steps, orc_trace_step *out);

/* SpaceRaytracer::trace_ray with ColorBuf (out_light_t: light rgb + transmittance),
 * DepthBuf (out_depth) accumulators (two independent traces). Returns cubes_traced of the
 * ColorBuf trace. */
uint64_t orc_trace_ray(const orc_space *space, const orc_options *opt, const double origin[3],
                       const double direction[3], int32_t include_sky, float out_light_t[4],
                       double *out_depth);

/* RtRenderer::draw_rgba equivalent (renderer.rs:282-308, 498-608) without the info-text
 * overlay. ui may be NULL. backdrop alpha 0 => no backdrop. Outputs (each may be NULL):
 * rgba8 [h][w][4], linear [h][w][4] = Rgba::from(ColorBuf) before post-processing,
 * aux [h][w]. rows [row_begin,row_end) only are rendered (others untouched). */
int32_t orc_render(const orc_space *world, const orc_options *world_opt, const orc_camera *world_cam,
                   const orc_space *ui, const orc_options *ui_opt, const orc_camera *ui_cam,
                   const float backdrop[4], uint32_t row_begin, uint32_t row_end, int32_t threads,
                   uint8_t *rgba8, float *linear, orc_pixel_aux *aux, orc_info *info);

/* SpaceRaytracer::to_text with CharacterBuf (sr.rs:367-472, text.rs:52-128):
 * out[h][w]: -2 '.', -1 ' ', otherwise the character code. */
int32_t orc_render_text(const orc_space *space, const orc_options *opt, const orc_camera *cam,
                        int32_t *out);

/* Camera restatement (camera_struct.rs:387-416,459-471; camera.rs:34-40; euclid 0.22). */
void orc_look_at_y_up(const double eye[3], const double target[3], double out_quat_ijkr[4]);
void orc_eye_for_look_at(const int32_t lo[3], const int32_t hi[3], const double direction[3],
                         double out_eye[3]);
/* returns 0 if not invertible */
int32_t orc_camera_matrices(double fov_y_degrees, double view_distance, double aspect,
                            const double quat_ijkr[4], const double translation[3],
                            double out_projection[16], double out_world_to_eye[16],
                            double out_inverse_projection_view[16]);
void orc_project_ndc_into_world(const double inverse_projection_view[16], double ndc_x,
                                double ndc_y, double out_origin[3], double out_direction[3]);
void orc_unproject(const double inverse_projection_view[16], const double ndc[3], double out[3]);

/* Colour helpers. */
void orc_apply_transmittance(const float color[4], float thickness, float out_color[4],
                             float *out_coeff);
void orc_to_srgb8(const float rgba[4], uint8_t out[4]);
float orc_packed_light_scalar_out(uint8_t v);
uint8_t orc_packed_light_scalar_in(float v);
void orc_block_sky(const orc_space *space, uint8_t out_faces_mean[7][4]);
/* get_interpolated_light (sr.rs:248-359) of one surface; returns the number of get_packed_light calls */
uint32_t orc_interpolated_light(const orc_space *space, const int32_t cube[3], const double surface_point[3], int32_t face,
                                int32_t lighting, float out_rgb[3]);
double orc_smoothstep(double x);
double orc_coarsestep(double x);

/* ---- the light updater (aic_light.inc) ---- */
/* compute_derived (block/eval/derived.rs:78-240) per block: out[n_blocks][32] = color rgba, 6 face colours rgba
 * (nx ny nz px py pz), emission rgb, visible; out_opaque[n_blocks][6]. */
void orc_compute_derived(const orc_space *space, float *out, uint8_t *out_opaque);
/* the light propagation chart (space/light/chart/generator.rs): node count; weights[n][6], children[n][6] if non-null */
uint32_t orc_light_chart(float *weights, uint32_t *children);
/* LightingOption::Bounce: the restated rand 0.10 SmallRng (xoshiro256++ / SplitMix64 seeding) and rand_distr UnitSphere, for tests */
void orc_xoshiro256pp(const uint64_t state[4], uint32_t n, uint64_t *out);
void orc_small_rng(uint64_t seed, uint32_t n, uint64_t out_state[4], uint64_t *out_u64, double *out_sphere);
/* LightStorage::compute_light (updater.rs:368-417) for one cube against space->light; returns the cost */
uint64_t orc_compute_light(const orc_space *space, int32_t maximum_distance, const int32_t cube[3], uint8_t out_texel[4]);
/* Mutation::fast_evaluate_light / evaluate_light (space.rs:1496-1540); see aic_light.inc */
void orc_set_light_threads(int32_t n);
uint64_t orc_evaluate_light(const orc_space *space, int32_t maximum_distance, int32_t fast, int32_t epsilon, int32_t batch,
                            uint64_t max_updates, uint8_t *light_inout, int32_t n_queue, const int32_t *queue_cubes,
                            const int32_t *queue_priorities, int32_t hb_width);

/* ---- axis-aligned rays and the orthographic renderer (aic_ortho.inc) ---- */
int32_t orc_aa_raycast(const int32_t origin[3], int32_t direction, const float *sub_origin, int32_t zoom_resolution,
                       const int32_t zoom_cube[3], int32_t use_bounds, const int32_t lo[3], const int32_t hi[3],
                       int32_t include_exit, int32_t max_steps, orc_rc_step *out, int32_t *ended, double ray_out[6]);
void orc_ortho_image_size(const int32_t lo[3], const int32_t size[3], int32_t resolution, uint32_t *w, uint32_t *h);
void orc_ortho_views(const int32_t lo[3], const int32_t size[3], int32_t resolution, uint32_t rect[5][4], double transform[5][16],
                     double direction[5][3]);
int32_t orc_render_orthographic(const orc_space *space, int32_t resolution, uint8_t *rgba8, orc_info *info);

#ifdef __cplusplus
}
#endif
#endif

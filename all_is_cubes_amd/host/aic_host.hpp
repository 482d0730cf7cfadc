// aic_host.hpp -- C++ host-side mirror of the reference's renderer interface for the
// raytracing path, sitting ABOVE the C ABI (include/aic_hip.h) exactly where the Rust shim
// crate `all-is-cubes-hip` would sit. It stands in for that shim in this
// repository because no Rust toolchain exists here; names, argument meaning and error
// behaviour follow the reference:
//
//   GraphicsOptions & enums   all-is-cubes-render/src/camera/graphics_options.rs:28-560
//   Viewport                  camera/viewport.rs:24-163
//   Camera, look_at_y_up      camera/camera_struct.rs:43-471
//   eye_for_look_at           all-is-cubes/src/camera.rs:34-40
//   StandardCameras/UiViewState  camera/stdcam.rs:21-271 (reduced to plain values)
//   PackedLight, Sky/BlockSky all-is-cubes/src/space/light/data.rs, space/sky.rs
//   Space + SpaceChange       all-is-cubes/src/space.rs:77-131,1062-1101 (the part the raytracer reads)
//   HeadlessRenderer, Rendering, Flaws, RenderError   all-is-cubes-render/src/{headless,flaws,lib}.rs
//   HipRtRenderer             mirrors RtRenderer (raytracer/renderer.rs:35-356) and
//                             UpdatingSpaceRaytracer (raytracer/updating.rs:22-219)
//
// euclid 0.22 (un-vendored dependency of the reference) is restated for Transform3D::{then,
// inverse, transform_point3d}, Rotation3D and RigidTransform3D; its rounding is pinned by the
// exact frustum-corner values of camera/tests.rs:78-108 (tests/test_host_mirror.py).
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <limits>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/aic_hip.h"

namespace aic::host {

struct Vec3 {
    double x = 0, y = 0, z = 0;
};
struct Quat {  // euclid Rotation3D {i,j,k,r}
    double i = 0, j = 0, k = 0, r = 1;
};
struct Mat4 {  // euclid Transform3D, m[row*4+col] = m<row+1><col+1>, row-vector convention
    double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    Mat4 then(const Mat4 &other) const;
    bool inverse(Mat4 *out) const;
    bool transform_point3d(const Vec3 &p, Vec3 *out) const;  // None unless w > 0
};
struct Ray {
    Vec3 origin, direction;
};
struct GridAab {
    int32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    static GridAab from_lower_size(const int32_t lo[3], const int32_t size[3]);
    Vec3 center() const;
    int64_t volume() const;
    bool contains_cube(int32_t x, int32_t y, int32_t z) const;
};

// ---- graphics options -------------------------------------------------------------------
enum class FogOption : int { None = 0, Abrupt, Compromise, Physical };
enum class ToneMappingOperator : int { Clamp = 0, Reinhard };
enum class AntialiasingOption : int { None = 0, IfCheap, Always };
struct TransparencyOption {
    enum Kind : int { Surface = 0, Volumetric, Threshold } kind = Volumetric;
    float threshold = 0.5f;
};
struct LightingOption {
    enum Kind : int { None = 0, Flat, Coarse, Linear, Smoothstep, Bounce } kind = Linear;
    uint8_t samples = 0;
};
struct ExposureOption {
    bool automatic = false;
    float fixed = 1.0f;
    float initial() const { return automatic ? 1.0f : fixed; }
};

struct GraphicsOptions {
    FogOption fog = FogOption::Abrupt;
    double fov_y = 90.0;
    ToneMappingOperator tone_mapping = ToneMappingOperator::Clamp;
    float maximum_intensity = std::numeric_limits<float>::infinity();
    ExposureOption exposure;
    float bloom_intensity = 0.125f;
    double view_distance = 200.0;
    LightingOption lighting_display;
    TransparencyOption transparency;
    bool show_ui = true;
    AntialiasingOption antialiasing = AntialiasingOption::None;
    bool debug_info_text = true;
    bool debug_pixel_cost = false;
    static GraphicsOptions unaltered_colors();  // UNALTERED_COLORS
    GraphicsOptions repair() const;             // clamps fov_y to 1..189, view_distance to 1..10000
    aic_options to_abi() const;
    bool operator==(const GraphicsOptions &o) const;
};

struct Viewport {
    double nominal_width = 0, nominal_height = 0;
    uint32_t framebuffer_width = 0, framebuffer_height = 0;
    static Viewport with_scale(double scale_factor, uint32_t w, uint32_t h);
    double nominal_aspect_ratio() const;
    double normalize_fb_x(size_t x) const;
    double normalize_fb_y(size_t y) const;
    double normalize_fb_x_edge(size_t x) const;
    double normalize_fb_y_edge(size_t y) const;
    bool is_empty() const { return framebuffer_width == 0 || framebuffer_height == 0; }
    size_t pixel_count() const { return (size_t)framebuffer_width * framebuffer_height; }
    bool operator==(const Viewport &o) const;
};

struct ViewTransform {  // RigidTransform3D<f64, Eye, Cube>
    Quat rotation;
    Vec3 translation;
    static ViewTransform identity() { return ViewTransform(); }
};
ViewTransform look_at_y_up(const Vec3 &eye, const Vec3 &target);
Vec3 eye_for_look_at(const GridAab &bounds, const Vec3 &direction);
Quat rotation_around_x(double radians);
Quat rotation_around_y(double radians);
Quat rotation_then(const Quat &first, const Quat &second);

class Camera {
  public:
    Camera(const GraphicsOptions &options, const Viewport &viewport);
    void set_options(const GraphicsOptions &options);
    const GraphicsOptions &options() const { return options_; }
    void set_viewport(const Viewport &viewport);
    Viewport viewport() const { return viewport_; }
    void set_view_transform(const ViewTransform &t);
    void look_at_y_up(const Vec3 &eye, const Vec3 &target) { set_view_transform(host::look_at_y_up(eye, target)); }
    ViewTransform view_transform() const { return eye_to_world_; }
    void set_measured_exposure(float value);
    double fov_y() const { return options_.fov_y; }
    double view_distance() const { return options_.view_distance; }
    double near_plane_distance() const { return 1.0 / 32.0; }
    Mat4 projection_matrix() const { return projection_; }
    Mat4 view_matrix() const { return world_to_eye_; }
    Mat4 inverse_projection_view() const { return inverse_projection_view_; }
    Vec3 view_position() const { return view_position_; }
    Ray project_ndc_into_world(double ndc_x, double ndc_y) const;
    Vec3 project_ndc3_into_world(const Vec3 &ndc) const;
    float exposure() const { return exposure_value_; }
    std::array<float, 4> post_process_color(const std::array<float, 4> &rgba) const;
    aic_camera to_abi() const;

  private:
    void compute_matrices();
    GraphicsOptions options_;
    Viewport viewport_;
    ViewTransform eye_to_world_;
    Mat4 world_to_eye_, projection_, inverse_projection_view_;
    Vec3 view_position_;
    float exposure_value_ = 1.0f;
};

// ---- scene data the raytracer reads -----------------------------------------------------
struct PackedLight {
    uint8_t r = 0, g = 0, b = 0, status = 0;  // status: 0 Uninitialized, 1 NoRays, 128 Opaque, 255 Visible
    static PackedLight some(float r, float g, float b);
    static uint8_t scalar_in(float value);
    static PackedLight one() { return PackedLight{144, 144, 144, 255}; }
    bool operator==(const PackedLight &o) const { return r == o.r && g == o.g && b == o.b && status == o.status; }
};
struct Sky {
    int kind = 0;  // 0 Uniform, 1 Octants
    float colors[8][3] = {};
    void for_blocks(uint8_t out[7][4]) const;  // BlockSky faces nx..pz + mean
};
struct Evoxel {
    float color[4] = {0, 0, 0, 0};
    float emission[3] = {0, 0, 0};
};
struct Evoxels {
    int32_t resolution = 1;
    int32_t vlo[3] = {0, 0, 0}, vsize[3] = {1, 1, 1};
    std::vector<uint16_t> indices;
    std::vector<Evoxel> palette;
    bool is_one = true;
    bool is_air = false;
    std::string display_name;  // BlockAttributes::display_name: its first character is what text renderings show (text.rs:27-38)
    static Evoxels from_one(const Evoxel &v);
    static Evoxels air();
};

// EveryLight is this mirror's coalesced form of a burst of CubeLight notifications: what a light-propagation
// step that touched most of the space sends (the reference emits one CubeLight per cube, space.rs light updater).
enum class SpaceChangeKind { EveryBlock, CubeBlock, CubeLight, BlockIndex, BlockEvaluation, EveryLight };
struct SpaceChange {
    SpaceChangeKind kind;
    int32_t cube[3];
    uint32_t block_index;
};
// What UpdatingSpaceRaytracer accumulates between updates (updating.rs:180-219).
struct SpaceRendererTodo {
    bool everything = true;
    bool every_light = false;
    std::set<uint32_t> blocks;
    std::set<std::array<int32_t, 3>> cubes;
    void receive(const SpaceChange &c);
    void clear() { everything = false; every_light = false; blocks.clear(); cubes.clear(); }
};

class Space {
  public:
    Space(const GridAab &bounds);
    const GridAab &bounds() const { return bounds_; }
    Sky sky;
    // block palette
    uint32_t add_block(const Evoxels &e);                 // emits BlockIndex
    void set_block_data(uint32_t index, const Evoxels &e);  // emits BlockEvaluation
    size_t n_blocks() const { return blocks_.size(); }
    const Evoxels &block(uint32_t i) const { return blocks_.at(i); }
    // cubes
    void set(int32_t x, int32_t y, int32_t z, uint32_t block_index);   // emits CubeBlock
    void set_light(int32_t x, int32_t y, int32_t z, PackedLight l);    // emits CubeLight
    void fill_all(uint32_t block_index);                               // emits EveryBlock
    void load_contents(const uint16_t *block_index, const uint8_t *light);  // emits EveryBlock
    void load_light(const uint8_t *light);                                  // emits EveryLight
    uint16_t get_block_index(int32_t x, int32_t y, int32_t z) const;
    PackedLight get_light(int32_t x, int32_t y, int32_t z) const;
    const std::vector<uint16_t> &contents() const { return contents_; }
    const std::vector<PackedLight> &light() const { return light_; }
    // listeners (listen::Source<SpaceChange>)
    void listen(const std::shared_ptr<SpaceRendererTodo> &todo);
    size_t index(int32_t x, int32_t y, int32_t z) const;

  private:
    void notify(const SpaceChange &c);
    GridAab bounds_;
    std::vector<uint16_t> contents_;
    std::vector<PackedLight> light_;
    std::vector<Evoxels> blocks_;
    std::vector<std::weak_ptr<SpaceRendererTodo>> listeners_;
};

// ---- renderer surface -------------------------------------------------------------------
namespace Flaws {  // flaws.rs:20-80
constexpr uint16_t OTHER = 1 << 0, TOO_COMPLEX = 1 << 1, UNFINISHED = 1 << 2, UNSUPPORTED = 1 << 3;
constexpr uint16_t OUT_OF_TIME = UNFINISHED | (1 << 4), OUT_OF_MEMORY = UNFINISHED | (1 << 5);
constexpr uint16_t NO_ANTIALIASING = UNSUPPORTED | (1 << 7), NO_BLOOM = UNSUPPORTED | (1 << 8), NO_CURSOR = UNSUPPORTED | (1 << 9);
}  // namespace Flaws

struct RenderError : std::runtime_error {
    int code;
    RenderError(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

struct ImageInfo {  // renderer.rs:617-647 + RaytraceInfo sr.rs:520-522
    uint64_t cubes_traced = 0;
    uint64_t n_outer = 0, n_inner = 0, n_hits = 0, n_light = 0;
    float kernel_ms = 0, total_ms = 0;
    uint32_t width = 0, height = 0, rows_rendered = 0;
    std::string status_text() const;
};
struct Rendering {  // headless.rs:52-67
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;  // [h][w][4] sRGB RGBA8
    uint16_t flaws = 0;
    ImageInfo info;
};

// The info-text overlay (renderer.rs:659-683 `draw_info_text`): `text` drawn with Builtin::FontSystem16 by
// FontDef::draw_str_monospaced (all-is-cubes/src/text/font.rs:178-203) at offset (5, 5) into an RGBA8 image of width x
// height, outline pixels in `outline`, glyph pixels in `foreground`, in the reference's call order (a later glyph's outline
// may overwrite an earlier glyph's edge, as it does there). Pixels outside the image are dropped.
void draw_info_text(uint8_t *rgba, uint32_t width, uint32_t height, const uint8_t outline[4], const uint8_t foreground[4], const std::string &text);
// Camera::post_process_color(color).to_srgb8() (camera_struct.rs:376-382, math/color.rs:669-676, 1038-1054) of an opaque
// colour: what the reference's encoder makes of the overlay's black and white paints
void encode_paint(const class Camera &camera, float exposure, const float rgb[3], uint8_t out[4]);

struct UiViewState {  // stdcam.rs UiViewState
    std::shared_ptr<Space> space;
    ViewTransform view_transform;
    float backdrop[4] = {0, 0, 0, 0};
    GraphicsOptions graphics_options;
};
struct StandardCameras {  // stdcam.rs:90-180, reduced to values the caller sets
    GraphicsOptions graphics_options;
    Viewport viewport;
    std::shared_ptr<Space> world_space;
    ViewTransform world_view_transform;
    float measured_exposure = 1.0f;
    UiViewState ui;
};
struct Cursor {};

class HeadlessRenderer {  // headless.rs:17-44
  public:
    virtual ~HeadlessRenderer() = default;
    virtual void update(const Cursor *cursor) = 0;
    virtual Rendering draw(const std::string &info_text) = 0;
};

class HipRtRenderer : public HeadlessRenderer {
  public:
    using SizePolicy = std::function<Viewport(Viewport)>;
    // device_id < 0: current device. Throws RenderError if no MI355X / HIP device is usable.
    HipRtRenderer(std::shared_ptr<StandardCameras> cameras, SizePolicy size_policy = nullptr, int device_id = -1);
    ~HipRtRenderer() override;
    HipRtRenderer(const HipRtRenderer &) = delete;
    HipRtRenderer &operator=(const HipRtRenderer &) = delete;

    // RtRenderer::update: returns whether anything changed.
    bool update_scene(const Cursor *cursor);
    void update(const Cursor *cursor) override { (void)update_scene(cursor); }
    Rendering draw(const std::string &info_text) override { return draw_rgba(info_text); }
    Rendering draw_rgba(const std::string &info_text);
    // Replaying a recorded frame (all_is_cubes_amd/replay.py, bench.py --workload replay:<file>): the world camera's
    // inverse_projection_view and exposure exactly as the recording has them, instead of the values derived from the
    // StandardCameras' view transform (which would be equal only up to rounding). nullptr: back to the cameras.
    void set_world_camera_override(const double *inverse_projection_view, float exposure);
    // SpaceRaytracer::<CharacterRtData>::to_text::<CharacterBuf> (sr.rs:367-472, text.rs:52-128): one ray through each
    // pixel centre; the first block hit shows the first character of its display name ('#' if it has none), a ray that
    // entered the space and hit nothing ' ', one that never entered it '.', one that ran out of steps 'X'
    std::string draw_text(const std::string &line_ending = "\n");
    // multi-GPU extension: render the rows of one partition into a device buffer (no read-back)
    ImageInfo draw_rows_to_device(void *device_out, uint32_t strip_rows, uint32_t n_parts, uint32_t part, bool counters = false,
                                  bool no_feedback = false);
    uint32_t partition_rows(uint32_t strip_rows, uint32_t n_parts, uint32_t part) const;
    void assemble_strips(const void *gathered_device, void *out_device, uint32_t strip_rows, uint32_t n_parts, bool wait = true);
    // streaming pair (aic_render_submit / aic_render_wait): up to AIC_MAX_IN_FLIGHT frames in flight
    void submit_rows_to_device(void *device_out, uint32_t strip_rows, uint32_t n_parts, uint32_t part, uint32_t slot);
    // aic_render_submit_batch: 1, 2, 4 or 8 frames traced by ONE launch, frame j into device_outs[j]. `inverse_projection_views` (16 doubles per frame, may be
    // empty) gives each frame its own world camera -- a camera path known ahead of time --; empty: every frame is the current view.
    void submit_rows_batch_to_device(const std::vector<void *> &device_outs, uint32_t strip_rows, uint32_t n_parts, uint32_t part, uint32_t slot,
                                     const std::vector<std::array<double, 16>> &inverse_projection_views = {});
    ImageInfo wait_rows(uint32_t slot);
    void synchronize();  // blocks until everything queued on the context's stream is done (aic_synchronize)
    Viewport modified_viewport() const;
    const StandardCameras &cameras() const { return *cameras_; }
    std::string device_name() const;
    void *stream() const;
    void wait_event(void *hip_event);  // aic_wait_event: later frames wait for a foreign event, the host does not
    void stream_wait_rows(uint32_t slot, void *hip_stream);  // aic_stream_wait_frame: a foreign stream waits for the slot's frame, the host does not
    // Light propagation on the device (aic_evaluate_light): Mutation::fast_evaluate_light (if `fast`) then
    // Mutation::evaluate_light(epsilon) (space.rs:1496-1540) on the WORLD space as uploaded, with
    // LightPhysics::Rays { maximum_distance }; the device's light volume is updated in place (the host Space's is not).
    struct LightUpdateInfo { uint64_t updates, batches, cost; double device_ms, total_ms; uint32_t queue_left; uint64_t bundles_visited; };
    // aic_evaluate_light_submit / _wait: the same update on the library's worker thread, beside the frames submitted meanwhile (which read the light as it
    // stood); the wait -- or the next update() that changes the scene -- publishes it
    void evaluate_light_submit(int maximum_distance, bool fast, int epsilon, int batch, int queue_order, int lanes_per_cube = 0, bool continue_queue = false,
                               uint64_t max_updates = 0);
    LightUpdateInfo evaluate_light_wait();
    bool evaluate_light_done();  // aic_evaluate_light_poll: false while a submitted update is still running
    // `continue_queue`: add nothing to the layer's update queue (what update() queued through aic_light_cubes_changed is
    // drained); `max_updates`: stop after that many cube updates (a per-frame light budget), 0 = run to the end.
    LightUpdateInfo evaluate_light(int maximum_distance, bool fast = true, int epsilon = 1, int batch = 32, int queue_order = 16, int lanes_per_cube = 0,
                                   bool continue_queue = false, uint64_t max_updates = 0);
    // true: the WORLD space's light lives on the device (evaluate_light): update() forwards block changes only, never the
    // host Space's light texels, and queues the changed cubes for relighting (aic_light_cubes_changed)
    bool device_light = false;
    int device_light_queue_order = 16;
    bool enable_counters = false;

  private:
    struct LayerState {
        std::shared_ptr<Space> space;
        std::shared_ptr<SpaceRendererTodo> todo;
        GraphicsOptions options;
        bool options_valid = false;
    };
    bool sync_space(int layer, const std::shared_ptr<Space> &space, const GraphicsOptions &options);
    void upload_full(int layer, const Space &space);
    void check(int rc, const char *what);
    aic_frame_desc make_frame() const;
    std::shared_ptr<StandardCameras> cameras_;
    SizePolicy size_policy_;
    aic_ctx *ctx_ = nullptr;
    LayerState layers_[2];
    bool cam_override_ = false;
    double cam_override_inv_[16] = {0};
    float cam_override_exposure_ = 1.0f;
    bool had_cursor_ = false;
    // snapshot taken by update(): draw() must not touch the scene objects (headless.rs:33-39)
    Camera world_camera_, ui_camera_;
    float backdrop_[4] = {0, 0, 0, 0};
    bool show_ui_ = true;
};

}  // namespace aic::host

// aic_host.cpp -- see aic_host.hpp. Host logic only: everything that touches the GPU goes
// through the C ABI (include/aic_hip.h). There is no CPU rendering path.
#include "aic_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace aic::host {

// ---- euclid 0.22 restatement ---------------------------------------------------------------
#define MM(t, r, c) ((t).m[((r)-1) * 4 + ((c)-1)])

Mat4 Mat4::then(const Mat4 &o) const {
    Mat4 out;
    for (int r = 1; r <= 4; r++)
        for (int c = 1; c <= 4; c++)
            MM(out, r, c) = MM(*this, r, 1) * MM(o, 1, c) + MM(*this, r, 2) * MM(o, 2, c) + MM(*this, r, 3) * MM(o, 3, c) +
                            MM(*this, r, 4) * MM(o, 4, c);
    return out;
}

bool Mat4::inverse(Mat4 *out) const {
    const double m11 = MM(*this, 1, 1), m12 = MM(*this, 1, 2), m13 = MM(*this, 1, 3), m14 = MM(*this, 1, 4);
    const double m21 = MM(*this, 2, 1), m22 = MM(*this, 2, 2), m23 = MM(*this, 2, 3), m24 = MM(*this, 2, 4);
    const double m31 = MM(*this, 3, 1), m32 = MM(*this, 3, 2), m33 = MM(*this, 3, 3), m34 = MM(*this, 3, 4);
    const double m41 = MM(*this, 4, 1), m42 = MM(*this, 4, 2), m43 = MM(*this, 4, 3), m44 = MM(*this, 4, 4);
    const double det = m14 * m23 * m32 * m41 - m13 * m24 * m32 * m41 - m14 * m22 * m33 * m41 + m12 * m24 * m33 * m41 +
                       m13 * m22 * m34 * m41 - m12 * m23 * m34 * m41 - m14 * m23 * m31 * m42 + m13 * m24 * m31 * m42 +
                       m14 * m21 * m33 * m42 - m11 * m24 * m33 * m42 - m13 * m21 * m34 * m42 + m11 * m23 * m34 * m42 +
                       m14 * m22 * m31 * m43 - m12 * m24 * m31 * m43 - m14 * m21 * m32 * m43 + m11 * m24 * m32 * m43 +
                       m12 * m21 * m34 * m43 - m11 * m22 * m34 * m43 - m13 * m22 * m31 * m44 + m12 * m23 * m31 * m44 +
                       m13 * m21 * m32 * m44 - m11 * m23 * m32 * m44 - m12 * m21 * m33 * m44 + m11 * m22 * m33 * m44;
    if (det == 0.0) return false;
    Mat4 a;
    MM(a, 1, 1) = m23 * m34 * m42 - m24 * m33 * m42 + m24 * m32 * m43 - m22 * m34 * m43 - m23 * m32 * m44 + m22 * m33 * m44;
    MM(a, 1, 2) = m14 * m33 * m42 - m13 * m34 * m42 - m14 * m32 * m43 + m12 * m34 * m43 + m13 * m32 * m44 - m12 * m33 * m44;
    MM(a, 1, 3) = m13 * m24 * m42 - m14 * m23 * m42 + m14 * m22 * m43 - m12 * m24 * m43 - m13 * m22 * m44 + m12 * m23 * m44;
    MM(a, 1, 4) = m14 * m23 * m32 - m13 * m24 * m32 - m14 * m22 * m33 + m12 * m24 * m33 + m13 * m22 * m34 - m12 * m23 * m34;
    MM(a, 2, 1) = m24 * m33 * m41 - m23 * m34 * m41 - m24 * m31 * m43 + m21 * m34 * m43 + m23 * m31 * m44 - m21 * m33 * m44;
    MM(a, 2, 2) = m13 * m34 * m41 - m14 * m33 * m41 + m14 * m31 * m43 - m11 * m34 * m43 - m13 * m31 * m44 + m11 * m33 * m44;
    MM(a, 2, 3) = m14 * m23 * m41 - m13 * m24 * m41 - m14 * m21 * m43 + m11 * m24 * m43 + m13 * m21 * m44 - m11 * m23 * m44;
    MM(a, 2, 4) = m13 * m24 * m31 - m14 * m23 * m31 + m14 * m21 * m33 - m11 * m24 * m33 - m13 * m21 * m34 + m11 * m23 * m34;
    MM(a, 3, 1) = m22 * m34 * m41 - m24 * m32 * m41 + m24 * m31 * m42 - m21 * m34 * m42 - m22 * m31 * m44 + m21 * m32 * m44;
    MM(a, 3, 2) = m14 * m32 * m41 - m12 * m34 * m41 - m14 * m31 * m42 + m11 * m34 * m42 + m12 * m31 * m44 - m11 * m32 * m44;
    MM(a, 3, 3) = m12 * m24 * m41 - m14 * m22 * m41 + m14 * m21 * m42 - m11 * m24 * m42 - m12 * m21 * m44 + m11 * m22 * m44;
    MM(a, 3, 4) = m14 * m22 * m31 - m12 * m24 * m31 - m14 * m21 * m32 + m11 * m24 * m32 + m12 * m21 * m34 - m11 * m22 * m34;
    MM(a, 4, 1) = m23 * m32 * m41 - m22 * m33 * m41 - m23 * m31 * m42 + m21 * m33 * m42 + m22 * m31 * m43 - m21 * m32 * m43;
    MM(a, 4, 2) = m12 * m33 * m41 - m13 * m32 * m41 + m13 * m31 * m42 - m11 * m33 * m42 - m12 * m31 * m43 + m11 * m32 * m43;
    MM(a, 4, 3) = m13 * m22 * m41 - m12 * m23 * m41 - m13 * m21 * m42 + m11 * m23 * m42 + m12 * m21 * m43 - m11 * m22 * m43;
    MM(a, 4, 4) = m12 * m23 * m31 - m13 * m22 * m31 + m13 * m21 * m32 - m11 * m23 * m32 - m12 * m21 * m33 + m11 * m22 * m33;
    const double inv_det = 1.0 / det;
    for (int i = 0; i < 16; i++) out->m[i] = a.m[i] * inv_det;
    return true;
}

bool Mat4::transform_point3d(const Vec3 &p, Vec3 *out) const {
    const double x = p.x * MM(*this, 1, 1) + p.y * MM(*this, 2, 1) + p.z * MM(*this, 3, 1) + MM(*this, 4, 1);
    const double y = p.x * MM(*this, 1, 2) + p.y * MM(*this, 2, 2) + p.z * MM(*this, 3, 2) + MM(*this, 4, 2);
    const double z = p.x * MM(*this, 1, 3) + p.y * MM(*this, 2, 3) + p.z * MM(*this, 3, 3) + MM(*this, 4, 3);
    const double w = p.x * MM(*this, 1, 4) + p.y * MM(*this, 2, 4) + p.z * MM(*this, 3, 4) + MM(*this, 4, 4);
    if (w > 0.0) {
        *out = Vec3{x / w, y / w, z / w};
        return true;
    }
    return false;
}

Quat rotation_around_x(double radians) {
    const double h = radians / 2.0;
    return Quat{std::sin(h), 0.0, 0.0, std::cos(h)};
}
Quat rotation_around_y(double radians) {
    const double h = radians / 2.0;
    return Quat{0.0, std::sin(h), 0.0, std::cos(h)};
}
Quat rotation_then(const Quat &s, const Quat &o) {  // Rotation3D::then
    return Quat{o.i * s.r + o.r * s.i + o.j * s.k - o.k * s.j, o.j * s.r + o.r * s.j + o.k * s.i - o.i * s.k,
                o.k * s.r + o.r * s.k + o.i * s.j - o.j * s.i, o.r * s.r - o.i * s.i - o.j * s.j - o.k * s.k};
}
static Vec3 quat_rotate(const Quat &q, const Vec3 &p) {  // Rotation3D::transform_point3d
    const double cx = (q.j * p.z - q.k * p.y) * 2.0, cy = (q.k * p.x - q.i * p.z) * 2.0, cz = (q.i * p.y - q.j * p.x) * 2.0;
    return Vec3{p.x + q.r * cx + q.j * cz - q.k * cy, p.y + q.r * cy + q.k * cx - q.i * cz, p.z + q.r * cz + q.i * cy - q.j * cx};
}
static Mat4 quat_to_transform(const Quat &q) {  // Rotation3D::to_transform
    const double i2 = q.i + q.i, j2 = q.j + q.j, k2 = q.k + q.k;
    const double ii = q.i * i2, ij = q.i * j2, ik = q.i * k2, jj = q.j * j2, jk = q.j * k2, kk = q.k * k2;
    const double ri = q.r * i2, rj = q.r * j2, rk = q.r * k2;
    Mat4 t;
    MM(t, 1, 1) = 1.0 - (jj + kk); MM(t, 1, 2) = ij + rk; MM(t, 1, 3) = ik - rj; MM(t, 1, 4) = 0.0;
    MM(t, 2, 1) = ij - rk; MM(t, 2, 2) = 1.0 - (ii + kk); MM(t, 2, 3) = jk + ri; MM(t, 2, 4) = 0.0;
    MM(t, 3, 1) = ik + rj; MM(t, 3, 2) = jk - ri; MM(t, 3, 3) = 1.0 - (ii + jj); MM(t, 3, 4) = 0.0;
    MM(t, 4, 1) = 0.0; MM(t, 4, 2) = 0.0; MM(t, 4, 3) = 0.0; MM(t, 4, 4) = 1.0;
    return t;
}

// ---- GridAab ----------------------------------------------------------------------------------
GridAab GridAab::from_lower_size(const int32_t lo_[3], const int32_t size[3]) {
    GridAab g;
    for (int a = 0; a < 3; a++) {
        g.lo[a] = lo_[a];
        g.hi[a] = lo_[a] + size[a];
    }
    return g;
}
Vec3 GridAab::center() const {  // grid_aab.rs:391-395
    return Vec3{((double)lo[0] + (double)hi[0]) / 2., ((double)lo[1] + (double)hi[1]) / 2., ((double)lo[2] + (double)hi[2]) / 2.};
}
int64_t GridAab::volume() const { return (int64_t)(hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]); }
bool GridAab::contains_cube(int32_t x, int32_t y, int32_t z) const {
    return x >= lo[0] && x < hi[0] && y >= lo[1] && y < hi[1] && z >= lo[2] && z < hi[2];
}

// ---- GraphicsOptions ----------------------------------------------------------------------------
GraphicsOptions GraphicsOptions::unaltered_colors() {  // graphics_options.rs:168-190
    GraphicsOptions o;
    o.fog = FogOption::None;
    o.bloom_intensity = 0.0f;
    o.lighting_display.kind = LightingOption::None;
    o.exposure = ExposureOption{false, 1.0f};
    return o;
}
GraphicsOptions GraphicsOptions::repair() const {  // graphics_options.rs:194-198
    GraphicsOptions o = *this;
    o.fov_y = std::min(std::max(o.fov_y, 1.0), 189.0);
    o.view_distance = std::min(std::max(o.view_distance, 1.0), 10000.0);
    return o;
}
aic_options GraphicsOptions::to_abi() const {
    aic_options a;
    std::memset(&a, 0, sizeof(a));
    a.fog = (int)fog;
    a.transparency = (int)transparency.kind;
    a.threshold = transparency.threshold;
    a.lighting = (int)lighting_display.kind;
    a.bounce_samples = lighting_display.samples;
    a.antialiasing = (int)antialiasing;
    a.debug_pixel_cost = debug_pixel_cost ? 1 : 0;
    a.tone_mapping = (int)tone_mapping;
    a.maximum_intensity = maximum_intensity;
    a.bloom_intensity = bloom_intensity;
    a.view_distance = view_distance;
    return a;
}
bool GraphicsOptions::operator==(const GraphicsOptions &o) const {
    return fog == o.fog && fov_y == o.fov_y && tone_mapping == o.tone_mapping && maximum_intensity == o.maximum_intensity &&
           exposure.automatic == o.exposure.automatic && exposure.fixed == o.exposure.fixed && bloom_intensity == o.bloom_intensity &&
           view_distance == o.view_distance && lighting_display.kind == o.lighting_display.kind &&
           lighting_display.samples == o.lighting_display.samples && transparency.kind == o.transparency.kind &&
           transparency.threshold == o.transparency.threshold && show_ui == o.show_ui && antialiasing == o.antialiasing &&
           debug_info_text == o.debug_info_text && debug_pixel_cost == o.debug_pixel_cost;
}

// ---- Viewport (viewport.rs:24-163) -------------------------------------------------------------
Viewport Viewport::with_scale(double scale_factor, uint32_t w, uint32_t h) {
    Viewport v;
    v.framebuffer_width = w;
    v.framebuffer_height = h;
    v.nominal_width = (double)w / scale_factor;
    v.nominal_height = (double)h / scale_factor;
    if (!(v.nominal_width >= 0.0) || !(v.nominal_height >= 0.0)) throw std::invalid_argument("scale_factor must be positive");
    return v;
}
double Viewport::nominal_aspect_ratio() const {
    const double ratio = nominal_width / nominal_height;
    return std::isfinite(ratio) ? ratio : 1.0;
}
double Viewport::normalize_fb_x(size_t x) const { return ((double)x + 0.5) / (double)framebuffer_width * 2.0 - 1.0; }
double Viewport::normalize_fb_y(size_t y) const { return -(((double)y + 0.5) / (double)framebuffer_height * 2.0 - 1.0); }
double Viewport::normalize_fb_x_edge(size_t x) const { return ((double)x) / (double)framebuffer_width * 2.0 - 1.0; }
double Viewport::normalize_fb_y_edge(size_t y) const { return -(((double)y) / (double)framebuffer_height * 2.0 - 1.0); }
bool Viewport::operator==(const Viewport &o) const {
    return nominal_width == o.nominal_width && nominal_height == o.nominal_height && framebuffer_width == o.framebuffer_width &&
           framebuffer_height == o.framebuffer_height;
}

// ---- Camera (camera_struct.rs) -----------------------------------------------------------------
ViewTransform look_at_y_up(const Vec3 &eye, const Vec3 &target) {  // camera_struct.rs:459-471
    const Vec3 look{target.x - eye.x, target.y - eye.y, target.z - eye.z};
    const double yaw = std::atan2(look.x, -look.z);
    const double pitch = std::atan2(-look.y, std::sqrt(look.x * look.x + look.z * look.z));
    ViewTransform t;
    t.rotation = rotation_then(rotation_around_x(-pitch), rotation_around_y(-yaw));
    t.translation = eye;
    return t;
}
Vec3 eye_for_look_at(const GridAab &b, const Vec3 &d) {  // all-is-cubes/src/camera.rs:34-40
    double radius = 0.0;
    for (int a = 0; a < 3; a++) radius = std::fmax(radius, (double)(b.hi[a] - b.lo[a]));
    const double len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    const Vec3 c = b.center();
    return Vec3{c.x + (d.x / len) * radius, c.y + (d.y / len) * radius, c.z + (d.z / len) * radius};
}

Camera::Camera(const GraphicsOptions &options, const Viewport &viewport) : options_(options.repair()), viewport_(viewport) {
    exposure_value_ = options_.exposure.initial();
    compute_matrices();
}
void Camera::set_options(const GraphicsOptions &options) {
    options_ = options.repair();
    exposure_value_ = options_.exposure.initial();
    compute_matrices();
}
void Camera::set_viewport(const Viewport &viewport) {
    if (!(viewport == viewport_)) {
        viewport_ = viewport;
        compute_matrices();
    }
}
void Camera::set_view_transform(const ViewTransform &t) {
    eye_to_world_ = t;
    compute_matrices();
}
void Camera::set_measured_exposure(float value) {  // camera_struct.rs:169-181
    if (!(value >= 0.0f)) return;
    if (!options_.exposure.automatic) return;
    exposure_value_ = options_.lighting_display.kind == LightingOption::None ? 1.0f : value;
}
void Camera::compute_matrices() {  // camera_struct.rs:387-416
    const double fov_cot = 1.0 / std::tan((fov_y() / 2.) * (M_PI / 180.0));
    const double aspect = viewport_.nominal_aspect_ratio();
    const double near = near_plane_distance();
    const double far = view_distance();
    Mat4 p;
    std::memset(p.m, 0, sizeof(p.m));
    MM(p, 1, 1) = fov_cot / aspect;
    MM(p, 2, 2) = fov_cot;
    MM(p, 3, 3) = far / (near - far);
    MM(p, 3, 4) = -1.0;
    MM(p, 4, 3) = (far * near) / (near - far);
    projection_ = p;
    // RigidTransform3D::inverse().to_transform()
    const Quat inv{-eye_to_world_.rotation.i, -eye_to_world_.rotation.j, -eye_to_world_.rotation.k, eye_to_world_.rotation.r};
    const Vec3 it = quat_rotate(inv, Vec3{-eye_to_world_.translation.x, -eye_to_world_.translation.y, -eye_to_world_.translation.z});
    Mat4 w2e = quat_to_transform(inv);
    MM(w2e, 4, 1) = it.x; MM(w2e, 4, 2) = it.y; MM(w2e, 4, 3) = it.z;
    world_to_eye_ = w2e;
    view_position_ = eye_to_world_.translation;
    if (!world_to_eye_.then(projection_).inverse(&inverse_projection_view_))
        throw std::runtime_error("projection and view matrix was not invertible");
}
Vec3 Camera::project_ndc3_into_world(const Vec3 &ndc) const {
    Vec3 out;
    if (!inverse_projection_view_.transform_point3d(ndc, &out)) {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        return Vec3{nan, nan, nan};
    }
    return out;
}
Ray Camera::project_ndc_into_world(double x, double y) const {  // camera_struct.rs:238-251
    const Vec3 n = project_ndc3_into_world(Vec3{x, y, 0.0}), f = project_ndc3_into_world(Vec3{x, y, 1.0});
    return Ray{n, Vec3{f.x - n.x, f.y - n.y, f.z - n.z}};
}
static inline float ps_mul(float a, float b) {
    const float v = a * b;
    return v != v ? 0.f : v;
}
std::array<float, 4> Camera::post_process_color(const std::array<float, 4> &c) const {  // camera_struct.rs:376-382
    float r = ps_mul(c[0], exposure_value_), g = ps_mul(c[1], exposure_value_), b = ps_mul(c[2], exposure_value_);
    const float m = options_.maximum_intensity;
    if (std::isfinite(m)) {
        if (options_.tone_mapping == ToneMappingOperator::Clamp) {
            r = std::min(std::max(r, 0.f), m); g = std::min(std::max(g, 0.f), m); b = std::min(std::max(b, 0.f), m);
        } else {
            const float lum = g * 0.7152f + (r * 0.2126f + b * 0.0722f);
            float scale = 1.0f / (1.0f + lum / m);
            scale = scale > 0.f ? scale : 0.f;
            r = ps_mul(r, scale); g = ps_mul(g, scale); b = ps_mul(b, scale);
        }
    }
    return {r, g, b, c[3]};
}
aic_camera Camera::to_abi() const {
    aic_camera c;
    std::memcpy(c.inverse_projection_view, inverse_projection_view_.m, sizeof(c.inverse_projection_view));
    c.exposure = exposure_value_;
    c.reserved = 0;
    return c;
}

// ---- PackedLight / Sky -------------------------------------------------------------------------
uint8_t PackedLight::scalar_in(float value) {  // light/data.rs:214-218
    const float x = std::round(std::log2(value) * 10.0f + 144.0f);
    if (!(x > 0.f)) return 0;
    if (x >= 255.f) return 255;
    return (uint8_t)x;
}
PackedLight PackedLight::some(float r, float g, float b) { return PackedLight{scalar_in(r), scalar_in(g), scalar_in(b), 255}; }

void Sky::for_blocks(uint8_t out[7][4]) const {  // sky.rs:45-82
    auto put = [&](int f, const PackedLight &p) { out[f][0] = p.r; out[f][1] = p.g; out[f][2] = p.b; out[f][3] = p.status; };
    if (kind == 0) {
        for (int f = 0; f < 7; f++) put(f, PackedLight::some(colors[0][0], colors[0][1], colors[0][2]));
        return;
    }
    // images of +X,+Y,+Z under Face::rotation_from_nz (face.rs:395-404): NX NY NZ PX PY PZ
    static const int B[6][3][3] = {{{0, 1, 0}, {0, 0, 1}, {1, 0, 0}},  {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}},  {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}},
                                   {{0, -1, 0}, {0, 0, 1}, {-1, 0, 0}}, {{0, 0, 1}, {-1, 0, 0}, {0, -1, 0}}, {{1, 0, 0}, {0, -1, 0}, {0, 0, -1}}};
    static const int P[4][3] = {{-1, -1, -1}, {-1, 1, -1}, {1, -1, -1}, {1, 1, -1}};
    for (int f = 0; f < 6; f++) {
        float acc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 4; k++) {
            int d[3];
            for (int a = 0; a < 3; a++) d[a] = P[k][0] * B[f][0][a] + P[k][1] * B[f][1][a] + P[k][2] * B[f][2][a];
            const int idx = ((d[0] >= 0 ? 1 : 0) << 2) + ((d[1] >= 0 ? 1 : 0) << 1) + (d[2] >= 0 ? 1 : 0);
            for (int c = 0; c < 3; c++) acc[c] = acc[c] + colors[idx][c];
        }
        put(f, PackedLight::some(ps_mul(acc[0], 0.25f), ps_mul(acc[1], 0.25f), ps_mul(acc[2], 0.25f)));
    }
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 8; k++)
        for (int c = 0; c < 3; c++) acc[c] = acc[c] + colors[k][c];
    put(6, PackedLight::some(ps_mul(acc[0], 1.0f / 8.0f), ps_mul(acc[1], 1.0f / 8.0f), ps_mul(acc[2], 1.0f / 8.0f)));
}

Evoxels Evoxels::from_one(const Evoxel &v) {
    Evoxels e;
    e.resolution = 1;
    e.is_one = true;
    e.indices = {0};
    e.palette = {v};
    return e;
}
Evoxels Evoxels::air() {
    Evoxels e = from_one(Evoxel{});
    e.is_air = true;
    return e;
}

// ---- Space ------------------------------------------------------------------------------------
void SpaceRendererTodo::receive(const SpaceChange &c) {  // updating.rs:201-219
    switch (c.kind) {
        case SpaceChangeKind::EveryBlock: everything = true; break;
        case SpaceChangeKind::EveryLight: every_light = true; break;
        case SpaceChangeKind::CubeBlock:
        case SpaceChangeKind::CubeLight: cubes.insert({c.cube[0], c.cube[1], c.cube[2]}); break;
        case SpaceChangeKind::BlockIndex:
        case SpaceChangeKind::BlockEvaluation: blocks.insert(c.block_index); break;
    }
}

Space::Space(const GridAab &bounds) : bounds_(bounds) {
    const int64_t n = std::max<int64_t>(bounds.volume(), 0);
    contents_.assign((size_t)n, 0);
    light_.assign((size_t)n, PackedLight::one());  // LightPhysics::None (space.rs:1241-1245)
    sky.colors[0][0] = 0.89626944f; sky.colors[0][1] = 0.89626944f; sky.colors[0][2] = 1.0f;  // DAY_SKY_COLOR palette.rs:63
}
size_t Space::index(int32_t x, int32_t y, int32_t z) const {  // vol.rs:988-1023
    if (!bounds_.contains_cube(x, y, z)) throw std::out_of_range("cube outside the space bounds");
    const size_t sy = (size_t)(bounds_.hi[1] - bounds_.lo[1]), sz = (size_t)(bounds_.hi[2] - bounds_.lo[2]);
    return ((size_t)(x - bounds_.lo[0]) * sy + (size_t)(y - bounds_.lo[1])) * sz + (size_t)(z - bounds_.lo[2]);
}
void Space::notify(const SpaceChange &c) {
    for (auto it = listeners_.begin(); it != listeners_.end();) {
        if (auto l = it->lock()) {
            l->receive(c);
            ++it;
        } else it = listeners_.erase(it);
    }
}
void Space::listen(const std::shared_ptr<SpaceRendererTodo> &todo) { listeners_.push_back(todo); }
uint32_t Space::add_block(const Evoxels &e) {
    if (blocks_.size() >= 65536) throw std::length_error("too many blocks for a u16 BlockIndex");
    blocks_.push_back(e);
    const uint32_t i = (uint32_t)blocks_.size() - 1;
    notify(SpaceChange{SpaceChangeKind::BlockIndex, {0, 0, 0}, i});
    return i;
}
void Space::set_block_data(uint32_t index, const Evoxels &e) {
    blocks_.at(index) = e;
    notify(SpaceChange{SpaceChangeKind::BlockEvaluation, {0, 0, 0}, index});
}
void Space::set(int32_t x, int32_t y, int32_t z, uint32_t block_index) {
    if (block_index >= blocks_.size()) throw std::out_of_range("block index not in the palette");
    contents_[index(x, y, z)] = (uint16_t)block_index;
    notify(SpaceChange{SpaceChangeKind::CubeBlock, {x, y, z}, 0});
}
void Space::set_light(int32_t x, int32_t y, int32_t z, PackedLight l) {
    light_[index(x, y, z)] = l;
    notify(SpaceChange{SpaceChangeKind::CubeLight, {x, y, z}, 0});
}
void Space::fill_all(uint32_t block_index) {
    if (block_index >= blocks_.size()) throw std::out_of_range("block index not in the palette");
    std::fill(contents_.begin(), contents_.end(), (uint16_t)block_index);
    notify(SpaceChange{SpaceChangeKind::EveryBlock, {0, 0, 0}, 0});
}
void Space::load_contents(const uint16_t *block_index, const uint8_t *light) {
    const size_t n = contents_.size();
    if (block_index) {
        for (size_t i = 0; i < n; i++)
            if (block_index[i] >= blocks_.size()) throw std::out_of_range("block index not in the palette");
        std::memcpy(contents_.data(), block_index, n * sizeof(uint16_t));
    }
    if (light) std::memcpy((void *)light_.data(), light, n * 4);
    notify(SpaceChange{SpaceChangeKind::EveryBlock, {0, 0, 0}, 0});
}
void Space::load_light(const uint8_t *light) {
    std::memcpy((void *)light_.data(), light, contents_.size() * 4);
    notify(SpaceChange{SpaceChangeKind::EveryLight, {0, 0, 0}, 0});
}
uint16_t Space::get_block_index(int32_t x, int32_t y, int32_t z) const { return contents_[index(x, y, z)]; }
PackedLight Space::get_light(int32_t x, int32_t y, int32_t z) const { return light_[index(x, y, z)]; }

// ---- renderer ---------------------------------------------------------------------------------
std::string ImageInfo::status_text() const {  // renderer.rs:633-646
    char buf[256];
    const double per = (width && height) ? (double)cubes_traced / ((double)width * height) : 0.0;
    std::snprintf(buf, sizeof(buf), "Traced %llu cubes, %.1f cubes/pixel in %u\xC3\x97%u image", (unsigned long long)cubes_traced, per, width, height);
    return buf;
}

static_assert(sizeof(PackedLight) == 4, "PackedLight is a 4-byte texel");

HipRtRenderer::HipRtRenderer(std::shared_ptr<StandardCameras> cameras, SizePolicy size_policy, int device_id)
    : cameras_(std::move(cameras)), size_policy_(std::move(size_policy)), world_camera_(GraphicsOptions(), Viewport::with_scale(1.0, 1, 1)),
      ui_camera_(GraphicsOptions(), Viewport::with_scale(1.0, 1, 1)) {
    if (!cameras_) throw std::invalid_argument("cameras must not be null");
    int status = 0;
    ctx_ = aic_create(device_id, &status);
    if (!ctx_) throw RenderError(status, status == AIC_ERR_NO_DEVICE ? "no usable HIP device (MI355X required)" : "aic_create failed");
}
HipRtRenderer::~HipRtRenderer() { aic_destroy(ctx_); }

void HipRtRenderer::check(int rc, const char *what) {
    if (rc != AIC_OK) throw RenderError(rc, std::string(what) + ": " + aic_last_error(ctx_));
}
Viewport HipRtRenderer::modified_viewport() const { return size_policy_ ? size_policy_(cameras_->viewport) : cameras_->viewport; }
std::string HipRtRenderer::device_name() const {
    char buf[256];
    aic_device_name(ctx_, buf, sizeof(buf));
    return buf;
}
void *HipRtRenderer::stream() const { return aic_stream(ctx_); }

void HipRtRenderer::upload_full(int layer, const Space &space) {  // SpaceRaytracer::new (sr.rs:64-88)
    aic_space_desc d;
    std::memset(&d, 0, sizeof(d));
    const GridAab &b = space.bounds();
    for (int a = 0; a < 3; a++) { d.lo[a] = b.lo[a]; d.size[a] = b.hi[a] - b.lo[a]; }
    d.block_index = space.contents().data();
    d.light = reinterpret_cast<const uint8_t *>(space.light().data());
    std::vector<aic_block_desc> blocks(space.n_blocks());
    std::vector<uint16_t> vox;
    std::vector<float> pal;
    for (size_t i = 0; i < blocks.size(); i++) {
        const Evoxels &e = space.block((uint32_t)i);
        aic_block_desc &bd = blocks[i];
        std::memset(&bd, 0, sizeof(bd));
        bd.resolution = e.resolution;
        for (int a = 0; a < 3; a++) { bd.vlo[a] = e.vlo[a]; bd.vsize[a] = e.vsize[a]; }
        bd.vox_off = (uint32_t)vox.size();
        bd.pal_off = (uint32_t)(pal.size() / 8);
        bd.pal_len = (uint32_t)e.palette.size();
        bd.flags = (e.is_one ? AIC_BLOCK_ONE : 0) | (e.is_air ? AIC_BLOCK_AIR : 0);
        vox.insert(vox.end(), e.indices.begin(), e.indices.end());
        for (const Evoxel &v : e.palette) {
            pal.insert(pal.end(), v.color, v.color + 4);
            pal.insert(pal.end(), v.emission, v.emission + 3);
            pal.push_back(0.f);
        }
    }
    d.n_blocks = (uint32_t)blocks.size();
    d.blocks = blocks.data();
    d.voxels = vox.data();
    d.n_voxels = vox.size();
    d.palette = pal.data();
    d.n_palette = pal.size() / 8;
    d.sky_kind = space.sky.kind;
    std::memcpy(d.sky, space.sky.colors, sizeof(d.sky));
    space.sky.for_blocks(d.block_sky);
    check(aic_upload_space(ctx_, layer, &d), "aic_upload_space");
}

// UpdatingSpaceRaytracer::update (updating.rs:107-172)
bool HipRtRenderer::sync_space(int layer, const std::shared_ptr<Space> &space, const GraphicsOptions &options) {
    LayerState &ls = layers_[layer];
    bool changed = false;
    if (!space) {
        if (ls.space) {
            check(aic_clear_space(ctx_, layer), "aic_clear_space");
            ls.space.reset();
            ls.todo.reset();
            changed = true;
        }
        return changed;
    }
    if (ls.space != space) {  // renderer.rs:122-133: replace the raytracer
        ls.space = space;
        ls.todo = std::make_shared<SpaceRendererTodo>();
        space->listen(ls.todo);
        changed = true;
    }
    if (!ls.options_valid || !(ls.options == options)) {
        aic_options ao = options.to_abi();
        check(aic_set_options(ctx_, layer, &ao), "aic_set_options");
        ls.options = options;
        ls.options_valid = true;
        changed = true;
    }
    SpaceRendererTodo &todo = *ls.todo;
    if (todo.everything) {
        upload_full(layer, *space);
        todo.clear();
        return true;
    }
    if (!todo.blocks.empty()) {
        for (uint32_t bi : todo.blocks) {
            const Evoxels &e = space->block(bi);
            aic_block_desc bd;
            std::memset(&bd, 0, sizeof(bd));
            bd.resolution = e.resolution;
            for (int a = 0; a < 3; a++) { bd.vlo[a] = e.vlo[a]; bd.vsize[a] = e.vsize[a]; }
            bd.pal_len = (uint32_t)e.palette.size();
            bd.flags = (e.is_one ? AIC_BLOCK_ONE : 0) | (e.is_air ? AIC_BLOCK_AIR : 0);
            std::vector<float> pal;
            for (const Evoxel &v : e.palette) {
                pal.insert(pal.end(), v.color, v.color + 4);
                pal.insert(pal.end(), v.emission, v.emission + 3);
                pal.push_back(0.f);
            }
            check(aic_replace_block(ctx_, layer, bi, &bd, e.indices.data(), pal.data()), "aic_replace_block");
        }
        changed = true;
    }
    const bool dev_light = device_light && layer == AIC_LAYER_WORLD;
    if (todo.every_light && !dev_light) {  // one H2D copy of the light volume instead of a scatter (BASELINE config 5)
        check(aic_update_light_volume(ctx_, layer, reinterpret_cast<const uint8_t *>(space->light().data())), "aic_update_light_volume");
        changed = true;
    }
    if (!todo.cubes.empty()) {
        std::vector<int32_t> xyz;
        std::vector<uint16_t> bi;
        std::vector<uint8_t> lt;
        xyz.reserve(todo.cubes.size() * 3);
        for (const auto &c : todo.cubes) {
            xyz.insert(xyz.end(), c.begin(), c.end());
            bi.push_back(space->get_block_index(c[0], c[1], c[2]));
            const PackedLight p = space->get_light(c[0], c[1], c[2]);
            lt.push_back(p.r); lt.push_back(p.g); lt.push_back(p.b); lt.push_back(p.status);
        }
        check(aic_update_cubes(ctx_, layer, (uint32_t)bi.size(), xyz.data(), bi.data(), dev_light ? nullptr : lt.data()), "aic_update_cubes");
        if (dev_light) check(aic_light_cubes_changed(ctx_, layer, (uint32_t)bi.size(), xyz.data(), device_light_queue_order), "aic_light_cubes_changed");
        changed = true;
    }
    todo.clear();
    return changed;
}

bool HipRtRenderer::update_scene(const Cursor *cursor) {  // renderer.rs:96-161
    had_cursor_ = cursor != nullptr;
    const StandardCameras &sc = *cameras_;
    const Viewport vp = modified_viewport();
    world_camera_ = Camera(sc.graphics_options, vp);
    world_camera_.set_view_transform(sc.world_view_transform);
    world_camera_.set_measured_exposure(sc.measured_exposure);
    ui_camera_ = Camera(sc.ui.graphics_options, vp);
    ui_camera_.set_view_transform(sc.ui.view_transform);
    std::memcpy(backdrop_, sc.ui.backdrop, sizeof(backdrop_));
    show_ui_ = sc.graphics_options.show_ui;
    bool changed = sync_space(AIC_LAYER_WORLD, sc.world_space, world_camera_.options());
    changed |= sync_space(AIC_LAYER_UI, show_ui_ ? sc.ui.space : nullptr, ui_camera_.options());
    return changed;
}

aic_frame_desc HipRtRenderer::make_frame() const {
    aic_frame_desc f;
    std::memset(&f, 0, sizeof(f));
    const Viewport vp = world_camera_.viewport();
    f.width = vp.framebuffer_width;
    f.height = vp.framebuffer_height;
    f.world = world_camera_.to_abi();
    if (cam_override_) {
        std::memcpy(f.world.inverse_projection_view, cam_override_inv_, sizeof(cam_override_inv_));
        f.world.exposure = cam_override_exposure_;
    }
    f.ui = ui_camera_.to_abi();
    std::memcpy(f.backdrop, backdrop_, sizeof(f.backdrop));
    f.partition = aic_partition{0, 1, 0, 0};
    f.flags = enable_counters ? AIC_FRAME_COUNTERS : 0;
    return f;
}

void HipRtRenderer::set_world_camera_override(const double *inverse_projection_view, float exposure) {
    cam_override_ = inverse_projection_view != nullptr;
    if (cam_override_) std::memcpy(cam_override_inv_, inverse_projection_view, sizeof(cam_override_inv_));
    cam_override_exposure_ = exposure;
}

static ImageInfo to_info(const aic_frame_info &fi, uint32_t w, uint32_t h) {
    ImageInfo i;
    i.cubes_traced = fi.cubes_traced; i.n_outer = fi.n_outer; i.n_inner = fi.n_inner; i.n_hits = fi.n_hits; i.n_light = fi.n_light;
    i.kernel_ms = fi.kernel_ms; i.total_ms = fi.total_ms; i.width = w; i.height = h; i.rows_rendered = fi.rows_rendered;
    return i;
}

Rendering HipRtRenderer::draw_rgba(const std::string &info_text) {  // renderer.rs:282-308
    aic_frame_desc f = make_frame();
    Rendering r;
    r.width = f.width;
    r.height = f.height;
    r.data.assign((size_t)f.width * f.height * 4, 0);
    aic_frame_info fi;
    check(aic_render(ctx_, &f, r.data.data(), 0, &fi), "aic_render");
    r.info = to_info(fi, f.width, f.height);
    if (fi.flaws & AIC_FLAW_UNSUPPORTED) r.flaws |= Flaws::UNSUPPORTED;
    if (fi.flaws & AIC_FLAW_NO_BLOOM) r.flaws |= Flaws::NO_BLOOM;
    if (had_cursor_) r.flaws |= Flaws::NO_CURSOR;
    // The info-text overlay (renderer.rs:205-217, 659-683): drawn over the finished frame on the host, in the encoder's
    // black and white, when the world camera's options ask for it.
    if (!info_text.empty() && world_camera_.options().debug_info_text && f.width && f.height) {
        uint8_t black[4], white[4];
        const float k0[3] = {0.f, 0.f, 0.f}, k1[3] = {1.f, 1.f, 1.f};
        // (the exposure the kernel's encoder used for this frame: the recorded one when a camera override is active -- ADVICE r03)
        const float paint_exposure = cam_override_ ? cam_override_exposure_ : world_camera_.exposure();
        encode_paint(world_camera_, paint_exposure, k0, black);
        encode_paint(world_camera_, paint_exposure, k1, white);
        draw_info_text(r.data.data(), f.width, f.height, black, white, info_text);
    }
    return r;
}

// ---- the info-text overlay ---------------------------------------------------------------------------------------------

#include "font_system16.inc"

namespace {
// FontDef::char_to_glyph_index (text/font.rs:214-229): ISO-8859-1 plus the curly quotes; anything else is '?'
uint32_t glyph_index_of(uint32_t c) {
    if (c == 0x2018u || c == 0x2019u) c = '\'';
    if (c == 0x201cu || c == 0x201du) c = '"';
    if (c >= 0x20u && c <= 0x7fu) return c - 0x20u;
    if (c >= 0xa0u && c <= 0xffu) return c - 0x40u;
    return 0x1fu;
}
bool glyph_bit(uint32_t g, int x, int y) { return x >= 0 && x < 7 && y >= 0 && y < 16 && ((kFontSystem16[g][y] >> x) & 1u) != 0u; }
}  // namespace

void draw_info_text(uint8_t *rgba, uint32_t width, uint32_t height, const uint8_t outline[4], const uint8_t foreground[4], const std::string &text) {
    constexpr int kCellW = 7, kCellH = 16;  // FONT_SYSTEM_16 metrics (font.rs:23-30)
    constexpr int kOriginX = 5, kOriginY = 5;  // renderer.rs:668
    // text::compute_layout with Left / BodyTop / Back in GridAab::ORIGIN_CUBE, no outline expansion (layout.rs:101-265): the
    // first line's glyph origins are at y = 0, every further line 16 lower (y is up there; draw_str_monospaced flips it), and
    // a glyph that draws nothing (space) still advances the cursor.
    int cursor_x = 0, line = 0;
    for (size_t i = 0; i < text.size();) {
        // next UTF-8 scalar (`str::chars`); malformed input cannot occur in a Rust &str, here it decodes to U+FFFD
        uint32_t c = (unsigned char)text[i];
        size_t len = 1;
        if (c >= 0xf0u) { len = 4; c &= 0x07u; } else if (c >= 0xe0u) { len = 3; c &= 0x0fu; } else if (c >= 0xc0u) { len = 2; c &= 0x1fu; } else if (c >= 0x80u) { c = 0xfffdu; }
        if (i + len > text.size()) { c = 0xfffdu; len = 1; }
        else for (size_t k = 1; k < len; k++) c = (c << 6) | ((unsigned char)text[i + k] & 0x3fu);
        i += len;
        if (c == '\n') { cursor_x = 0; line++; continue; }
        const uint32_t g = glyph_index_of(c);
        const int gx0 = cursor_x, gy0 = line * kCellH;
        cursor_x += kCellW;
        // Glyphs::new / Glyphs::get (font.rs:340-520): the stored box is the bounding box of the glyph's set pixels grown by one
        // pixel on every side; inside it a set pixel is Foreground, a pixel 8-adjacent to one (within this glyph) is Outline
        int minx = 99, miny = 99, maxx = -1, maxy = -1;
        for (int y = 0; y < kCellH; y++)
            for (int x = 0; x < kCellW; x++)
                if (glyph_bit(g, x, y)) { minx = std::min(minx, x); maxx = std::max(maxx, x); miny = std::min(miny, y); maxy = std::max(maxy, y); }
        if (maxx < 0) continue;  // draws nothing
        for (int y = miny - 1; y <= maxy + 1; y++)
            for (int x = minx - 1; x <= maxx + 1; x++) {
                const uint8_t *paint;
                if (glyph_bit(g, x, y)) paint = foreground;
                else {
                    bool near = false;
                    for (int dy = -1; dy <= 1 && !near; dy++)
                        for (int dx = -1; dx <= 1; dx++)
                            if (glyph_bit(g, x + dx, y + dy)) { near = true; break; }
                    if (!near) continue;
                    paint = outline;
                }
                const int px = gx0 + x + kOriginX, py = gy0 + y + kOriginY;
                if (px < 0 || py < 0 || (uint32_t)px >= width || (uint32_t)py >= height) continue;
                std::memcpy(rgba + ((size_t)py * width + (size_t)px) * 4, paint, 4);
            }
    }
}

void encode_paint(const Camera &camera, float exposure, const float rgb_in[3], uint8_t out[4]) {
    const GraphicsOptions &o = camera.options();
    float c[3];
    for (int k = 0; k < 3; k++) {  // rgb * exposure (PositiveSign: 0 * inf = 0)
        const float v = rgb_in[k] * exposure;
        c[k] = (v != v) ? 0.f : v;
    }
    if (std::isfinite(o.maximum_intensity)) {  // ToneMappingOperator::apply (graphics_options.rs:352-368)
        if (o.tone_mapping == ToneMappingOperator::Clamp) {
            for (int k = 0; k < 3; k++) c[k] = c[k] > o.maximum_intensity ? o.maximum_intensity : c[k];
        } else {
            const float lum = c[1] * 0.7152f + (c[0] * 0.2126f + c[2] * 0.0722f);  // Rgb::luminance (color.rs)
            const float scale = 1.0f / (1.0f + lum / o.maximum_intensity);
            for (int k = 0; k < 3; k++) { const float v = c[k] * scale; c[k] = (v != v) ? 0.f : v; }
        }
    }
    for (int k = 0; k < 3; k++) {  // component_to_srgb8 (color.rs:1038-1054)
        const float e = c[k] <= 0.0031308f ? c[k] * (323.f / 25.f) : (211.f * std::pow(c[k], 5.f / 12.f) - 11.f) / 200.f;
        const float r = std::round(e * 255.f);
        out[k] = !(r > 0.f) ? 0 : (r >= 255.f ? 255 : (uint8_t)r);
    }
    out[3] = 255;
}

std::string HipRtRenderer::draw_text(const std::string &line_ending) {
    aic_frame_desc f = make_frame();
    f.flags |= AIC_FRAME_AUX | AIC_FRAME_PIXEL_CENTERS;
    const size_t n = (size_t)f.width * f.height;
    std::vector<uint8_t> rgba(n * 4);
    std::vector<aic_pixel_aux> aux(n);
    aic_frame_info fi;
    check(aic_render(ctx_, &f, rgba.data(), 0, &fi), "aic_render");
    if (n) check(aic_read_aux(ctx_, aux.data(), n), "aic_read_aux");
    std::string out;
    for (uint32_t y = 0; y < f.height; y++) {
        for (uint32_t x = 0; x < f.width; x++) {
            const aic_pixel_aux &a = aux[(size_t)y * f.width + x];
            if (a.hit == 1) {
                // the character comes from the block data of the layer that was hit (text.rs:52-128): aux.layer
                const std::shared_ptr<Space> &space = layers_[a.layer == 1u ? AIC_LAYER_UI : AIC_LAYER_WORLD].space;
                const std::string &name = (space && (size_t)a.block_index < space->n_blocks()) ? space->block((uint32_t)a.block_index).display_name : std::string();
                if (name.empty()) out += '#';
                else {  // first UTF-8 scalar (graphemes of more than one scalar are beyond this mirror)
                    size_t len = 1;
                    const unsigned char c0 = (unsigned char)name[0];
                    if (c0 >= 0xf0) len = 4; else if (c0 >= 0xe0) len = 3; else if (c0 >= 0xc0) len = 2;
                    out.append(name, 0, std::min(len, name.size()));
                }
            } else if (a.cubes_traced > 1000u) out += 'X';      // Exception::Incomplete (sr.rs:643-651)
            else out += (a.cubes_traced > 0u ? ' ' : '.');       // EnteredSpace / Empty (text.rs:111-119)
        }
        out += line_ending;
    }
    return out;
}

uint32_t HipRtRenderer::partition_rows(uint32_t strip_rows, uint32_t n_parts, uint32_t part) const {
    aic_partition p{strip_rows, n_parts, part, 0};
    return aic_partition_rows(world_camera_.viewport().framebuffer_height, &p);
}
ImageInfo HipRtRenderer::draw_rows_to_device(void *device_out, uint32_t strip_rows, uint32_t n_parts, uint32_t part, bool counters,
                                             bool no_feedback) {
    aic_frame_desc f = make_frame();
    f.partition = aic_partition{strip_rows, n_parts, part, 0};
    if (counters) f.flags |= AIC_FRAME_COUNTERS;
    if (no_feedback) f.flags |= AIC_FRAME_NO_FEEDBACK;
    aic_frame_info fi;
    check(aic_render(ctx_, &f, device_out, 1, &fi), "aic_render");
    return to_info(fi, f.width, f.height);
}
void HipRtRenderer::assemble_strips(const void *gathered_device, void *out_device, uint32_t strip_rows, uint32_t n_parts, bool wait) {
    const Viewport vp = world_camera_.viewport();
    if (wait)
        check(aic_assemble_strips(ctx_, gathered_device, out_device, vp.framebuffer_width, vp.framebuffer_height, strip_rows, n_parts),
              "aic_assemble_strips");
    else
        check(aic_assemble_strips_async(ctx_, gathered_device, out_device, vp.framebuffer_width, vp.framebuffer_height, strip_rows, n_parts),
              "aic_assemble_strips_async");
}

void HipRtRenderer::submit_rows_to_device(void *device_out, uint32_t strip_rows, uint32_t n_parts, uint32_t part, uint32_t slot) {
    aic_frame_desc f = make_frame();
    f.partition = aic_partition{strip_rows, n_parts, part, 0};
    check(aic_render_submit(ctx_, &f, device_out, slot), "aic_render_submit");
}
void HipRtRenderer::submit_rows_batch_to_device(const std::vector<void *> &device_outs, uint32_t strip_rows, uint32_t n_parts, uint32_t part, uint32_t slot,
                                                const std::vector<std::array<double, 16>> &inverse_projection_views) {
    if (!inverse_projection_views.empty() && inverse_projection_views.size() != device_outs.size())
        throw std::invalid_argument("submit_rows_batch_to_device: one camera per frame, or none");
    std::vector<aic_frame_desc> frames(device_outs.size(), make_frame());
    for (size_t j = 0; j < frames.size(); j++) {
        frames[j].partition = aic_partition{strip_rows, n_parts, part, 0};
        if (!inverse_projection_views.empty()) std::memcpy(frames[j].world.inverse_projection_view, inverse_projection_views[j].data(), sizeof(double) * 16);
    }
    check(aic_render_submit_batch(ctx_, (uint32_t)frames.size(), frames.data(), device_outs.data(), slot), "aic_render_submit_batch");
}
ImageInfo HipRtRenderer::wait_rows(uint32_t slot) {
    aic_frame_info fi;
    check(aic_render_wait(ctx_, slot, &fi), "aic_render_wait");
    const Viewport vp = world_camera_.viewport();
    return to_info(fi, vp.framebuffer_width, vp.framebuffer_height);
}
void HipRtRenderer::synchronize() { check(aic_synchronize(ctx_), "aic_synchronize"); }
HipRtRenderer::LightUpdateInfo HipRtRenderer::evaluate_light(int maximum_distance, bool fast, int epsilon, int batch, int queue_order, int lanes_per_cube,
                                                              bool continue_queue, uint64_t max_updates) {
    aic_light_params p;
    std::memset(&p, 0, sizeof(p));
    p.maximum_distance = maximum_distance;
    p.fast = fast ? 1 : 0;
    p.epsilon = epsilon;
    p.batch = batch;
    p.queue_order = queue_order;
    p.n_queue = continue_queue ? 0 : -1;
    p.max_updates = max_updates;
    p.lanes_per_cube = lanes_per_cube;
    aic_light_info info;
    check(aic_evaluate_light(ctx_, AIC_LAYER_WORLD, &p, &info), "aic_evaluate_light");
    return LightUpdateInfo{info.updates, info.batches, info.cost, info.device_ms, info.total_ms, info.queue_left, info.bundles_visited};
}

void HipRtRenderer::evaluate_light_submit(int maximum_distance, bool fast, int epsilon, int batch, int queue_order, int lanes_per_cube, bool continue_queue,
                                          uint64_t max_updates) {
    aic_light_params p;
    std::memset(&p, 0, sizeof(p));
    p.maximum_distance = maximum_distance;
    p.fast = fast ? 1 : 0;
    p.epsilon = epsilon;
    p.batch = batch;
    p.queue_order = queue_order;
    p.n_queue = continue_queue ? 0 : -1;
    p.max_updates = max_updates;
    p.lanes_per_cube = lanes_per_cube;
    check(aic_evaluate_light_submit(ctx_, AIC_LAYER_WORLD, &p), "aic_evaluate_light_submit");
}
bool HipRtRenderer::evaluate_light_done() {
    int done = 1;
    check(aic_evaluate_light_poll(ctx_, AIC_LAYER_WORLD, &done), "aic_evaluate_light_poll");
    return done != 0;
}
HipRtRenderer::LightUpdateInfo HipRtRenderer::evaluate_light_wait() {
    aic_light_info info;
    check(aic_evaluate_light_wait(ctx_, AIC_LAYER_WORLD, &info), "aic_evaluate_light_wait");
    return LightUpdateInfo{info.updates, info.batches, info.cost, info.device_ms, info.total_ms, info.queue_left, info.bundles_visited};
}

void HipRtRenderer::wait_event(void *hip_event) { check(aic_wait_event(ctx_, hip_event), "aic_wait_event"); }
void HipRtRenderer::stream_wait_rows(uint32_t slot, void *hip_stream) { check(aic_stream_wait_frame(ctx_, slot, hip_stream), "aic_stream_wait_frame"); }

}  // namespace aic::host

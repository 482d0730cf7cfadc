// py_module.cpp -- pybind11 exposure of the C++ host mirror (aic_host.hpp) so that the Python
// tests and bench.py can drive the same classes a C++ (or, via the C ABI, Rust) caller uses.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>

#include "aic_host.hpp"

namespace py = pybind11;
using namespace aic::host;

static Vec3 v3(const std::array<double, 3> &a) { return Vec3{a[0], a[1], a[2]}; }
static std::array<double, 3> a3(const Vec3 &v) { return {v.x, v.y, v.z}; }
static py::array_t<double> mat(const Mat4 &m) {
    py::array_t<double> out({4, 4});
    std::memcpy(out.mutable_data(), m.m, sizeof(m.m));
    return out;
}

PYBIND11_MODULE(_host, m) {
    m.doc() = "C++ host mirror of the all-is-cubes raytracer interface above the MI355X C ABI";

    py::register_exception<RenderError>(m, "RenderError");

    py::enum_<FogOption>(m, "FogOption").value("None_", FogOption::None).value("Abrupt", FogOption::Abrupt)
        .value("Compromise", FogOption::Compromise).value("Physical", FogOption::Physical);
    py::enum_<ToneMappingOperator>(m, "ToneMappingOperator").value("Clamp", ToneMappingOperator::Clamp).value("Reinhard", ToneMappingOperator::Reinhard);
    py::enum_<AntialiasingOption>(m, "AntialiasingOption").value("None_", AntialiasingOption::None).value("IfCheap", AntialiasingOption::IfCheap)
        .value("Always", AntialiasingOption::Always);
    py::enum_<TransparencyOption::Kind>(m, "TransparencyKind").value("Surface", TransparencyOption::Surface)
        .value("Volumetric", TransparencyOption::Volumetric).value("Threshold", TransparencyOption::Threshold);
    py::enum_<LightingOption::Kind>(m, "LightingKind").value("None_", LightingOption::None).value("Flat", LightingOption::Flat)
        .value("Coarse", LightingOption::Coarse).value("Linear", LightingOption::Linear).value("Smoothstep", LightingOption::Smoothstep)
        .value("Bounce", LightingOption::Bounce);

    py::class_<TransparencyOption>(m, "TransparencyOption").def(py::init<>())
        .def(py::init([](TransparencyOption::Kind k, float t) { TransparencyOption o; o.kind = k; o.threshold = t; return o; }), py::arg("kind"), py::arg("threshold") = 0.5f)
        .def_readwrite("kind", &TransparencyOption::kind).def_readwrite("threshold", &TransparencyOption::threshold);
    py::class_<LightingOption>(m, "LightingOption").def(py::init<>())
        .def(py::init([](LightingOption::Kind k, int s) { LightingOption o; o.kind = k; o.samples = (uint8_t)s; return o; }), py::arg("kind"), py::arg("samples") = 0)
        .def_readwrite("kind", &LightingOption::kind).def_readwrite("samples", &LightingOption::samples);
    py::class_<ExposureOption>(m, "ExposureOption").def(py::init<>())
        .def_readwrite("automatic", &ExposureOption::automatic).def_readwrite("fixed", &ExposureOption::fixed);

    py::class_<GraphicsOptions>(m, "GraphicsOptions")
        .def(py::init<>())
        .def_static("unaltered_colors", &GraphicsOptions::unaltered_colors)
        .def("repair", &GraphicsOptions::repair)
        .def_readwrite("fog", &GraphicsOptions::fog).def_readwrite("fov_y", &GraphicsOptions::fov_y)
        .def_readwrite("tone_mapping", &GraphicsOptions::tone_mapping).def_readwrite("maximum_intensity", &GraphicsOptions::maximum_intensity)
        .def_readwrite("exposure", &GraphicsOptions::exposure).def_readwrite("bloom_intensity", &GraphicsOptions::bloom_intensity)
        .def_readwrite("view_distance", &GraphicsOptions::view_distance).def_readwrite("lighting_display", &GraphicsOptions::lighting_display)
        .def_readwrite("transparency", &GraphicsOptions::transparency).def_readwrite("show_ui", &GraphicsOptions::show_ui)
        .def_readwrite("antialiasing", &GraphicsOptions::antialiasing).def_readwrite("debug_info_text", &GraphicsOptions::debug_info_text)
        .def_readwrite("debug_pixel_cost", &GraphicsOptions::debug_pixel_cost);

    py::class_<Viewport>(m, "Viewport")
        .def(py::init<>())
        .def_static("with_scale", &Viewport::with_scale)
        .def_readwrite("nominal_width", &Viewport::nominal_width).def_readwrite("nominal_height", &Viewport::nominal_height)
        .def_readwrite("framebuffer_width", &Viewport::framebuffer_width).def_readwrite("framebuffer_height", &Viewport::framebuffer_height)
        .def("nominal_aspect_ratio", &Viewport::nominal_aspect_ratio)
        .def("normalize_fb_x", &Viewport::normalize_fb_x).def("normalize_fb_y", &Viewport::normalize_fb_y)
        .def("normalize_fb_x_edge", &Viewport::normalize_fb_x_edge).def("normalize_fb_y_edge", &Viewport::normalize_fb_y_edge)
        .def("is_empty", &Viewport::is_empty).def("pixel_count", &Viewport::pixel_count);

    py::class_<ViewTransform>(m, "ViewTransform")
        .def(py::init<>())
        .def_static("identity", &ViewTransform::identity)
        .def_property("rotation", [](const ViewTransform &t) { return std::array<double, 4>{t.rotation.i, t.rotation.j, t.rotation.k, t.rotation.r}; },
                      [](ViewTransform &t, const std::array<double, 4> &q) { t.rotation = Quat{q[0], q[1], q[2], q[3]}; })
        .def_property("translation", [](const ViewTransform &t) { return a3(t.translation); },
                      [](ViewTransform &t, const std::array<double, 3> &v) { t.translation = v3(v); });
    m.def("look_at_y_up", [](const std::array<double, 3> &eye, const std::array<double, 3> &target) { return look_at_y_up(v3(eye), v3(target)); });
    m.def("eye_for_look_at", [](const std::array<int32_t, 3> &lo, const std::array<int32_t, 3> &hi, const std::array<double, 3> &dir) {
        GridAab b;
        for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; }
        return a3(eye_for_look_at(b, v3(dir)));
    });
    m.def("draw_info_text", [](py::array_t<uint8_t, py::array::c_style> image, const std::string &text, const std::array<uint8_t, 4> &outline,
                               const std::array<uint8_t, 4> &foreground) {
        if (image.ndim() != 3 || image.shape(2) != 4) throw std::invalid_argument("draw_info_text: image must be [h][w][4] uint8");
        draw_info_text(image.mutable_data(), (uint32_t)image.shape(1), (uint32_t)image.shape(0), outline.data(), foreground.data(), text);
    }, py::arg("image"), py::arg("text"), py::arg("outline") = std::array<uint8_t, 4>{0, 0, 0, 255}, py::arg("foreground") = std::array<uint8_t, 4>{255, 255, 255, 255});
    m.def("rotation_around_y", [](double radians) { Quat q = rotation_around_y(radians); return std::array<double, 4>{q.i, q.j, q.k, q.r}; });

    py::class_<Camera>(m, "Camera")
        .def(py::init<const GraphicsOptions &, const Viewport &>())
        .def("set_options", &Camera::set_options).def("options", &Camera::options)
        .def("set_viewport", &Camera::set_viewport).def("viewport", &Camera::viewport)
        .def("set_view_transform", &Camera::set_view_transform).def("view_transform", &Camera::view_transform)
        .def("look_at_y_up", [](Camera &c, const std::array<double, 3> &eye, const std::array<double, 3> &target) { c.look_at_y_up(v3(eye), v3(target)); })
        .def("set_measured_exposure", &Camera::set_measured_exposure).def("exposure", &Camera::exposure)
        .def("fov_y", &Camera::fov_y).def("view_distance", &Camera::view_distance).def("near_plane_distance", &Camera::near_plane_distance)
        .def("projection_matrix", [](const Camera &c) { return mat(c.projection_matrix()); })
        .def("view_matrix", [](const Camera &c) { return mat(c.view_matrix()); })
        .def("inverse_projection_view", [](const Camera &c) { return mat(c.inverse_projection_view()); })
        .def("view_position", [](const Camera &c) { return a3(c.view_position()); })
        .def("project_ndc_into_world", [](const Camera &c, double x, double y) { Ray r = c.project_ndc_into_world(x, y); return py::make_tuple(a3(r.origin), a3(r.direction)); })
        .def("project_ndc3_into_world", [](const Camera &c, const std::array<double, 3> &p) { return a3(c.project_ndc3_into_world(v3(p))); })
        .def("post_process_color", &Camera::post_process_color);

    py::class_<PackedLight>(m, "PackedLight")
        .def(py::init<>())
        .def_static("some", &PackedLight::some).def_static("scalar_in", &PackedLight::scalar_in).def_static("one", &PackedLight::one)
        .def(py::init([](int r, int g, int b, int s) { return PackedLight{(uint8_t)r, (uint8_t)g, (uint8_t)b, (uint8_t)s}; }))
        .def("as_texel", [](const PackedLight &p) { return std::array<int, 4>{p.r, p.g, p.b, p.status}; });

    py::class_<Sky>(m, "Sky")
        .def(py::init<>())
        .def_readwrite("kind", &Sky::kind)
        .def("set_uniform", [](Sky &s, const std::array<float, 3> &c) { s.kind = 0; std::memset(s.colors, 0, sizeof(s.colors)); for (int i = 0; i < 3; i++) s.colors[0][i] = c[i]; })
        .def("set_octants", [](Sky &s, py::array_t<float, py::array::c_style | py::array::forcecast> c) {
            if (c.size() != 24) throw std::invalid_argument("octants needs 8x3 floats");
            s.kind = 1;
            std::memcpy(s.colors, c.data(), sizeof(s.colors));
        })
        .def("for_blocks", [](const Sky &s) {
            uint8_t out[7][4];
            s.for_blocks(out);
            py::array_t<uint8_t> a({7, 4});
            std::memcpy(a.mutable_data(), out, sizeof(out));
            return a;
        });

    py::class_<Evoxels>(m, "Evoxels")
        .def_static("from_one", [](const std::array<float, 4> &rgba, const std::array<float, 3> &em) {
            Evoxel v;
            for (int i = 0; i < 4; i++) v.color[i] = rgba[i];
            for (int i = 0; i < 3; i++) v.emission[i] = em[i];
            return Evoxels::from_one(v);
        }, py::arg("rgba"), py::arg("emission") = std::array<float, 3>{0, 0, 0})
        .def_static("air", &Evoxels::air)
        .def_static("paletted", [](int resolution, const std::array<int32_t, 3> &vlo, py::array_t<uint16_t, py::array::c_style | py::array::forcecast> idx,
                                   py::array_t<float, py::array::c_style | py::array::forcecast> pal) {
            if (idx.ndim() != 3) throw std::invalid_argument("indices must be 3-D");
            if (pal.ndim() != 2 || pal.shape(1) != 8) throw std::invalid_argument("palette must be [n][8]");
            Evoxels e;
            e.resolution = resolution;
            e.is_one = false;
            for (int a = 0; a < 3; a++) { e.vlo[a] = vlo[a]; e.vsize[a] = (int32_t)idx.shape(a); }
            e.indices.assign(idx.data(), idx.data() + idx.size());
            e.palette.resize((size_t)pal.shape(0));
            for (size_t i = 0; i < e.palette.size(); i++) {
                const float *p = pal.data() + 8 * i;
                std::memcpy(e.palette[i].color, p, 16);
                std::memcpy(e.palette[i].emission, p + 4, 12);
            }
            return e;
        })
        .def_readwrite("resolution", &Evoxels::resolution).def_readwrite("is_air", &Evoxels::is_air).def_readwrite("is_one", &Evoxels::is_one)
        .def_readwrite("display_name", &Evoxels::display_name);

    py::class_<Space, std::shared_ptr<Space>>(m, "Space")
        .def(py::init([](const std::array<int32_t, 3> &lo, const std::array<int32_t, 3> &size) {
            return std::make_shared<Space>(GridAab::from_lower_size(lo.data(), size.data()));
        }))
        .def_readwrite("sky", &Space::sky)
        .def("add_block", &Space::add_block).def("set_block_data", &Space::set_block_data).def("n_blocks", &Space::n_blocks)
        .def("set", &Space::set).def("set_light", &Space::set_light).def("fill_all", &Space::fill_all)
        .def("get_block_index", &Space::get_block_index)
        .def("load_contents", [](Space &s, py::array_t<uint16_t, py::array::c_style | py::array::forcecast> bi, py::object light) {
            if ((size_t)bi.size() != s.contents().size()) throw std::invalid_argument("block_index size mismatch");
            const uint8_t *lp = nullptr;
            py::array_t<uint8_t, py::array::c_style | py::array::forcecast> la;
            if (!light.is_none()) {
                la = light.cast<py::array_t<uint8_t, py::array::c_style | py::array::forcecast>>();
                if ((size_t)la.size() != s.contents().size() * 4) throw std::invalid_argument("light size mismatch");
                lp = la.data();
            }
            s.load_contents(bi.data(), lp);
        }, py::arg("block_index"), py::arg("light") = py::none())
        .def("load_light", [](Space &s, py::array_t<uint8_t, py::array::c_style | py::array::forcecast> la) {
            if ((size_t)la.size() != s.contents().size() * 4) throw std::invalid_argument("light size mismatch");
            s.load_light(la.data());
        }, py::arg("light"))
        .def("bounds", [](const Space &s) { const GridAab &b = s.bounds(); return py::make_tuple(std::array<int32_t, 3>{b.lo[0], b.lo[1], b.lo[2]}, std::array<int32_t, 3>{b.hi[0], b.hi[1], b.hi[2]}); });

    py::class_<UiViewState>(m, "UiViewState")
        .def(py::init<>())
        .def_readwrite("space", &UiViewState::space).def_readwrite("view_transform", &UiViewState::view_transform)
        .def_readwrite("graphics_options", &UiViewState::graphics_options)
        .def_property("backdrop", [](const UiViewState &u) { return std::array<float, 4>{u.backdrop[0], u.backdrop[1], u.backdrop[2], u.backdrop[3]}; },
                      [](UiViewState &u, const std::array<float, 4> &b) { for (int i = 0; i < 4; i++) u.backdrop[i] = b[i]; });
    py::class_<StandardCameras, std::shared_ptr<StandardCameras>>(m, "StandardCameras")
        .def(py::init([]() { return std::make_shared<StandardCameras>(); }))
        .def_readwrite("graphics_options", &StandardCameras::graphics_options).def_readwrite("viewport", &StandardCameras::viewport)
        .def_readwrite("world_space", &StandardCameras::world_space).def_readwrite("world_view_transform", &StandardCameras::world_view_transform)
        .def_readwrite("measured_exposure", &StandardCameras::measured_exposure).def_readwrite("ui", &StandardCameras::ui);

    py::class_<ImageInfo>(m, "ImageInfo")
        .def_readonly("cubes_traced", &ImageInfo::cubes_traced).def_readonly("n_outer", &ImageInfo::n_outer).def_readonly("n_inner", &ImageInfo::n_inner)
        .def_readonly("n_hits", &ImageInfo::n_hits).def_readonly("n_light", &ImageInfo::n_light).def_readonly("kernel_ms", &ImageInfo::kernel_ms)
        .def_readonly("total_ms", &ImageInfo::total_ms).def_readonly("width", &ImageInfo::width).def_readonly("height", &ImageInfo::height)
        .def_readonly("rows_rendered", &ImageInfo::rows_rendered).def("status_text", &ImageInfo::status_text);
    py::class_<Rendering>(m, "Rendering")
        .def_readonly("width", &Rendering::width).def_readonly("height", &Rendering::height).def_readonly("flaws", &Rendering::flaws)
        .def_readonly("info", &Rendering::info)
        .def_property_readonly("data", [](const Rendering &r) {
            py::array_t<uint8_t> a({(py::ssize_t)r.height, (py::ssize_t)r.width, (py::ssize_t)4});
            if (!r.data.empty()) std::memcpy(a.mutable_data(), r.data.data(), r.data.size());
            return a;
        });

    auto flaws = m.def_submodule("Flaws");
    flaws.attr("OTHER") = Flaws::OTHER; flaws.attr("UNSUPPORTED") = Flaws::UNSUPPORTED; flaws.attr("NO_BLOOM") = Flaws::NO_BLOOM;
    flaws.attr("NO_CURSOR") = Flaws::NO_CURSOR; flaws.attr("OUT_OF_MEMORY") = Flaws::OUT_OF_MEMORY;

    py::class_<Cursor>(m, "Cursor").def(py::init<>());

    py::class_<HipRtRenderer>(m, "HipRtRenderer")
        .def(py::init([](std::shared_ptr<StandardCameras> cams, py::object policy, int device_id) {
            HipRtRenderer::SizePolicy sp = nullptr;
            if (!policy.is_none()) sp = policy.cast<HipRtRenderer::SizePolicy>();
            return std::make_unique<HipRtRenderer>(std::move(cams), sp, device_id);
        }), py::arg("cameras"), py::arg("size_policy") = py::none(), py::arg("device_id") = -1)
        .def("update", [](HipRtRenderer &r, py::object cursor) { Cursor c; return r.update_scene(cursor.is_none() ? nullptr : &c); }, py::arg("cursor") = py::none())
        .def("draw", [](HipRtRenderer &r, const std::string &t) { py::gil_scoped_release rel; return r.draw(t); }, py::arg("info_text") = "")
        .def("draw_text", [](HipRtRenderer &r, const std::string &le) { py::gil_scoped_release rel; return r.draw_text(le); }, py::arg("line_ending") = "\n")
        .def("set_world_camera_override", [](HipRtRenderer &r, py::object inv, float exposure) {
            if (inv.is_none()) { r.set_world_camera_override(nullptr, 1.0f); return; }
            const auto m = inv.cast<std::array<double, 16>>();
            r.set_world_camera_override(m.data(), exposure);
        }, py::arg("inverse_projection_view"), py::arg("exposure") = 1.0f)
        .def("draw_rgba", [](HipRtRenderer &r, const std::string &t) { py::gil_scoped_release rel; return r.draw_rgba(t); }, py::arg("info_text") = "")
        .def("draw_rows_to_device", [](HipRtRenderer &r, uintptr_t ptr, uint32_t strip_rows, uint32_t n_parts, uint32_t part, bool counters, bool no_feedback) {
            py::gil_scoped_release rel;
            return r.draw_rows_to_device(reinterpret_cast<void *>(ptr), strip_rows, n_parts, part, counters, no_feedback);
        }, py::arg("device_ptr"), py::arg("strip_rows"), py::arg("n_parts"), py::arg("part"), py::arg("counters") = false, py::arg("no_feedback") = false)
        .def("partition_rows", &HipRtRenderer::partition_rows)
        .def("submit_rows_to_device", [](HipRtRenderer &r, uintptr_t ptr, uint32_t strip_rows, uint32_t n_parts, uint32_t part, uint32_t slot) {
            r.submit_rows_to_device(reinterpret_cast<void *>(ptr), strip_rows, n_parts, part, slot);
        }, py::arg("device_ptr"), py::arg("strip_rows"), py::arg("n_parts"), py::arg("part"), py::arg("slot"))
        .def("submit_rows_batch_to_device", [](HipRtRenderer &r, const std::vector<uintptr_t> &ptrs, uint32_t strip_rows, uint32_t n_parts, uint32_t part, uint32_t slot,
                                               const std::vector<std::array<double, 16>> &views) {
            std::vector<void *> outs;
            for (uintptr_t p : ptrs) outs.push_back(reinterpret_cast<void *>(p));
            r.submit_rows_batch_to_device(outs, strip_rows, n_parts, part, slot, views);
        }, py::arg("device_ptrs"), py::arg("strip_rows"), py::arg("n_parts"), py::arg("part"), py::arg("slot"), py::arg("inverse_projection_views") = std::vector<std::array<double, 16>>())
        .def("wait_rows", [](HipRtRenderer &r, uint32_t slot) { py::gil_scoped_release rel; return r.wait_rows(slot); }, py::arg("slot"))
        .def("synchronize", [](HipRtRenderer &r) { py::gil_scoped_release rel; r.synchronize(); })
        .def("assemble_strips", [](HipRtRenderer &r, uintptr_t gathered, uintptr_t out, uint32_t strip_rows, uint32_t n_parts, bool wait) {
            r.assemble_strips(reinterpret_cast<const void *>(gathered), reinterpret_cast<void *>(out), strip_rows, n_parts, wait);
        }, py::arg("gathered"), py::arg("out"), py::arg("strip_rows"), py::arg("n_parts"), py::arg("wait") = true)
        .def("modified_viewport", &HipRtRenderer::modified_viewport)
        .def("device_name", &HipRtRenderer::device_name)
        .def("stream", [](const HipRtRenderer &r) { return reinterpret_cast<uintptr_t>(r.stream()); })
        .def("wait_event", [](HipRtRenderer &r, uintptr_t ev) { r.wait_event(reinterpret_cast<void *>(ev)); })
        .def("stream_wait_rows", [](HipRtRenderer &r, uint32_t slot, uintptr_t stream) { r.stream_wait_rows(slot, reinterpret_cast<void *>(stream)); },
             py::arg("slot"), py::arg("hip_stream"))
        .def("evaluate_light", [](HipRtRenderer &r, int maximum_distance, bool fast, int epsilon, int batch, int queue_order, int lanes_per_cube, bool continue_queue,
                                  uint64_t max_updates) {
            const HipRtRenderer::LightUpdateInfo i = r.evaluate_light(maximum_distance, fast, epsilon, batch, queue_order, lanes_per_cube, continue_queue, max_updates);
            py::dict d;
            d["updates"] = i.updates; d["batches"] = i.batches; d["cost"] = i.cost; d["device_ms"] = i.device_ms;
            d["total_ms"] = i.total_ms; d["queue_left"] = i.queue_left; d["bundles_visited"] = i.bundles_visited;
            return d;
        }, py::arg("maximum_distance"), py::arg("fast") = true, py::arg("epsilon") = 1, py::arg("batch") = 32, py::arg("queue_order") = 16, py::arg("lanes_per_cube") = 0, py::arg("continue_queue") = false,
           py::arg("max_updates") = 0)
        .def("evaluate_light_submit", [](HipRtRenderer &r, int maximum_distance, bool fast, int epsilon, int batch, int queue_order, int lanes_per_cube, bool continue_queue,
                                         uint64_t max_updates) { r.evaluate_light_submit(maximum_distance, fast, epsilon, batch, queue_order, lanes_per_cube, continue_queue, max_updates); },
             py::arg("maximum_distance"), py::arg("fast") = true, py::arg("epsilon") = 1, py::arg("batch") = 32, py::arg("queue_order") = 16, py::arg("lanes_per_cube") = 0,
             py::arg("continue_queue") = false, py::arg("max_updates") = 0)
        .def("evaluate_light_done", &HipRtRenderer::evaluate_light_done)
        .def("evaluate_light_wait", [](HipRtRenderer &r) {
            py::gil_scoped_release rel;
            const HipRtRenderer::LightUpdateInfo i = r.evaluate_light_wait();
            py::gil_scoped_acquire acq;
            py::dict d;
            d["updates"] = i.updates; d["batches"] = i.batches; d["cost"] = i.cost; d["device_ms"] = i.device_ms;
            d["total_ms"] = i.total_ms; d["queue_left"] = i.queue_left; d["bundles_visited"] = i.bundles_visited;
            return d;
        })
        .def_readwrite("device_light", &HipRtRenderer::device_light)
        .def_readwrite("device_light_queue_order", &HipRtRenderer::device_light_queue_order)
        .def_readwrite("enable_counters", &HipRtRenderer::enable_counters);
}

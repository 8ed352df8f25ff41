// host/host_api.cpp - C exports of include/mlrt_host.h: worlds registry, Camera::new, image height, P3 writer.
#include <cmath>
#include <cstdio>
#include <string>

#include "host_internal.hpp"

namespace mlrt {
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }
}  // namespace mlrt

extern "C" {

int32_t mlrt_world_count(void) { return (int32_t)mlrt::world_registry().size(); }
const char* mlrt_world_name(int32_t i) {
    const auto& names = mlrt::world_registry();
    if (i < 0 || (size_t)i >= names.size()) return nullptr;
    return names[(size_t)i].c_str();
}

mlrt_world* mlrt_world_build(const char* name, uint64_t seed, const uint8_t* earth_rgb, int32_t earth_w, int32_t earth_h) {
    if (!name) { mlrt::set_last_error("mlrt_world_build: name is NULL"); return nullptr; }
    std::string err;
    auto store = mlrt::build_world_store(name, seed, earth_rgb, earth_w, earth_h, err);
    if (!store) { mlrt::set_last_error(err); return nullptr; }
    mlrt_world* w = new mlrt_world();
    w->store = std::move(store);
    return w;
}
void mlrt_world_free(mlrt_world* w) { delete w; }
const mlrt_scene* mlrt_world_scene(const mlrt_world* w) { return w ? &w->store->scene : nullptr; }
void mlrt_world_camera(const mlrt_world* w, double lookfrom[3], double lookat[3], double* field_of_view) {
    for (int k = 0; k < 3; k++) { lookfrom[k] = w->store->lookfrom[k]; lookat[k] = w->store->lookat[k]; }
    *field_of_view = w->store->field_of_view;
}
void mlrt_world_background(const mlrt_world* w, mlrt_params* params) {
    params->background_kind = w->store->background_kind;
    // GradientBackground::default() = new(top = blueish, bottom = white), raytrace.rs:21-26
    const double top[3] = {0.5, 0.7, 1.0}, bottom[3] = {1.0, 1.0, 1.0};
    for (int k = 0; k < 3; k++) { params->background_top[k] = top[k]; params->background_bottom[k] = bottom[k]; }
}
void mlrt_world_stream_probes(const mlrt_world* w, uint64_t out[2]) {
    out[0] = w->store->probe_before_last_bvh;
    out[1] = w->store->probe_after_build;
}

// Camera::new, camera.rs:15-38 (operation order kept)
void mlrt_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov, double aspect_ratio,
                     double aperture, double focus_dist, mlrt_camera* out) {
    const double pi = 3.14159265358979323846264338327950288;
    const double theta = vfov * pi / 180.0;
    const double h = std::tan(theta / 2.0);
    const double viewport_height = 2.0 * h;
    const double viewport_width = aspect_ratio * viewport_height;
    auto unit = [](const double a[3], double r[3]) {
        const double len = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        for (int k = 0; k < 3; k++) r[k] = a[k] / len;
    };
    auto cross = [](const double a[3], const double b[3], double r[3]) {
        r[0] = a[1] * b[2] - a[2] * b[1];
        r[1] = a[2] * b[0] - a[0] * b[2];
        r[2] = a[0] * b[1] - a[1] * b[0];
    };
    double diff[3], w[3], c[3], u[3], v[3];
    for (int k = 0; k < 3; k++) diff[k] = lookfrom[k] - lookat[k];
    unit(diff, w);
    cross(vup, w, c);
    unit(c, u);
    cross(w, u, v);
    const double hs = focus_dist * viewport_width, vs = focus_dist * viewport_height;
    for (int k = 0; k < 3; k++) {
        out->origin[k] = lookfrom[k];
        out->horizontal[k] = u[k] * hs;
        out->vertical[k] = v[k] * vs;
        out->u[k] = u[k];
        out->v[k] = v[k];
        out->lower_left_corner[k] = lookfrom[k] - out->horizontal[k] / 2.0 - out->vertical[k] / 2.0 - w[k] * focus_dist;
    }
    out->lens_radius = aperture / 2.0;
}

int32_t mlrt_image_height(int32_t image_width, double aspect_ratio) { return (int32_t)((double)image_width / aspect_ratio); }  // main.rs:121

// main.rs:144,175-179.  The reference prints one `println!` per pixel (11 MB of text for 1200x800, slower than the GPU
// render); here every row is formatted into one buffer with a 0..255 lookup table and written at once.
static void format_rows(std::string& buf, const int32_t* rgb, int32_t width, int32_t row) {
    static char lut[256][4];
    static int lut_len[256];
    static bool ready = false;
    if (!ready) {
        for (int v = 0; v < 256; v++) lut_len[v] = std::snprintf(lut[v], sizeof lut[v], "%d", v);
        ready = true;
    }
    const int32_t* p = &rgb[(size_t)row * width * 3];
    for (int32_t i = 0; i < width; i++, p += 3) {
        for (int c = 0; c < 3; c++) {
            const int32_t v = p[c];
            if (v >= 0 && v < 256) buf.append(lut[v], (size_t)lut_len[v]);
            else buf.append(std::to_string(v));  // not produced by to_rgb, but the format is "{}" of an i32
            buf.push_back(c == 2 ? '\n' : ' ');
        }
    }
}
int32_t mlrt_write_p3(FILE* f, const int32_t* rgb, int32_t width, int32_t height) {
    if (!f || !rgb) { mlrt::set_last_error("mlrt_write_p3: NULL argument"); return MLRT_ERR_INVALID; }
    std::fprintf(f, "P3\n%d %d\n255\n", width, height);
    std::string buf;
    buf.reserve((size_t)width * 12 * 16 + 64);
    for (int32_t j = height - 1; j >= 0; j--) {
        format_rows(buf, rgb, width, j);
        if (buf.size() > (1u << 20) || j == 0) {
            if (std::fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { mlrt::set_last_error("mlrt_write_p3: write failed"); return MLRT_ERR_INVALID; }
            buf.clear();
        }
    }
    return MLRT_OK;
}
int32_t mlrt_write_p3_path(const char* path, const int32_t* rgb, int32_t width, int32_t height) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { mlrt::set_last_error(std::string("cannot open ") + path); return MLRT_ERR_INVALID; }
    const int32_t rc = mlrt_write_p3(f, rgb, width, height);
    std::fclose(f);
    return rc;
}

void mlrt_pcg64_stream(uint64_t seed, uint64_t site_id, int32_t n, uint64_t* out) {
    mlrt::Pcg64 g = mlrt::Pcg64::seed_from_u64(seed + site_id);
    for (int32_t i = 0; i < n; i++) out[i] = g.next_u64();
}

}  // extern "C"

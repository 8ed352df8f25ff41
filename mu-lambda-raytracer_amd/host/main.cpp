// host/main.cpp - the `mlrt` command line: a drop-in for the reference binary's CLI on the seeded path.
//
// Mirrors /root/reference/src/main.rs: args() :64-132 (same flag names and defaults), do_it() :181-207 (world from
// rng(0), camera), do_tracing() :134-180 (P3 header, render, "Rendered in X s" on stderr, rows top to bottom).
// The render itself is mlrt_context_render (include/mlrt.h) on the MI355X; there is no CPU path here.
// Flags beyond the reference: --precision=fp32|fp64, --device=N, --earthmap=<P6 file> (decoded earthmap.jpg).
// Not supported: running without --seed and --randomized_rendering (ThreadRng, non-deterministic).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mlrt.h"
#include "../../include/mlrt_host.h"

namespace {

bool read_p6(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    const bool header_ok = std::fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) == 4 && std::strcmp(magic, "P6") == 0 && maxv == 255;
    bool ok = false;
    if (header_ok) {
        std::fgetc(f);
        rgb.resize((size_t)w * h * 3);
        ok = std::fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
    }
    std::fclose(f);
    return ok;
}

bool parse_vector(const std::string& s, double out[3]) {  // main.rs:54-62
    return std::sscanf(s.c_str(), "%lf,%lf,%lf", &out[0], &out[1], &out[2]) == 3;
}

struct Progress {
    std::chrono::steady_clock::time_point start;
    long last_logged_ms = 0;
};
void on_progress(int32_t done, int32_t total, void* user) {  // main.rs:157-173
    Progress* p = static_cast<Progress*>(user);
    const int32_t remaining = total - done;
    if (remaining == 0) { std::fprintf(stderr, "\r%-50s", "Done!"); return; }
    const long elapsed = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - p->start).count();
    if (elapsed - p->last_logged_ms > 300) {
        p->last_logged_ms = elapsed;
        std::fprintf(stderr, "\rRemaining: %3d%%  ", (int)((long)remaining * 100 / total));
    }
}

int fail(const char* what) {
    std::fprintf(stderr, "mlrt: %s: %s\n", what, mlrt_last_error());
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    std::map<std::string, std::string> opt = {{"aspect_ratio", "16:9"}, {"image_width", "400"}, {"samples_per_pixel", "200"}, {"max_depth", "50"},
                                              {"up", "0,1.0,0"},        {"aperture", "0.0"},    {"world", "simple"},          {"precision", "fp32"},
                                              {"device", "0"},          {"earthmap", "earthmap.ppm"}};
    for (int i = 1; i < argc; i++) {
        std::string a(argv[i]);
        if (a == "--randomized_rendering" || a == "-r") { std::fprintf(stderr, "mlrt: --randomized_rendering (ThreadRng) is not supported by the device path\n"); return 2; }
        if (a.rfind("--", 0) != 0) { std::fprintf(stderr, "mlrt: unexpected argument '%s'\n", a.c_str()); return 2; }
        a = a.substr(2);
        const size_t eq = a.find('=');
        if (eq != std::string::npos) opt[a.substr(0, eq)] = a.substr(eq + 1);
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { std::fprintf(stderr, "mlrt: missing value for --%s\n", a.c_str()); return 2; }
    }
    if (!opt.count("seed")) { std::fprintf(stderr, "mlrt: --seed is required (the unseeded ThreadRng path of the reference is non-deterministic and not accelerated)\n"); return 2; }
    const uint64_t seed = std::strtoull(opt["seed"].c_str(), nullptr, 10);

    bool known = false;
    for (int32_t i = 0; i < mlrt_world_count(); i++) known = known || opt["world"] == mlrt_world_name(i);
    if (!known) {  // clap possible_values, main.rs:80-84
        std::fprintf(stderr, "error: '%s' isn't a valid value for '--world <world>'\n\t[possible values:", opt["world"].c_str());
        for (int32_t i = 0; i < mlrt_world_count(); i++) std::fprintf(stderr, " %s", mlrt_world_name(i));
        std::fprintf(stderr, "]\n");
        return 2;
    }
    int aw = 0, ah = 0;  // parse_aspect_ratio, main.rs:49-52
    if (std::sscanf(opt["aspect_ratio"].c_str(), "%d:%d", &aw, &ah) != 2 || ah == 0) { std::fprintf(stderr, "mlrt: bad --aspect_ratio\n"); return 2; }
    const double aspect_ratio = (double)aw / (double)ah;

    mlrt_params params;
    std::memset(&params, 0, sizeof params);
    params.image_width = std::atoi(opt["image_width"].c_str());
    params.image_height = mlrt_image_height(params.image_width, aspect_ratio);
    params.samples_per_pixel = std::atoi(opt["samples_per_pixel"].c_str());
    params.max_depth = std::atoi(opt["max_depth"].c_str());
    params.seed = seed;

    std::vector<uint8_t> earth;
    int ew = 0, eh = 0;
    if (opt["world"] == "earth" || opt["world"] == "final_scene") {
        if (!read_p6(opt["earthmap"], earth, ew, eh)) {
            std::fprintf(stderr, "mlrt: cannot read the decoded earth map (binary P6) at '%s'; see assets/earthmap.ppm\n", opt["earthmap"].c_str());
            return 2;
        }
    }
    // World (main.rs:185-189): built from rng(0)
    mlrt_world* world = mlrt_world_build(opt["world"].c_str(), seed, earth.empty() ? nullptr : earth.data(), ew, eh);
    if (!world) return fail("world build");
    mlrt_world_background(world, &params);

    double lookfrom[3], lookat[3], up[3], fov;  // main.rs:104-112
    mlrt_world_camera(world, lookfrom, lookat, &fov);
    if (opt.count("lookfrom") && !parse_vector(opt["lookfrom"], lookfrom)) { std::fprintf(stderr, "mlrt: bad --lookfrom\n"); return 2; }
    if (opt.count("lookat") && !parse_vector(opt["lookat"], lookat)) { std::fprintf(stderr, "mlrt: bad --lookat\n"); return 2; }
    if (!parse_vector(opt["up"], up)) { std::fprintf(stderr, "mlrt: bad --up\n"); return 2; }
    if (opt.count("field_of_view")) fov = std::atof(opt["field_of_view"].c_str());
    double focus_dist;
    if (opt.count("focus_dist")) focus_dist = std::atof(opt["focus_dist"].c_str());
    else {
        const double dx = lookat[0] - lookfrom[0], dy = lookat[1] - lookfrom[1], dz = lookat[2] - lookfrom[2];
        focus_dist = std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    mlrt_camera camera;
    mlrt_camera_new(lookfrom, lookat, up, fov, aspect_ratio, std::atof(opt["aperture"].c_str()), focus_dist, &camera);

    const int32_t precision = opt["precision"] == "fp64" ? MLRT_PRECISION_FP64 : MLRT_PRECISION_FP32;
    std::vector<int32_t> rgb((size_t)params.image_width * params.image_height * 3);

    // do_tracing, main.rs:134-180.  (The reference prints the header before rendering; mlrt_write_p3 emits header and
    // pixels together after it - same bytes on stdout.)
    Progress progress;
    progress.start = std::chrono::steady_clock::now();
    mlrt_context* ctx = nullptr;
    if (mlrt_context_create(mlrt_world_scene(world), std::atoi(opt["device"].c_str()), precision, &ctx) != MLRT_OK) return fail("context");
    if (mlrt_context_render(ctx, &camera, &params, nullptr, 0, rgb.data(), on_progress, &progress) != MLRT_OK) return fail("render");
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - progress.start).count();
    std::fprintf(stderr, "\nRendered in %.3fs\n", secs);
    if (mlrt_write_p3(stdout, rgb.data(), params.image_width, params.image_height) != MLRT_OK) return fail("write");
    mlrt_context_destroy(ctx);
    mlrt_world_free(world);
    return 0;
}

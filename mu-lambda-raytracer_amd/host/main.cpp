// host/main.cpp - the `mlrt` command line: a drop-in for the reference binary's CLI on the seeded path.
//
// Mirrors /root/reference/src/main.rs: args() :64-132 (same flag names and defaults), do_it() :181-207 (world from
// rng(0), camera), do_tracing() :134-180 (P3 header, render, "Rendered in X s" on stderr, rows top to bottom).
// The render itself is mlrt_context_render (include/mlrt.h) on the MI355X; there is no CPU path here.
// Flags beyond the reference: --precision=fp64|fp32 (default fp64: the reference's arithmetic, /root/reference/src/vec.rs:5-8;
// fp32 is the fast mixed-precision mode), --device=N, --gpus=N (the frame's rows over devices 0..N-1 of this node, one
// gather: mlrt_multi_*) or --devices=a,b,c, --format=p3|p6|png (p6 / png: the same image as binary PPM / PNG), --earthmap=<earthmap.jpg or a
// binary P6> (default: earthmap.jpg in the working directory, as the reference; then earthmap.ppm).
// Progress ("Remaining: NN%", main.rs:157-173) is shown only when stderr is a terminal: it splits the frame into a few
// synchronised launches, which a timed or piped run should not pay for.
// Without --seed / with --randomized_rendering (the reference's ThreadRng runs, different every time by design) the seeds come
// from the OS's entropy source and are printed.
//
// Frame-sequence mode (replaces /root/reference/movie.py, which re-launches the binary - world rebuild and all - once
// per frame): --movie_frames=N [--movie_step_degrees=0.5] [--movie_origin=x,y,z] [--movie_prefix=_movie/frame] renders N
// frames with the camera orbiting `movie_origin` exactly as movie.py:56-69 moves it (lookat = origin, lookfrom = origin +
// the start offset with its polar angle advanced by step degrees per frame) and writes <prefix>NNNN.ppm.  The world,
// the packed scene and the device context stay resident; only the camera changes.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <thread>
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

// PNG (8-bit RGB, filter 0 on every row, deflate in STORED blocks: no compression library in this image and none needed - the
// file is for viewing, 3 bytes per pixel like the P6).  rgb: [H][W][3], row 0 = bottom.
int32_t write_png(FILE* f, const int32_t* rgb, int32_t width, int32_t height) {
    static uint32_t crc_table[256];
    if (!crc_table[1])
        for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
    auto be32 = [](std::vector<uint8_t>& v, uint32_t x) { for (int k = 3; k >= 0; k--) v.push_back((uint8_t)(x >> (8 * k))); };
    bool ok = true;
    auto chunk = [&](const char type[4], const std::vector<uint8_t>& data) {
        std::vector<uint8_t> out;
        be32(out, (uint32_t)data.size());
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        uint32_t c = 0xFFFFFFFFu;
        for (size_t i = 4; i < out.size(); i++) c = crc_table[(c ^ out[i]) & 0xFFu] ^ (c >> 8);
        be32(out, c ^ 0xFFFFFFFFu);
        ok = ok && std::fwrite(out.data(), 1, out.size(), f) == out.size();
    };
    static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    ok = std::fwrite(signature, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk("IHDR", ihdr);
    std::vector<uint8_t> raw;  // filter byte + row, top row first
    raw.reserve((size_t)height * ((size_t)width * 3 + 1));
    for (int32_t j = height - 1; j >= 0; j--) {
        raw.push_back(0);
        for (size_t e = 0; e < (size_t)width * 3; e++) raw.push_back((uint8_t)rgb[(size_t)j * width * 3 + e]);
    }
    std::vector<uint8_t> z = {0x78, 0x01};  // zlib header: deflate, 32 KiB window, no preset dictionary
    uint32_t a = 1, b = 0;                  // Adler-32 of the raw bytes
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    }
    be32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});
    return ok ? MLRT_OK : MLRT_ERR_INVALID;
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
                                              {"up", "0,1.0,0"},        {"aperture", "0.0"},    {"world", "simple"},          {"precision", "fp64"},
                                              {"device", "0"},          {"format", "p3"},       {"tracer", "recursive"},
                                              {"light_source", "14.0,3.0,3.0"}, {"light_intensity", "1.0"}};
    bool randomized_rendering = false;
    for (int i = 1; i < argc; i++) {
        std::string a(argv[i]);
        if (a == "--randomized_rendering" || a == "-r" || a == "-rr") { randomized_rendering = true; continue; }  // clap short("rr"): -r
        if (a.rfind("--", 0) != 0) { std::fprintf(stderr, "mlrt: unexpected argument '%s'\n", a.c_str()); return 2; }
        a = a.substr(2);
        const size_t eq = a.find('=');
        if (eq != std::string::npos) opt[a.substr(0, eq)] = a.substr(eq + 1);
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { std::fprintf(stderr, "mlrt: missing value for --%s\n", a.c_str()); return 2; }
    }
    // main.rs:202-216: without --seed the reference draws the world AND the samples from ThreadRng, with --randomized_rendering only the
    // samples (the world still follows --seed): runs that differ from each other by design.  Here: a seed from the OS's entropy
    // source in the same places, printed so the run can be repeated with --seed.
    auto entropy_seed = [] { std::random_device rd; return ((uint64_t)rd() << 32) ^ (uint64_t)rd(); };
    const uint64_t seed = opt.count("seed") ? std::strtoull(opt["seed"].c_str(), nullptr, 10) : entropy_seed();
    const uint64_t render_seed = (randomized_rendering && opt.count("seed")) ? entropy_seed() : seed;
    if (!opt.count("seed")) std::fprintf(stderr, "mlrt: no --seed: using --seed=%llu\n", (unsigned long long)seed);
    else if (randomized_rendering) std::fprintf(stderr, "mlrt: --randomized_rendering: world from --seed=%llu, samples from seed %llu\n", (unsigned long long)seed, (unsigned long long)render_seed);

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
    params.seed = render_seed;
    // The Renderer's RayTracer: main.rs:150-153 constructs RecursiveRayTracer{max_depth} and keeps SingleLightSourceRayTracer{(14, 3, 3), 1.0}
    // in a comment - a recompile there, a flag here (its light defaults to that comment's).
    if (opt["tracer"] != "recursive" && opt["tracer"] != "single_light") { std::fprintf(stderr, "mlrt: --tracer must be recursive or single_light\n"); return 2; }
    mlrt_options options;
    mlrt_options_from_env(&options);
    if (opt["tracer"] == "single_light") {
        params.tracer = MLRT_TRACER_SINGLE_LIGHT;
        options.single_light_tracer = 1;
        if (!parse_vector(opt["light_source"], params.light_source)) { std::fprintf(stderr, "mlrt: bad --light_source\n"); return 2; }
        params.light_intensity = std::atof(opt["light_intensity"].c_str());
    }

    std::vector<uint8_t> earth;
    int ew = 0, eh = 0;
    if (opt["world"] == "earth" || opt["world"] == "final_scene") {
        // worlds.rs:180,441: image::open("earthmap.jpg") relative to the working directory.  --earthmap overrides the
        // path; a .ppm (binary P6, e.g. assets/earthmap.ppm) is accepted as well.
        const bool given = opt.count("earthmap") != 0;
        const std::string path = given ? opt["earthmap"] : "earthmap.jpg";
        const bool is_ppm = path.size() > 4 && path.compare(path.size() - 4, 4, ".ppm") == 0;
        bool ok = false;
        if (is_ppm) ok = read_p6(path, earth, ew, eh);
        else {
            uint8_t* px = nullptr;
            int32_t w32 = 0, h32 = 0;
            if (mlrt_decode_jpeg_file(path.c_str(), &px, &w32, &h32) == MLRT_OK) {
                earth.assign(px, px + (size_t)w32 * h32 * 3);
                mlrt_free(px);
                ew = w32; eh = h32; ok = true;
            } else if (!given && read_p6("earthmap.ppm", earth, ew, eh)) ok = true;
        }
        if (!ok) {
            std::fprintf(stderr, "mlrt: cannot read the earth map at '%s' (%s); pass --earthmap=<earthmap.jpg | assets/earthmap.ppm>\n", path.c_str(), mlrt_last_error());
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

    if (opt["precision"] != "fp64" && opt["precision"] != "fp32") { std::fprintf(stderr, "mlrt: --precision must be fp64 or fp32\n"); return 2; }
    if (opt["format"] != "p3" && opt["format"] != "p6" && opt["format"] != "png") { std::fprintf(stderr, "mlrt: --format must be p3, p6 or png\n"); return 2; }
    const int32_t precision = opt["precision"] == "fp64" ? MLRT_PRECISION_FP64 : MLRT_PRECISION_FP32;
    std::vector<int32_t> rgb((size_t)params.image_width * params.image_height * 3);
    std::vector<int32_t> devices;  // --gpus=N | --devices=a,b,c: the multi-device path
    if (opt.count("devices")) {
        for (const char* q = opt["devices"].c_str(); *q;) {
            devices.push_back(std::atoi(q));
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
    } else if (opt.count("gpus")) {
        const int n = std::atoi(opt["gpus"].c_str());
        if (n < 1 || n > mlrt_device_count()) { std::fprintf(stderr, "mlrt: --gpus=%d but %d device(s) are visible\n", n, (int)mlrt_device_count()); return 2; }
        for (int k = 0; k < n; k++) devices.push_back(k);
    }
    // The process's first GPU operation (HIP runtime, queues: ~0.2 s whatever it is) is not part of "renderer construction + render":
    // the reference's timer (main.rs:145) starts in a process that is already up.  Reported on its own line.
    {
        const auto t_dev = std::chrono::steady_clock::now();
        if (devices.empty()) { if (mlrt_device_warm_up(std::atoi(opt["device"].c_str())) != MLRT_OK) return fail("device"); }
        else for (int32_t dv : devices) if (mlrt_device_warm_up(dv) != MLRT_OK) return fail("device");
        std::fprintf(stderr, "Device ready in %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dev).count());
    }
    auto write_image = [&](FILE* f) -> int32_t {
        if (opt["format"] == "p3") return mlrt_write_p3(f, rgb.data(), params.image_width, params.image_height);
        if (opt["format"] == "png") return write_png(f, rgb.data(), params.image_width, params.image_height);
        // binary PPM of the same image, top row first (the reference prints P3 only, main.rs:144,175-179)
        std::fprintf(f, "P6\n%d %d\n255\n", params.image_width, params.image_height);
        std::vector<uint8_t> line((size_t)params.image_width * 3);
        for (int32_t j = params.image_height - 1; j >= 0; j--) {
            for (size_t e = 0; e < line.size(); e++) line[e] = (uint8_t)rgb[(size_t)j * line.size() + e];
            if (std::fwrite(line.data(), 1, line.size(), f) != line.size()) return MLRT_ERR_INVALID;
        }
        return MLRT_OK;
    };

    if (opt.count("movie_frames")) {
        const int frames = std::atoi(opt["movie_frames"].c_str());
        const double step_grad = opt.count("movie_step_degrees") ? std::atof(opt["movie_step_degrees"].c_str()) : 0.5;  // movie.py:10
        double origin[3] = {278.0, 278.0, 400.0};                                                                       // movie.py:56
        if (opt.count("movie_origin") && !parse_vector(opt["movie_origin"], origin)) { std::fprintf(stderr, "mlrt: bad --movie_origin\n"); return 2; }
        const std::string prefix = opt.count("movie_prefix") ? opt["movie_prefix"] : "_movie/frame";
        // movie.py:57-60: delta = start - origin in polar form (r, theta, phi); `start` is the camera position
        const double dx = lookfrom[0] - origin[0], dy = lookfrom[1] - origin[1], dz = lookfrom[2] - origin[2];
        const double r = std::sqrt(dx * dx + dy * dy + dz * dz);
        double theta = std::atan2(std::sqrt(dx * dx + dy * dy), dz);
        const double phi = std::atan2(dy, dx);
        const double grad = 3.141592653589793 / 180.0;  // movie.py:62 (math.pi)
        mlrt_context* mctx = nullptr;
        if (mlrt_context_create_opts(mlrt_world_scene(world), std::atoi(opt["device"].c_str()), precision, &options, &mctx) != MLRT_OK) return fail("context");
        const auto t0 = std::chrono::steady_clock::now();
        // A frame's P3 text (10 MB for 1200x800: ~30 ms) is written by a second thread while the next frame renders: two image buffers,
        // at most one write in flight.
        std::vector<int32_t> rgb_b(rgb.size());
        std::thread writer;
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_on_every_way_out{writer};
        int32_t write_rc = MLRT_OK;
        for (int i = 0; i < frames; i++) {
            std::vector<int32_t>& frame = (i & 1) ? rgb_b : rgb;
            // movie.py:41-46 polar_to_cart, :64-69
            const double from[3] = {origin[0] + r * std::cos(phi) * std::sin(theta), origin[1] + r * std::sin(phi) * std::sin(theta), origin[2] + r * std::cos(theta)};
            double fd = focus_dist;
            if (!opt.count("focus_dist")) { const double ex = origin[0] - from[0], ey = origin[1] - from[1], ez = origin[2] - from[2]; fd = std::sqrt(ex * ex + ey * ey + ez * ez); }
            mlrt_camera cam_i;
            mlrt_camera_new(from, origin, up, fov, aspect_ratio, std::atof(opt["aperture"].c_str()), fd, &cam_i);
            const int32_t render_rc = mlrt_context_render(mctx, &cam_i, &params, nullptr, 0, frame.data(), nullptr, nullptr);
            if (writer.joinable()) writer.join();  // (the previous frame's file; its buffer is the one the NEXT frame renders into)
            if (render_rc != MLRT_OK) return fail("render");
            if (write_rc != MLRT_OK) return fail("write");
            char name[512];
            std::snprintf(name, sizeof name, "%s%04d.ppm", prefix.c_str(), i);
            writer = std::thread([&write_rc, &frame, &params, path = std::string(name)] { write_rc = mlrt_write_p3_path(path.c_str(), frame.data(), params.image_width, params.image_height); });
            std::fprintf(stderr, "Frame %d: (%.17g, %.17g, %.17g)\n", i, from[0] - origin[0], from[1] - origin[1], from[2] - origin[2]);
            theta += step_grad * grad;
        }
        if (writer.joinable()) writer.join();
        if (write_rc != MLRT_OK) return fail("write");
        std::fprintf(stderr, "Rendered %d frames in %.3fs\n", frames, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        mlrt_context_destroy(mctx);
        mlrt_world_free(world);
        return 0;
    }

    // do_tracing, main.rs:134-180.  (The reference prints the header before rendering; mlrt_write_p3 emits header and
    // pixels together after it - same bytes on stdout.)
    Progress progress;
    progress.start = std::chrono::steady_clock::now();
    const bool show_progress = isatty(2) != 0;
    mlrt_context* ctx = nullptr;
    mlrt_multi* multi = nullptr;
    if (!devices.empty()) {
        if (mlrt_multi_create(mlrt_world_scene(world), devices.data(), (int32_t)devices.size(), precision, &options, MLRT_GATHER_AUTO, &multi) != MLRT_OK) return fail("multi-device context");
        if (mlrt_multi_render(multi, &camera, &params, rgb.data(), show_progress ? on_progress : nullptr, &progress) != MLRT_OK) return fail("render");
    } else {
        if (mlrt_context_create_opts(mlrt_world_scene(world), std::atoi(opt["device"].c_str()), precision, &options, &ctx) != MLRT_OK) return fail("context");
        if (mlrt_context_render(ctx, &camera, &params, nullptr, 0, rgb.data(), show_progress ? on_progress : nullptr, &progress) != MLRT_OK) return fail("render");
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - progress.start).count();
    std::fprintf(stderr, "%sRendered in %.3fs\n", show_progress ? "\n" : "", secs);
    if (write_image(stdout) != MLRT_OK) return fail("write");
    if (ctx) mlrt_context_destroy(ctx);
    if (multi) mlrt_multi_destroy(multi);
    mlrt_world_free(world);
    return 0;
}

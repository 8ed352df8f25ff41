// oracle/oracle_main.cpp - CLI of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
// Same flags, defaults and P3 output as the reference binary (main.rs:64-132, :144, :175-179), so BASELINE
// config 1 (--world=random --seed=42 --aspect_ratio=3:2 --image_width=400 --samples_per_pixel=8) can be diffed.
// Extra flags (not in the reference): --rng=pcg|philox, --threads=N, --earthmap=<P6 file>.
#include <chrono>
#include <map>
#include <thread>

#include "oracle_core.hpp"

extern "C" {
void* orc_world_new(const char* name, uint64_t seed, const uint8_t* earth_rgb, int earth_w, int earth_h);
void orc_world_free(void* h);
void orc_world_camera(void* h, double out[7]);
void orc_camera_new(const double from[3], const double at[3], const double up[3], double vfov, double aspect, double aperture,
                    double focus_dist, double out[19]);
double orc_render(void* h, const double camera19[19], int width, int height, int spp, int max_depth, uint64_t seed, int rng_mode,
                  const int* rows, int n_rows, int threads, int32_t* rgb, double* accum, uint64_t* stats_out);
}

static bool read_p6(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int maxv = 0;
    if (std::fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) != 4 || std::string(magic) != "P6" || maxv != 255) { std::fclose(f); return false; }
    std::fgetc(f);
    rgb.resize((size_t)w * h * 3);
    bool ok = std::fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
    std::fclose(f);
    return ok;
}

static void parse_vec(const std::string& s, double out[3]) {  // main.rs:54-62
    size_t a = s.find(','), b = s.find(',', a + 1);
    out[0] = std::atof(s.substr(0, a).c_str());
    out[1] = std::atof(s.substr(a + 1, b - a - 1).c_str());
    out[2] = std::atof(s.substr(b + 1).c_str());
}

int main(int argc, char** argv) {
    std::map<std::string, std::string> opt = {{"aspect_ratio", "16:9"}, {"image_width", "400"}, {"samples_per_pixel", "200"},
                                              {"max_depth", "50"},     {"up", "0,1.0,0"},      {"aperture", "0.0"},
                                              {"world", "simple"},     {"rng", "pcg"},         {"threads", "0"},
                                              {"earthmap", "earthmap.ppm"}};
    for (int i = 1; i < argc; i++) {
        std::string a(argv[i]);
        if (a.rfind("--", 0) != 0) { std::fprintf(stderr, "unexpected argument %s\n", a.c_str()); return 2; }
        a = a.substr(2);
        size_t eq = a.find('=');
        if (eq != std::string::npos) opt[a.substr(0, eq)] = a.substr(eq + 1);
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { std::fprintf(stderr, "missing value for --%s\n", a.c_str()); return 2; }
    }
    if (!opt.count("seed")) { std::fprintf(stderr, "the oracle needs --seed (the unseeded ThreadRng path is non-deterministic and out of scope)\n"); return 2; }
    uint64_t seed = std::strtoull(opt["seed"].c_str(), nullptr, 10);
    size_t colon = opt["aspect_ratio"].find(':');  // main.rs:49-52
    double aspect = (double)std::atoi(opt["aspect_ratio"].substr(0, colon).c_str()) / (double)std::atoi(opt["aspect_ratio"].substr(colon + 1).c_str());
    int width = std::atoi(opt["image_width"].c_str());
    int height = (int)((double)width / aspect);  // main.rs:121
    int spp = std::atoi(opt["samples_per_pixel"].c_str());
    int max_depth = std::atoi(opt["max_depth"].c_str());

    std::vector<uint8_t> earth;
    int ew = 0, eh = 0;
    const std::string wn = opt["world"];
    if (wn == "earth" || wn == "final_scene") {
        if (!read_p6(opt["earthmap"], earth, ew, eh)) { std::fprintf(stderr, "cannot read decoded earth map (P6) at %s\n", opt["earthmap"].c_str()); return 2; }
    }
    void* world = orc_world_new(wn.c_str(), seed, earth.empty() ? nullptr : earth.data(), ew, eh);
    if (!world) { std::fprintf(stderr, "unknown world %s\n", wn.c_str()); return 2; }
    double wc[7];
    orc_world_camera(world, wc);
    double from[3] = {wc[0], wc[1], wc[2]}, at[3] = {wc[3], wc[4], wc[5]}, up[3];
    double fov = wc[6];
    if (opt.count("lookfrom")) parse_vec(opt["lookfrom"], from);
    if (opt.count("lookat")) parse_vec(opt["lookat"], at);
    if (opt.count("field_of_view")) fov = std::atof(opt["field_of_view"].c_str());
    parse_vec(opt["up"], up);
    double focus_dist;
    if (opt.count("focus_dist")) focus_dist = std::atof(opt["focus_dist"].c_str());
    else focus_dist = orc::Vec3(at[0] - from[0], at[1] - from[1], at[2] - from[2]).length();  // main.rs:109-112
    double cam[19];
    orc_camera_new(from, at, up, fov, aspect, std::atof(opt["aperture"].c_str()), focus_dist, cam);

    std::printf("P3\n%d %d\n255\n", width, height);  // main.rs:144
    std::vector<int32_t> rgb((size_t)width * height * 3);
    int rng_mode = opt["rng"] == "philox" ? 1 : 0;
    double secs = orc_render(world, cam, width, height, spp, max_depth, seed, rng_mode, nullptr, 0, std::atoi(opt["threads"].c_str()), rgb.data(),
                             nullptr, nullptr);
    std::fprintf(stderr, "\nRendered in %.3fs\n", secs);  // main.rs:174
    for (int j = height - 1; j >= 0; j--)                   // main.rs:175-179
        for (int i = 0; i < width; i++) {
            const int32_t* p = &rgb[((size_t)j * width + i) * 3];
            std::printf("%d %d %d\n", p[0], p[1], p[2]);
        }
    orc_world_free(world);
    return 0;
}

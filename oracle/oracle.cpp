// oracle/oracle.cpp - world builders, renderer and the C API of the CPU oracle.
// TEST INFRASTRUCTURE ONLY (see oracle_core.hpp for the full header and the parity-pinning statement).
#include "oracle_core.hpp"

#include <atomic>
#include <chrono>
#include <thread>

namespace orc {

thread_local Stats g_stats;

// Inverse-CDF samplers of the device contract (see oracle_core.hpp).  Draw order: u1, u2, u3 / u1, u2.
Vec3 closed_form_unit_sphere(Rng& rng) {
    PhiloxStream& ps = static_cast<PhiloxStream&>(rng);
    const double z = ps.uniform01() * 2.0 + -1.0;
    const uint32_t k = ps.uniform_24_bits();  // phi = 2 pi k 2^-24
    const double r = cbrt24(ps.uniform_24_bits() + 1u);
    double cs_phi, sn_phi;
    turn_cos_sin(k, cs_phi, sn_phi);
    const double s = std::sqrt(std::fmax(0.0, 1.0 - z * z));
    const double rs = r * s;
    return Vec3(rs * cs_phi, rs * sn_phi, r * z);
}
Vec3 closed_form_unit_disk(Rng& rng) {
    PhiloxStream& ps = static_cast<PhiloxStream&>(rng);
    const double r = std::sqrt(ps.uniform_0_1_closed_top());
    double cs_phi, sn_phi;
    turn_cos_sin(ps.uniform_24_bits(), cs_phi, sn_phi);
    return Vec3(r * cs_phi, r * sn_phi, 0.0);
}

// ------------------------------------------------------------------------------------------------
// worlds.rs
// ------------------------------------------------------------------------------------------------
namespace {

struct Builder {
    World& w;
    explicit Builder(World& world) : w(world) {}
    static std::shared_ptr<Texture> solid(double r, double g, double b) { return std::make_shared<SolidColor>(r, g, b); }
    std::shared_ptr<Sphere> sphere(const Point3& c, double r, std::shared_ptr<Material> m) {
        PrimRecord pr{};
        pr.kind = 0;
        pr.g[0] = c.e[0]; pr.g[1] = c.e[1]; pr.g[2] = c.e[2]; pr.g[3] = r;
        m->describe(pr.mat);
        w.prims.push_back(pr);
        return std::make_shared<Sphere>(c, r, m);
    }
    std::shared_ptr<Block> block(const Point3& p0, const Point3& p1, std::shared_ptr<Material> m) {
        PrimRecord pr{};
        pr.kind = 1;
        for (int i = 0; i < 3; i++) { pr.g[i] = p0.e[i]; pr.g[3 + i] = p1.e[i]; }
        m->describe(pr.mat);
        w.prims.push_back(pr);
        return std::make_shared<Block>(p0, p1, m);
    }
    std::shared_ptr<RectShape> rect(std::shared_ptr<RectShape> r) {
        PrimRecord pr{};
        pr.kind = 2;
        pr.g[0] = r->r.a0; pr.g[1] = r->r.a1; pr.g[2] = r->r.a0_v0; pr.g[3] = r->r.a0_v1; pr.g[4] = r->r.a1_v0; pr.g[5] = r->r.a1_v1;
        pr.g[6] = r->r.aplane_v;
        r->material->describe(pr.mat);
        w.prims.push_back(pr);
        return r;
    }
    std::shared_ptr<BHV> bvh(SceneBuilder& sb, Rng& rng) {
        auto b = std::make_shared<BHV>(sb, rng);
        w.bvhs.push_back(b.get());
        return b;
    }
};

double rnd01(Rng& rng) { return rng.gen_range(0.0, 1.0); }  // worlds.rs:62-64

void camera_default(World& w, double fx, double fy, double fz, double ax, double ay, double az, double fov) {
    w.camera.lookfrom = Point3(fx, fy, fz);
    w.camera.lookat = Point3(ax, ay, az);
    w.camera.field_of_view = fov;
}

// worlds.rs:79-112 (Random) and :129-162 (RandomChk); they differ only in the ground texture.
void build_random(World& w, Builder& B, Pcg64& rng, bool checker) {
    SceneBuilder world;
    std::shared_ptr<Texture> ground_tex;
    if (checker) ground_tex = std::make_shared<Checker>(Builder::solid(0.2, 0.3, 0.1), Builder::solid(0.9, 0.9, 0.9));
    else ground_tex = Builder::solid(0.5, 0.5, 0.5);
    world.add(B.sphere(Point3(0.0, -1000.0, 0.0), 1000.0, std::make_shared<Lambertian>(ground_tex)));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            double choose_mat = rnd01(rng);
            double cx = (double)a + 0.9 * rnd01(rng);
            double cz = (double)b + 0.9 * rnd01(rng);
            Point3 center(cx, 0.2, cz);
            if ((center - Point3(4.0, 0.2, 0.0)).length() > 0.9) {
                if (choose_mat < 0.8) {
                    Color c1 = vec3_random(0.0, 1.0, rng);  // Color::random_unit(rng) * Color::random_unit(rng)
                    Color c2 = vec3_random(0.0, 1.0, rng);
                    Color albedo = c1 * c2;
                    world.add(B.sphere(center, 0.2, std::make_shared<Lambertian>(std::make_shared<SolidColor>(albedo))));
                } else if (choose_mat < 0.95) {
                    Color albedo = vec3_random(0.5, 1.0, rng);
                    double fuzz = rng.gen_range(0.0, 0.5);
                    world.add(B.sphere(center, 0.2, std::make_shared<Metal>(albedo, fuzz)));
                } else {
                    world.add(B.sphere(center, 0.2, std::make_shared<Dielectric>(1.5)));
                }
            }
        }
    }
    world.add(B.sphere(Point3(0.0, 1.0, 0.0), 1.0, std::make_shared<Dielectric>(1.5)));
    world.add(B.sphere(Point3(-4.0, 1.0, 0.0), 1.0, std::make_shared<Lambertian>(Builder::solid(0.4, 0.2, 0.1))));
    world.add(B.sphere(Point3(4.0, 1.0, 0.0), 1.0, std::make_shared<Metal>(Color(0.7, 0.6, 0.5), 0.0)));
    { Pcg64 probe = rng; w.raw_u64_before_last_bvh = probe.next_u64(); }
    w.root = B.bvh(world, rng);
    w.background.kind = BG_GRADIENT;
    camera_default(w, 13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 20.0);
}

void build_cornell(World& w, Builder& B, bool smoke) {  // worlds.rs:260-287, :308-335
    auto shapes = std::make_shared<HittableList>();
    auto red = std::make_shared<Lambertian>(Builder::solid(0.65, 0.05, 0.05));
    auto white = std::make_shared<Lambertian>(Builder::solid(0.73, 0.73, 0.73));
    auto green = std::make_shared<Lambertian>(Builder::solid(0.12, 0.45, 0.15));
    auto light = std::make_shared<DiffuseLight>(Builder::solid(7.0, 7.0, 7.0));
    shapes->add(B.rect(yz_rect(0.0, 555.0, 0.0, 555.0, 555.0, green)));
    shapes->add(B.rect(yz_rect(0.0, 555.0, 0.0, 555.0, 0.0, red)));
    shapes->add(B.rect(xz_rect(113.0, 443.0, 127.0, 432.0, 554.0, light)));
    shapes->add(B.rect(xz_rect(0.0, 555.0, 0.0, 555.0, 0.0, white)));
    shapes->add(B.rect(xz_rect(0.0, 555.0, 0.0, 555.0, 555.0, white)));
    shapes->add(B.rect(xy_rect(0.0, 555.0, 0.0, 555.0, 555.0, white)));
    std::shared_ptr<Bounded> large = B.block(Point3(0, 0, 0), Point3(165.0, 330.0, 165.0), white);
    large = std::make_shared<Rotate>(1, 15.0, large);
    large = std::make_shared<Translate>(Vec3(265.0, 0.0, 295.0), large);
    std::shared_ptr<Bounded> small_b = B.block(Point3(0, 0, 0), Point3(165.0, 165.0, 165.0), white);
    small_b = std::make_shared<Rotate>(1, -18.0, small_b);
    small_b = std::make_shared<Translate>(Vec3(130.0, 0.0, 65.0), small_b);
    if (smoke) {
        shapes->add(std::make_shared<ConstantMedium>(large, 0.01, Builder::solid(0, 0, 0)));
        shapes->add(std::make_shared<ConstantMedium>(small_b, 0.01, Builder::solid(1, 1, 1)));
    } else {
        shapes->add(large);
        shapes->add(small_b);
    }
    w.root = shapes;
    w.background.kind = BG_BLACK;
    camera_default(w, 278.0, 278.0, -800.0, 278.0, 278.0, 0.0, 40.0);
}

void build_final_scene(World& w, Builder& B, Pcg64& rng, std::shared_ptr<ImageData> earth) {  // worlds.rs:386-468
    auto shapes = std::make_shared<HittableList>();
    {  // Light.
        auto light = std::make_shared<DiffuseLight>(Builder::solid(9.0, 9.0, 9.0));
        shapes->add(B.rect(xz_rect(123.0, 423.0, 147.0, 412.0, 554.0, light)));
    }
    {  // Ground.
        auto ground = std::make_shared<Lambertian>(Builder::solid(0.48, 0.83, 0.53));
        const int BLOCKS_PER_SIDE = 20;
        SceneBuilder ground_blocks;
        for (int i = 0; i < BLOCKS_PER_SIDE; i++) {
            for (int j = 0; j < BLOCKS_PER_SIDE; j++) {
                double wd = 100.0;
                double x0 = -1000.0 + (double)i * wd;
                double z0 = -1000.0 + (double)j * wd;
                double y0 = 0.0;
                double x1 = x0 + wd;
                double y1 = rng.gen_range(1.0, 70.0);
                double z1 = z0 + wd;
                ground_blocks.add(B.block(Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
            }
        }
        shapes->add(B.bvh(ground_blocks, rng));
    }
    shapes->add(B.sphere(Point3(400.0, 400.0, 400.0), 50.0, std::make_shared<Lambertian>(Builder::solid(0.7, 0.3, 0.1))));
    shapes->add(B.sphere(Point3(260.0, 150.0, 45.0), 50.0, std::make_shared<Dielectric>(1.5)));
    shapes->add(B.sphere(Point3(0.0, 150.0, 145.0), 50.0, std::make_shared<Metal>(Color(0.8, 0.8, 0.9), 1.0)));
    {  // Glass sphere filled with blue-ish smoke
        auto boundary = B.sphere(Point3(360.0, 150.0, 145.0), 70.0, std::make_shared<Dielectric>(1.5));
        shapes->add(boundary);
        shapes->add(std::make_shared<ConstantMedium>(std::make_shared<Sphere>(*boundary), 0.2, Builder::solid(0.2, 0.4, 0.9)));
    }
    {  // Fill everything with fog.
        auto boundary = std::make_shared<Sphere>(Point3(0, 0, 0), 1000.0, std::make_shared<Dielectric>(1.5));
        shapes->add(std::make_shared<ConstantMedium>(boundary, 0.0001, Builder::solid(1, 1, 1)));
    }
    {  // Earth.
        if (!earth || earth->rgb.empty()) { std::fprintf(stderr, "oracle: final_scene needs the decoded earthmap image\n"); std::abort(); }
        shapes->add(B.sphere(Point3(400.0, 200.0, 400.0), 100.0, std::make_shared<Lambertian>(std::make_shared<ImageTexture>(earth))));
    }
    {  // Noise.
        auto pertext = std::make_shared<NoiseTexture>(0.1, rng);
        w.perlins.push_back(pertext->noise);
        shapes->add(B.sphere(Point3(220.0, 280.0, 300.0), 80.0, std::make_shared<Lambertian>(pertext)));
    }
    {  // Foam.
        SceneBuilder foam;
        auto white = std::make_shared<Lambertian>(Builder::solid(0.73, 0.73, 0.73));
        for (int i = 0; i < 1000; i++) foam.add(B.sphere(vec3_random(0.0, 165.0, rng), 10.0, white));
        { Pcg64 probe = rng; w.raw_u64_before_last_bvh = probe.next_u64(); }
        std::shared_ptr<Bounded> foam_bvh = B.bvh(foam, rng);
        shapes->add(std::make_shared<Translate>(Vec3(-100.0, 270.0, 395.0), std::make_shared<Rotate>(1, 15.0, foam_bvh)));
    }
    w.root = shapes;
    w.background.kind = BG_BLACK;
    camera_default(w, 478.0, 278.0, -600.0, 278.0, 278.0, 0.0, 40.0);
}

}  // namespace

const std::vector<std::string>& world_names() {  // worlds.rs:471-484
    static const std::vector<std::string> names = {"simple", "random", "random_chk", "two_spheres", "simple_light",
                                                   "cornell_box", "cornell_smoke", "earth", "debug_perlin", "final_scene"};
    return names;
}

std::unique_ptr<World> build_world(const std::string& name, uint64_t seed, std::shared_ptr<ImageData> earth) {
    std::unique_ptr<World> wp(new World());
    World& w = *wp;
    w.name = name;
    Builder B(w);
    Pcg64 rng = Pcg64::seed_from_u64(seed + 0);  // main.rs:185: rngator.rng(0)
    if (name == "simple") {  // worlds.rs:42-59
        SceneBuilder world;
        auto mat_left = std::make_shared<Dielectric>(1.5);
        world.add(B.sphere(Point3(0.0, -100.5, -1.0), 100.0, std::make_shared<Lambertian>(Builder::solid(0.8, 0.8, 0.0))));
        world.add(B.sphere(Point3(0.0, 0.0, -1.0), 0.5, std::make_shared<Lambertian>(Builder::solid(0.1, 0.3, 0.5))));
        world.add(B.sphere(Point3(-1.0, 0.0, -1.0), 0.5, mat_left));
        world.add(B.sphere(Point3(-1.0, 0.0, -1.0), -0.4, mat_left));
        world.add(B.sphere(Point3(1.0, 0.0, -1.0), 0.5, std::make_shared<Metal>(Color(0.8, 0.6, 0.2), 0.0)));
        w.root = B.bvh(world, rng);
        w.background.kind = BG_GRADIENT;
        camera_default(w, -2.0, 2.0, 1.0, 0.0, 0.0, -1.0, 20.0);
    } else if (name == "random") {
        build_random(w, B, rng, false);
    } else if (name == "random_chk") {
        build_random(w, B, rng, true);
    } else if (name == "earth") {  // worlds.rs:179-186
        if (!earth || earth->rgb.empty()) { std::fprintf(stderr, "oracle: earth needs the decoded earthmap image\n"); std::abort(); }
        w.root = B.sphere(Point3(0, 0, 0), 2.0, std::make_shared<Lambertian>(std::make_shared<ImageTexture>(earth)));
        w.background.kind = BG_GRADIENT;
        camera_default(w, 13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 20.0);
    } else if (name == "two_spheres" || name == "simple_light") {  // worlds.rs:203-210, :227-239
        auto shapes = std::make_shared<HittableList>();
        auto pertext = std::make_shared<NoiseTexture>(4.0, rng);
        w.perlins.push_back(pertext->noise);
        shapes->add(B.sphere(Point3(0.0, -1000.0, 0.0), 1000.0, std::make_shared<Lambertian>(pertext)));
        shapes->add(B.sphere(Point3(0.0, 2.0, 0.0), 2.0, std::make_shared<Lambertian>(pertext)));
        if (name == "simple_light") {
            shapes->add(B.rect(xy_rect(3.0, 5.0, 1.0, 3.0, -2.0, std::make_shared<DiffuseLight>(Builder::solid(0.0, 7.0, 0.0)))));
            shapes->add(B.sphere(Point3(0.0, 6.0, 0.0), 1.5, std::make_shared<DiffuseLight>(Builder::solid(7.0, 0.0, 0.0))));
            w.background.kind = BG_BLACK;
            camera_default(w, 20.0, 3.0, 6.0, 0.0, 2.0, 0.0, 20.0);
        } else {
            w.background.kind = BG_GRADIENT;
            camera_default(w, 13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 20.0);
        }
        w.root = shapes;
    } else if (name == "cornell_box") {
        build_cornell(w, B, false);
    } else if (name == "cornell_smoke") {
        build_cornell(w, B, true);
    } else if (name == "debug_perlin") {  // worlds.rs:355-365
        auto shapes = std::make_shared<HittableList>();
        auto pertext = std::make_shared<NoiseTexture>(0.1, rng);
        w.perlins.push_back(pertext->noise);
        shapes->add(B.sphere(Point3(278.0, 278.0, 0.0), 80.0, std::make_shared<Lambertian>(pertext)));
        w.root = shapes;
        w.background.kind = BG_GRADIENT;
        camera_default(w, 278.0, 278.0, -600.0, 278.0, 278.0, 0.0, 40.0);
    } else if (name == "final_scene") {
        build_final_scene(w, B, rng, earth);
    } else {
        return nullptr;
    }
    w.next_raw_u64_after_build = rng.next_u64();
    return wp;
}

// ------------------------------------------------------------------------------------------------
// raytrace.rs:137-199 Renderer
// ------------------------------------------------------------------------------------------------
enum RngMode { RNG_PCG_ROW = 0, RNG_PHILOX_SAMPLE = 1 };

struct RenderJob {
    const World* world;
    Camera camera;
    int width, height, spp, max_depth;
    uint64_t seed;
    int rng_mode;
    bool single_light = false;  // the RayTracer of the Renderer: RecursiveRayTracer{max_depth} or SingleLightSourceRayTracer{light_source, intensity}
    Point3 light_source;
    double intensity = 0.0;
    Color trace(const Ray& r, Rng& rng) const {  // RayTracer::trace, raytrace.rs:70-72
        return single_light ? trace_single_light(r, *world->root, world->background, light_source, intensity, rng)
                            : trace_internal(r, *world->root, world->background, max_depth, rng);
    }
    std::vector<int> rows;   // rows to render (j = 0 is the bottom row)
    int32_t* rgb;            // [H][W][3] (only listed rows are written)
    double* accum;           // optional [H][W][3] colour sums
};

static void render_pixel(const RenderJob& job, int i, int j, Rng* row_rng) {  // raytrace.rs:188-198
    Color pixel_color(0, 0, 0);
    const double wm1 = (double)job.width - 1.0, hm1 = (double)job.height - 1.0;
    for (int s = 0; s < job.spp; s++) {
        g_stats.samples++;
        if (job.rng_mode == RNG_PCG_ROW) {
            double u = ((double)i + row_rng->gen_range(0.0, 1.0)) / wm1;
            double v = ((double)j + row_rng->gen_range(0.0, 1.0)) / hm1;
            Ray r = job.camera.get_ray(u, v, *row_rng);
            pixel_color = pixel_color + job.trace(r, *row_rng);
        } else {
            PhiloxStream rng(job.seed, (uint32_t)(j * job.width + i), (uint32_t)s);
            double u = ((double)i + rng.gen_range(0.0, 1.0)) / wm1;
            double v = ((double)j + rng.gen_range(0.0, 1.0)) / hm1;
            Ray r = job.camera.get_ray(u, v, rng);
            pixel_color = pixel_color + job.trace(r, rng);
        }
    }
    size_t idx = ((size_t)j * job.width + i) * 3;
    if (job.accum) { job.accum[idx] = pixel_color.e[0]; job.accum[idx + 1] = pixel_color.e[1]; job.accum[idx + 2] = pixel_color.e[2]; }
    to_rgb(pixel_color, job.spp, job.rgb + idx);
}

static void render_rows(const RenderJob& job, int threads, Stats* stats_out) {  // raytrace.rs:172-186
    std::atomic<size_t> next(0);
    std::vector<Stats> per_thread((size_t)threads);
    auto worker = [&](int tid) {
        g_stats = Stats();
        for (;;) {
            size_t k = next.fetch_add(1);
            if (k >= job.rows.size()) break;
            int j = job.rows[k];
            Pcg64 row_rng = Pcg64::seed_from_u64(job.seed + (uint64_t)j);  // rngator.rs:29-30, raytrace.rs:179
            for (int i = 0; i < job.width; i++) render_pixel(job, i, j, &row_rng);
        }
        per_thread[(size_t)tid] = g_stats;
    };
    if (threads <= 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
        for (auto& th : pool) th.join();
    }
    if (stats_out) {
        Stats total;
        for (const auto& s : per_thread) total.add(s);
        *stats_out = total;
    }
}

}  // namespace orc

// ------------------------------------------------------------------------------------------------
// C API (ctypes) - used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline only.
// ------------------------------------------------------------------------------------------------
using namespace orc;

struct OrcWorld {
    std::unique_ptr<World> world;
    std::shared_ptr<ImageData> earth;
};

namespace orc { const Hittable* orc_scene_root(void* scene); }  // oracle_scene.cpp

extern "C" {

void orc_pcg64_seed_from_u64_stream(uint64_t seed, int n, uint64_t* out) {
    Pcg64 p = Pcg64::seed_from_u64(seed);
    for (int i = 0; i < n; i++) out[i] = p.next_u64();
}
void orc_pcg64_state_after_seed(uint64_t seed, uint64_t out[4]) {
    Pcg64 p = Pcg64::seed_from_u64(seed);
    out[0] = (uint64_t)p.state; out[1] = (uint64_t)(p.state >> 64); out[2] = (uint64_t)p.inc; out[3] = (uint64_t)(p.inc >> 64);
}
void orc_pcg64_new_stream(uint64_t state_lo, uint64_t state_hi, uint64_t stream_lo, uint64_t stream_hi, int n, uint64_t* out) {
    Pcg64 p = Pcg64::from_state_stream(((unsigned __int128)state_hi << 64) | state_lo, ((unsigned __int128)stream_hi << 64) | stream_lo);
    for (int i = 0; i < n; i++) out[i] = p.next_u64();
}
void orc_pcg64_uniform(uint64_t seed, double lo, double hi, int n, double* out) {
    Pcg64 p = Pcg64::seed_from_u64(seed);
    for (int i = 0; i < n; i++) out[i] = p.gen_range(lo, hi);
}
void orc_pcg64_usize(uint64_t seed, uint64_t lo, uint64_t hi, int n, uint64_t* out) {
    Pcg64 p = Pcg64::seed_from_u64(seed);
    for (int i = 0; i < n; i++) out[i] = p.gen_range_usize(lo, hi);
}
void orc_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) { philox4x32_10(key, ctr, out); }
void orc_philox_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, int n, double* out) {
    PhiloxStream s(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = s.uniform01();
}

// n samples of the unit-ball / unit-disk samplers: mode 0 = reference rejection on Pcg64(seed), 1 = device contract on Philox(seed, pixel 0, sample i)
// The contract's own transcendental arithmetic against long-double libm over ALL 2^24 inputs: out = max relative error of cbrt24,
// max absolute errors of turn_cos_sin's cos and sin, and the sampler values at the ends (cbrt24(1), cbrt24(2^24), cos / sin of k = 0, 2^22).
void orc_contract_function_errors(double out[10]) {
    long double worst_cbrt = 0.0L, worst_cos = 0.0L, worst_sin = 0.0L;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (uint32_t n = 1; n <= (1u << 24); n++) {
        const long double t = cbrtl((long double)n / 16777216.0L);
        worst_cbrt = std::fmax(worst_cbrt, fabsl(((long double)cbrt24(n) - t) / t));
        double cs, sn;
        turn_cos_sin(n - 1u, cs, sn);
        const long double a = two_pi * (long double)(n - 1u) / 16777216.0L;
        worst_cos = std::fmax(worst_cos, fabsl((long double)cs - cosl(a)));
        worst_sin = std::fmax(worst_sin, fabsl((long double)sn - sinl(a)));
    }
    out[0] = (double)worst_cbrt; out[1] = (double)worst_cos; out[2] = (double)worst_sin;
    out[3] = cbrt24(1u); out[4] = cbrt24(1u << 24);
    turn_cos_sin(0u, out[5], out[6]);
    turn_cos_sin(1u << 22, out[7], out[8]);  // a quarter turn
    out[9] = 0.0;
}
void orc_sample_ball(uint64_t seed, int mode, int n, double* out) {
    if (mode == 0) { Pcg64 g = Pcg64::seed_from_u64(seed); for (int i = 0; i < n; i++) { Vec3 p = random_in_unit_sphere(g); out[3 * i] = p.e[0]; out[3 * i + 1] = p.e[1]; out[3 * i + 2] = p.e[2]; } }
    else for (int i = 0; i < n; i++) { PhiloxStream g(seed, 0, (uint32_t)i); Vec3 p = random_in_unit_sphere(g); out[3 * i] = p.e[0]; out[3 * i + 1] = p.e[1]; out[3 * i + 2] = p.e[2]; }
}
void orc_sample_disk(uint64_t seed, int mode, int n, double* out) {
    if (mode == 0) { Pcg64 g = Pcg64::seed_from_u64(seed); for (int i = 0; i < n; i++) { Vec3 p = random_in_unit_disk(g); out[2 * i] = p.e[0]; out[2 * i + 1] = p.e[1]; } }
    else for (int i = 0; i < n; i++) { PhiloxStream g(seed, 0, (uint32_t)i); Vec3 p = random_in_unit_disk(g); out[2 * i] = p.e[0]; out[2 * i + 1] = p.e[1]; }
}

int orc_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3], double tmin, double tmax) {
    AABB b(Point3(mn[0], mn[1], mn[2]), Point3(mx[0], mx[1], mx[2]));
    return b.hit(Ray(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2])), tmin, tmax) ? 1 : 0;
}
void orc_sphere_uv(const double n[3], double out[2]) { sphere_uv(Vec3(n[0], n[1], n[2]), out[0], out[1]); }
void orc_to_rgb(const double c[3], int spp, int32_t out[3]) { to_rgb(Color(c[0], c[1], c[2]), spp, out); }
// out: origin, lower_left_corner, horizontal, vertical, u, v (3 each) + lens_radius = 19 doubles
void orc_camera_new(const double from[3], const double at[3], const double up[3], double vfov, double aspect, double aperture,
                    double focus_dist, double out[19]) {
    Camera c(Point3(from[0], from[1], from[2]), Point3(at[0], at[1], at[2]), Vec3(up[0], up[1], up[2]), vfov, aspect, aperture, focus_dist);
    const Vec3* v[6] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v};
    for (int i = 0; i < 6; i++) for (int k = 0; k < 3; k++) out[i * 3 + k] = v[i]->e[k];
    out[18] = c.lens_radius;
}
// Sphere::hit known-answer helper: returns 1 and fills t, p, normal, u, v, front_face on hit.
int orc_sphere_hit(const double c[3], double radius, const double o[3], const double d[3], double tmin, double tmax, double out[10]) {
    auto m = std::make_shared<Dielectric>(1.5);
    Sphere s(Point3(c[0], c[1], c[2]), radius, m);
    Hit h;
    Pcg64 rng;
    if (!s.hit(Ray(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2])), tmin, tmax, rng, h)) return 0;
    out[0] = h.t; for (int k = 0; k < 3; k++) { out[1 + k] = h.p.e[k]; out[4 + k] = h.normal.e[k]; }
    out[7] = h.u; out[8] = h.v; out[9] = h.front_face ? 1.0 : 0.0;
    return 1;
}

int orc_world_count(void) { return (int)world_names().size(); }
const char* orc_world_name(int i) { return world_names()[(size_t)i].c_str(); }

// earth_rgb may be NULL for worlds that do not use the image texture.
void* orc_world_new(const char* name, uint64_t seed, const uint8_t* earth_rgb, int earth_w, int earth_h) {
    OrcWorld* ow = new OrcWorld();
    if (earth_rgb) {
        ow->earth = std::make_shared<ImageData>();
        ow->earth->width = earth_w; ow->earth->height = earth_h;
        ow->earth->rgb.assign(earth_rgb, earth_rgb + (size_t)earth_w * earth_h * 3);
    }
    std::string n(name);
    bool needs_earth = (n == "earth" || n == "final_scene");
    if (needs_earth && !earth_rgb) { delete ow; return nullptr; }
    ow->world = build_world(n, seed, ow->earth);
    if (!ow->world) { delete ow; return nullptr; }
    return ow;
}
void orc_world_free(void* h) { delete (OrcWorld*)h; }
int orc_world_prim_count(void* h) { return (int)((OrcWorld*)h)->world->prims.size(); }
// per prim: kind, g[8], mat[5] = 14 doubles
void orc_world_prims(void* h, double* out) {
    const auto& prims = ((OrcWorld*)h)->world->prims;
    for (size_t i = 0; i < prims.size(); i++) {
        double* o = out + i * 14;
        o[0] = prims[i].kind;
        for (int k = 0; k < 8; k++) o[1 + k] = prims[i].g[k];
        for (int k = 0; k < 5; k++) o[9 + k] = prims[i].mat[k];
    }
}
uint64_t orc_world_next_raw_u64(void* h) { return ((OrcWorld*)h)->world->next_raw_u64_after_build; }
uint64_t orc_world_raw_u64_before_last_bvh(void* h) { return ((OrcWorld*)h)->world->raw_u64_before_last_bvh; }
void orc_world_camera(void* h, double out[7]) {
    const WorldCamera& c = ((OrcWorld*)h)->world->camera;
    for (int k = 0; k < 3; k++) { out[k] = c.lookfrom.e[k]; out[3 + k] = c.lookat.e[k]; }
    out[6] = c.field_of_view;
}
int orc_world_background(void* h) { return ((OrcWorld*)h)->world->background.kind; }
int orc_world_bvh_count(void* h) { return (int)((OrcWorld*)h)->world->bvhs.size(); }

static void bvh_stats(const BvhNode* n, int depth, int* nodes, int* leaves, int* max_depth) {
    (*nodes)++;
    if (depth > *max_depth) *max_depth = depth;
    if (n->leaf) { (*leaves)++; return; }
    bvh_stats(n->left.get(), depth + 1, nodes, leaves, max_depth);
    bvh_stats(n->right.get(), depth + 1, nodes, leaves, max_depth);
}
void orc_world_bvh_info(void* h, int which, int out[3]) {
    out[0] = out[1] = out[2] = 0;
    bvh_stats(((OrcWorld*)h)->world->bvhs[(size_t)which]->root.get(), 0, &out[0], &out[1], &out[2]);
}
// Pre-order dump of one BVH: per node 8 doubles: is_leaf, then for an inner node its bounds (6) and 0;
// for a leaf the bounding box of its shape (6) and 0.  Used to compare topologies with the product's flattened tree.
static void bvh_dump(const BvhNode* n, std::vector<double>& out) {
    AABB b = n->bounding_box();
    out.push_back(n->leaf ? 1.0 : 0.0);
    for (int k = 0; k < 3; k++) out.push_back(b.minimum.e[k]);
    for (int k = 0; k < 3; k++) out.push_back(b.maximum.e[k]);
    out.push_back(0.0);
    if (!n->leaf) { bvh_dump(n->left.get(), out); bvh_dump(n->right.get(), out); }
}
int orc_world_bvh_dump(void* h, int which, double* out, int max_nodes) {
    std::vector<double> v;
    bvh_dump(((OrcWorld*)h)->world->bvhs[(size_t)which]->root.get(), v);
    int n = (int)(v.size() / 8);
    if (out && n <= max_nodes) std::memcpy(out, v.data(), v.size() * sizeof(double));
    return n;
}
int orc_world_perlin_count(void* h) { return (int)((OrcWorld*)h)->world->perlins.size(); }
// ranvec: 1024*3 doubles; perms: 3*1024 int32
void orc_world_perlin(void* h, int which, double* ranvec, int32_t* perms) {
    const Perlin& p = *((OrcWorld*)h)->world->perlins[(size_t)which];
    for (int i = 0; i < 1024; i++) {
        for (int k = 0; k < 3; k++) ranvec[i * 3 + k] = p.ranvec[i].e[k];
        perms[i] = (int32_t)p.perm_x[i]; perms[1024 + i] = (int32_t)p.perm_y[i]; perms[2048 + i] = (int32_t)p.perm_z[i];
    }
}
void orc_perlin_eval(void* h, int which, const double p[3], double out[2]) {
    const Perlin& pn = *((OrcWorld*)h)->world->perlins[(size_t)which];
    out[0] = pn.noise(Point3(p[0], p[1], p[2]));
    out[1] = pn.turbulence(Point3(p[0], p[1], p[2]));
}

// Trace one explicit ray with a Philox stream (pixel, sample) whose first `skip_draws` draws are skipped.
// Returns the colour.  Used for ray-level known answers against the HIP path.
void orc_trace_ray(void* h, const double o[3], const double d[3], int max_depth, uint64_t seed, uint32_t pixel, uint32_t sample, double out[3]) {
    const World& w = *((OrcWorld*)h)->world;
    PhiloxStream rng(seed, pixel, sample);
    Color c = trace_internal(Ray(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2])), *w.root, w.background, max_depth, rng);
    out[0] = c.e[0]; out[1] = c.e[1]; out[2] = c.e[2];
}

// camera19: as produced by orc_camera_new.  rows: list of row indices (j=0 bottom) or NULL for all rows.
// rgb: [H][W][3] int32 (only the listed rows are written); accum: optional [H][W][3] f64 colour sums.
// stats_out: optional 10 uint64: samples, rays, aabb, sphere, rect, scatters, draws, noise, texel, medium.
// Returns the render wall time in seconds (the reference's timer window, main.rs:145-174).
static double render_world(const World* world, const double camera19[19], int width, int height, int spp, int max_depth, uint64_t seed, int rng_mode,
                           const int* rows, int n_rows, int threads, int32_t* rgb, double* accum, uint64_t* stats_out, const double* single_light = nullptr) {
    RenderJob job;
    if (single_light) { job.single_light = true; job.light_source = Point3(single_light[0], single_light[1], single_light[2]); job.intensity = single_light[3]; }
    job.world = world;
    const double* c = camera19;
    job.camera.origin = Point3(c[0], c[1], c[2]);
    job.camera.lower_left_corner = Point3(c[3], c[4], c[5]);
    job.camera.horizontal = Vec3(c[6], c[7], c[8]);
    job.camera.vertical = Vec3(c[9], c[10], c[11]);
    job.camera.u = Vec3(c[12], c[13], c[14]);
    job.camera.v = Vec3(c[15], c[16], c[17]);
    job.camera.lens_radius = c[18];
    job.width = width; job.height = height; job.spp = spp; job.max_depth = max_depth; job.seed = seed; job.rng_mode = rng_mode;
    if (rows) job.rows.assign(rows, rows + n_rows);
    else for (int j = 0; j < height; j++) job.rows.push_back(j);
    job.rgb = rgb; job.accum = accum;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    Stats st;
    auto t0 = std::chrono::steady_clock::now();
    render_rows(job, threads, &st);
    auto t1 = std::chrono::steady_clock::now();
    if (stats_out) {
        stats_out[0] = st.samples; stats_out[1] = st.rays; stats_out[2] = st.aabb_tests; stats_out[3] = st.sphere_tests; stats_out[4] = st.rect_tests;
        stats_out[5] = st.scatters; stats_out[6] = st.draws; stats_out[7] = st.noise_evals; stats_out[8] = st.texel_fetches; stats_out[9] = st.medium_tests;
    }
    return std::chrono::duration<double>(t1 - t0).count();
}
double orc_render(void* h, const double camera19[19], int width, int height, int spp, int max_depth, uint64_t seed, int rng_mode,
                  const int* rows, int n_rows, int threads, int32_t* rgb, double* accum, uint64_t* stats_out) {
    return render_world(((OrcWorld*)h)->world.get(), camera19, width, height, spp, max_depth, seed, rng_mode, rows, n_rows, threads, rgb, accum, stats_out);
}
// The same loop with Renderer's other RayTracer, SingleLightSourceRayTracer{light_source, intensity} (raytrace.rs:110-135); light4 = x, y, z, intensity.
double orc_render_single_light(void* h, const double camera19[19], int width, int height, int spp, const double light4[4], uint64_t seed, int rng_mode,
                               const int* rows, int n_rows, int threads, int32_t* rgb, double* accum) {
    return render_world(((OrcWorld*)h)->world.get(), camera19, width, height, spp, 1, seed, rng_mode, rows, n_rows, threads, rgb, accum, nullptr, light4);
}
void orc_trace_ray_single_light(void* h, const double o[3], const double d[3], const double light4[4], uint64_t seed, uint32_t pixel, uint32_t sample, double out[3]) {
    const World& w = *((OrcWorld*)h)->world;
    PhiloxStream rng(seed, pixel, sample);
    Color c = trace_single_light(Ray(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2])), *w.root, w.background, Point3(light4[0], light4[1], light4[2]), light4[3], rng);
    out[0] = c.e[0]; out[1] = c.e[1]; out[2] = c.e[2];
}
// The same render over an arbitrary flattened scene (oracle_scene.cpp, orc_scene_new): the randomised parity tests.
// background_kind: 0 black, 1 gradient (bottom .. top, raytrace.rs:29-35).
double orc_scene_render(void* scene, int background_kind, const double top[3], const double bottom[3], const double camera19[19], int width, int height,
                        int spp, int max_depth, uint64_t seed, int rng_mode, const int* rows, int n_rows, int threads, int32_t* rgb, double* accum) {
    World w;
    w.name = "scene";
    w.root = std::shared_ptr<Hittable>(std::shared_ptr<Hittable>(), const_cast<Hittable*>(orc_scene_root(scene)));  // not owned
    w.background.kind = background_kind == 0 ? BG_BLACK : BG_GRADIENT;
    w.background.top = Color(top[0], top[1], top[2]);
    w.background.bottom = Color(bottom[0], bottom[1], bottom[2]);
    return render_world(&w, camera19, width, height, spp, max_depth, seed, rng_mode, rows, n_rows, threads, rgb, accum, nullptr);
}

}  // extern "C"

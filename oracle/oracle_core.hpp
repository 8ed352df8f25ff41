// oracle/oracle_core.hpp
//
// TEST INFRASTRUCTURE ONLY.  This is the CPU oracle: an f64 restatement of the hot path of the
// reference CPU path tracer (mu-lambda-raytracer, /root/reference/src/*.rs).  It is the checker for
// the HIP path and the timed "cpu_baseline" (kind "port") in bench.py.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
// mu-lambda-raytracer_amd/ links, includes or calls it.
//
// Parity status: the reference is Rust and cannot be built here (no rustc/cargo, crates not vendored),
// so this restatement is pinned by (1) the reference's own unit tests (bhv.rs:168-221, shapes.rs:200-214),
// (2) published PCG64 / Philox known answers for the third-party generators (rand_pcg 0.3.0,
// rand_core 0.6.2, rand 0.8.3 - restated from their published algorithms), and (3) the three README
// renders (sample.jpg, sample_blur.jpg, final_scene.jpg) whose decoded rows are committed under
// tests/golden/.  Functions the reference's tests do not cover (Sphere::hit, materials, media, textures,
// camera, to_rgb) are pinned only through those images: "parity unpinned by tests" for them.
//
// Every function cites the reference file:line it follows.  Arithmetic order is kept as in the
// reference (left-to-right, division not reciprocal-multiply) and the file is compiled with
// -ffp-contract=off so no FMA contraction changes roundings.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

namespace orc {

static const double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI
static const double INF = std::numeric_limits<double>::infinity();

// ------------------------------------------------------------------------------------------------
// Instrumentation (per thread).  Feeds SURVEY.md section 8(d): algorithmic bytes per sample.
// ------------------------------------------------------------------------------------------------
struct Stats {
    uint64_t samples = 0, rays = 0, aabb_tests = 0, sphere_tests = 0, rect_tests = 0, scatters = 0;
    uint64_t draws = 0, noise_evals = 0, texel_fetches = 0, medium_tests = 0;
    void add(const Stats& o) {
        samples += o.samples; rays += o.rays; aabb_tests += o.aabb_tests; sphere_tests += o.sphere_tests;
        rect_tests += o.rect_tests; scatters += o.scatters; draws += o.draws; noise_evals += o.noise_evals;
        texel_fetches += o.texel_fetches; medium_tests += o.medium_tests;
    }
};
extern thread_local Stats g_stats;

// ------------------------------------------------------------------------------------------------
// vec.rs
// ------------------------------------------------------------------------------------------------
struct Vec3 {  // vec.rs:5-8
    double e[3];
    Vec3() : e{0, 0, 0} {}
    Vec3(double a, double b, double c) : e{a, b, c} {}
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }  // vec.rs:83-85
    double length() const { return std::sqrt(length_squared()); }                      // vec.rs:63-65
    double dot(const Vec3& v) const { return e[0] * v.e[0] + e[1] * v.e[1] + e[2] * v.e[2]; }  // vec.rs:71-73
    Vec3 cross(const Vec3& v) const {                                                          // vec.rs:75-81
        return Vec3(e[1] * v.e[2] - e[2] * v.e[1], e[2] * v.e[0] - e[0] * v.e[2], e[0] * v.e[1] - e[1] * v.e[0]);
    }
    Vec3 unit() const;  // vec.rs:67-69
    bool near_zero() const {  // vec.rs:54-57
        const double S = 1e-8;
        return std::fabs(e[0]) < S && std::fabs(e[1]) < S && std::fabs(e[2]) < S;
    }
};
typedef Vec3 Point3;
typedef Vec3 Color;
inline Vec3 operator-(const Vec3& a) { return Vec3(-a.e[0], -a.e[1], -a.e[2]); }                              // vec.rs:114-119
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }  // vec.rs:135-140
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }  // vec.rs:149-154
inline Vec3 operator*(const Vec3& a, double s) { return Vec3(a.e[0] * s, a.e[1] * s, a.e[2] * s); }            // vec.rs:156-161
inline Vec3 operator*(double s, const Vec3& a) { return a * s; }                                               // vec.rs:198-203 (rhs * self)
inline Vec3 operator*(const Vec3& a, const Vec3& b) { return Vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }  // vec.rs:163-168
inline Vec3 operator/(const Vec3& a, double s) { return Vec3(a.e[0] / s, a.e[1] / s, a.e[2] / s); }            // vec.rs:184-189
inline Vec3 Vec3::unit() const { return *this / length(); }

struct Ray {  // vec.rs:215-228
    Point3 orig;
    Vec3 dir;
    Ray() {}
    Ray(const Point3& o, const Vec3& d) : orig(o), dir(d) {}
    Point3 at(double t) const { return orig + (t * dir); }
};

// ------------------------------------------------------------------------------------------------
// Random sources.
//
// Rng mirrors `&mut dyn rand::RngCore` + the `rand::Rng::gen_range` call sites.  Two generators:
//  * Pcg64: rand_pcg 0.3.0 `Pcg64 = Lcg128Xsl64` seeded through rand_core 0.6.2 `seed_from_u64`
//    (rngator.rs:27-31).  Third-party crates, sources absent from /root/reference; restated from
//    their published algorithms and pinned by the pcg64 demo vector + SURVEY.md section 8c known answers.
//  * PhiloxStream: Philox4x32-10 (Salmon et al., SC'11), the DEVICE's generator: one stream per
//    (pixel, sample); draw k is 32-bit word k%4 of block k/4; uniform = (word >> 8) * 2^-24.  A Material::scatter
//    starts on a block boundary and owns words 0-2 of its block, the draw after it is word 3 (begin_scatter /
//    end_scatter), so the device evaluates Philox exactly once per scatter with no per-lane refill logic.
//    The HIP kernel consumes the identical words in the identical order, which makes
//    oracle[Philox] vs GPU[Philox] comparable sample for sample.
// ------------------------------------------------------------------------------------------------
struct Vec3;
struct Rng {
    virtual ~Rng() {}
    virtual uint64_t next_u64() = 0;
    // vec.rs:23-30 / :45-52.  The reference (and the Pcg64 mode here) draws these by rejection; the device contract
    // (PhiloxStream) draws the SAME distributions by inverse CDF from exactly 3 / 2 uniforms, see below.
    virtual bool closed_form_samplers() const { return false; }
    // Device contract hooks around every Material::scatter call (no-ops for the reference's Pcg64 streams): see PhiloxStream.
    virtual void begin_scatter() {}
    virtual void end_scatter() {}
    // uniform in [0,1): the value rand 0.8.3's UniformFloat::sample_single forms before scaling.
    virtual double uniform01() = 0;
    // rand 0.8.3 UniformFloat<f64>::sample_single: value0_1 * (high-low) + low, retried if >= high.
    double gen_range(double lo, double hi) {
        const double scale = hi - lo;
        for (;;) {
            double r = uniform01() * scale + lo;
            if (r < hi) return r;
        }
    }
    // rand 0.8.3 UniformInt<usize>::sample_single (64-bit usize): widening multiply with the
    // "conservative zone" rejection.  Draw COUNT is data dependent (25% rejections for range 3).
    uint64_t gen_range_usize(uint64_t lo, uint64_t hi) {
        const uint64_t range = hi - lo;
        const uint64_t zone = (range << __builtin_clzll(range)) - 1;
        for (;;) {
            uint64_t v = next_u64();
            unsigned __int128 m = (unsigned __int128)v * (unsigned __int128)range;
            uint64_t mlo = (uint64_t)m, mhi = (uint64_t)(m >> 64);
            if (mlo <= zone) return lo + mhi;
        }
    }
};

struct Pcg64 : Rng {
    unsigned __int128 state, inc;
    static unsigned __int128 mult() {
        return ((unsigned __int128)0x2360ED051FC65DA4ULL << 64) | (unsigned __int128)0x4385DF649FCCF645ULL;
    }
    Pcg64() : state(0), inc(1) {}
    // Lcg128Xsl64::new(state, stream): increment = (stream << 1) | 1
    static Pcg64 from_state_stream(unsigned __int128 st, unsigned __int128 stream) {
        Pcg64 p;
        p.inc = (stream << 1) | 1;
        p.state = st + p.inc;
        p.step();
        return p;
    }
    // rand_core 0.6.2 SeedableRng::seed_from_u64 (PCG32 expansion to 32 seed bytes) followed by
    // Lcg128Xsl64::from_seed (state = w0|w1<<64, increment = (w2|w3<<64)|1).
    static Pcg64 seed_from_u64(uint64_t s) {
        const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
        uint32_t words[8];
        for (int i = 0; i < 8; i++) {
            s = s * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((s >> 18) ^ s) >> 27);
            uint32_t rot = (uint32_t)(s >> 59);
            words[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
        uint64_t w[4];
        for (int i = 0; i < 4; i++) w[i] = (uint64_t)words[2 * i] | ((uint64_t)words[2 * i + 1] << 32);
        Pcg64 p;
        p.state = (unsigned __int128)w[0] | ((unsigned __int128)w[1] << 64);
        p.inc = ((unsigned __int128)w[2] | ((unsigned __int128)w[3] << 64)) | 1;
        p.state = p.state + p.inc;
        p.step();
        return p;
    }
    void step() { state = state * mult() + inc; }
    uint64_t next_u64() override {
        step();
        uint32_t rot = (uint32_t)(state >> 122);
        uint64_t xsl = (uint64_t)(state >> 64) ^ (uint64_t)state;
        return (xsl >> rot) | (xsl << ((64 - rot) & 63));
    }
    double uniform01() override {
        g_stats.draws++;
        uint64_t bits = (next_u64() >> 12) | (1023ULL << 52);
        double v;
        std::memcpy(&v, &bits, 8);
        return v - 1.0;
    }
};

inline void philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct PhiloxStream : Rng {
    uint32_t key[2], pixel, sample, draw;
    uint32_t buf[4];
    uint32_t scatter_start = 0;
    void begin_scatter() override {
        if ((draw & 3) != 0) draw = (draw + 3u) & ~3u;  // the next draw refills
        scatter_start = draw;
    }
    // the scatter owns words 0-2 of its block; word 3 is the next draw of the stream (a medium draw of the scattered ray)
    void end_scatter() override {
        if (draw == scatter_start) {  // nothing drawn (Dielectric that cannot refract, a light): the block is still this scatter's
            uint32_t ctr[4] = {scatter_start >> 2, sample, pixel, 0};
            philox4x32_10(key, ctr, buf);
        }
        draw = scatter_start + 3u;
    }
    PhiloxStream(uint64_t seed, uint32_t pixel_, uint32_t sample_) : pixel(pixel_), sample(sample_), draw(0) {
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    }
    uint32_t next_u32() {
        if ((draw & 3) == 0) {
            uint32_t ctr[4] = {draw >> 2, sample, pixel, 0};
            philox4x32_10(key, ctr, buf);
        }
        return buf[(draw++) & 3];
    }
    uint64_t next_u64() override { std::fprintf(stderr, "oracle: PhiloxStream::next_u64 is not part of the render path\n"); std::abort(); }
    bool closed_form_samplers() const override { return true; }
    // the radius draw maps word w to (w>>8 + 1) * 2^-24 in (0, 1]: r = cbrt(.) is never 0 (a zero vector would make an
    // Isotropic scatter direction degenerate with probability 2^-24 instead of the rejection loop's ~2^-72)
    double uniform_0_1_closed_top() {
        g_stats.draws++;
        return ((double)(next_u32() >> 8) + 1.0) * (1.0 / 16777216.0);
    }
    double uniform01() override {
        g_stats.draws++;
        return (double)(next_u32() >> 8) * (1.0 / 16777216.0);
    }
    uint32_t uniform_24_bits() {  // the k of uniform01() = k 2^-24 (the samplers' angle: turn_cos_sin)
        g_stats.draws++;
        return next_u32() >> 8;
    }
};

// vec.rs:15-52 random helpers.  Rust evaluates call arguments left to right: x, then y, then z.
inline Vec3 vec3_random(double lo, double hi, Rng& rng) {  // vec.rs:15-17
    double a = rng.gen_range(lo, hi);
    double b = rng.gen_range(lo, hi);
    double c = rng.gen_range(lo, hi);
    return Vec3(a, b, c);
}
// Device contract (Philox mode): uniform point in the unit ball by inverse CDF from exactly three draws -
// z = 2*u1 - 1, phi = 2*pi*u2, r = cbrt(u3') - the same distribution as the reference's rejection loop, which stays
// in force for the Pcg64 mode (tests/test_oracle_samplers.py checks the two statistically and at image level).
struct PhiloxStream;
// cos and sin of 2 pi k / 2^24 for the 24-bit k of a sampler's uniform: the device contract defines them as THIS arithmetic (mlrt_kernel.hip
// turn_cos_sin, the same lines): the nearest of 64 tabulated angles, Taylor polynomials of the remainder |x| <= pi / 64, angle-sum
// formulas; every operation an exactly rounded mul or fma, so oracle and device get the very same doubles.  |error| < 7e-16 over all 2^24 inputs.
inline void turn_cos_sin(uint32_t k, double& cs, double& sn) {
    static const double kTurnTable[64][2] = {
        {0x0.0p+0, 0x1.0000000000000p+0},
        {0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1},
        {0x1.8f8b83c69a60ap-3, 0x1.f6297cff75cb0p-1},
        {0x1.294062ed59f05p-2, 0x1.e9f4156c62ddap-1},
        {0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1},
        {0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1},
        {0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1},
        {0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1},
        {0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1},
        {0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1},
        {0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c9p-1},
        {0x1.c38b2f180bdb0p-1, 0x1.e2b5d3806f63ep-2},
        {0x1.d906bcf328d46p-1, 0x1.87de2a6aea964p-2},
        {0x1.e9f4156c62ddbp-1, 0x1.294062ed59f05p-2},
        {0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60dp-3},
        {0x1.fd88da3d12525p-1, 0x1.917a6bc29b438p-4},
        {0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.fd88da3d12526p-1, -0x1.917a6bc29b42fp-4},
        {0x1.f6297cff75cb0p-1, -0x1.8f8b83c69a608p-3},
        {0x1.e9f4156c62ddbp-1, -0x1.294062ed59f02p-2},
        {0x1.d906bcf328d46p-1, -0x1.87de2a6aea962p-2},
        {0x1.c38b2f180bdb1p-1, -0x1.e2b5d3806f63cp-2},
        {0x1.a9b66290ea1a5p-1, -0x1.1c73b39ae68c6p-1},
        {0x1.8bc806b151742p-1, -0x1.44cf325091dd5p-1},
        {0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1},
        {0x1.44cf325091dd6p-1, -0x1.8bc806b151741p-1},
        {0x1.1c73b39ae68c8p-1, -0x1.a9b66290ea1a4p-1},
        {0x1.e2b5d3806f63fp-2, -0x1.c38b2f180bdb0p-1},
        {0x1.87de2a6aea965p-2, -0x1.d906bcf328d46p-1},
        {0x1.294062ed59f06p-2, -0x1.e9f4156c62ddap-1},
        {0x1.8f8b83c69a617p-3, -0x1.f6297cff75cb0p-1},
        {0x1.917a6bc29b43cp-4, -0x1.fd88da3d12525p-1},
        {0x0.0p+0, -0x1.0000000000000p+0},
        {-0x1.917a6bc29b42bp-4, -0x1.fd88da3d12526p-1},
        {-0x1.8f8b83c69a60ep-3, -0x1.f6297cff75cb0p-1},
        {-0x1.294062ed59f01p-2, -0x1.e9f4156c62ddbp-1},
        {-0x1.87de2a6aea961p-2, -0x1.d906bcf328d47p-1},
        {-0x1.e2b5d3806f63bp-2, -0x1.c38b2f180bdb1p-1},
        {-0x1.1c73b39ae68c6p-1, -0x1.a9b66290ea1a5p-1},
        {-0x1.44cf325091dd4p-1, -0x1.8bc806b151742p-1},
        {-0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1},
        {-0x1.8bc806b15173ep-1, -0x1.44cf325091ddap-1},
        {-0x1.a9b66290ea1a3p-1, -0x1.1c73b39ae68c8p-1},
        {-0x1.c38b2f180bdb0p-1, -0x1.e2b5d3806f63fp-2},
        {-0x1.d906bcf328d44p-1, -0x1.87de2a6aea96dp-2},
        {-0x1.e9f4156c62ddap-1, -0x1.294062ed59f07p-2},
        {-0x1.f6297cff75cafp-1, -0x1.8f8b83c69a619p-3},
        {-0x1.fd88da3d12526p-1, -0x1.917a6bc29b421p-4},
        {-0x1.0000000000000p+0, 0x0.0p+0},
        {-0x1.fd88da3d12526p-1, 0x1.917a6bc29b407p-4},
        {-0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60cp-3},
        {-0x1.e9f4156c62ddbp-1, 0x1.294062ed59f00p-2},
        {-0x1.d906bcf328d45p-1, 0x1.87de2a6aea967p-2},
        {-0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63ap-2},
        {-0x1.a9b66290ea1a5p-1, 0x1.1c73b39ae68c5p-1},
        {-0x1.8bc806b151740p-1, 0x1.44cf325091dd7p-1},
        {-0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1},
        {-0x1.44cf325091ddap-1, 0x1.8bc806b15173ep-1},
        {-0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1},
        {-0x1.e2b5d3806f640p-2, 0x1.c38b2f180bdafp-1},
        {-0x1.87de2a6aea96ep-2, 0x1.d906bcf328d44p-1},
        {-0x1.294062ed59f08p-2, 0x1.e9f4156c62ddap-1},
        {-0x1.8f8b83c69a61bp-3, 0x1.f6297cff75cafp-1},
        {-0x1.917a6bc29b425p-4, 0x1.fd88da3d12526p-1}
    };
    const uint32_t j = (k + (1u << 17)) >> 18;
    const int dk = (int)k - (int)(j << 18);
    const double x = (double)dk * 0x1.921fb54442d18p-22;  // 2 pi / 2^24
    const double x2 = x * x;
    const double ps = std::fma(x2, std::fma(x2, std::fma(x2, -0x1.a01a01a01a01ap-13, 0x1.1111111111111p-7), -0x1.5555555555555p-3), 1.0);
    const double sx = x * ps;
    const double cx = std::fma(x2, std::fma(x2, std::fma(x2, std::fma(x2, 0x1.a01a01a01a01ap-16, -0x1.6c16c16c16c17p-10), 0x1.5555555555555p-5), -0.5), 1.0);
    const double ta_sin = kTurnTable[j & 63u][0], ta_cos = kTurnTable[j & 63u][1];
    sn = std::fma(ta_sin, cx, ta_cos * sx);
    cs = std::fma(ta_cos, cx, -(ta_sin * sx));
}
// cbrt(n 2^-24), n = 1 .. 2^24: the radius of the contract's unit-ball sample, defined as this arithmetic (mlrt_kernel.hip cbrt24, the
// same lines): relative error <= 5.5e-16 over all inputs, exact at both ends.
inline double cbrt24(uint32_t n) {
    const int e = 31 - __builtin_clz(n);
    double f = (double)n;
    uint64_t b;
    std::memcpy(&b, &f, 8); b -= (uint64_t)e << 52; std::memcpy(&f, &b, 8);  // n 2^-e in [1, 2)
    const int q3 = (e * 11) >> 5;  // e / 3 for e <= 24
    const int rem = e - 3 * q3;
    double z = std::fma(f, std::fma(f, std::fma(f, -0x1.9239a09c3fa5fp-5, 0x1.407a7b055bc3ep-2), -0x1.9a4cbd01c7c23p-1), 0x1.897a11517ead3p+0);
    for (int it = 0; it < 2; it++) {
        const double z2 = z * z;
        const double h = std::fma(-f * z2, z, 1.0);
        const double p = std::fma(h, 0x1.c71c71c71c71cp-3, 0x1.5555555555555p-2);
        z = std::fma(z * h, p, z);
    }
    double y = f * z * z;
    y *= rem == 0 ? 1.0 : (rem == 1 ? 0x1.428a2f98d728bp+0 : 0x1.965fea53d6e3dp+0);  // cbrt(2), cbrt(4)
    std::memcpy(&b, &y, 8); b += (uint64_t)(int64_t)(q3 - 8) << 52; std::memcpy(&y, &b, 8);  // 2^(q - 8)
    return y;
}
Vec3 closed_form_unit_sphere(Rng& rng);
Vec3 closed_form_unit_disk(Rng& rng);
inline Vec3 random_in_unit_sphere(Rng& rng) {  // vec.rs:23-30
    if (rng.closed_form_samplers()) return closed_form_unit_sphere(rng);
    for (;;) {
        Vec3 p = vec3_random(-1.0, 1.0, rng);
        if (p.length_squared() < 1.0) return p;
    }
}
inline Vec3 random_in_hemisphere(const Vec3& normal, Rng& rng) {  // vec.rs:36-43
    Vec3 s = random_in_unit_sphere(rng);
    if (normal.dot(s) > 0.0) return s;
    return -s;
}
inline Vec3 random_in_unit_disk(Rng& rng) {  // vec.rs:45-52
    if (rng.closed_form_samplers()) return closed_form_unit_disk(rng);
    for (;;) {
        double a = rng.gen_range(-1.0, 1.0);
        double b = rng.gen_range(-1.0, 1.0);
        Vec3 p(a, b, 0.0);
        if (p.length_squared() < 1.0) return p;
    }
}

// ------------------------------------------------------------------------------------------------
// textures.rs, image_texture.rs
// ------------------------------------------------------------------------------------------------
struct Texture {  // textures.rs:4-6
    virtual ~Texture() {}
    virtual Color value(double u, double v, const Point3& p) const = 0;
};
struct SolidColor : Texture {  // textures.rs:8-26
    Color color;
    explicit SolidColor(const Color& c) : color(c) {}
    SolidColor(double r, double g, double b) : color(r, g, b) {}
    Color value(double, double, const Point3&) const override { return color; }
};
struct Checker : Texture {  // textures.rs:28-49
    std::shared_ptr<Texture> odd, even;
    Checker(std::shared_ptr<Texture> o, std::shared_ptr<Texture> e) : odd(o), even(e) {}
    Color value(double u, double v, const Point3& p) const override {
        double sines = std::sin(5.0 * p.x()) * std::sin(5.0 * p.y()) * std::sin(5.0 * p.z());
        if (sines < 0.0) return odd->value(u, v, p);
        return even->value(u, v, p);
    }
};
struct Perlin {  // textures.rs:51-149
    static const int POINT_COUNT = 1024;
    Vec3 ranvec[POINT_COUNT];
    uint64_t perm_x[POINT_COUNT], perm_y[POINT_COUNT], perm_z[POINT_COUNT];
    int turbulence_depth;
    Perlin() : turbulence_depth(7) {}  // tables filled in by the caller (oracle_scene.cpp: a scene that carries its tables)
    explicit Perlin(Rng& rng) {  // textures.rs:62-74
        for (int i = 0; i < POINT_COUNT; i++) ranvec[i] = vec3_random(-1.0, 1.0, rng).unit();
        permute(rng, perm_x);
        permute(rng, perm_y);
        permute(rng, perm_z);
        turbulence_depth = 7;
    }
    static void permute(Rng& rng, uint64_t* result) {  // textures.rs:136-148
        for (int i = 0; i < POINT_COUNT; i++) result[i] = (uint64_t)i;
        for (int i = POINT_COUNT - 1; i >= 1; i--) {
            uint64_t j = rng.gen_range_usize(0, (uint64_t)i);
            uint64_t tmp = result[i];
            result[i] = result[j];
            result[j] = tmp;
        }
    }
    double turbulence(const Point3& p) const {  // textures.rs:76-88
        double accum = 0.0;
        Point3 temp_p = p;
        double weight = 1.0;
        for (int i = 0; i < turbulence_depth; i++) {
            accum += weight * noise(temp_p);
            weight *= 0.5;
            temp_p = 2.0 * temp_p;
        }
        return std::fabs(accum);
    }
    double noise(const Point3& p) const {  // textures.rs:90-113
        g_stats.noise_evals++;
        double u = p.x() - std::floor(p.x());
        double v = p.y() - std::floor(p.y());
        double w = p.z() - std::floor(p.z());
        int64_t i = (int64_t)std::floor(p.x());
        int64_t j = (int64_t)std::floor(p.y());
        int64_t k = (int64_t)std::floor(p.z());
        Vec3 c[2][2][2];
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++) {
                    // rem_euclid(1024) == & 1023 for two's complement
                    uint64_t ii = (uint64_t)((i + di) & 1023), jj = (uint64_t)((j + dj) & 1023), kk = (uint64_t)((k + dk) & 1023);
                    c[di][dj][dk] = ranvec[perm_x[ii] ^ perm_y[jj] ^ perm_z[kk]];
                }
        return trilinear_interpolation(c, u, v, w);
    }
    static double trilinear_interpolation(const Vec3 c[2][2][2], double u, double v, double w) {  // textures.rs:115-134
        double uu = u * u * (3.0 - 2.0 * u);
        double vv = v * v * (3.0 - 2.0 * v);
        double ww = w * w * (3.0 - 2.0 * w);
        double accum = 0.0;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    Vec3 weight(u - (double)i, v - (double)j, w - (double)k);
                    accum += ((double)i * uu + (double)(1 - i) * (1.0 - uu)) * ((double)j * vv + (double)(1 - j) * (1.0 - vv)) *
                             ((double)k * ww + (double)(1 - k) * (1.0 - ww)) * weight.dot(c[i][j][k]);
                }
        return accum;
    }
};
struct NoiseTexture : Texture {  // textures.rs:151-167
    std::shared_ptr<Perlin> noise;
    double scale;
    NoiseTexture(double s, Rng& rng) : noise(std::make_shared<Perlin>(rng)), scale(s) {}
    NoiseTexture(std::shared_ptr<Perlin> tables, double s) : noise(tables), scale(s) {}
    Color value(double, double, const Point3& p) const override {
        // Color::ONE * 0.5 * (1.0 + sin(scale*p.z + 10*turb(scale*p)))  (left-assoc: (ONE*0.5)*x)
        double x = 1.0 + std::sin(scale * p.z() + 10.0 * noise->turbulence(scale * p));
        return (Color(1.0, 1.0, 1.0) * 0.5) * x;
    }
};
struct ImageData {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;
};
struct ImageTexture : Texture {  // image_texture.rs:16-28
    std::shared_ptr<ImageData> image;
    explicit ImageTexture(std::shared_ptr<ImageData> im) : image(im) {}
    Color value(double u, double v, const Point3&) const override {
        g_stats.texel_fetches++;
        u = std::min(std::max(u, 0.0), 1.0);
        v = std::min(std::max(1.0 - v, 0.0), 1.0);
        uint32_t width = (uint32_t)image->width, height = (uint32_t)image->height;
        uint32_t i = (uint32_t)(u * (double)width);
        uint32_t j = (uint32_t)(v * (double)height);
        i = std::min(i, width - 1);
        j = std::min(j, height - 1);
        const uint8_t* px = &image->rgb[((size_t)j * width + i) * 3];
        return Color((double)px[0] / 255.0, (double)px[1] / 255.0, (double)px[2] / 255.0);
    }
};

// ------------------------------------------------------------------------------------------------
// hittable.rs, materials.rs
// ------------------------------------------------------------------------------------------------
struct Material;
struct Hit {  // hittable.rs:7-15
    Point3 p;
    Vec3 normal;
    double t = 0, u = 0, v = 0;
    bool front_face = false;
    const Material* material = nullptr;
    static Hit new_with_face_normal(const Point3& p, double t, double u, double v, const Vec3& outward_normal, const Ray& r,
                                    const Material* m) {  // hittable.rs:18-30
        Hit h;
        h.front_face = outward_normal.dot(r.dir) < 0.0;
        h.normal = h.front_face ? outward_normal : -outward_normal;
        h.p = p; h.t = t; h.u = u; h.v = v; h.material = m;
        return h;
    }
};

enum MatKind { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3, MAT_ISOTROPIC = 4 };

struct Material {  // materials.rs:6-12
    virtual ~Material() {}
    virtual bool scatter(const Ray& ray, const Hit& h, Rng& rng, Color& attenuation, Ray& scattered) const = 0;
    virtual Color emit(double, double, const Point3&) const { return Color(0, 0, 0); }
    // kind, then up to 4 parameters (albedo rgb / fuzz / ior); for textured materials the solid colour if any.
    virtual void describe(double out[5]) const = 0;
};
inline void describe_texture(const Texture* t, double out[5]) {
    if (const SolidColor* s = dynamic_cast<const SolidColor*>(t)) { out[1] = s->color.e[0]; out[2] = s->color.e[1]; out[3] = s->color.e[2]; out[4] = 0; }
    else if (dynamic_cast<const NoiseTexture*>(t)) { out[1] = out[2] = out[3] = -1; out[4] = 2; }
    else if (dynamic_cast<const ImageTexture*>(t)) { out[1] = out[2] = out[3] = -1; out[4] = 1; }
    else { out[1] = out[2] = out[3] = -1; out[4] = 3; }
}
struct Lambertian : Material {  // materials.rs:14-34
    std::shared_ptr<Texture> albedo;
    explicit Lambertian(std::shared_ptr<Texture> a) : albedo(a) {}
    bool scatter(const Ray&, const Hit& h, Rng& rng, Color& att, Ray& scattered) const override {
        Vec3 scatter_direction = h.normal + random_in_hemisphere(h.normal, rng);  // quirk Q1
        if (scatter_direction.near_zero()) scatter_direction = h.normal;
        att = albedo->value(h.u, h.v, h.p);
        scattered = Ray(h.p, scatter_direction);
        return true;
    }
    void describe(double out[5]) const override { out[0] = MAT_LAMBERTIAN; describe_texture(albedo.get(), out); }
};
inline Vec3 reflect(const Vec3& v, const Vec3& n) { return v - 2.0 * v.dot(n) * n; }  // materials.rs:47-49
struct Metal : Material {  // materials.rs:36-61
    Color albedo;
    double fuzz;
    Metal(const Color& a, double f) : albedo(a), fuzz(f) {}
    bool scatter(const Ray& ray, const Hit& h, Rng& rng, Color& att, Ray& scattered) const override {
        Vec3 reflected = reflect(ray.dir.unit(), h.normal);
        scattered = Ray(h.p, reflected + fuzz * random_in_unit_sphere(rng));  // quirk Q7: always draws
        if (scattered.dir.dot(h.normal) > 0.0) { att = albedo; return true; }
        return false;
    }
    void describe(double out[5]) const override { out[0] = MAT_METAL; out[1] = albedo.e[0]; out[2] = albedo.e[1]; out[3] = albedo.e[2]; out[4] = fuzz; }
};
inline Vec3 refract(const Vec3& uv, const Vec3& n, double etai_over_etat) {  // materials.rs:63-68
    double cos_theta = std::min((-uv).dot(n), 1.0);
    Vec3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    Vec3 r_out_parallel = -std::sqrt(std::fabs(1.0 - r_out_perp.length_squared())) * n;
    return r_out_perp + r_out_parallel;
}
inline double powi5(double x) {
    // f64::powi(5) lowers to llvm.powi -> compiler-rt __powidf2: square-and-multiply, b=5: r=x; x2; x4; r=x*x4
    double a = x, r = 1.0;
    int b = 5;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}
inline double reflectance(double cos_theta, double refraction_ratio) {  // materials.rs:81-86
    double r0 = (1.0 - refraction_ratio) / (1.0 + refraction_ratio);
    double r0_sq = r0 * r0;
    return r0_sq + (1.0 - r0_sq) * powi5(1.0 - cos_theta);
}
struct Dielectric : Material {  // materials.rs:70-106
    double index_of_refraction;
    explicit Dielectric(double ior) : index_of_refraction(ior) {}
    bool scatter(const Ray& ray, const Hit& h, Rng& rng, Color& att, Ray& scattered) const override {
        att = Color(1.0, 1.0, 1.0);
        double refraction_ratio = !h.front_face ? index_of_refraction : 1.0 / index_of_refraction;
        Vec3 unit_direction = ray.dir.unit();
        double cos_theta = std::min(h.normal.dot(-unit_direction), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        Vec3 direction;
        // quirk Q8: the draw is short-circuited when refraction is impossible
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng.gen_range(0.0, 1.0))
            direction = reflect(unit_direction, h.normal);
        else
            direction = refract(unit_direction, h.normal, refraction_ratio);
        scattered = Ray(h.p, direction);
        return true;
    }
    void describe(double out[5]) const override { out[0] = MAT_DIELECTRIC; out[1] = index_of_refraction; out[2] = out[3] = out[4] = 0; }
};
struct DiffuseLight : Material {  // materials.rs:108-127
    std::shared_ptr<Texture> texture;
    explicit DiffuseLight(std::shared_ptr<Texture> t) : texture(t) {}
    bool scatter(const Ray&, const Hit&, Rng&, Color&, Ray&) const override { return false; }
    Color emit(double u, double v, const Point3& p) const override { return texture->value(u, v, p); }
    void describe(double out[5]) const override { out[0] = MAT_DIFFUSE_LIGHT; describe_texture(texture.get(), out); }
};
struct Isotropic : Material {  // volumes.rs:67-83
    std::shared_ptr<Texture> albedo;
    explicit Isotropic(std::shared_ptr<Texture> a) : albedo(a) {}
    bool scatter(const Ray&, const Hit& h, Rng& rng, Color& att, Ray& scattered) const override {
        scattered = Ray(h.p, random_in_unit_sphere(rng));  // un-normalised (quirk Q9)
        att = albedo->value(h.u, h.v, h.p);
        return true;
    }
    void describe(double out[5]) const override { out[0] = MAT_ISOTROPIC; describe_texture(albedo.get(), out); }
};

struct Hittable {  // hittable.rs:33-35
    virtual ~Hittable() {}
    virtual bool hit(const Ray& r, double t_min, double t_max, Rng& rng, Hit& out) const = 0;
};
struct HittableList : Hittable {  // hittable.rs:37-69
    std::vector<std::shared_ptr<Hittable>> contents;
    void add(std::shared_ptr<Hittable> h) { contents.push_back(h); }
    bool hit(const Ray& r, double t_min, double t_max, Rng& rng, Hit& out) const override {
        bool any = false;
        double closest_so_far = t_max;
        Hit h;
        for (const auto& o : contents) {
            if (o->hit(r, t_min, closest_so_far, rng, h)) {
                closest_so_far = h.t;
                out = h;
                any = true;
            }
        }
        return any;
    }
};

// ------------------------------------------------------------------------------------------------
// bhv.rs
// ------------------------------------------------------------------------------------------------
struct AABB {  // bhv.rs:10-53
    Point3 minimum, maximum;
    AABB() {}
    AABB(const Point3& a, const Point3& b) {  // bhv.rs:16-20
        for (int i = 0; i < 3; i++) { minimum.e[i] = std::fmin(a.e[i], b.e[i]); maximum.e[i] = std::fmax(a.e[i], b.e[i]); }
    }
    bool hit(const Ray& r, double tmin, double tmax) const {  // bhv.rs:29-42 (quirk Q5)
        g_stats.aabb_tests++;
        for (int a = 0; a < 3; a++) {
            double t0 = (minimum.e[a] - r.orig.e[a]) / r.dir.e[a];
            double t1 = (maximum.e[a] - r.orig.e[a]) / r.dir.e[a];
            tmin = std::fmax(std::fmin(t0, t1), tmin);  // f64::min/max ignore NaN like fmin/fmax
            tmax = std::fmin(std::fmax(t0, t1), tmax);
            if (tmax <= tmin) return false;
        }
        return true;
    }
    AABB surround(const AABB& o) const {  // bhv.rs:44-53
        Point3 mn, mx;
        for (int a = 0; a < 3; a++) { mn.e[a] = std::fmin(minimum.e[a], o.minimum.e[a]); mx.e[a] = std::fmax(maximum.e[a], o.maximum.e[a]); }
        return AABB(mn, mx);
    }
};
struct Bounded : Hittable {  // bhv.rs:61-63
    virtual AABB bounding_box() const = 0;
};
struct Empty : Bounded {  // shapes.rs:8-23
    bool hit(const Ray&, double, double, Rng&, Hit&) const override { return false; }
    AABB bounding_box() const override { return AABB(Point3(0, 0, 0), Point3(0, 0, 0)); }
};
struct BvhNode {  // bhv.rs:109-112
    bool leaf = true;
    std::shared_ptr<Bounded> shape;  // Leaf
    AABB bounds;                     // Inner
    std::unique_ptr<BvhNode> left, right;
    AABB bounding_box() const { return leaf ? shape->bounding_box() : bounds; }  // bhv.rs:115-120
    // bhv.rs:122-145: random axis, STABLE sort by bbox-min on that axis, split at len/2, pre-order recursion.
    static std::unique_ptr<BvhNode> build(std::shared_ptr<Bounded>* shapes, size_t n, Rng& rng) {
        std::unique_ptr<BvhNode> node(new BvhNode());
        if (n == 0) { node->leaf = true; node->shape = std::make_shared<Empty>(); return node; }
        if (n == 1) { node->leaf = true; node->shape = shapes[0]; return node; }
        size_t axis = (size_t)rng.gen_range_usize(0, 3);
        std::stable_sort(shapes, shapes + n, [axis](const std::shared_ptr<Bounded>& a, const std::shared_ptr<Bounded>& b) {
            // partial_cmp: None (NaN) => Equal; Less => a before b
            return a->bounding_box().minimum.e[axis] < b->bounding_box().minimum.e[axis];
        });
        size_t half = n / 2;
        node->leaf = false;
        node->left = build(shapes, half, rng);
        node->right = build(shapes + half, n - half, rng);
        node->bounds = node->left->bounding_box().surround(node->right->bounding_box());
        return node;
    }
    bool hit(const Ray& r, double tmin, double tmax, Rng& rng, Hit& out) const {  // bhv.rs:147-165 (quirk Q4)
        if (leaf) return shape->hit(r, tmin, tmax, rng, out);
        if (!bounds.hit(r, tmin, tmax)) return false;
        Hit hl;
        bool got_left = left->hit(r, tmin, tmax, rng, hl);
        double tmax_for_right = got_left ? hl.t : tmax;
        Hit hr;
        if (right->hit(r, tmin, tmax_for_right, rng, hr)) { out = hr; return true; }
        if (got_left) { out = hl; return true; }
        return false;
    }
};
struct SceneBuilder {  // bhv.rs:65-82
    std::vector<std::shared_ptr<Bounded>> contents;
    SceneBuilder& add(std::shared_ptr<Bounded> b) { contents.push_back(b); return *this; }
};
struct BHV : Bounded {  // bhv.rs:84-107
    std::unique_ptr<BvhNode> root;
    BHV(SceneBuilder& scene, Rng& rng) {
        root = BvhNode::build(scene.contents.data(), scene.contents.size(), rng);
        scene.contents.clear();
    }
    bool hit(const Ray& r, double tmin, double tmax, Rng& rng, Hit& out) const override { return root->hit(r, tmin, tmax, rng, out); }
    AABB bounding_box() const override { return root->bounding_box(); }
};

// ------------------------------------------------------------------------------------------------
// shapes.rs, aarects.rs
// ------------------------------------------------------------------------------------------------
inline void sphere_uv(const Vec3& normal, double& u, double& v) {  // shapes.rs:44-56
    double theta = std::acos(-normal.y());
    double phi = std::atan2(-normal.z(), normal.x()) + PI;
    u = phi / (2.0 * PI);
    v = theta / PI;
}
struct Sphere : Bounded {  // shapes.rs:25-89
    Point3 center;
    double radius;
    std::shared_ptr<Material> material;
    Sphere(const Point3& c, double r, std::shared_ptr<Material> m) : center(c), radius(r), material(m) {}
    bool hit(const Ray& r, double t_min, double t_max, Rng&, Hit& out) const override {  // shapes.rs:58-82
        g_stats.sphere_tests++;
        Vec3 oc = r.orig - center;
        double a = r.dir.length_squared();
        double half_b = oc.dot(r.dir);
        double c = oc.length_squared() - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0) return false;
        double sqrtd = std::sqrt(discriminant);
        double root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) return false;
        }
        double t = root;
        Point3 p = r.at(t);
        Vec3 normal = (p - center) / radius;
        double u, v;
        sphere_uv(normal, u, v);  // quirk Q11: computed for every accepted intersection
        out = Hit::new_with_face_normal(p, t, u, v, normal, r, material.get());
        return true;
    }
    AABB bounding_box() const override {  // shapes.rs:84-89
        Vec3 rad_v(radius, radius, radius);
        return AABB(center - rad_v, center + rad_v);
    }
};
struct AARect {  // aarects.rs:19-78
    int a0, a1, aplane;
    double a0_v0, a0_v1, a1_v0, a1_v1, aplane_v;
    AARect() {}
    AARect(int a0_, double v00, double v01, int a1_, double v10, double v11, double plane_v) {  // aarects.rs:31-43
        a0 = a0_; a1 = a1_;
        aplane = 3 - a0 - a1;  // other(): (X,Y)->Z, (X,Z)->Y, (Y,Z)->X
        a0_v0 = std::fmin(v00, v01);
        a0_v1 = std::fmax(v01, v00);
        a1_v0 = std::fmin(v10, v10);  // quirk Q2: a1_v0.min(a1_v0)
        a1_v1 = std::fmax(v11, v10);
        aplane_v = plane_v;
    }
    bool hit(const Ray& r, double tmin, double tmax, const Material* material, Hit& out) const {  // aarects.rs:45-64
        g_stats.rect_tests++;
        double t = (aplane_v - r.orig.e[aplane]) / r.dir.e[aplane];
        if (t < tmin || t > tmax) return false;
        double a0_v = r.orig.e[a0] + t * r.dir.e[a0];
        double a1_v = r.orig.e[a1] + t * r.dir.e[a1];
        if (a0_v < a0_v0 || a0_v > a0_v1 || a1_v < a1_v0 || a1_v > a1_v1) return false;
        double u = (a0_v - a0_v0) / (a0_v1 - a0_v0);
        double v = (a1_v - a1_v0) / (a1_v1 - a1_v0);
        Vec3 outward_normal(0, 0, 0);
        outward_normal.e[aplane] = 1.0;
        out = Hit::new_with_face_normal(r.at(t), t, u, v, outward_normal, r, material);
        return true;
    }
    AABB bounding_box() const {  // aarects.rs:66-77 (quirk Q2: degenerate; never used by a BVH in the reference)
        Point3 mn(0, 0, 0), mx(0, 0, 0);
        mn.e[a0] = a0_v0; mn.e[a1] = a1_v0; mn.e[aplane] = aplane_v - 0.001;
        mx.e[a0] = a0_v0; mx.e[a1] = a1_v0; mx.e[aplane] = aplane_v + 0.001;
        return AABB(mn, mx);
    }
};
struct RectShape : Bounded {  // shapes.rs:91-164 (XYRect / XZRect / YZRect)
    AARect r;
    std::shared_ptr<Material> material;
    RectShape(const AARect& rr, std::shared_ptr<Material> m) : r(rr), material(m) {}
    bool hit(const Ray& ray, double tmin, double tmax, Rng&, Hit& out) const override { return r.hit(ray, tmin, tmax, material.get(), out); }
    AABB bounding_box() const override { return r.bounding_box(); }
};
inline std::shared_ptr<RectShape> xy_rect(double x0, double x1, double y0, double y1, double z, std::shared_ptr<Material> m) {
    return std::make_shared<RectShape>(AARect(0, x0, x1, 1, y0, y1, z), m);  // shapes.rs:98-101
}
inline std::shared_ptr<RectShape> xz_rect(double x0, double x1, double z0, double z1, double y, std::shared_ptr<Material> m) {
    return std::make_shared<RectShape>(AARect(0, x0, x1, 2, z0, z1, y), m);  // shapes.rs:123-126
}
inline std::shared_ptr<RectShape> yz_rect(double y0, double y1, double z0, double z1, double x, std::shared_ptr<Material> m) {
    return std::make_shared<RectShape>(AARect(1, y0, y1, 2, z0, z1, x), m);  // shapes.rs:148-151
}
struct Block : Bounded {  // shapes.rs:166-198
    Point3 min, max;
    HittableList sides;
    std::shared_ptr<Material> material;
    Block(const Point3& p0, const Point3& p1, std::shared_ptr<Material> m) : min(p0), max(p1), material(m) {
        sides.add(xy_rect(p0.x(), p1.x(), p0.y(), p1.y(), p1.z(), m));
        sides.add(xy_rect(p0.x(), p1.x(), p0.y(), p1.y(), p0.z(), m));
        sides.add(xz_rect(p0.x(), p1.x(), p0.z(), p1.z(), p0.y(), m));
        sides.add(xz_rect(p0.x(), p1.x(), p0.z(), p1.z(), p1.y(), m));
        sides.add(yz_rect(p0.y(), p1.y(), p0.z(), p1.z(), p0.x(), m));
        sides.add(yz_rect(p0.y(), p1.y(), p0.z(), p1.z(), p1.x(), m));
    }
    bool hit(const Ray& r, double tmin, double tmax, Rng& rng, Hit& out) const override { return sides.hit(r, tmin, tmax, rng, out); }
    AABB bounding_box() const override { return AABB(min, max); }
};

// ------------------------------------------------------------------------------------------------
// transforms.rs
// ------------------------------------------------------------------------------------------------
struct Translate : Bounded {  // transforms.rs:20-49
    std::shared_ptr<Bounded> original;
    Vec3 offset;
    Translate(const Vec3& off, std::shared_ptr<Bounded> o) : original(o), offset(off) {}
    bool hit(const Ray& r, double t_min, double t_max, Rng& rng, Hit& out) const override {
        Ray moved_r(r.orig - offset, r.dir);
        Hit h;
        if (!original->hit(moved_r, t_min, t_max, rng, h)) return false;
        out = Hit::new_with_face_normal(h.p + offset, h.t, h.u, h.v, h.normal, moved_r, h.material);
        return true;
    }
    AABB bounding_box() const override {
        AABB b = original->bounding_box();
        return AABB(b.minimum + offset, b.maximum + offset);
    }
};
struct Rotate : Bounded {  // transforms.rs:51-148
    int a1;
    double sin_theta, cos_theta;
    AABB bbox;
    std::shared_ptr<Bounded> original;
    Rotate(int axis, double angle, std::shared_ptr<Bounded> o) : original(o) {  // transforms.rs:59-97
        a1 = axis;
        int a2 = (a1 + 1) % 3, a0 = (a1 + 2) % 3;
        double theta = angle * PI / 180.0;
        sin_theta = std::sin(theta);
        cos_theta = std::cos(theta);
        AABB b = original->bounding_box();
        Point3 mn(-INF, -INF, -INF), mx(-INF, -INF, -INF);  // quirk Q3: min starts at -inf
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    double a0_v = i == 1 ? b.maximum.e[a0] : b.minimum.e[a0];
                    double a1_v = j == 1 ? b.maximum.e[a1] : b.minimum.e[a1];
                    double a2_v = k == 1 ? b.maximum.e[a2] : b.minimum.e[a2];
                    double new_a0_v = cos_theta * a0_v + sin_theta * a2_v;
                    double new_a2_v = -sin_theta * a0_v + cos_theta * a2_v;
                    double tester[3];
                    tester[a0] = new_a0_v; tester[a1] = a1_v; tester[a2] = new_a2_v;
                    for (int c = 0; c < 3; c++) { mn.e[c] = std::fmin(mn.e[c], tester[c]); mx.e[c] = std::fmax(mx.e[c], tester[c]); }
                }
        bbox = AABB(mn, mx);
    }
    int a0() const { return (a1 + 2) % 3; }
    int a2() const { return (a1 + 1) % 3; }
    Vec3 rotate_back(const Vec3& v) const {  // transforms.rs:106-114
        Vec3 r = v;
        r.e[a0()] = cos_theta * v.e[a0()] - sin_theta * v.e[a2()];
        r.e[a2()] = sin_theta * v.e[a0()] + cos_theta * v.e[a2()];
        return r;
    }
    Vec3 rotate(const Vec3& v) const {  // transforms.rs:116-124
        Vec3 r = v;
        r.e[a0()] = cos_theta * v.e[a0()] + sin_theta * v.e[a2()];
        r.e[a2()] = -sin_theta * v.e[a0()] + cos_theta * v.e[a2()];
        return r;
    }
    bool hit(const Ray& r, double t_min, double t_max, Rng& rng, Hit& out) const override {  // transforms.rs:127-142 (quirk Q10)
        Ray rotated_r(rotate_back(r.orig), rotate_back(r.dir));
        Hit h;
        if (!original->hit(rotated_r, t_min, t_max, rng, h)) return false;
        Point3 p = rotate(h.p);
        Vec3 normal = rotate(h.normal);
        out = Hit::new_with_face_normal(p, h.t, h.u, h.v, normal, rotated_r, h.material);
        return true;
    }
    AABB bounding_box() const override { return bbox; }
};

// ------------------------------------------------------------------------------------------------
// volumes.rs
// ------------------------------------------------------------------------------------------------
struct ConstantMedium : Hittable {  // volumes.rs:7-65
    std::shared_ptr<Hittable> boundary;
    std::shared_ptr<Isotropic> phase_function;
    double neg_inv_density;
    ConstantMedium(std::shared_ptr<Hittable> b, double d, std::shared_ptr<Texture> tex)
        : boundary(b), phase_function(std::make_shared<Isotropic>(tex)), neg_inv_density(-1.0 / d) {}
    bool hit(const Ray& r, double t_min, double t_max, Rng& rng, Hit& out) const override {  // volumes.rs:26-65 (quirk Q9)
        g_stats.medium_tests++;
        Hit h1, h2;
        if (!boundary->hit(r, -INF, INF, rng, h1)) return false;
        if (!boundary->hit(r, h1.t + 0.001, INF, rng, h2)) return false;
        h1.t = std::fmax(h1.t, t_min);
        h2.t = std::fmin(h2.t, t_max);
        if (h1.t >= h2.t) return false;
        h1.t = std::fmax(h1.t, 0.0);
        double ray_scale = r.dir.length();
        double distance_inside_boundary = (h2.t - h1.t) * ray_scale;
        double hit_distance = neg_inv_density * std::log(rng.gen_range(0.0, 1.0));
        if (hit_distance > distance_inside_boundary) return false;
        double t = h1.t + hit_distance / ray_scale;
        out.p = r.at(t);
        out.t = t; out.u = 0.0; out.v = 0.0;
        out.normal = Vec3(1.0, 0.0, 0.0);
        out.front_face = true;
        out.material = phase_function.get();
        return true;
    }
};

// ------------------------------------------------------------------------------------------------
// camera.rs, raytrace.rs
// ------------------------------------------------------------------------------------------------
struct Camera {  // camera.rs:3-12
    Point3 origin, lower_left_corner;
    Vec3 horizontal, vertical, u, v;
    double lens_radius;
    Camera() : lens_radius(0) {}
    Camera(const Point3& lookfrom, const Point3& lookat, const Vec3& vup, double vfov, double aspect_ratio, double aperture,
           double focus_dist) {  // camera.rs:15-38
        double theta = vfov * PI / 180.0;
        double h = std::tan(theta / 2.0);
        double viewport_height = 2.0 * h;
        double viewport_width = aspect_ratio * viewport_height;
        Vec3 w = (lookfrom - lookat).unit();
        u = vup.cross(w).unit();
        v = w.cross(u);
        origin = lookfrom;
        horizontal = focus_dist * viewport_width * u;
        vertical = focus_dist * viewport_height * v;
        lower_left_corner = origin - horizontal / 2.0 - vertical / 2.0 - focus_dist * w;
        lens_radius = aperture / 2.0;
    }
    Ray get_ray(double s, double t, Rng& rng) const {  // camera.rs:40-48 (quirk Q6: always draws)
        Vec3 rd = lens_radius * random_in_unit_disk(rng);
        Vec3 offset = u * rd.x() + v * rd.y();
        return Ray(origin + offset, lower_left_corner + s * horizontal + t * vertical - origin - offset);
    }
};

enum BackgroundKind { BG_BLACK = 0, BG_GRADIENT = 1 };
struct Background {  // raytrace.rs:8-48
    int kind = BG_GRADIENT;
    Color top = Color(0.5, 0.7, 1.0), bottom = Color(1.0, 1.0, 1.0);  // GradientBackground::default(): new(blueish, white)
    Color color(const Ray& ray) const {
        if (kind == BG_BLACK) return Color(0, 0, 0);
        Vec3 unit_direction = ray.dir.unit();
        double t = 0.5 * (unit_direction.y() + 1.0);
        return (1.0 - t) * bottom + t * top;
    }
};

inline void to_rgb(const Color& color, int samples_per_pixel, int32_t out[3]) {  // raytrace.rs:59-68
    double scale = 1.0 / (double)samples_per_pixel;
    for (int c = 0; c < 3; c++) {
        double x = std::sqrt(color.e[c] * scale);
        // f64::clamp keeps NaN; `NaN as i32` == 0; `as i32` truncates toward zero and saturates
        double y;
        if (x != x) y = x;
        else y = x < 0.0 ? 0.0 : (x > 0.99999999 ? 0.99999999 : x);
        double z = 255.999 * y;
        out[c] = (z != z) ? 0 : (int32_t)z;
    }
}

inline Color trace_internal(const Ray& ray, const Hittable& world, const Background& bg, int depth, Rng& rng) {  // raytrace.rs:79-102
    if (depth <= 0) return Color(0, 0, 0);
    g_stats.rays++;
    Hit h;
    if (world.hit(ray, 0.001, INF, rng, h)) {
        Color attenuation;
        Ray scattered;
        rng.begin_scatter();
        const bool scatters = h.material->scatter(ray, h, rng, attenuation, scattered);
        rng.end_scatter();
        if (scatters) {
            g_stats.scatters++;
            return attenuation * trace_internal(scattered, world, bg, depth - 1, rng);
        }
        return h.material->emit(h.u, h.v, h.p);
    }
    return bg.color(ray);
}

// f64::powi(n), n > 0, as llvm.powi lowers it (compiler-rt __powidf2, and the same product sequence when LLVM expands a constant
// exponent inline): square-and-multiply from the low bit
inline double powi_pos(double x, int n) {
    double a = x, r = 1.0;
    for (;;) {
        if (n & 1) r *= a;
        n /= 2;
        if (n == 0) break;
        a *= a;
    }
    return r;
}
// SingleLightSourceRayTracer::trace (raytrace.rs:110-135): one bounce - the hit's scatter decides between the emitter branch and a
// Lambert + Blinn-Phong term towards one point light (the scattered ray is dropped, its RNG draws are not).  The reference holds no
// output of this tracer (it is constructed only in a comment, main.rs:153): parity unpinned beyond this restatement.
inline Color trace_single_light(const Ray& ray, const Hittable& world, const Background& bg, const Point3& light_source, double intensity, Rng& rng) {
    g_stats.rays++;
    Hit hit;
    if (world.hit(ray, 0.001, INF, rng, hit)) {
        Color attenuation;
        Ray scattered;
        rng.begin_scatter();
        const bool scatters = hit.material->scatter(ray, hit, rng, attenuation, scattered);
        rng.end_scatter();
        if (scatters) {
            g_stats.scatters++;
            const Vec3 l = (light_source - hit.p).unit();
            const Vec3 v = -ray.dir.unit();
            const Vec3 h = (l + v).unit();
            // "All materials are Lambertian."
            const Color lambertian = attenuation * intensity * std::fmax(l.dot(hit.normal), 0.0);
            const Color blinn_phong = 0.5 * Color(1, 1, 1) * intensity * powi_pos(std::fmax(h.dot(hit.normal), 0.0), 100);
            return lambertian + blinn_phong;
        }
        return hit.material->emit(hit.u, hit.v, hit.p);
    }
    return bg.color(ray);
}

// worlds.rs
struct WorldCamera { Point3 lookfrom, lookat; double field_of_view; };
struct PrimRecord {  // registry used by tests to compare against the product's flattened scene
    int kind;        // 0 sphere, 1 block, 2 rect
    double g[8];     // sphere: cx,cy,cz,r ; block: min xyz, max xyz ; rect: a0,a1(as doubles),a0v0,a0v1,a1v0,a1v1,plane
    double mat[5];
};
struct World {
    std::string name;
    std::shared_ptr<Hittable> root;
    Background background;
    WorldCamera camera;
    std::vector<PrimRecord> prims;       // creation order
    std::vector<const BHV*> bvhs;        // creation order
    std::vector<std::shared_ptr<Perlin>> perlins;
    uint64_t next_raw_u64_after_build = 0;
    uint64_t raw_u64_before_last_bvh = 0;  // stream probe taken just before the last BHV::new (SURVEY.md 8c known answers)
};
std::unique_ptr<World> build_world(const std::string& name, uint64_t seed, std::shared_ptr<ImageData> earth);
const std::vector<std::string>& world_names();

}  // namespace orc

// csrc/mlrt_kernel.hip - the render megakernel for gfx950 (CDNA4, wave64).  Hand-written HIP; no MFMA (branchy
// intersection arithmetic, not a contraction), no CUDA shims, no second back end.
//
// Compiled three times (see mu-lambda-raytracer_amd/build.py), all with -ffp-contract=off:
//   -DMLRT_REAL=float                  -> fast path (f32):    f64 ray state + sphere quadratic, f32 boxes/rects/shading
//   -DMLRT_REAL=double                 -> product path (f64): every operation that decides a sample's value is f64 in the
//                                         reference's operation order; only the CULLING tests differ from the reference
//                                         (SAH topology, box wrappers, slab test by reciprocal + fma with an outward slack,
//                                         primary candidate lists): none of them can change a closest hit
//   -DMLRT_REAL=double -DMLRT_STRICT=1 -> strict path (f64):  additionally the reference's BVH topology and its slab test
//                                         with divisions (bhv.rs:29-42), i.e. exactly the tests the reference performs
//
// What it replaces (reference file:line):
//   Renderer::render / render_line / render_pixel   raytrace.rs:162-198   -> tile queue + per-lane path regeneration
//   Camera::get_ray                                 camera.rs:40-48       -> gen_camera_ray (f64)
//   RecursiveRayTracer::trace_internal              raytrace.rs:79-102    -> iterative throughput loop
//   HittableList::hit                               hittable.rs:53-69     -> the order of the linear entry array
//   BHV::hit / Node::hit / AABB::hit                bhv.rs:29-42,147-165  -> threaded (stackless) BVH walk + slab_hit
//   Sphere::hit / sphere_uv                         shapes.rs:44-82       -> sphere_hit / sphere_uv
//   AARect::hit, Block::hit                         aarects.rs:45-64, shapes.rs:172-192 -> rect_hit, box_hit
//   Translate::hit / Rotate::hit                    transforms.rs:31-42,127-142 -> to_object / rotate_fwd
//   ConstantMedium::hit, Isotropic                  volumes.rs:26-83      -> special_entry (MEDIUM), scatter
//   Lambertian / Metal / Dielectric / DiffuseLight  materials.rs          -> hit_attributes, scatter
//   SolidColor / Checker / Perlin / Image           textures.rs, image_texture.rs -> texture_value
//   to_rgb                                          raytrace.rs:59-68     -> mlrt_resolve_kernel
//   Rngator (Pcg64 per row)                         rngator.rs:27-31      -> Philox4x32-10 per (pixel, sample)
//
// Execution model: persistent workgroups.  Each workgroup stages the packed scene blob into LDS once
// (16-byte coalesced loads), then each WAVE pulls tiles (a pixel group x a chunk of its samples) from the frame's
// queue (interleaved sub-queues, chunk-major tapering passes: see KParams).  Inside a tile the samples form one
// queue; a lane whose path has ended takes the next entry in the next shading batch (__ballot + mbcnt prefix),
// and a wave keeps several tiles open so that no lane waits for a tile's last path.  Finished samples are added
// to per-tile accumulators in LDS as 64-bit fixed point (2^-32), finished tiles to the frame buffer in HBM:
// integer adds are associative, so the image is bit-identical however samples are distributed over lanes,
// waves or GPUs.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/mlrt.h"
#include "mlrt_device.hpp"

#ifndef MLRT_REAL
#error "compile with -DMLRT_REAL=float or -DMLRT_REAL=double"
#endif
#ifndef MLRT_BLOCK
#define MLRT_BLOCK 512
#endif
#define DEV __device__ __forceinline__
// Cold code kept OUT of line: sphere_uv's f64 acos / atan2 (0.07 texel fetches per sample of final_scene, none in most scenes) run for
// a few lanes in a few shading batches, but inlined their registers are the kernel's: the f64 FULL variant needs 255 VGPRs with them
// inline and 200 without (a kernel with an image texture alone: 193 / 129).  Out of line they cost a call and 32 bytes of stack for
// callee-saved registers when they run.  (Perlin turbulence out of line made things worse: it takes the scene view by reference, which
// then lives in memory: 157 -> 178 VGPRs for a noise-only kernel.)  -DMLRT_COLD_UV_INLINE puts it back (experiments).
#ifdef MLRT_COLD_UV_INLINE
#define MLRT_COLD_UV DEV
#else
#define MLRT_COLD_UV __device__ __attribute__((noinline))
#endif
// Branch weights for the register allocator: the traversal step runs every iteration of the kernel loop, the batches every
// ~10th, a tile is opened every ~500th.  Without them a value the step reads (the slab ray) is as good a spill candidate
// as one only the shading code reads, and at 80 registers the difference is 2x in frame time.
#define MLRT_RARELY(x) __builtin_expect(!!(x), 0)
// Instrumentation (in-kernel counters MLRT_STATS=1, the per-sample path printer MLRT_DEBUG_PIXEL=p,s) exists only in the
// debug build of the library (build.py --debug -> libmlrt_hip_debug.so).  Even switched off at run time it costs the
// render kernel 8 %: its wave-uniform counters and the printf call's registers push tile and camera state out of the
// scalar register file.
#ifdef MLRT_DEBUG_BUILD
#define MLRT_STATS_ON(P) ((P).stats != nullptr)
#define MLRT_DEBUG_ON(P) ((P).debug_pixel >= 0)
#else
#define MLRT_STATS_ON(P) false
#define MLRT_DEBUG_ON(P) false
#endif
#define MLRT_CAT_(a, b) a##b
#define MLRT_CAT(a, b) MLRT_CAT_(a, b)
#ifndef MLRT_SUFFIX
#error "compile with -DMLRT_SUFFIX=f32, f64 or f64s"
#endif
#define MLRT_NS MLRT_CAT(mlrt_, MLRT_SUFFIX)

// one namespace per precision build: the two translation units define same-named types of different layout
namespace MLRT_NS {
using namespace mlrt;

typedef MLRT_REAL real;
constexpr bool kFast = sizeof(real) == 4;
#ifdef MLRT_STRICT
constexpr bool kStrict = true;   // reference slab arithmetic (the packer also keeps the reference topology)
#else
constexpr bool kStrict = false;
#endif
static_assert(!(kFast && kStrict), "the strict build is an f64 build");
constexpr bool kProduct64 = !kFast && !kStrict;
// The entries' own type: the bounds of inner entries (and of Block entries) and the arithmetic of the
// per-step culling test.  float: 32-byte entries, primitives' f64 parameters in payload slots (mlrt_device.hpp).
#ifndef MLRT_CULL
#error "compile with -DMLRT_CULL=float or -DMLRT_CULL=double"
#endif
typedef MLRT_CULL creal;
constexpr bool kCull32 = sizeof(creal) == 4;
static_assert(!kFast || kCull32, "the fast build has f32 entries");
static_assert(!kStrict || !kCull32, "the strict build has f64 entries");
constexpr bool kPayloadAll = kCull32 && !kFast;    // product f64 build: every primitive's parameters are f64 payload slots
constexpr bool kRigorousCull = kCull32 && !kFast;
#ifndef MLRT_LEAN_MATERIALS_HBM
#define MLRT_LEAN_MATERIALS_HBM 0
#endif
// The sphere-only variant reads its materials (one per sphere: a fifth of `random`'s f64 blob) from HBM / L2, once per shading, and
// stages only the entries in LDS: the blob then leaves room for two workgroups per CU (KParams::lds_blob_bytes stops at the materials).
constexpr bool kLeanMaterialsInHbm = MLRT_LEAN_MATERIALS_HBM != 0;
constexpr bool kPerlinInHbm = !kFast;              // f64 builds: the Perlin tables stay in HBM (they end the blob; KParams::lds_blob_bytes stops before them)  // ... and its f32 culling test may never reject what an f64 primitive test accepts
// Outward slack of the product f64 build's culling tests: reciprocal + fma instead of the reference's division moves a
// slab distance by a few 1e-16 relative; 1e-12 is far outside that and far inside anything a box can wrongly admit.
constexpr double kCullSlack64 = 1.0 + 1e-12;
// Per wave: kAccBytesPerWave bytes of LDS holding the accumulator banks, one per tile in flight (the one being handed out
// and older ones whose last paths are still running): [remaining | out | values][kMaxBanks] ints, the
// candidate records, then the sums - n_banks x bank_values of them, cut at run time (KParams): one-pixel tiles get all 16
// banks, four-pixel tiles 6.  Many banks matter with small tiles: one 50-bounce path holds its bank while the other
// lanes finish ~8 tiles' worth of 64-sample tiles.
constexpr int kAccBytesPerWave = 1344;
// The variants other than the sphere-only one have no candidate records (kMaxBanks x kCandRecord u16 = 512 bytes of the area): with the
// smaller area final_scene's f64 entries stay LDS resident under a 768-thread workgroup (3 waves per SIMD).
constexpr int kAccBytesPerWaveFull = 1344 - 512;
constexpr int acc_bytes_per_wave(uint32_t feat) { return feat == 0 ? kAccBytesPerWave : kAccBytesPerWaveFull; }
constexpr int kMaxBanks = 16;
// + per bank: the tile's primary candidate list (count, then up to kCand entry indices as u16; count 0xFF = not usable)
constexpr int kCand = 14;
constexpr int kCandInvalid = 0xFF;
constexpr int kPrimaryFlag = 1 << 30;  // in `depth`: the lane's primary ray is working through its tile's list (index in bits 16..23)
constexpr int kPixBankShift = 10;      // `pix` = bank << 10 | index of the pixel's first sum in the wave's acc array

// Per-frame parameters only the sample generation and the path ends read (camera.rs:3-12, the background): every
// workgroup copies them from the kernel arguments into LDS once.  Left in the argument struct they sit in ~50 scalar
// registers for the whole kernel (the compiler hoists the loads), which spilled scalar registers into vector lanes.
struct FrameParams {
    double cam_origin[3], cam_llc[3], cam_horizontal[3], cam_vertical[3], cam_u[3], cam_v[3];
    double lens_radius, inv_wm1, inv_hm1;
    MLRT_REAL bg_top[3], bg_bottom[3];
    float through0[3], hor_f[3], ver_f[3], u_f[3], v_f[3];  // fast build: llc - origin, horizontal, vertical, u, v in f32
    unsigned int stall_cap;                                  // watchdog (see the end of the kernel loop)
};
constexpr int kFrameParamBytes = 320;
static_assert(sizeof(FrameParams) <= kFrameParamBytes, "FrameParams must fit its LDS slot");
// the kernel copies cam_origin ... inv_hm1 as 21 consecutive doubles from the argument struct
static_assert(offsetof(FrameParams, inv_hm1) - offsetof(FrameParams, cam_origin) == 20 * sizeof(double), "FrameParams: camera block layout");
static_assert(offsetof(mlrt::KParams<MLRT_REAL>, inv_hm1) - offsetof(mlrt::KParams<MLRT_REAL>, cam_origin) == 20 * sizeof(double), "KParams: camera block layout");

// ---------------------------------------------------------------------------------------------- math helpers
DEV int magic_div(int n, uint32_t magic, int shift) { return shift < 0 ? n : (int)(__umulhi((uint32_t)n, magic) >> shift); }  // KParams: n / divisor
DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV double fast_rcp(double x) { return 1.0 / x; }
DEV float sqrt_r(float x) { return __builtin_amdgcn_sqrtf(x); }
DEV double sqrt_r(double x) { return sqrt(x); }
DEV float div_r(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
DEV double div_r(double a, double b) { return a / b; }
// Several f64 quotients over ONE divisor (the three components of Vec3 / f64, vec.rs:184-189; the two roots of the sphere
// quadratic over |d|^2, shapes.rs:66-73; a slab's two face distances, aarects.rs:46).  The compiler expands every x / a on
// its own into v_div_scale_f64 x 2, v_rcp_f64, four fma (two Newton steps on the reciprocal), mul, fma, v_div_fmas_f64,
// v_div_fixup_f64.  Everything up to the refined reciprocal depends on the divisor alone: Divisor does it once, over() is the
// tail (mul, fma, fma) per numerator - the SAME instructions on the same values whenever v_div_scale_f64 scales nothing and
// v_div_fixup_f64 changes nothing but the sign of a zero, i.e. for a divisor in 2^-300 .. 2^300 (checked: anything else -
// zero, denormal, huge, NaN - divides the long way) and numerators that are zero or in 2^-600 .. 2^600 (not checked: what the
// kernel divides are differences and products of scene coordinates and ray directions, which validate() bounds by 2^100 and the
// materials keep above 1e-8; a quotient of zero takes its sign from x * (1/a), as v_div_fixup_f64 gives it).  So the quotient is
// bit for bit the compiler's - which tests/test_gpu_f64_modes.py holds to the strict build (plain divisions only) over every
// world and option, tests/test_gpu_fuzz.py to the oracle at four scales, and the probe of tests/test_gpu_directed.py to x / a
// itself on 1e8 operand pairs.  (Range tests on every quotient were measured first: they cost what the sharing saves.)
#ifndef MLRT_SHARED_DIVISOR
#define MLRT_SHARED_DIVISOR 1
#endif
struct Divisor {
    double a, y;
    bool ok;
    DEV explicit Divisor(double divisor) : a(divisor) {
        const uint32_t exponent = ((uint32_t)__double2hiint(divisor) >> 20) & 0x7FFu;
        ok = exponent - 723u < 600u;  // 2^-300 <= |a| < 2^300
        double r = __builtin_amdgcn_rcp(divisor);
        double e = __builtin_fma(-divisor, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-divisor, r, 1.0);
        y = __builtin_fma(r, e, r);
    }
    // a divisor whose refined reciprocal was worked out before (the frame's width - 1 and height - 1: FrameParams)
    DEV Divisor(double divisor, double refined_reciprocal) : a(divisor), y(refined_reciprocal), ok(true) {}
    DEV double one(double x) const { return kShared ? tail(x) : x / a; }
    DEV double tail(double x) const {
        const double q0 = x * y;
        const double r = __builtin_fma(-a, q0, x);
        return __builtin_copysign(__builtin_fma(r, y, q0), q0);
    }
    static constexpr bool kShared = MLRT_SHARED_DIVISOR && !kStrict;
    DEV void over(double x0, double x1, double& q0, double& q1) const {
        if (kShared && __builtin_expect(ok, 1)) { q0 = tail(x0); q1 = tail(x1); }
        else { q0 = x0 / a; q1 = x1 / a; }
    }
    DEV void over(double x0, double x1, double x2, double& q0, double& q1, double& q2) const {
        if (kShared && __builtin_expect(ok, 1)) { q0 = tail(x0); q1 = tail(x1); q2 = tail(x2); }
        else { q0 = x0 / a; q1 = x1 / a; q2 = x2 / a; }
    }
};
DEV float min_r(float a, float b) { return fminf(a, b); }
DEV double min_r(double a, double b) { return fmin(a, b); }
DEV float max_r(float a, float b) { return fmaxf(a, b); }
DEV double max_r(double a, double b) { return fmax(a, b); }
DEV float abs_r(float a) { return fabsf(a); }
DEV double abs_r(double a) { return fabs(a); }
// (f32 mode: the sine of a Noise / Checker texture by v_sin_f32 on the fraction of a / 2 pi - three instructions for the library's ~60;
// its error, ~1e-6 plus the argument's own f32 rounding, is far inside the mode's tolerance)
DEV float sin_r(float a) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(a * 0.15915494309189535f)); }
DEV double sin_r(double a) { return sin(a); }
// ln of a medium's uniform draw (volumes.rs:48, f64 builds): x = 0, or normal and in [2^-32, 1).  The classic argument reduction
// x = 2^k m, m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), ln m = 2 atanh(s) = f - f^2/2 + s (f^2/2 + R(s^2)) with R a degree-7
// minimax polynomial in s^2 (the coefficients of fdlibm's e_log.c; error below 1 ulp), ln 2 added in two parts.  ~40 instructions where
// the device library's log - double-double arithmetic, correct to the last bit or so - takes ~85: the medium's log is issued in nearly
// every special batch of a scene with fog.  The oracle calls glibc's log (what Rust's f64::ln calls): either way the last bit may differ,
// which is what the <= 0.01 % tolerance of the worlds with media is for (tests/test_gpu_parity.py).
#ifndef MLRT_MEDIUM_LOG_LIBM
#define MLRT_MEDIUM_LOG_LIBM 0
#endif
DEV double log_uniform(double x) {
    if (MLRT_MEDIUM_LOG_LIBM) return log(x);
    if (x == 0.0) return -(double)INFINITY;
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
DEV float floor_r(float a) { return floorf(a); }
DEV double floor_r(double a) { return floor(a); }

struct V3 {
    real x, y, z;
};
DEV V3 mk(real x, real y, real z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 ld3(const real* p) { return mk(p[0], p[1], p[2]); }
DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
DEV V3 operator*(V3 a, real s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV V3 operator*(real s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV real length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV real comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
DEV V3 axis_vec(int i, real s) { return mk(i == 0 ? s : (real)0, i == 1 ? s : (real)0, i == 2 ? s : (real)0); }
// Geometry state (ray origin, direction, hit point) is f64 in BOTH builds: an f32 ulp of a ~500-unit coordinate
// (6e-5) is enough to flip accept/reject decisions against the oracle in dense geometry (final_scene foam).
struct P3 {
    double x, y, z;
};
// a*b + c for the f64 geometry code.  The fast build fuses EXPLICITLY and pins the compiler's own contraction off
// in the functions that use it: the reference relies on evaluating the same quadratic twice giving the identical
// root (a ConstantMedium boundary that is a clone of a listed sphere, worlds.rs:428-430 -> volumes.rs:36 `h1.t >=
// h2.t`), and with -ffp-contract=fast two inlined copies may be fused differently and differ in the last bit.
// The parity build keeps the reference's separate multiply and add.
DEV double dfma(double a, double b, double c) {
#pragma clang fp contract(off)
    if (sizeof(MLRT_REAL) == 4) return __builtin_fma(a, b, c);
    return a * b + c;
}
DEV P3 mkp(double x, double y, double z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV P3 to_p3(V3 a) { return mkp((double)a.x, (double)a.y, (double)a.z); }
DEV V3 to_v3(P3 a) { return mk((real)a.x, (real)a.y, (real)a.z); }
DEV P3 ld3p(const double* p) { return mkp(p[0], p[1], p[2]); }
DEV P3 operator+(P3 a, P3 b) { return mkp(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV P3 operator-(P3 a, P3 b) { return mkp(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV P3 operator*(P3 a, double s) { return mkp(a.x * s, a.y * s, a.z * s); }
DEV P3 operator*(double s, P3 a) { return mkp(a.x * s, a.y * s, a.z * s); }
DEV double compp(P3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// Vec3::unit = self / self.length() (vec.rs:67-69): per-component division in the parity build
DEV V3 unit(V3 a) {
    if (kFast) {
        const real inv = (real)__builtin_amdgcn_rsqf((float)length_squared(a));
        return a * inv;
    }
    const real len = sqrt_r(length_squared(a));
    double qx, qy, qz;
    Divisor((double)len).over((double)a.x, (double)a.y, (double)a.z, qx, qy, qz);
    return mk((real)qx, (real)qy, (real)qz);
}

// ---------------------------------------------------------------------------------------------- Philox4x32-10
// One stream per (pixel, sample): counter = (draw/4, sample, pixel, 0), key = seed; draw k is word k%4 of block k/4.
// Identical to PhiloxStream in oracle/oracle_core.hpp, word for word.  next3/next4 read several consecutive words
// with ONE refill site: the draw order is unchanged, only the number of divergent refill branches drops.
struct Rng {
    uint32_t pixel, sample, draw;
    uint32_t b3;  // word 3 of the block `draw` is in (valid when draw % 4 == 3): all that is kept of a block between draws
};
struct RngKey {
    uint32_t k0, k1;
};
DEV void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t& o0, uint32_t& o1,
                       uint32_t& o2, uint32_t& o3) {
#ifdef MLRT_PROBE_PHILOX_ROUNDS  // cost probe (experiments only): 20 rounds instead of 10 cost C2 +10 %: that is Philox's share
    constexpr int kRounds = MLRT_PROBE_PHILOX_ROUNDS;
#else
    constexpr int kRounds = 10;
#endif
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        // one 32x32->64 multiply (v_mad_u64_u32) yields both halves: integer multiplies are quarter rate on CDNA
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
// One draw of the stream outside a scatter (a ConstantMedium's, during traversal).  Word 3 of a block is served from the
// lane's state - after a scatter that is the usual case - any other word evaluates its block (a second medium draw in one
// ray segment is rare: keeping all four words cost three registers in the variant whose shading batch is short of them).
DEV uint32_t rng_next(Rng& g, RngKey key) {
    const uint32_t i = g.draw & 3u;
    uint32_t w = g.b3;
    if (i != 3u) {
        uint32_t w0, w1, w2, w3;
        philox4x32_10(key.k0, key.k1, g.draw >> 2, g.sample, g.pixel, 0u, w0, w1, w2, w3);
        g.b3 = w3;
        w = i == 0 ? w0 : (i == 1 ? w1 : w2);
    }
    g.draw++;
    return w;
}
// The words of one Material::scatter.  Contract (mirrored by the oracle's PhiloxStream::begin_scatter/end_scatter): a
// scatter starts on a block boundary and owns words 0-2 of its block, whatever the material draws from them (Lambertian /
// Metal / Isotropic: the unit-ball sample; Dielectric: word 0 when it can refract, materials.rs:98); word 3 is the next
// draw of the stream (the first ConstantMedium draw of the scattered ray, which so needs no Philox of its own).  So a
// shading batch runs Philox exactly once, with no refill branch and no word selection; and in a scene without media
// (whose draws happen during traversal, volumes.rs:47) the block number is just 1 + the bounce number: the stream
// position is not per-lane state at all.
template <uint32_t FEAT>
DEV void rng_scatter_words(Rng& g, RngKey key, int bounce, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
    const uint32_t block = (FEAT & FEAT_MEDIUM) ? (g.draw + 3u) >> 2 : 1u + (uint32_t)bounce;
    uint32_t w3;
    philox4x32_10(key.k0, key.k1, block, g.sample, g.pixel, 0u, w0, w1, w2, w3);
    if (FEAT & FEAT_MEDIUM) { g.draw = (block << 2) + 3u; g.b3 = w3; }
}
DEV real u01(uint32_t w) { return (real)(w >> 8) * (real)(1.0 / 16777216.0); }
DEV double u01d(uint32_t w) { return (double)(w >> 8) * (1.0 / 16777216.0); }
// rand's UniformFloat::sample_single: u * (high - low) + low (the >= high retry cannot trigger for 24-bit u)
DEV real sym(uint32_t w) { return u01(w) * (real)2 + (real)-1; }  // gen_range(-1.0..1.0)
// cos and sin of 2 pi k / 2^24 for the 24-bit k of a sampler's uniform (unit_ball_from_words, the lens disk of gen_camera_ray): part of
// the DEVICE CONTRACT, defined as arithmetic - not "whatever libm returns" - so that the oracle (oracle_core.hpp turn_cos_sin, the same
// lines with std::fma) gets the very same doubles: the nearest of 64 tabulated angles (sin, cos exact at the axes and diagonals), then
// the Taylor polynomials of the remainder |x| <= pi / 64 (next terms 5e-18 and 2e-20) and the angle-sum formulas, every operation
// an exactly rounded mul or fma.  |error| < 7e-16 (all 2^24 inputs against long-double libm: tests/test_oracle_samplers.py); 15 f64 instructions where ocml's sincospi takes 29 plus as many integer ones.
__device__ const double kTurnTable[64][2] = {
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
DEV void turn_cos_sin(uint32_t k, double& cs, double& sn) {
    const uint32_t j = (k + (1u << 17)) >> 18;
    const int dk = (int)k - (int)(j << 18);
    const double x = (double)dk * 0x1.921fb54442d18p-22;  // 2 pi / 2^24
    const double x2 = x * x;
    const double ps = __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, -0x1.a01a01a01a01ap-13, 0x1.1111111111111p-7), -0x1.5555555555555p-3), 1.0);
    const double sx = x * ps;
    const double cx = __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, 0x1.a01a01a01a01ap-16, -0x1.6c16c16c16c17p-10), 0x1.5555555555555p-5), -0.5), 1.0);
    const double ta_sin = kTurnTable[j & 63u][0], ta_cos = kTurnTable[j & 63u][1];
    sn = __builtin_fma(ta_sin, cx, ta_cos * sx);
    cs = __builtin_fma(ta_cos, cx, -(ta_sin * sx));
}
// cbrt(n 2^-24) for n = 1 .. 2^24, the radius of a unit-ball sample: arithmetic of the device contract like turn_cos_sin (the
// oracle's cbrt24 is the same lines).  n = f 2^e with f in [1, 2); e = 3 q + rem; z ~ f^(-1/3) from a cubic seed (5e-4) and two
// third-order steps z <- z (1 + h/3 + 2 h^2/9), h = 1 - f z^3; cbrt(f) = f z^2; times cbrt(2^rem) and 2^(q - 8).  Relative
// error <= 5.5e-16 over all 2^24 inputs (checked exhaustively against cbrtl), exact at both ends (2^-8 and 1).
DEV double cbrt24(uint32_t n) {
    const int e = 31 - __builtin_clz(n);
    const double fn = (double)n;
    const double f = __hiloint2double(__double2hiint(fn) - (e << 20), __double2loint(fn));  // n 2^-e
    const int q3 = (e * 11) >> 5;  // e / 3 for e <= 24
    const int rem = e - 3 * q3;
    double z = __builtin_fma(f, __builtin_fma(f, __builtin_fma(f, -0x1.9239a09c3fa5fp-5, 0x1.407a7b055bc3ep-2), -0x1.9a4cbd01c7c23p-1), 0x1.897a11517ead3p+0);
#pragma unroll 1
    for (int it = 0; it < 2; it++) {
        const double z2 = z * z;
        const double h = __builtin_fma(-f * z2, z, 1.0);
        const double p = __builtin_fma(h, 0x1.c71c71c71c71cp-3, 0x1.5555555555555p-2);
        z = __builtin_fma(z * h, p, z);
    }
    double y = f * z * z;
    y *= rem == 0 ? 1.0 : (rem == 1 ? 0x1.428a2f98d728bp+0 : 0x1.965fea53d6e3dp+0);  // cbrt(2), cbrt(4)
    return __hiloint2double(__double2hiint(y) + ((q3 - 8) << 20), __double2loint(y));  // 2^(q - 8)
}
// Uniform point in the unit ball.  The reference draws it by rejection (vec.rs:23-30); the device contract (and the
// oracle's Philox mode, oracle/oracle.cpp closed_form_unit_sphere) draws the same distribution by inverse CDF from
// exactly three words - z = 2*u1 - 1, phi = 2*pi*u2, r = cbrt((w3>>8 + 1) * 2^-24) - so there is no data-dependent
// loop: in a 37-lane shading batch the rejection loop ran ~5 times for the unluckiest lane, each time re-running
// Philox for a handful of lanes.  Fast build: hardware sin/cos (argument in turns), log2/exp2 for the cube root.
DEV V3 unit_ball_from_words(uint32_t w0, uint32_t w1, uint32_t w2) {
    if (kFast) {
        const float z = (float)sym(w0);
        const float turn = (float)u01(w1);
        const float u3 = ((float)(w2 >> 8) + 1.0f) * (float)(1.0 / 16777216.0);
        const float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u3) * (float)(1.0 / 3.0));
        const float sn = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - z * z));
        const float rs = r * sn;
        return mk((real)(rs * __builtin_amdgcn_cosf(turn)), (real)(rs * __builtin_amdgcn_sinf(turn)), (real)(r * z));
    }
    const double z = u01d(w0) * 2.0 + -1.0;
    const double r = cbrt24((w2 >> 8) + 1u);
    double sn_phi, cs_phi;
    turn_cos_sin(w1 >> 8, cs_phi, sn_phi);  // phi = 2 pi (w1 >> 8) 2^-24
    const double sn = sqrt(fmax(0.0, 1.0 - z * z));
    const double rs = r * sn;
    return mk((real)(rs * cs_phi), (real)(rs * sn_phi), (real)(r * z));
}

// ---------------------------------------------------------------------------------------------- scene view
struct Scene {
    const DNode<creal>* nodes;      // the linear threaded entry array (see mlrt_device.hpp)
    const DMaterial<real>* materials;
    const DTexture<real>* textures;
    const real* perlin_vec;         // [tables][1024][3]
    const uint16_t* perlin_perm;    // [tables][3][1024]
    const DImage* images;           // HBM (L2-resident)
    // the pair walk (mlrt_device.hpp DPair; sphere-only scenes): `nodes` is then unused
    const DPair* pairs;
    const real* prims;              // [n_prims][4]: centre, radius
    const uint16_t* prim_mat;       // [n_prims]
};
// The f64 parameters of the primitive whose entry is `idx` (mlrt_device.hpp: payload slots behind an f32 entry, inline in an
// f64 entry; the fast build computes boxes, rects, media and transforms from the entry's inline f32 values).
DEV const double* payload(const Scene& S, int idx) { return reinterpret_cast<const double*>(&S.nodes[idx + 1]); }
struct SphereP {
    double cx, cy, cz, r;
};
DEV SphereP sphere_params(const Scene& S, int idx, const DNode<creal>& n) {
    SphereP s;
    if (kPayloadAll) { const double* p = payload(S, idx); s.cx = p[0]; s.cy = p[1]; s.cz = p[2]; s.r = p[3]; }
    else { s.cx = (double)n.v[0]; s.cy = (double)n.v[1]; s.cz = (double)n.v[2]; s.r = (double)n.v[3]; }
    return s;
}
DEV SphereP pair_sphere(const Scene& S, int prim) {  // pair walk: primitive `prim` of S.prims (one 16- or 32-byte read)
    SphereP s;
    const real* q = S.prims + 4 * prim;
    s.cx = (double)q[0]; s.cy = (double)q[1]; s.cz = (double)q[2]; s.r = (double)q[3];
    return s;
}
template <int N>
DEV void prim_params(const Scene& S, int idx, const DNode<creal>& n, double (&out)[N]) {
    if (kPayloadAll) {
        const double* p = payload(S, idx);
#pragma unroll
        for (int k = 0; k < N; k++) out[k] = p[k];
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) out[k] = (double)n.v[k];
    }
}
DEV int slots_of(int tag) { return entry_slots(tag & TAG_KIND_MASK, kCull32, kPayloadAll); }

// What the cheap per-step tests need of the ray in the current (world or object) space.
struct SlabRay {
    // f32 entries and the cull_f64 product build: -origin/direction and 1/direction (one fma per plane); strict build: origin and direction.
    // of2*: product f64 build over f32 entries only - the offsets of the max planes, see slab_ray (elsewhere unused: of* serves both planes)
    creal ofx, ofy, ofz, ix, iy, iz, of2x, of2y, of2z;
};
// t = (k - origin) / direction for a plane coordinate k.  f32 entries: ONE explicit fma per plane, k * (1/d) + (-o/d) (the
// explicit builtin keeps the rounding independent of the compiler's contraction setting); cull_f64 product build: the same in
// f64 (culling only, see kCullSlack64); strict build: the reference's subtraction and division (bhv.rs:33-35).
DEV creal slab_t(creal k, creal of, creal invd) {
    if (kCull32) return (creal)__builtin_fmaf((float)k, (float)invd, (float)of);
    if (!kStrict) return (creal)__builtin_fma((double)k, (double)invd, (double)of);
    return (k - of) / invd;
}
// The f64 ray for the primitive tests.
struct Ray {
    P3 o, d;
    double a;      // |d|^2
    double inv_a;  // fast build: 1/|d|^2
};

// Translate(offset, Rotate(axis, angle, .)) of an XFORM entry
struct XfP {
    double sin_t, cos_t, ox, oy, oz;
    int tag;
};
DEV XfP xform_params(const Scene& S, int idx) {
    const DNode<creal> n = S.nodes[idx];
    double v[5];
    prim_params<5>(S, idx, n, v);
    XfP x;
    x.sin_t = v[0]; x.cos_t = v[1]; x.ox = v[2]; x.oy = v[3]; x.oz = v[4]; x.tag = n.tag;
    return x;
}
// Translate::hit / Rotate::hit ray mapping (transforms.rs:33, 106-114, 128-132), in f64
DEV P3 rotate_back(const XfP& xf, P3 v) {
#pragma clang fp contract(off)
    const int a1 = (xf.tag >> 8) & 3, a2 = (a1 + 1) % 3, a0 = (a1 + 2) % 3;
    const double sin_t = xf.sin_t, cos_t = xf.cos_t;
    // (about y - every rotation of the reference's worlds - without the component selects of the general case, which are nine tenths of
    // its instructions; the same products and sums: a branch the other lanes skip)
    if (a1 == 1) return mkp(dfma(cos_t, v.x, -(sin_t * v.z)), v.y, dfma(sin_t, v.x, cos_t * v.z));
    const double v0 = compp(v, a0), v2 = compp(v, a2);
    const double n0 = dfma(cos_t, v0, -(sin_t * v2));
    const double n2 = dfma(sin_t, v0, cos_t * v2);
    return mkp(a0 == 0 ? n0 : (a2 == 0 ? n2 : v.x), a0 == 1 ? n0 : (a2 == 1 ? n2 : v.y), a0 == 2 ? n0 : (a2 == 2 ? n2 : v.z));
}
DEV P3 rotate_fwd(const XfP& xf, P3 v) {  // transforms.rs:116-124
#pragma clang fp contract(off)
    const int a1 = (xf.tag >> 8) & 3, a2 = (a1 + 1) % 3, a0 = (a1 + 2) % 3;
    const double sin_t = xf.sin_t, cos_t = xf.cos_t;
    if (a1 == 1) return mkp(dfma(cos_t, v.x, sin_t * v.z), v.y, dfma(-sin_t, v.x, cos_t * v.z));
    const double v0 = compp(v, a0), v2 = compp(v, a2);
    const double n0 = dfma(cos_t, v0, sin_t * v2);
    const double n2 = dfma(-sin_t, v0, cos_t * v2);
    return mkp(a0 == 0 ? n0 : (a2 == 0 ? n2 : v.x), a0 == 1 ? n0 : (a2 == 1 ? n2 : v.y), a0 == 2 ? n0 : (a2 == 2 ? n2 : v.z));
}
DEV bool xf_translates(const XfP& xf) { return ((xf.tag >> 4) & XF_TRANSLATE) != 0; }
DEV bool xf_rotates(const XfP& xf) { return ((xf.tag >> 4) & XF_ROTATE) != 0; }
DEV void to_object(const XfP& xf, P3& o, P3& d) {
#pragma clang fp contract(off)
    if (xf_translates(xf)) o = o - mkp(xf.ox, xf.oy, xf.oz);
    if (xf_rotates(xf)) { o = rotate_back(xf, o); d = rotate_back(xf, d); }
}
// The ray in the space of the bracket `xf` (entry index, -1 = world).
template <uint32_t FEAT>
DEV Ray local_ray(const Scene& S, int xf, P3 o, P3 d) {
#pragma clang fp contract(off)
    Ray r;
    r.o = o; r.d = d;
    if ((FEAT & FEAT_XFORM) && xf >= 0) to_object(xform_params(S, xf), r.o, r.d);
    r.a = dfma(r.d.z, r.d.z, dfma(r.d.y, r.d.y, r.d.x * r.d.x));  // length_squared: x*x + y*y + z*z
    if (kFast) {
        // 1/|d|^2: f32 seed (1e-7) + ONE Newton step (1e-14: the roots carry that relative error, far below the f32 shading
        // that consumes them; no f64 divide in the hot path)
        double x = (double)__builtin_amdgcn_rcpf((float)r.a);
        x = x * dfma(-r.a, x, 2.0);
        r.inv_a = x;
    } else {
        r.inv_a = 0.0;
    }
    return r;
}
// Relative slack of the f32 culling test on the far bound and on closest-so-far.  Fast build: 2-3 ulp (decisions may flip in
// grazing cases, inside the build's RMSE tolerance).  Product f64 build: the test must NEVER reject what the f64 primitive
// test would accept, so the slack covers the worst-case relative error of both interval ends (2 x 4 ulp/2) and the absolute
// error is folded into the ray (slab_ray).
constexpr float kSlack32 = kRigorousCull ? 1.000001f : 1.0000003f;
DEV SlabRay slab_ray(const Ray& r) {
    SlabRay q;
    if (kCull32) {
        // an axis-parallel ray (d = 0 on an axis) keeps finite terms: |d| is floored at 1e-30, so t stays +-1e30-sized
        // with the right sign instead of inf - inf
        const float tiny = 1e-30f;
        const float dx = (float)r.d.x, dy = (float)r.d.y, dz = (float)r.d.z;
        const float ix = fast_rcp(fabsf(dx) < tiny ? copysignf(tiny, dx) : dx), iy = fast_rcp(fabsf(dy) < tiny ? copysignf(tiny, dy) : dy),
                    iz = fast_rcp(fabsf(dz) < tiny ? copysignf(tiny, dz) : dz);
        const float ofx = -((float)r.o.x * ix), ofy = -((float)r.o.y * iy), ofz = -((float)r.o.z * iz);
        q.ix = (creal)ix; q.iy = (creal)iy; q.iz = (creal)iz;
        if (kRigorousCull) {
            // Error of the computed t' = fma(k', r, of') against the exact (k' - o) / d with f64 o, d (k' is the entry's f32
            // bound, already moved outward by the packer): r = 1/d (1 + dr), |dr| <= 3 * 2^-24 (d rounded to f32, v_rcp_f32 1 ulp);
            // of' = -o (1 + do)(1 + dm) r, |do|, |dm| <= 2^-24; one rounding in the fma.  So
            //     |t' - t| <= |t| * 4 * 2^-24  +  |o / d| * 2 * 2^-24   (+ second order).
            // The first term is relative and covered by kSlack32.  The second is absolute and can dwarf t (an origin 1000
            // units from the world origin, a sphere hit at t = 1): e = |o' r| * 2^-21 (4x the bound) widens the slab on
            // this axis - the near plane's t is lowered, the far plane's raised.  t0 comes from the entry's min plane, which is
            // the near one when r > 0: of (min plane) = of' - sign(r) e, of2 (max plane) = of' + sign(r) e.
            const float ex = copysignf(fabsf(ofx) * 4.76837158e-7f, ix), ey = copysignf(fabsf(ofy) * 4.76837158e-7f, iy), ez = copysignf(fabsf(ofz) * 4.76837158e-7f, iz);
            q.ofx = (creal)(ofx - ex); q.ofy = (creal)(ofy - ey); q.ofz = (creal)(ofz - ez);
            q.of2x = (creal)(ofx + ex); q.of2y = (creal)(ofy + ey); q.of2z = (creal)(ofz + ez);
        } else {
            q.ofx = (creal)ofx; q.ofy = (creal)ofy; q.ofz = (creal)ofz;
            q.of2x = q.ofx; q.of2y = q.ofy; q.of2z = q.ofz;
        }
    } else if (!kStrict) {
        // the same form in f64: 1/d by v_rcp_f64 + one Newton step (~1e-16; the slack is 1e-12), |d| floored at 1e-300
        const double tiny = 1e-300;
        const double dx = fabs(r.d.x) < tiny ? copysign(tiny, r.d.x) : r.d.x, dy = fabs(r.d.y) < tiny ? copysign(tiny, r.d.y) : r.d.y,
                     dz = fabs(r.d.z) < tiny ? copysign(tiny, r.d.z) : r.d.z;
        double ix = __builtin_amdgcn_rcp(dx), iy = __builtin_amdgcn_rcp(dy), iz = __builtin_amdgcn_rcp(dz);
        ix = ix * __builtin_fma(-dx, ix, 2.0); iy = iy * __builtin_fma(-dy, iy, 2.0); iz = iz * __builtin_fma(-dz, iz, 2.0);
        q.ix = (creal)ix; q.iy = (creal)iy; q.iz = (creal)iz;
        q.ofx = (creal)-(r.o.x * ix); q.ofy = (creal)-(r.o.y * iy); q.ofz = (creal)-(r.o.z * iz);
        q.of2x = q.ofx; q.of2y = q.ofy; q.of2z = q.ofz;
    } else {
        q.ofx = (creal)r.o.x; q.ofy = (creal)r.o.y; q.ofz = (creal)r.o.z;
        q.ix = (creal)r.d.x; q.iy = (creal)r.d.y; q.iz = (creal)r.d.z;
        q.of2x = q.ofx; q.of2y = q.ofy; q.of2z = q.ofz;
    }
    return q;
}

// AABB::hit, bhv.rs:29-42 (quirk Q5), for f64 entries.  Evaluating the three slabs and testing once equals the reference's
// per-axis early exit because the interval only shrinks.
DEV bool slab_hit(const DNode<creal>& n, const SlabRay& r, double tmin, double tmax) {
    double t0 = slab_t(n.v[0], r.ofx, r.ix), t1 = slab_t(n.v[3], r.ofx, r.ix);
    tmin = fmax(fmin(t0, t1), tmin); tmax = fmin(fmax(t0, t1), tmax);
    t0 = slab_t(n.v[1], r.ofy, r.iy); t1 = slab_t(n.v[4], r.ofy, r.iy);
    tmin = fmax(fmin(t0, t1), tmin); tmax = fmin(fmax(t0, t1), tmax);
    t0 = slab_t(n.v[2], r.ofz, r.iz); t1 = slab_t(n.v[5], r.ofz, r.iz);
    tmin = fmax(fmin(t0, t1), tmin); tmax = fmin(fmax(t0, t1), tmax);
    if (!kStrict) return !(tmax * kCullSlack64 <= tmin);
    return !(tmax <= tmin);
}
// The same decision over f32 entries with two instructions less per step: the far bound min(far, closest) * slack is
// min(far * slack, closest * slack) (rounding is monotone), and closest * slack changes only when a hit is accepted, so
// the caller passes it in: no min with - and no canonicalisation of - a value the compiler cannot see the origin of.
DEV bool slab_hit_fast(const DNode<creal>& n, const SlabRay& r, float closest_slack) {
    float t0 = slab_t(n.v[0], r.ofx, r.ix), t1 = slab_t(n.v[3], kRigorousCull ? r.of2x : r.ofx, r.ix);
    float tnear = fmaxf(fminf(t0, t1), 0.001f), tfar = fmaxf(t0, t1);
    t0 = slab_t(n.v[1], r.ofy, r.iy); t1 = slab_t(n.v[4], kRigorousCull ? r.of2y : r.ofy, r.iy);
    tnear = fmaxf(fminf(t0, t1), tnear); tfar = fminf(fmaxf(t0, t1), tfar);
    t0 = slab_t(n.v[2], r.ofz, r.iz); t1 = slab_t(n.v[5], kRigorousCull ? r.of2z : r.ofz, r.iz);
    tnear = fmaxf(fminf(t0, t1), tnear); tfar = fminf(fmaxf(t0, t1), tfar);
    // (closest_slack and tnear are positive and not NaN - tnear >= 0.001 by the fmaxf above - so the second comparison is done
    // on the bit patterns: written on floats the compiler folds the two back into one min, canonicalisation included)
    return !(tfar * kSlack32 <= tnear) && !(__builtin_bit_cast(int, closest_slack) <= __builtin_bit_cast(int, tnear));
}
// One child box of a pair node (f32 culling builds only): the decision of slab_hit_fast, and the entry distance the
// near-first order is taken from.
DEV bool slab_child(const float (&v)[6], const SlabRay& r, float closest_slack, float& tnear_out) {
    float t0 = __builtin_fmaf(v[0], (float)r.ix, (float)r.ofx), t1 = __builtin_fmaf(v[3], (float)r.ix, (float)(kRigorousCull ? r.of2x : r.ofx));
    float tnear = fmaxf(fminf(t0, t1), 0.001f), tfar = fmaxf(t0, t1);
    t0 = __builtin_fmaf(v[1], (float)r.iy, (float)r.ofy); t1 = __builtin_fmaf(v[4], (float)r.iy, (float)(kRigorousCull ? r.of2y : r.ofy));
    tnear = fmaxf(fminf(t0, t1), tnear); tfar = fminf(fmaxf(t0, t1), tfar);
    t0 = __builtin_fmaf(v[2], (float)r.iz, (float)r.ofz); t1 = __builtin_fmaf(v[5], (float)r.iz, (float)(kRigorousCull ? r.of2z : r.ofz));
    tnear = fmaxf(fminf(t0, t1), tnear); tfar = fminf(fmaxf(t0, t1), tfar);
    tnear_out = tnear;
    return !(tfar * kSlack32 <= tnear) && !(__builtin_bit_cast(int, closest_slack) <= __builtin_bit_cast(int, tnear));
}
// closest-so-far as the slab tests of the current build want it
DEV creal slab_far_bound(double closest) { return kCull32 ? (creal)((float)closest * kSlack32) : (creal)closest; }  // (f64 entries: slab_hit applies the slack)
DEV bool slab_hit_any(const DNode<creal>& n, const SlabRay& r, creal far_bound) {
    if (kCull32) return slab_hit_fast(n, r, (float)far_bound);
    return slab_hit(n, r, 0.001, (double)far_bound);
}

// Sphere::hit, shapes.rs:58-82.  The quadratic is evaluated in f64 in EVERY build: with the radius-1000 ground
// sphere of `random` (worlds.rs:83) |oc|^2 - r^2 cancels catastrophically in f32.  MI355X runs f64 FMA at half
// the f32 rate, so this is affordable; the fast build avoids the f64 divide and square-root sequences.
// The quadratic of Sphere::hit: discriminant >= 0, and its two roots evaluated exactly as sphere_hit's callers always did
// (near root first; the far root by the same expression with the sign of the square root flipped).
DEV bool sphere_roots(const SphereP& s, const Ray& r, double& near_root, double& far_root) {
#pragma clang fp contract(off)
    const double ocx = r.o.x - s.cx, ocy = r.o.y - s.cy, ocz = r.o.z - s.cz;
    const double radius = s.r;
    const double half_b = dfma(ocz, r.d.z, dfma(ocy, r.d.y, ocx * r.d.x));           // oc.dot(dir)
    const double c = dfma(ocz, ocz, dfma(ocy, ocy, ocx * ocx)) - radius * radius;   // oc.length_squared() - r*r
    const double discriminant = dfma(half_b, half_b, -(r.a * c));                   // half_b*half_b - a*c
    if (discriminant < 0.0) return false;
    double sqrtd;
    if (kFast) {
        // sqrt to 1e-14: f32 rsqrt seed (1e-7), one Newton step on y = 1/sqrt(x), s = x*y (the same argument as for 1/|d|^2)
        if (discriminant == 0.0) {
            sqrtd = 0.0;
        } else {
            double y = (double)__builtin_amdgcn_rsqf((float)discriminant);
            y = y * dfma(-0.5 * discriminant, y * y, 1.5);
            sqrtd = discriminant * y;
        }
        near_root = (-half_b - sqrtd) * r.inv_a;
        far_root = (-half_b + sqrtd) * r.inv_a;
    } else {
        sqrtd = sqrt(discriminant);
        Divisor(r.a).over(-half_b - sqrtd, -half_b + sqrtd, near_root, far_root);
    }
    return true;
}
DEV bool sphere_hit(const SphereP& s, const Ray& r, double tmin, double tmax, double& t_out) {
    double root, far_root;
    if (!sphere_roots(s, r, root, far_root)) return false;
    if (root < tmin || tmax < root) {
        root = far_root;
        if (root < tmin || tmax < root) return false;
    }
    t_out = root;
    return true;
}

// Block::hit = HittableList of six AARects (shapes.rs:172-192, hittable.rs:53-69): the nearest face hit inside
// [tmin, tmax].  The faces a line can hit inside their rectangles are the slab entry and exit faces, so the six
// rect tests collapse into one slab evaluation with the same t = (k - o)/d per face.
template <typename T>
DEV bool box_faces(const T (&lo)[3], const T (&hi)[3], double tmin, double tmax, double& t_out, int& axis_out) {
    T te = lo[0]; int ae = 0;
    if (lo[1] > te) { te = lo[1]; ae = 1; }
    if (lo[2] > te) { te = lo[2]; ae = 2; }
    T tx = hi[0]; int ax = 0;
    if (hi[1] < tx) { tx = hi[1]; ax = 1; }
    if (hi[2] < tx) { tx = hi[2]; ax = 2; }
    if (te > tx) return false;
    if ((double)te >= tmin && (double)te <= tmax) { t_out = (double)te; axis_out = ae; return true; }
    if ((double)tx >= tmin && (double)tx <= tmax) { t_out = (double)tx; axis_out = ax; return true; }
    return false;
}
// f64 builds: the reference's own t = (k - origin) / direction per face (aarects.rs:46) - the distance of an accepted hit must
// not carry a culling form's rounding
DEV bool box_hit_exact(const double (&b)[6], P3 o, P3 d, double tmin, double tmax, double& t_out, int& axis_out) {
    double lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double oa = compp(o, a), da = compp(d, a);
        double t0, t1;
        Divisor(da).over(b[a] - oa, b[3 + a] - oa, t0, t1);
        lo[a] = fmin(t0, t1); hi[a] = fmax(t0, t1);
    }
    return box_faces(lo, hi, tmin, tmax, t_out, axis_out);
}
// fast build: the culling form's t (one fma per face) from the entry's inline f32 bounds
DEV bool box_hit_fast(const DNode<creal>& n, const SlabRay& r, double tmin, double tmax, double& t_out, int& axis_out) {
    float lo[3], hi[3];
    const float of[3] = {(float)r.ofx, (float)r.ofy, (float)r.ofz}, iv[3] = {(float)r.ix, (float)r.iy, (float)r.iz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float t0 = __builtin_fmaf((float)n.v[a], iv[a], of[a]), t1 = __builtin_fmaf((float)n.v[3 + a], iv[a], of[a]);
        lo[a] = fminf(t0, t1); hi[a] = fmaxf(t0, t1);
    }
    return box_faces(lo, hi, tmin, tmax, t_out, axis_out);
}

// AARect::hit, aarects.rs:45-64
DEV bool rect_hit(const Scene& S, int idx, const DNode<creal>& n, const Ray& r, double tmin, double tmax, double& t_out) {
    const int a0 = (n.tag >> 4) & 3, a1 = (n.tag >> 6) & 3, ap = (n.tag >> 8) & 3;
    if (kFast) {
        const float t = ((float)n.v[4] - (float)compp(r.o, ap)) * fast_rcp((float)compp(r.d, ap));
        if ((double)t < tmin || (double)t > tmax) return false;
        const float a0_v = (float)compp(r.o, a0) + t * (float)compp(r.d, a0);
        const float a1_v = (float)compp(r.o, a1) + t * (float)compp(r.d, a1);
        if (a0_v < (float)n.v[0] || a0_v > (float)n.v[1] || a1_v < (float)n.v[2] || a1_v > (float)n.v[3]) return false;
        t_out = (double)t;
        return true;
    }
    double v[5];
    prim_params<5>(S, idx, n, v);
    const double t = (v[4] - compp(r.o, ap)) / compp(r.d, ap);
    if (t < tmin || t > tmax) return false;
    const double a0_v = compp(r.o, a0) + t * compp(r.d, a0);
    const double a1_v = compp(r.o, a1) + t * compp(r.d, a1);
    if (a0_v < v[0] || a0_v > v[1] || a1_v < v[2] || a1_v > v[3]) return false;
    t_out = t;
    return true;
}

// The primitive whose entry is `idx` (n = that entry) against the f64 ray.
template <uint32_t FEAT>
DEV bool prim_hit(const Scene& S, int idx, const DNode<creal>& n, const Ray& r, double tmin, double tmax, double& t_out, int& aux) {
    const int kind = n.tag & TAG_KIND_MASK;
    if (kind == TAG_SPHERE) return sphere_hit(sphere_params(S, idx, n), r, tmin, tmax, t_out);
    if ((FEAT & FEAT_BOX) && kind == TAG_BOX) {
        if (kFast) return box_hit_fast(n, slab_ray(r), tmin, tmax, t_out, aux);
        double b[6];
        prim_params<6>(S, idx, n, b);
        return box_hit_exact(b, r.o, r.d, tmin, tmax, t_out, aux);
    }
    if ((FEAT & FEAT_RECT) && kind == TAG_RECT) return rect_hit(S, idx, n, r, tmin, tmax, t_out);
    return false;  // TAG_EMPTY: shapes.rs:8-23
}

// The per-lane traversal cursor: one ray walking the linear entry array.
struct Cursor {
    int node;        // next entry to look at
    int xf;          // entry index of the open XFORM bracket or -1 (world space)
    double closest;  // closest_so_far (hittable.rs:56, bhv.rs:155-158)
    int hit_node;    // entry of the closest hit so far, -1 = none
    int hit_aux;     // box: face axis
    int hit_xf;      // bracket the hit was found in
    SlabRay sr;      // ray image for the slab tests in the current space
};

// Entries that are not plain box tests: primitives with f64 arithmetic, media, transform brackets.  Executed in
// batches (see the kernel loop), so the long code below is issued for many lanes at once.
template <uint32_t FEAT>
DEV void special_entry(const Scene& S, const DNode<creal>& n, Cursor& c, P3 o, P3 d, Rng& g, RngKey key) {
    const double t_min = 0.001;  // raytrace.rs:90
    const int kind = n.tag & TAG_KIND_MASK;
    if ((FEAT & FEAT_XFORM) && (kind == TAG_XFORM || kind == TAG_XFORM_END)) {
        c.xf = kind == TAG_XFORM ? c.node : -1;
        c.sr = slab_ray(local_ray<FEAT>(S, c.xf, o, d));
        c.node += slots_of(n.tag);
        return;
    }
    const Ray r = local_ray<FEAT>(S, c.xf, o, d);
    if ((FEAT & FEAT_MEDIUM) && kind == TAG_MEDIUM) {
        // ConstantMedium::hit, volumes.rs:26-65 (quirk Q9: the draw happens only when the clamped interval is non-empty)
        const int b_idx = c.node + slots_of(n.tag);
        const DNode<creal> boundary = S.nodes[b_idx];
        double t1, t2; int aux = 0;
        bool through;
        if ((boundary.tag & TAG_KIND_MASK) == TAG_SPHERE) {
            // boundary.hit(-inf, inf) is the near root and boundary.hit(t1 + 0.001, inf) the far one when it is not below
            // t1 + 0.001 (shapes.rs:66-73): ONE evaluation of the quadratic instead of two identical ones
            through = sphere_roots(sphere_params(S, b_idx, boundary), r, t1, t2) && !(t2 < t1 + 0.001);
        } else {
            through = prim_hit<FEAT>(S, b_idx, boundary, r, -(double)INFINITY, (double)INFINITY, t1, aux) &&
                      prim_hit<FEAT>(S, b_idx, boundary, r, t1 + 0.001, (double)INFINITY, t2, aux);
        }
        if (through) {
            t1 = fmax(t1, t_min);
            t2 = fmin(t2, c.closest);
            if (t1 < t2) {
                t1 = fmax(t1, 0.0);
                // r.dir.length() of the WORLD ray and its reciprocal; fast build: f32 rsqrt seed + Newton instead of the f64
                // square root and division sequences
                const double dd = d.x * d.x + d.y * d.y + d.z * d.z;
                double ray_scale, inv_ray_scale = 0.0;
                if (kFast) {
                    double y = (double)__builtin_amdgcn_rsqf((float)dd);
                    y = y * dfma(-0.5 * dd, y * y, 1.5);  // 1e-14
                    inv_ray_scale = y;
                    ray_scale = dd * y;
                } else {
                    ray_scale = sqrt(dd);
                }
                const double distance_inside = (t2 - t1) * ray_scale;
                const double xi = u01d(rng_next(g, key));
                double density[1];
                prim_params<1>(S, c.node, n, density);  // neg_inv_density
                const double hit_distance = density[0] * (kFast ? (double)(__builtin_amdgcn_logf((float)xi) * 0.693147180559945f) : log_uniform(xi));  // (v_log_f32 is log2; xi = 0 gives -inf: no hit, as with ln)
                if (!(hit_distance > distance_inside)) {
                    c.closest = kFast ? dfma(hit_distance, inv_ray_scale, t1) : t1 + hit_distance / ray_scale;
                    c.hit_node = c.node; c.hit_aux = 0; c.hit_xf = c.xf;
                }
            }
        }
        c.node = b_idx + slots_of(boundary.tag);
        return;
    }
    double t; int aux = 0;
    if (prim_hit<FEAT>(S, c.node, n, r, t_min, c.closest, t, aux)) { c.closest = t; c.hit_node = c.node; c.hit_aux = aux; c.hit_xf = c.xf; }
    c.node += slots_of(n.tag);
}

// ---------------------------------------------------------------------------------------------- textures
DEV real perlin_noise(const Scene& S, int table, V3 p) {  // textures.rs:90-134
    const real fx = floor_r(p.x), fy = floor_r(p.y), fz = floor_r(p.z);
    const real u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const real uu = u * u * ((real)3 - (real)2 * u), vv = v * v * ((real)3 - (real)2 * v), ww = w * w * ((real)3 - (real)2 * w);
    const real* vec = S.perlin_vec + (size_t)table * 3072;
    const uint16_t* perm = S.perlin_perm + (size_t)table * 3072;
    real accum = (real)0;
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
        for (int dj = 0; dj < 2; dj++)
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                const int idx = perm[(i + di) & 1023] ^ perm[1024 + ((j + dj) & 1023)] ^ perm[2048 + ((k + dk) & 1023)];
                const V3 c = mk(vec[idx * 3], vec[idx * 3 + 1], vec[idx * 3 + 2]);
                const V3 weight = mk(u - (real)di, v - (real)dj, w - (real)dk);
                accum += ((real)di * uu + (real)(1 - di) * ((real)1 - uu)) * ((real)dj * vv + (real)(1 - dj) * ((real)1 - vv)) *
                         ((real)dk * ww + (real)(1 - dk) * ((real)1 - ww)) * dot(weight, c);
            }
    return accum;
}
DEV real perlin_turbulence(const Scene& S, int table, V3 p) {  // textures.rs:76-88, depth 7
    real accum = (real)0, weight = (real)1;
    V3 tp = p;
    for (int i = 0; i < 7; i++) {
        accum += weight * perlin_noise(S, table, tp);
        weight *= (real)0.5;
        tp = (real)2 * tp;
    }
    return abs_r(accum);
}
// A scene of SolidColor textures only has the colour in the material (`solid`) and no texture table (mlrt_api.cpp pack).
// The same turbulence for ONE point with the whole wave: its 7 octaves x 8 lattice corners are 56 independent terms, one
// per lane, summed by a butterfly.  A marble hit is rare (0.07 per sample of final_scene) but half of the shading batches
// have one, and the serial evaluation above is ~1000 instructions the other lanes of the batch wait through; this is ~80.
// Call in wave-uniform control flow with a uniform point.  Fast build only: the order of the sum differs from
// textures.rs:76-134 in the last bits (the parity build keeps the serial order).
DEV float wave_turbulence(const Scene& S, int table, float px, float py, float pz, int lane) {
    const int octave = lane >> 3;
    const int di = (lane >> 2) & 1, dj = (lane >> 1) & 1, dk = lane & 1;
    const float s = (float)(1 << (octave & 7));
    const float qx = px * s, qy = py * s, qz = pz * s;  // textures.rs:84: p doubles per octave
    const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
    const float u = qx - fx, v = qy - fy, w = qz - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
    const real* vec = S.perlin_vec + (size_t)table * 3072;
    const uint16_t* perm = S.perlin_perm + (size_t)table * 3072;
    const int idx = perm[(i + di) & 1023] ^ perm[1024 + ((j + dj) & 1023)] ^ perm[2048 + ((k + dk) & 1023)];
    const float cx = (float)vec[idx * 3], cy = (float)vec[idx * 3 + 1], cz = (float)vec[idx * 3 + 2];
    const float wx = u - (float)di, wy = v - (float)dj, wz = w - (float)dk;
    float term = (di ? uu : 1.0f - uu) * (dj ? vv : 1.0f - vv) * (dk ? ww : 1.0f - ww) * (wx * cx + wy * cy + wz * cz);
    term = octave < 7 ? term / s : 0.0f;  // textures.rs:83: the weight halves per octave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) term += __shfl_xor(term, off);
    return fabsf(term);
}
template <uint32_t FEAT>
DEV V3 texture_value(const Scene& S, int tex, V3 solid, real u, real v, V3 p, bool have_turbulence = false, real turbulence = (real)0) {
    if ((FEAT & (FEAT_TEX_IMAGE | FEAT_TEX_NOISE | FEAT_TEX_CHECKER)) == 0) return solid;
    DTexture<real> t = S.textures[tex];
    bool direct = true;  // t is the material's own texture
    for (int guard = 0; (FEAT & FEAT_TEX_CHECKER) && guard < 8 && t.kind == MLRT_TEX_CHECKER; guard++) {  // textures.rs:40-49
        const real sines = sin_r((real)5 * p.x) * sin_r((real)5 * p.y) * sin_r((real)5 * p.z);
        t = S.textures[sines < (real)0 ? t.a : t.b];
        direct = false;
    }
    if (t.kind == MLRT_TEX_SOLID) return ld3(t.color);
    if ((FEAT & FEAT_TEX_NOISE) && t.kind == MLRT_TEX_NOISE) {  // textures.rs:163-167
        // (have_turbulence: only for a NoiseTexture that is the material's own texture, not one behind a Checker)
        const real turb = (have_turbulence && direct) ? turbulence : perlin_turbulence(S, t.a, t.scale * p);
        const real x = (real)1 + sin_r(t.scale * p.z + (real)10 * turb);
        return mk((real)0.5 * x, (real)0.5 * x, (real)0.5 * x);
    }
    if ((FEAT & FEAT_TEX_IMAGE) && t.kind == MLRT_TEX_IMAGE) {  // image_texture.rs:16-28
        const DImage im = S.images[t.a];
        const real uc = min_r(max_r(u, (real)0), (real)1);
        const real vc = min_r(max_r((real)1 - v, (real)0), (real)1);
        unsigned int i = (unsigned int)(uc * (real)im.width);
        unsigned int j = (unsigned int)(vc * (real)im.height);
        i = min(i, (unsigned int)im.width - 1u);
        j = min(j, (unsigned int)im.height - 1u);
        const uint8_t* px = im.rgb + ((size_t)j * (size_t)im.width + i) * 3;
        return mk((real)px[0] / (real)255, (real)px[1] / (real)255, (real)px[2] / (real)255);
    }
    return mk(0, 0, 0);
}

// ---------------------------------------------------------------------------------------------- shading
DEV V3 reflect(V3 v, V3 n) { return v - ((real)2 * dot(v, n)) * n; }  // materials.rs:47-49
DEV V3 refract(V3 uv, V3 n, real etai_over_etat) {                    // materials.rs:63-68
    const real cos_theta = min_r(dot(-uv, n), (real)1);
    const V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    const V3 r_out_parallel = (-sqrt_r(abs_r((real)1 - length_squared(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}
DEV real reflectance(real cos_theta, real refraction_ratio) {  // materials.rs:81-86
    const real r0 = div_r((real)1 - refraction_ratio, (real)1 + refraction_ratio);
    const real r0_sq = r0 * r0;
    const real x = (real)1 - cos_theta;
    const real x2 = x * x;
    return r0_sq + ((real)1 - r0_sq) * (x * (x2 * x2));  // powi(5)
}
MLRT_COLD_UV void sphere_uv(V3 n, real& u, real& v) {  // shapes.rs:44-56
    const double pi = 3.14159265358979323846264338327950288;
    if (kFast) {
        const float theta = acosf(-(float)n.y);
        const float phi = atan2f(-(float)n.z, (float)n.x) + (float)pi;
        u = (real)(phi * (float)(1.0 / (2.0 * pi)));
        v = (real)(theta * (float)(1.0 / pi));
    } else {
        const double theta = acos(-(double)n.y);
        const double phi = atan2(-(double)n.z, (double)n.x) + pi;
        u = (real)(phi / (2.0 * pi));
        v = (real)(theta / pi);
    }
}

// Ray::at (vec.rs): origin + t * direction; one fma per component in the fast build
DEV P3 ray_at(P3 o, P3 d, double t) {
    if (kFast) return mkp(__builtin_fma(t, d.x, o.x), __builtin_fma(t, d.y, o.y), __builtin_fma(t, d.z, o.z));
    return o + t * d;
}
// Shading in two halves.  hit_attributes: the closest hit the cursor found (point, normal, uv, material).  scatter: the
// material response; returns true when the path continues with (o, d) replaced and `thr` multiplied by the attenuation,
// false when it ends with radiance `emitted`.  Between the two the batch evaluates, for the whole wave at once, what a
// few of its lanes need and is long (wave_turbulence).
struct HitAttr {
    P3 p;
    V3 normal;
    real u, v;
    bool front_face;
    DMaterial<real> m;
};
template <uint32_t FEAT, bool PAIR = false>
DEV HitAttr hit_attributes(const Scene& S, const Cursor& c, P3 o, P3 d) {
    if constexpr (PAIR) {  // pair walk: c.hit_node is a primitive index, every primitive a sphere with a solid-colour material
        const SphereP sp = pair_sphere(S, c.hit_node);
        const P3 po = ray_at(o, d, c.closest);
        V3 outward;
        if (kFast) outward = to_v3(po - mkp(sp.cx, sp.cy, sp.cz)) * fast_rcp((real)sp.r);
        else {
            double nx, ny, nz;
            Divisor(sp.r).over(po.x - sp.cx, po.y - sp.cy, po.z - sp.cz, nx, ny, nz);
            outward = mk((real)nx, (real)ny, (real)nz);
        }
        HitAttr h;
        h.front_face = dot(outward, to_v3(d)) < (real)0;  // Hit::new_with_face_normal, hittable.rs:18-30
        h.normal = h.front_face ? outward : -outward;
        h.p = po; h.u = (real)0; h.v = (real)0;
        h.m = S.materials[S.prim_mat[c.hit_node]];
        return h;
    }
    const DNode<creal> n = S.nodes[c.hit_node];
    const int kind = n.tag & TAG_KIND_MASK;
    P3 p;
    V3 normal;
    real u = (real)0, v = (real)0;
    bool front_face = true;
    const int mat_id = n.link;
    const V3 dw = to_v3(d);  // world direction in shading precision
    if ((FEAT & FEAT_MEDIUM) && kind == TAG_MEDIUM) {  // volumes.rs:55-63
        p = ray_at(o, d, c.closest);
        normal = mk((real)1, (real)0, (real)0);
    } else {
        P3 oo = o, od = d;
        XfP xf;
        xf.tag = 0; xf.sin_t = 0.0; xf.cos_t = 1.0; xf.ox = xf.oy = xf.oz = 0.0;
        if ((FEAT & FEAT_XFORM) && c.hit_xf >= 0) { xf = xform_params(S, c.hit_xf); to_object(xf, oo, od); }
        const P3 po = ray_at(oo, od, c.closest);
        const V3 odf = to_v3(od);
        V3 outward = mk(0, 0, 1);
        if (kind == TAG_SPHERE) {
            const SphereP sp = sphere_params(S, c.hit_node, n);
            if (kFast) outward = to_v3(po - mkp(sp.cx, sp.cy, sp.cz)) * fast_rcp((real)sp.r);
            else {
                double nx, ny, nz;
                Divisor(sp.r).over(po.x - sp.cx, po.y - sp.cy, po.z - sp.cz, nx, ny, nz);
                outward = mk((real)nx, (real)ny, (real)nz);
            }
            if ((FEAT & (FEAT_TEX_IMAGE | FEAT_TEX_CHECKER)) && (S.materials[mat_id].kind & MATFLAG_NEEDS_UV)) sphere_uv(outward, u, v);
        } else if ((FEAT & FEAT_BOX) && kind == TAG_BOX) {
            outward = axis_vec(c.hit_aux, (real)1);
            if ((FEAT & (FEAT_TEX_IMAGE | FEAT_TEX_CHECKER)) && (S.materials[mat_id].kind & MATFLAG_NEEDS_UV)) {
                // the face is one of the Block's six rects (shapes.rs:172-192): u, v over its two in-plane axes, in axis order (aarects.rs:56-57)
                double b[6];
                prim_params<6>(S, c.hit_node, n, b);
                const int a0 = c.hit_aux == 0 ? 1 : 0, a1 = c.hit_aux == 2 ? 1 : 2;
                u = div_r((real)compp(po, a0) - (real)b[a0], (real)b[3 + a0] - (real)b[a0]);
                v = div_r((real)compp(po, a1) - (real)b[a1], (real)b[3 + a1] - (real)b[a1]);
            }
        } else if (FEAT & FEAT_RECT) {
            const int a0 = (n.tag >> 4) & 3, a1 = (n.tag >> 6) & 3, ap = (n.tag >> 8) & 3;
            outward = axis_vec(ap, (real)1);
            if ((FEAT & (FEAT_TEX_IMAGE | FEAT_TEX_CHECKER)) && (S.materials[mat_id].kind & MATFLAG_NEEDS_UV)) {  // (aarects.rs:56-57; only a texture reads them)
                double rv[5];
                prim_params<5>(S, c.hit_node, n, rv);
                u = div_r((real)compp(po, a0) - (real)rv[0], (real)rv[1] - (real)rv[0]);
                v = div_r((real)compp(po, a1) - (real)rv[2], (real)rv[3] - (real)rv[2]);
            }
        }
        front_face = dot(outward, odf) < (real)0;  // Hit::new_with_face_normal, hittable.rs:18-30
        normal = front_face ? outward : -outward;
        p = po;
        if ((FEAT & FEAT_XFORM) && c.hit_xf >= 0) {
            if (xf_rotates(xf)) {  // Rotate::hit, transforms.rs:134-139 (quirk Q10: world normal vs object-space ray)
                p = rotate_fwd(xf, p);
                const V3 nw = to_v3(rotate_fwd(xf, to_p3(normal)));
                front_face = dot(nw, odf) < (real)0;
                normal = front_face ? nw : -nw;
            }
            if (xf_translates(xf)) {  // Translate::hit, transforms.rs:36-39 (moved_r.dir == r.dir)
                p = p + mkp(xf.ox, xf.oy, xf.oz);
                front_face = dot(normal, dw) < (real)0;
                normal = front_face ? normal : -normal;
            }
        }
    }
    HitAttr h;
    h.p = p; h.normal = normal; h.u = u; h.v = v; h.front_face = front_face;
    h.m = S.materials[mat_id];
    return h;
}
// `turbulence`: the NoiseTexture turbulence of the hit point when the batch has already evaluated it (have_turbulence).
// The colour a material's texture contributes at the hit (Lambertian / Isotropic albedo, DiffuseLight emission; materials.rs:25-34, 119-127,
// volumes.rs:77-83): evaluated BEFORE scatter, while only the hit attributes are live - inside scatter's branches the Perlin code's
// registers came on top of the sampler's and the new ray's.
template <uint32_t FEAT>
DEV V3 material_texture(const Scene& S, const HitAttr& h, bool have_turbulence, real turbulence) {
    const int kind = h.m.kind & 0xFF;
    if ((FEAT & (FEAT_TEX_IMAGE | FEAT_TEX_NOISE | FEAT_TEX_CHECKER)) == 0 || kind == MLRT_MAT_METAL || kind == MLRT_MAT_DIELECTRIC) return ld3(h.m.albedo);
    return texture_value<FEAT>(S, h.m.tex, ld3(h.m.albedo), h.u, h.v, to_v3(h.p), have_turbulence, turbulence);
}
template <uint32_t FEAT>
DEV bool scatter(const Scene& S, const HitAttr& h, V3 texture_colour, P3& o, P3& d, V3& thr, V3& emitted, Rng& g, RngKey key, int bounce) {
    const P3 p = h.p;
    const V3 normal = h.normal;
    const bool front_face = h.front_face;
    const DMaterial<real> m = h.m;
    const V3 dw = to_v3(d);  // world direction in shading precision
    const int mk_kind = m.kind & 0xFF;
    // Lambertian, Metal and Isotropic all begin their RNG use with one random_in_unit_sphere (materials.rs:27,54,
    // volumes.rs:79) and Dielectric may draw one word: the scatter's block, evaluated once for the whole batch.
    const bool three = mk_kind == MLRT_MAT_LAMBERTIAN || mk_kind == MLRT_MAT_METAL || ((FEAT & FEAT_MEDIUM) && mk_kind == MLRT_MAT_ISOTROPIC);
    uint32_t w0, w1, w2;
    rng_scatter_words<FEAT>(g, key, bounce, w0, w1, w2);
    V3 s = mk(0, 0, 0);
    if (three) s = unit_ball_from_words(w0, w1, w2);
    // (measured and not kept: the unit direction and its reflection, which Metal and Dielectric both start from, evaluated once in front of
    // the two branches - fewer instructions per batch on paper, six more live registers in practice: C2 f64 +2.7 %, f32 +1.9 %)
    if (mk_kind == MLRT_MAT_LAMBERTIAN) {  // materials.rs:25-34 (quirk Q1)
        if (!(dot(normal, s) > (real)0)) s = -s;  // random_in_hemisphere, vec.rs:36-43
        V3 dir = normal + s;
        const real eps = (real)1e-8;
        if (abs_r(dir.x) < eps && abs_r(dir.y) < eps && abs_r(dir.z) < eps) dir = normal;
        thr = thr * texture_colour;
        o = p; d = to_p3(dir);
        return true;
    }
    // (measured and not kept, twice: the unit direction and its reflection, which Metal and Dielectric both start from, evaluated once for the
    // two kinds - in front of the branches, or in a branch of their own around them.  A batch issues the code of every kind it holds, so it
    // is ~45 instructions fewer per batch on paper; in practice the sphere-only f64 kernel, at 127 of 128 registers, starts to spill
    // (20 B of scratch): C2 f64 +2.6 %.  Schlick's r0^2 from the packer, one of two values per material: +0.7 %, same cause.)
    if (mk_kind == MLRT_MAT_METAL) {  // materials.rs:51-61 (quirk Q7)
        const V3 reflected = reflect(unit(dw), normal);
        const V3 dir = reflected + m.param * s;
        if (dot(dir, normal) > (real)0) {
            thr = thr * ld3(m.albedo);
            o = p; d = to_p3(dir);
            return true;
        }
        emitted = mk(0, 0, 0);  // Material::emit default, materials.rs:9-11
        return false;
    }
    if (mk_kind == MLRT_MAT_DIELECTRIC) {  // materials.rs:88-106 (quirk Q8)
        const real refraction_ratio = !front_face ? m.param : m.albedo[0];  // 1.0 / ir, divided once by the packer (mlrt_api.cpp pack_materials)
        const V3 unit_direction = unit(dw);
        const real cos_theta = min_r(dot(normal, -unit_direction), (real)1);
        const real sin_theta = sqrt_r((real)1 - cos_theta * cos_theta);
        const bool cannot_refract = refraction_ratio * sin_theta > (real)1;
        V3 dir;
        bool reflects = cannot_refract;
        if (!cannot_refract) {  // the draw happens only on this side of the `||` (materials.rs:98)
            reflects = reflectance(cos_theta, refraction_ratio) > u01(w0);
        }
        if (reflects) dir = reflect(unit_direction, normal);
        else dir = refract(unit_direction, normal, refraction_ratio);
        o = p; d = to_p3(dir);
        return true;
    }
    if ((FEAT & FEAT_MEDIUM) && mk_kind == MLRT_MAT_ISOTROPIC) {  // volumes.rs:77-83
        const V3 dir = s;
        thr = thr * texture_colour;
        o = p; d = to_p3(dir);
        return true;
    }
    emitted = texture_colour;  // DiffuseLight, materials.rs:119-127
    return false;
}

DEV V3 background_color(const KParams<real>& P, const FrameParams& F, P3 d) {  // raytrace.rs:29-35, 44-48
    if (P.bg_kind == MLRT_BG_BLACK) return mk(0, 0, 0);
    const V3 ud = unit(to_v3(d));
    const real t = (real)0.5 * (ud.y + (real)1);
    return ((real)1 - t) * ld3(F.bg_bottom) + t * ld3(F.bg_top);
}

// KParams::ray_table read from the kernel-argument segment at its point of use, and only when bit 1 of KParams::has_lens announces a
// table (a test instrument must not cost the render kernels a scalar register pair, nor a load per regeneration)
DEV const double* ray_table_pointer() {
#if defined(__HIP_DEVICE_COMPILE__)
    const char* args = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const double* const volatile*>(args + offsetof(KParams<real>, ray_table));
#else
    return nullptr;
#endif
}
// SingleLightSourceRayTracer::trace after a scatter (raytrace.rs:119-126): Lambert and Blinn-Phong terms towards one point light.
// `view` = -ray.dir.unit() of the ray that hit, taken before the scatter replaced it; `attenuation` = the scatter's.  The light is read
// from the kernel-argument segment here (KParams::single_light): the path tracer's frames do not carry it in registers.  The f64 builds
// keep the reference's operation order (powi(100) as compiler-rt's square-and-multiply from the low bit); the f32 build computes in f32.
DEV V3 single_light_colour(V3 attenuation, const HitAttr& hit, V3 view) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double* q = reinterpret_cast<const double*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KParams<real>, single_light));
    const double lx = q[0], ly = q[1], lz = q[2];
    const real intensity = (real)q[3];
#else
    const double lx = 0, ly = 0, lz = 0;
    const real intensity = 0;
#endif
    const V3 l = unit(to_v3(mkp(lx, ly, lz) - hit.p));
    const V3 h = unit(l + view);
    const real kd = dot(l, hit.normal), ks = dot(h, hit.normal);
    real a = ks > (real)0 ? ks : (real)0, r = (real)1;  // f64::max(x, 0.0): a NaN becomes 0
#pragma unroll
    for (int n = 100;;) {
        if (n & 1) r *= a;
        n /= 2;
        if (n == 0) break;
        a *= a;
    }
    const V3 lambertian = (attenuation * intensity) * (kd > (real)0 ? kd : (real)0);
    const V3 blinn_phong = (mk((real)0.5, (real)0.5, (real)0.5) * intensity) * r;
    return lambertian + blinn_phong;
}
// raytrace.rs:191-193 + Camera::get_ray, camera.rs:40-48, in f64 in both builds: the uniforms are 24-bit values
// exact in either precision, so the primary ray is bit-identical to the oracle's (parity build; the fast build
// evaluates the lens offset's sin/cos in f32).  A fresh sample's four draws - u, v and the lens sample, drawn by
// inverse CDF from two words (r = sqrt(u1'), phi = 2*pi*u2; the reference's rejection loop is vec.rs:45-52, quirk Q6:
// the draws are consumed even when lens_radius == 0) - are exactly Philox block 0: no loop, no refill branch.
// f64 builds: F.inv_wm1 / F.inv_hm1 become the device's own refined reciprocals of width - 1 and height - 1 (Divisor), not the host's
// 1 / x: gen_camera_ray divides by them with the tail of the compiler's division sequence.  Called by the threads that filled F.
DEV void frame_reciprocals(const KParams<real>& P, FrameParams& F) {
    if (!kFast && threadIdx.x < 2) (&F.inv_wm1)[threadIdx.x] = Divisor((double)(threadIdx.x == 0 ? P.width : P.height) - 1.0).y;
}
DEV void gen_camera_ray(const KParams<real>& P, const FrameParams& F, Rng& g, RngKey key, int i, int j, P3& o, P3& d) {
    uint32_t b0, b1, b2, b3;
    philox4x32_10(key.k0, key.k1, 0u, g.sample, g.pixel, 0u, b0, b1, b2, b3);
    g.draw = 4u;
    if (kFast) {
        // Fast build: the primary ray in f32 (the f64 sequence below is ~45 double-rate instructions per sample; sub-pixel
        // position to 2^-13 pixel and direction to 1e-7 are far inside what a path tracer resolves, and inside the margins of
        // the primary candidate lists); the origin stays exact.
        const float sf = ((float)i + (float)(b0 >> 8) * (float)(1.0 / 16777216.0)) * (float)F.inv_wm1;
        const float tf = ((float)j + (float)(b1 >> 8) * (float)(1.0 / 16777216.0)) * (float)F.inv_hm1;
        float dx = F.through0[0] + sf * F.hor_f[0] + tf * F.ver_f[0];
        float dy = F.through0[1] + sf * F.hor_f[1] + tf * F.ver_f[1];
        float dz = F.through0[2] + sf * F.hor_f[2] + tf * F.ver_f[2];
        o = ld3p(F.cam_origin);
        if (P.has_lens & 1) {
            const float u1 = ((float)(b2 >> 8) + 1.0f) * (float)(1.0 / 16777216.0);
            const float turn = (float)(b3 >> 8) * (float)(1.0 / 16777216.0);
            const float r = (float)F.lens_radius * __builtin_amdgcn_sqrtf(u1);
            const float rdx = r * __builtin_amdgcn_cosf(turn), rdy = r * __builtin_amdgcn_sinf(turn);
            const float ox = F.u_f[0] * rdx + F.v_f[0] * rdy, oy = F.u_f[1] * rdx + F.v_f[1] * rdy, oz = F.u_f[2] * rdx + F.v_f[2] * rdy;
            o = o + mkp((double)ox, (double)oy, (double)oz);
            dx -= ox; dy -= oy; dz -= oz;
        }
        d = mkp((double)dx, (double)dy, (double)dz);
        return;
    }
    // (fast build: times the reciprocal - an f64 division is ~25 instructions, twice per sample)
    // f64 builds: the division of raytrace.rs:191-192 over the frame's two constant divisors (1 .. 2^31: always inside Divisor's domain),
    // whose refined reciprocals frame_reciprocals() left in F
    const double s = kFast ? ((double)i + u01d(b0)) * F.inv_wm1 : Divisor((double)P.width - 1.0, F.inv_wm1).one((double)i + u01d(b0));
    const double t = kFast ? ((double)j + u01d(b1)) * F.inv_hm1 : Divisor((double)P.height - 1.0, F.inv_hm1).one((double)j + u01d(b1));
    const P3 origin = ld3p(F.cam_origin);
    const P3 through = ld3p(F.cam_llc) + s * ld3p(F.cam_horizontal) + t * ld3p(F.cam_vertical) - origin;
    if (P.has_lens & 1) {  // wave-uniform; with a pinhole the offset is exactly zero whatever the lens sample is: nothing to add
        const double u1 = ((double)(b2 >> 8) + 1.0) * (1.0 / 16777216.0);
        double cs, sn;
        if (kFast) { const double turn = u01d(b3); cs = (double)__builtin_amdgcn_cosf((float)turn); sn = (double)__builtin_amdgcn_sinf((float)turn); }
        else turn_cos_sin(b3 >> 8, cs, sn);
        const double r = sqrt(u1);
        const double rdx = F.lens_radius * (r * cs);
        const double rdy = F.lens_radius * (r * sn);
        const P3 offset = ld3p(F.cam_u) * rdx + ld3p(F.cam_v) * rdy;
        o = origin + offset;
        d = through - offset;
    } else {
        o = origin;
        d = through;
    }
}

// 2^-32 fixed point; integer accumulation is order independent (within a tile in LDS, across tiles in HBM).  NaN/negative radiance cannot occur on this
// path (near_zero guard, abs() under the refraction root); they are clamped to 0 rather than poisoning the sum.
DEV unsigned long long to_fixed(double c) {
    double x = c;
    if (!(x > 0.0)) x = 0.0;
    if (x > 1048576.0) x = 1048576.0;
    return (unsigned long long)(x * 4294967296.0 + 0.5);
}
// Fast build: the same fixed point from f32 with five 32-bit instructions per channel instead of the f64 -> u64 sequence
// (three conversions per finished sample): whole part and fraction converted separately, both exact (c - trunc(c) and the
// scaling by 2^32 are exact in f32), truncating the bits below 2^-32.  v_cvt_u32_f32 maps NaN and negatives to 0 and
// saturates, which is the clamp.
DEV unsigned long long to_fixed(float c) {
    const uint32_t whole = (uint32_t)c;
    const uint32_t frac = (uint32_t)((c - (float)whole) * 4294967296.0f);
    return ((unsigned long long)whole << 32) | (unsigned long long)frac;
}

#ifndef MLRT_BOX_SPECIAL
#define MLRT_BOX_SPECIAL 0
#endif
// Batch thresholds (lanes parked on special entries / on shading before the batch runs), per variant: measured optima
// (A/B in one process): sphere-only variant 24 / 40 (C2: 45.1 ms at 32 / 32, 43.7 at 24 / 40: its special batch is short,
// its shading batch long), full variant 32 / 32 (final_scene: 24 / 32 -1.5 %, 32 / 40 and 40 / 32 -5 %).  The smaller feature subsets -
// every reference world but final_scene - take 40 / 40 (MLRT_T_*_SMALL): cornell_smoke -3 %, random_chk f64 -2.6 %, cornell_box -1.1 %,
// the others -0.5 %, where final_scene loses 6 % (profiles/r03_ab_instruction_trims.txt).
#ifndef MLRT_T_SPECIAL
#define MLRT_T_SPECIAL 32
#endif
#ifndef MLRT_T_SHADE
#define MLRT_T_SHADE 32
#endif
#ifndef MLRT_T_SPECIAL_SMALL
#define MLRT_T_SPECIAL_SMALL 40
#endif
#ifndef MLRT_T_SHADE_SMALL
#define MLRT_T_SHADE_SMALL 40
#endif
// (a variant with both media and image textures is final_scene's, or the one with every feature)
constexpr bool big_world_variant(uint32_t feat) { return (feat & (FEAT_MEDIUM | FEAT_TEX_IMAGE)) == (FEAT_MEDIUM | FEAT_TEX_IMAGE); }
#ifndef MLRT_T_SPECIAL_LEAN
#define MLRT_T_SPECIAL_LEAN 24
#endif
#ifndef MLRT_T_SHADE_LEAN
#define MLRT_T_SHADE_LEAN 40
#endif
enum { PH_TRAV = 0, PH_SPECIAL = 1, PH_SHADE = 2, PH_IDLE = 3, PH_WAIT = 4 };
static_assert(PH_TRAV == 0 && PH_SPECIAL == 1, "the traversal step computes the phase from the tag's sign bit");  // WAIT: no sample to be had until a bank of the ring drains

// The traversal is latency bound (rocprof: waves wait on LDS 40 % of their cycles at 4 waves per SIMD), so the entry for
// a lane's NEXT step is fetched in ONE round trip (both 16-byte halves, all lanes, clamped index) right after the cursor
// moved, and the batch bookkeeping runs while it is in flight.  pin_entry marks where it must have arrived, which also
// stops the compiler from sinking the loads behind the tag test (two dependent round trips per step).
DEV DNode<creal> fetch_entry(const Scene& S, int node, int n_nodes) { return S.nodes[node]; }  // node <= n_nodes: the sentinel
DEV void pin_entry(const DNode<creal>& n) {
    asm volatile("" : : "v"(n.v[0]), "v"(n.v[1]), "v"(n.v[2]), "v"(n.v[3]), "v"(n.v[4]), "v"(n.v[5]), "v"(n.link), "v"(n.tag));
}
// pair walk: the node a lane visits next (four ds_read_b128 in one round trip); a lane parked on a leaf or finished reads node 0
DEV DPair fetch_pair(const Scene& S, int ref) { return S.pairs[ref < 0 ? 0 : ref]; }
DEV void pin_entry(const DPair& n) {
    asm volatile("" : : "v"(n.a[0]), "v"(n.a[1]), "v"(n.a[2]), "v"(n.a[3]), "v"(n.a[4]), "v"(n.a[5]), "v"(n.b[0]), "v"(n.b[1]), "v"(n.b[2]), "v"(n.b[3]), "v"(n.b[4]), "v"(n.b[5]),
                 "v"(n.ref_a), "v"(n.ref_b));
}
template <bool PAIR> struct EntryOf { typedef DNode<creal> type; };
template <> struct EntryOf<true> { typedef DPair type; };
#ifndef MLRT_PAIR_SPECIAL_ROUNDS
#define MLRT_PAIR_SPECIAL_ROUNDS 1
#endif
// Traversal steps between two looks at the batches (MLRT_STEPS for the threaded list).  A pair visit does the work of ~2.3 list
// steps, so the batches want looking at twice as often (measured, C2 kernel ms f64 / f32: 4 steps 53.5 / 37.8, 2 steps 52.3 / 36.4,
// 3 steps in between; the threaded list on the same box 55.6 / 38.5).
#ifndef MLRT_PAIR_STEPS
#define MLRT_PAIR_STEPS 2
#endif

// Primary rays of a pixel group (1-4 neighbouring pixels; pinhole camera, for a thin lens see below) all lie inside the
// pyramid from the camera origin through the group's footprint on the image plane, so they can only hit the spheres
// that touch that pyramid: typically the ground and one or two others.  A pre-pass, once per frame, tests the sphere
// entries against the four side planes (outward margins) and leaves the touching ones, in entry order, in a 32-byte
// record per group: u16 count (kCandInvalid when more than kCand touch: that group's samples walk the BVH as usual), the
// entry indices, and the group's image row.  A wave opening a tile copies the record into the tile's bank, and a fresh
// sample tests exactly those entries instead of walking the BVH (43 % of all traversal steps of `random` are
// primary-ray steps).  The closest hit is the one the walk finds: the walk visits sphere entries in the same order, the
// list is a superset of the spheres any of the group's rays can hit, and Sphere::hit does not depend on what was tested
// before it except through closest-so-far (GPU test: the image equals the one rendered with MLRT_NO_PRIMARY_LISTS).
constexpr int kCandRecord = 16;  // u16 per record: count + kCand entries + row
constexpr int kCoarseMax = 256;
// Two levels, one wave per 64 consecutive pixel groups of a row: the wave first scans the entry array against the
// pyramid of the whole 64-group span (64 entries per round, survivors compacted into LDS in entry order: typically 10-30),
// then every lane tests only those against its own group's pyramid.  (A flat scan of all entries per group cost 0.5 ms per
// C2 frame.)  A span with more than kCoarseMax survivors falls back to the flat scan.
template <bool PAIR>
__global__ __launch_bounds__(64) void mlrt_primary_lists_kernel(const KParams<real> P, uint16_t* records) {
    __shared__ uint16_t coarse[kCoarseMax];
    const int lane = threadIdx.x;
    const DNode<creal>* nodes = reinterpret_cast<const DNode<creal>*>(P.blob + P.off_nodes);
    const int spans_per_row = (P.tiles_per_row + 63) >> 6;
    const int row_k = (int)blockIdx.x / spans_per_row;
    const int gx0 = ((int)blockIdx.x - row_k * spans_per_row) << 6;
    const int gx = gx0 + lane;
    const int j = P.rows ? P.rows[row_k] : row_k;
    const double inv_w = 1.0 / ((double)P.width - 1.0), inv_h = 1.0 / ((double)P.height - 1.0);
    const P3 org = ld3p(P.cam_origin);
    const P3 base = ld3p(P.cam_llc) - org;
    const P3 hor = ld3p(P.cam_horizontal), ver = ld3p(P.cam_vertical);
    const V3 of = mk((real)org.x, (real)org.y, (real)org.z);
    // Thin lens (camera.rs:41-47): a ray leaves origin + delta, |delta| <= lens_radius, towards the point its pinhole twin hits
    // on the focus plane, so at distance t it is |delta| * |1 - t / f| away from that twin, f >= focus_dist the twin's distance
    // to the focus plane: the spheres it can hit are within lens_radius * max(1, t / focus_dist) of the pinhole pyramid.
    const P3 fw = mkp(-base.x - 0.5 * hor.x - 0.5 * ver.x, -base.y - 0.5 * hor.y - 0.5 * ver.y, -base.z - 0.5 * hor.z - 0.5 * ver.z);  // focus_dist * w
    const real lens = (real)(P.lens_radius * 1.001);
    const real inv_focus = (P.has_lens & 1) ? (real)(1.0 / sqrt(fw.x * fw.x + fw.y * fw.y + fw.z * fw.z)) : (real)0;
    // the four side planes (inward unit normals through the camera origin) of the pyramid over pixels [xa, xb) of row j,
    // footprint in the (s, t) coordinates of Camera::get_ray, 1/100 pixel wider on every side
    struct Pyramid { V3 left, right, bottom, top; };
    auto pyramid = [&](int xa, int xb) {
        const double s0 = ((double)xa - 0.01) * inv_w, s1 = ((double)xb + 0.01) * inv_w;
        const double t0 = ((double)j - 0.01) * inv_h, t1 = ((double)j + 1.01) * inv_h;
        auto corner = [&](double s, double t) {
            const P3 q = base + s * hor + t * ver;
            return mk((real)q.x, (real)q.y, (real)q.z);
        };
        const V3 d00 = corner(s0, t0), d10 = corner(s1, t0), d01 = corner(s0, t1), d11 = corner(s1, t1);
        const V3 mid = d00 + d11;
        auto plane = [&](V3 a, V3 b) {
            V3 nrm = mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
            nrm = unit(nrm);
            return dot(nrm, mid) < (real)0 ? -nrm : nrm;
        };
        Pyramid py;
        py.left = plane(d00, d01); py.right = plane(d10, d11); py.bottom = plane(d00, d10); py.top = plane(d01, d11);
        return py;
    };
    // (the sphere entries are listed in the blob: payload slots make the entry array unsafe to scan blindly)
    const uint16_t* sphere_index = reinterpret_cast<const uint16_t*>(P.blob + P.off_sphere_index);
    auto touches = [&](const Pyramid& py, int e) {
        double cx, cy, cz, rad;
        if (PAIR) { const real* q = reinterpret_cast<const real*>(P.blob + P.off_prims) + 4 * e; cx = (double)q[0]; cy = (double)q[1]; cz = (double)q[2]; rad = (double)q[3]; }
        else if (kPayloadAll) { const double* pl = reinterpret_cast<const double*>(&nodes[e + 1]); cx = pl[0]; cy = pl[1]; cz = pl[2]; rad = pl[3]; }
        else { const DNode<creal> q = nodes[e]; cx = (double)q.v[0]; cy = (double)q.v[1]; cz = (double)q.v[2]; rad = (double)q.v[3]; }
        const V3 cc = mk((real)cx, (real)cy, (real)cz) - of;
        const real radius = abs_r((real)rad);
        const real far = abs_r(cc.x) + abs_r(cc.y) + abs_r(cc.z);  // >= |cc|
        const real reach = radius * (real)1.00001 + (real)1e-3 + (real)1e-5 * far + lens * max_r((real)1, (far + radius) * inv_focus);
        return dot(py.left, cc) >= -reach && dot(py.right, cc) >= -reach && dot(py.bottom, cc) >= -reach && dot(py.top, cc) >= -reach;
    };
    // level 1: the span's survivors, in entry order
    const Pyramid span = pyramid(gx0 * P.pix_per_tile, min(P.width, (gx0 + 64) * P.pix_per_tile));
    int n_coarse = 0;  // wave-uniform
    const int n_index = PAIR ? P.n_prims : P.n_sphere_index;  // (pair walk: the candidates are primitive indices, every primitive a sphere)
    for (int k0 = 0; k0 < n_index; k0 += 64) {
        const int k = k0 + lane;
        const int e = PAIR ? (k < n_index ? k : 0) : (int)sphere_index[k < n_index ? k : 0];
        const bool in = k < n_index && touches(span, e);
        const unsigned long long mask = __ballot(in);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
        if (in && n_coarse + rank < kCoarseMax) coarse[n_coarse + rank] = (uint16_t)e;
        n_coarse += (int)__popcll(mask);
    }
    __syncthreads();
    const bool flat = n_coarse > kCoarseMax;
    const int n_candidates = flat ? n_index : n_coarse;
    // level 2: the lane's own pixel group
    if (gx >= P.tiles_per_row) return;
    const int x0 = gx * P.pix_per_tile;
    const Pyramid own = pyramid(x0, min(P.width, x0 + P.pix_per_tile));
    uint16_t list[kCandRecord];
#pragma unroll
    for (int k = 0; k < kCandRecord; k++) list[k] = 0;
    int count = 0;
    for (int k = 0; k < n_candidates; k++) {
        const int e = flat ? (PAIR ? k : (int)sphere_index[k]) : (int)coarse[k];
        if (touches(own, e)) {
#pragma unroll
            for (int m = 0; m < kCand; m++) if (m == count) list[1 + m] = (uint16_t)e;
            count++;
        }
    }
    list[0] = (uint16_t)(count <= kCand ? count : kCandInvalid);
    list[kCandRecord - 1] = (uint16_t)j;  // the group's image row rides along (image height < 65536)
    uint32_t* out = reinterpret_cast<uint32_t*>(records + ((size_t)row_k * P.tiles_per_row + gx) * kCandRecord);
#pragma unroll
    for (int k = 0; k < kCandRecord / 2; k++) out[k] = (uint32_t)list[2 * k] | ((uint32_t)list[2 * k + 1] << 16);
}

// KParams::status read from the kernel-argument segment at the point of use (the by-value argument struct sits at its start).
DEV unsigned int* watchdog_record() {
#if defined(__HIP_DEVICE_COMPILE__)
    const char* args = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<unsigned int* const volatile*>(args + offsetof(KParams<real>, status));
#else
    return nullptr;
#endif
}

// The megakernel.  Every lane runs a small state machine over its current path:
//   PH_TRAV    one cheap step per loop iteration: fetch an entry, slab-test it, follow the skip link
//   PH_SPECIAL parked on an entry that needs the long code (f64 sphere quadratic, rect, medium, transform bracket)
//   PH_SHADE   traversal finished: evaluate hit / background, scatter or end the path, take a new sample
// SPECIAL and SHADE work is executed only when at least MLRT_T_* lanes are parked on it (or nothing else can
// run), so the long divergent code paths are issued for many lanes at once while the short traversal step keeps
// most lanes busy every iteration.  Lanes never wait for the slowest ray of a "bounce".
#ifndef MLRT_LEAN_WAVES_PER_EU
#define MLRT_LEAN_WAVES_PER_EU 6
#endif
#ifndef MLRT_FULL_WAVES_PER_EU
#define MLRT_FULL_WAVES_PER_EU 1
#endif
// Workgroup shapes of the variants that are not sphere-only, by what their code needs in registers (tools/kernel_resources.py): waves per
// SIMD the register allocator is held to, and threads per workgroup (waves per SIMD x 256, split over two workgroups per CU above 1024
// threads - their scenes are small).  The strict and cull_f64 builds keep one shape (MLRT_BLOCK, MLRT_FULL_WAVES_PER_EU).
//   f32:  image only 79 VGPRs -> 6 waves (2 x 768: earth 5.6 -> 5.2 ms); everything else 4 waves (1024).  (cornell_box / cornell_smoke / checker
//         need 89 - 93 VGPRs, but held to the 96 of 5 waves, 2 x 640 threads, they lose: cornell_box 18.9 -> 24.1 ms, cornell_smoke 30.5 -> 38.4)
//   f64:  cornell_box 129 -> held to 128: 4 waves (1024); image only 128 -> 4; the rest 137 - 196 -> held to 168: 3 waves (768):
//         cornell_box 37.5 -> 28.1 ms, cornell_smoke 57.6 -> 46.6, final_scene (spp 200) 124.8 -> 109.4 against 512 threads at 2 waves
#ifndef MLRT_PER_VARIANT_SHAPES
#define MLRT_PER_VARIANT_SHAPES 1
#endif
constexpr bool kShaped = MLRT_PER_VARIANT_SHAPES != 0 && kCull32 && !kStrict;
constexpr int full_waves(uint32_t feat) {
    if (!kShaped) return MLRT_FULL_WAVES_PER_EU;
    const bool noise = (feat & FEAT_TEX_NOISE) != 0;
    const bool geometry = (feat & (FEAT_BOX | FEAT_RECT | FEAT_XFORM | FEAT_MEDIUM)) != 0;
    if (kFast) return (feat == FEAT_TEX_IMAGE && !geometry && !noise) ? 6 : 4;
    if (feat == (FEAT_BOX | FEAT_RECT | FEAT_XFORM) || (feat == FEAT_TEX_IMAGE && !geometry)) return 4;
    return 3;
}
constexpr int full_block(uint32_t feat) {
    if (!kShaped) return MLRT_BLOCK;
    const int w = full_waves(feat);
    return w <= 4 ? w * 256 : w * 128;  // above 4 waves per SIMD: two workgroups per CU
}
// The split-role kernel (mlrt_split.inc) exists in the builds with f32 culling entries, like the pair walk it is built on.
#ifndef MLRT_HAVE_SPLIT
#ifdef MLRT_STRICT
#define MLRT_HAVE_SPLIT 0
#else
#define MLRT_HAVE_SPLIT 1
#endif
#endif
template <bool LDS_SCENE, uint32_t FEAT, int BLOCK, bool PAIR = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FEAT == 0 ? MLRT_LEAN_WAVES_PER_EU : full_waves(FEAT))))
void mlrt_render_kernel(const KParams<real> P) {
    static_assert(!PAIR || (FEAT == 0 && kCull32), "the pair walk serves sphere-only scenes in the builds with f32 culling");
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the wave's LDS addresses are SGPR + immediate

    const unsigned char* base;
    unsigned char* acc_base;
    if (LDS_SCENE) {
        // stage the packed scene: 16 B per lane per step, fully coalesced
        const uint4* src = reinterpret_cast<const uint4*>(P.blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const int n16 = P.lds_blob_bytes >> 4;  // (the f64 builds leave the Perlin tables, the blob's tail, in HBM: see kPerlinInHbm)
        for (int i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
        base = smem;
        acc_base = smem + P.lds_blob_bytes;
    } else {
        base = P.blob;
        acc_base = smem;
    }
    constexpr int kAccBytes = acc_bytes_per_wave(FEAT);
    FrameParams& F = *reinterpret_cast<FrameParams*>(acc_base + (BLOCK / 64) * kAccBytes);
    if (threadIdx.x < 21) (&F.cam_origin[0])[threadIdx.x] = (&P.cam_origin[0])[threadIdx.x];  // cam_origin ... inv_hm1 are contiguous in both structs
    frame_reciprocals(P, F);
    if (threadIdx.x < 3) {
        F.bg_top[threadIdx.x] = P.bg_top[threadIdx.x]; F.bg_bottom[threadIdx.x] = P.bg_bottom[threadIdx.x];
        F.through0[threadIdx.x] = (float)(P.cam_llc[threadIdx.x] - P.cam_origin[threadIdx.x]);
        F.hor_f[threadIdx.x] = (float)P.cam_horizontal[threadIdx.x]; F.ver_f[threadIdx.x] = (float)P.cam_vertical[threadIdx.x];
        F.u_f[threadIdx.x] = (float)P.cam_u[threadIdx.x]; F.v_f[threadIdx.x] = (float)P.cam_v[threadIdx.x];
    }
    if (threadIdx.x == 0) F.stall_cap = P.stall_cap;
    __syncthreads();
    Scene S;
    S.nodes = reinterpret_cast<const DNode<creal>*>(base + P.off_nodes);
    S.materials = reinterpret_cast<const DMaterial<real>*>(((FEAT == 0 && kLeanMaterialsInHbm) ? P.blob : base) + P.off_materials);
    S.textures = reinterpret_cast<const DTexture<real>*>(base + P.off_textures);
    // Perlin tables (textures.rs:51-74; 18 KB in f32, 30 KB in f64; read by marble hits only, 0.07 per sample of final_scene): in
    // LDS with the rest of the scene in the fast build, whose wave-level turbulence gathers from them in every other shading
    // batch; read from HBM / L2 in the f64 builds, where they are what keeps final_scene's 181 KB blob out of the 160 KB of LDS
    // (C4 f64: 702 -> ? ms with the entries LDS resident)
    S.perlin_vec = reinterpret_cast<const real*>((kPerlinInHbm ? P.blob : base) + P.off_perlin_vec);
    S.perlin_perm = reinterpret_cast<const uint16_t*>((kPerlinInHbm ? P.blob : base) + P.off_perlin_perm);
    S.images = P.images;
    S.pairs = reinterpret_cast<const DPair*>(base + P.off_nodes);
    S.prims = reinterpret_cast<const real*>(base + P.off_prims);
    S.prim_mat = reinterpret_cast<const uint16_t*>(base + P.off_prim_mat);
    // Pair walk: the lane's traversal stack, rows of BLOCK 16-bit references behind the frame parameters (row r of lane l at
    // [r * BLOCK + l]; P.stack_levels rows are ever written).  The newest reference lives in `top`; rows 1 .. sp / BLOCK - 1 hold the
    // older ones, kPairEnd first.  Row 0 is only ever read (by the pop that takes kPairEnd out of `top`: the value is never looked at).
    // (Measured and not kept: `top` stored and the row below it re-read by every lane after every step, so that neither push nor pop
    // waits for the LDS inside a step and the step has no branch but the outer one: 3 % slower in both builds - the visit is bound by
    // the instructions it issues, not by the pop's round trip.  Only traversing lanes fetching their next node: +0.5 %.  Two rounds of
    // the sphere batch for lanes whose pop is another leaf: +2 %.  Metal / Dielectric hits - a fifth of `random`'s hits, a third of a
    // shading batch's code - shaded in batches of their own once 8 / 16 / 24 such lanes wait: +8 / +10 / +14 %: a lane that waits costs
    // more than the code its absence lets a batch skip.  The two children's plane distances as six v_pk_fma_f32 (boxes interleaved in the
    // node): +4 % in both builds - the pairs of ray constants cost registers (16 B of scratch appear) and copies.)
    int16_t* const stack_lane = reinterpret_cast<int16_t*>(acc_base + (BLOCK / 64) * kAccBytes + kFrameParamBytes) + threadIdx.x;
    int sp = BLOCK, top = kPairEnd;

    const int n_banks = P.n_banks, bank_vals = P.bank_values;  // wave-uniform
    int* bank_remaining = reinterpret_cast<int*>(acc_base + wave * kAccBytes);  // samples of the bank's tile not yet finished
    int* bank_out = bank_remaining + kMaxBanks;                                         // first index into acc_global
    int* bank_values = bank_remaining + 2 * kMaxBanks;                                  // npix * 3
    uint16_t* cand = reinterpret_cast<uint16_t*>(bank_remaining + 3 * kMaxBanks);       // [kMaxBanks][kCandRecord]: the tile's primary candidates (sphere-only variants)
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(cand + (FEAT == 0 ? kMaxBanks * kCandRecord : 0));  // [n_banks][bank_vals]
    const RngKey key = {P.seed_lo, P.seed_hi};
    const int n_nodes = P.n_nodes;

    if (lane < n_banks) bank_remaining[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // The wave's stream of tiles.  A tile = pix_per_tile consecutive pixels of a row x one chunk of samples, pulled from
    // the global queue; its pixels*samples form a queue handed out to lanes in order.  Tiles OVERLAP: when the current
    // tile's queue is exhausted the next one is opened at once (into the next accumulator bank of the ring) while the
    // last paths of older tiles are still running, so there is no per-tile drain.  All of this state is wave-uniform.
    int t_bank = 0, t_total = 0, t_next = 0, t_ns = 1, t_s0 = 0, t_x0 = 0, t_j = 0;
    bool queue_done = false;
    // The queue is n_queues interleaved sub-queues (sub-queue k holds tiles k, k + n_queues, ...), each with its own
    // counter on its own cache line: with ONE counter for all waves of the GPU and 64-sample tiles the tail passes ran at
    // the rate of same-address device-scope atomics (one every ~13 ns was needed).  A wave starts on the sub-queue of its
    // workgroup and moves on when that one is exhausted.
    int q_id = (int)blockIdx.x & (P.n_queues - 1), q_tried = 0;  // the wave's sub-queue; sub-queues found exhausted
    const bool lists = FEAT == 0 && P.primary_lists != nullptr;
    auto queue_pop = [&]() -> int {  // synchronous; n_tiles when every sub-queue is exhausted
        for (;;) {
            int i = 0;
            if (lane == 0) i = (int)atomicAdd(P.tile_counter + q_id * 16, 1u);
            const int t = __builtin_amdgcn_readfirstlane(i) * P.n_queues + q_id;
            if (t < P.n_tiles) return t;
            if (++q_tried >= P.n_queues) return P.n_tiles;
            q_id = (q_id + 1) & (P.n_queues - 1);
        }
    };
    unsigned int st_psteps = 0, st_pspecial = 0;  // MLRT_STATS: traversal steps / special entries of primary rays
    unsigned long long st_wait = 0ull;            // MLRT_STATS: lane-iterations spent starved (PH_WAIT: the wave's banks are all taken and its tile is handed out)
    unsigned long long st_c_trav = 0, st_c_special = 0, st_c_shade = 0, st_c0 = 0;  // MLRT_STATS: shader clocks per loop section
    unsigned long long st_c_end = 0, st_c_regen = 0, st_c1 = 0;  // MLRT_STATS: within the shading batch: paths ended/scattered, regeneration (the rest: new-ray setup)
    unsigned int st_iter = 0, st_steps = 0, st_sb = 0, st_sl = 0, st_hb = 0, st_hl = 0, st_samples = 0, st_tiles = 0;  // MLRT_STATS

#ifdef MLRT_DEBUG_BUILD
    unsigned int stall = F.stall_cap;  // wave-uniform countdown: iterations left before watchdog (2) fires, reloaded whenever a path of this wave ends
#endif
    int phase = PH_SHADE;  // every lane starts by asking for a sample
    bool has_path = false;
    int pix = 0, depth = 0;  // pix: the lane's bank << kPixBankShift | index of its pixel's first sum in acc
    P3 o = mkp(0, 0, 0), d = mkp(0, 0, 1);
    V3 thr = mk(1, 1, 1);
    Rng g = {0u, 0u, 0u, 0u};
    Cursor c;
    c.node = 0; c.xf = -1; c.closest = 0.0; c.hit_node = -1; c.hit_aux = 0; c.hit_xf = -1;
    c.sr.ofx = c.sr.ofy = c.sr.ofz = (creal)0; c.sr.ix = c.sr.iy = c.sr.iz = (creal)1; c.sr.of2x = c.sr.of2y = c.sr.of2z = (creal)0;

    // Tiles whose last sample has finished: the WAVE looks at the banks' counts once per shading batch (one LDS read for
    // all banks; LDS operations of a wave execute in order, so every add of a tile is visible when its count reads zero)
    // and flushes finished tiles into the frame buffer - integer atomics, so the frame is bit-identical however pixels'
    // samples were split into tiles and whichever wave / GPU rendered them.  (Before, every finishing lane waited for
    // the value its atomic decrement returned, and the flush loop sat in the divergent per-lane code.)
    unsigned int open_banks = 0u;  // wave-uniform: banks holding a tile that is not flushed yet
    auto flush_finished_banks = [&]() -> bool {
        const int remaining = lane < n_banks ? bank_remaining[lane] : 1;
        unsigned int finished = (unsigned int)__ballot(remaining == 0) & open_banks;
        if (finished == 0u) return false;
        open_banks &= ~finished;
        for (; finished != 0u; finished &= finished - 1u) {
            const int bank = (int)__builtin_ctz(finished);
            const int n_values = __builtin_amdgcn_readfirstlane(bank_values[bank]);
            const int out0 = __builtin_amdgcn_readfirstlane(bank_out[bank]);
            if (lane < n_values) atomicAdd(&P.acc_global[(size_t)out0 + lane], acc[bank * bank_vals + lane]);
        }
        return true;
    };

    // A wave that gives up (watchdogs below) records what its lanes were doing.  (The record's address is read from the kernel
    // arguments only at that point - a volatile load: hoisted out of the loop it would occupy two scalar registers of a kernel
    // that has none to spare.  A macro, not a lambda: capturing the lane state by reference sent it to scratch.)
#define MLRT_GIVE_UP()                                                                                                                        \
    do {                                                                                                                                      \
        unsigned int* status_ = watchdog_record();                                                                                            \
        if (lane == 0) { atomicAdd(&status_[0], 1u); status_[1] = (unsigned int)blockIdx.x; status_[2] = (unsigned int)wave; }               \
        status_[4 + lane] = (unsigned int)phase | ((unsigned int)has_path << 8) | ((unsigned int)(c.node & 0xFFFF) << 16);                  \
    } while (0)
    typename EntryOf<PAIR>::type n;  // always the entry at c.node (any lane, any phase)
    if constexpr (PAIR) n = fetch_pair(S, 0); else n = fetch_entry(S, 0, n_nodes);
    for (;;) {
        // (A) MLRT_STEPS traversal steps between two looks at the batches: the ballots and threshold tests of (B) and (C)
        // are a fifth of an iteration's instructions, and a lane that parks waits at most three steps longer for
        // its batch (measured, C2 / final_scene spp 200 kernel ms: 1 step 43.6 / 73.7, 2: 41.3 / 69.7, 3: 42.1 / 70.1,
        // 4: 40.8 / 67.7, 6: 42.1 / 68.3, 8: 43.4 / 70.2; the thresholds' optima do not move)
#ifndef MLRT_STEPS
#define MLRT_STEPS 4
#endif
        if (MLRT_STATS_ON(P)) st_c0 = __builtin_readcyclecounter();
        creal far_bound = slab_far_bound(c.closest);  // (hits are accepted in the batches, and by box entries in the full variant's step)
#pragma unroll
        for (int sub = 0; sub < (PAIR ? MLRT_PAIR_STEPS : MLRT_STEPS); sub++) {
        pin_entry(n);
        if (MLRT_STATS_ON(P)) { st_steps += (unsigned int)__popcll(__ballot(phase == PH_TRAV)); st_psteps += (unsigned int)__popcll(__ballot(phase == PH_TRAV && depth == P.max_depth));
                                 st_wait += (unsigned long long)__popcll(__ballot(phase == PH_WAIT)); }
        // The array ends in a TAG_END sentinel, so there is no bounds test.  Two shapes of the same step (measured): the
        // sphere-only variant runs at 6 waves per SIMD and is bound by vector-instruction issue, where two nested
        // branches cost fewer vector instructions than selects; the full variant (4 waves, box entries resolved in the
        // step) gains 7 % from a branch-free step.
        if constexpr (PAIR) {
            // One visit of a pair node: both children's boxes, the nearer hit child next, the other one pushed; nothing hit: pop.
            if (phase == PH_TRAV) {
                float tn_a, tn_b;
                const bool hit_a = slab_child(n.a, c.sr, (float)far_bound, tn_a), hit_b = slab_child(n.b, c.sr, (float)far_bound, tn_b);
                const bool b_first = hit_b && (!hit_a || tn_b < tn_a);
                int first = b_first ? n.ref_b : n.ref_a;
                const int second = b_first ? n.ref_a : n.ref_b;
                if (hit_a && hit_b) { stack_lane[sp] = (int16_t)top; sp += BLOCK; top = second; }
                else if (!hit_a && !hit_b) { first = top; sp -= BLOCK; top = (int)stack_lane[sp]; }
                c.node = first;
                if (first < 0) phase = first == kPairEnd ? PH_SHADE : PH_SPECIAL;  // a leaf parks the lane for the sphere batch
            }
        } else if (FEAT == 0) {
            if (phase == PH_TRAV) {
                // (inner entries and the sentinel carry no flag bits: the tag is compared whole)
                if (n.tag == TAG_INNER) c.node = slab_hit_any(n, c.sr, far_bound) ? c.node + 1 : n.link;
                else phase = n.tag == TAG_END ? PH_SHADE : PH_SPECIAL;
            }
        } else {
            const int kind = n.tag & TAG_KIND_MASK;
            const bool trav = phase == PH_TRAV;
            const bool inner = trav && kind == TAG_INNER;
            const bool inside = slab_hit_any(n, c.sr, far_bound);
            c.node = inner ? (inside ? c.node + 1 : n.link) : c.node;
            bool handled = inner;
            if ((FEAT & FEAT_BOX) && !MLRT_BOX_SPECIAL && !kProduct64) {
                // Block entries resolved in the step: the fast build from the same slab evaluation (culling form, f32), the strict
                // build with the reference's divisions (its c.sr is the ray's own origin and direction)
                const bool box = trav && kind == TAG_BOX;
                if (box) {
                    double t; int aux = 0;
                    bool hit_box;
                    if (kFast) {
                        hit_box = box_hit_fast(n, c.sr, 0.001, c.closest, t, aux);
                    } else {
                        const double b[6] = {(double)n.v[0], (double)n.v[1], (double)n.v[2], (double)n.v[3], (double)n.v[4], (double)n.v[5]};
                        hit_box = box_hit_exact(b, mkp((double)c.sr.ofx, (double)c.sr.ofy, (double)c.sr.ofz), mkp((double)c.sr.ix, (double)c.sr.iy, (double)c.sr.iz), 0.001, c.closest, t, aux);
                    }
                    if (hit_box) { c.closest = t; c.hit_node = c.node; c.hit_aux = aux; c.hit_xf = c.xf; far_bound = slab_far_bound(t); }
                    c.node += 1;
                }
                handled = handled || box;
            }
            if ((FEAT & FEAT_BOX) && kProduct64) {
                // product f64 builds: a Block whose (outward) bounds the culling test misses is passed in the step; one the ray may
                // hit parks the lane for the special batch, where the face distances are the reference's divisions
                const bool box_missed = trav && kind == TAG_BOX && !inside;
                c.node = box_missed ? c.node + slots_of(n.tag) : c.node;
                handled = handled || box_missed;
            }
            phase = (trav && !handled) ? (kind == TAG_END ? PH_SHADE : PH_SPECIAL) : phase;
        }
        if constexpr (PAIR) n = fetch_pair(S, c.node); else n = fetch_entry(S, c.node, n_nodes);
        }
        if (MLRT_STATS_ON(P)) { st_iter++; const unsigned long long t = __builtin_readcyclecounter(); st_c_trav += t - st_c0; st_c0 = t; }
        const unsigned long long m_trav = __ballot(phase == PH_TRAV);
        // (B) batched special entries
        const unsigned long long m_special = __ballot(phase == PH_SPECIAL);
        // (measured and not kept: the batches also run as soon as fewer than 8 / 16 / 24 / 32 lanes still traverse - C2 f64 +1 / +1.5 / +2.3 / +4 %,
        // final_scene f64 +0.6 / +4.7 / +8.5 / +12 %: a batch that runs early runs with fewer lanes, and it is the batches that cost)
        if (MLRT_RARELY(m_special != 0ull && (__popcll(m_special) >= (FEAT == 0 ? MLRT_T_SPECIAL_LEAN : (big_world_variant(FEAT) ? MLRT_T_SPECIAL : MLRT_T_SPECIAL_SMALL)) || m_trav == 0ull))) {
            if (MLRT_STATS_ON(P)) { st_sb++; st_sl += (unsigned int)__popcll(m_special); st_pspecial += (unsigned int)__popcll(__ballot(phase == PH_SPECIAL && depth == P.max_depth)); }
            if constexpr (PAIR) {
#pragma unroll
                for (int round = 0; round < MLRT_PAIR_SPECIAL_ROUNDS; round++) {
                    if (round > 0 && __ballot(phase == PH_SPECIAL) == 0ull) break;
                    if (phase == PH_SPECIAL) {
                        // Sphere::hit of the leaf's primitive.  A hit at exactly the closest distance so far counts only from a primitive the
                        // reference visits later (mlrt_device.hpp DPair): the result is independent of the device's visiting order.
                        const int prim = c.node - kPairLeaf;
                        const Ray r = local_ray<FEAT>(S, -1, o, d);
                        double t;
                        if (sphere_hit(pair_sphere(S, prim), r, 0.001, c.closest, t) && (t < c.closest || prim > c.hit_node)) { c.closest = t; c.hit_node = prim; }
                        phase = PH_TRAV;
                        if (depth & kPrimaryFlag) {  // next candidate of the tile's list, or done
                            const int k = ((depth >> 16) & 0xFF) + 1;
                            const int bank = pix >> kPixBankShift;
                            if (k < cand[bank * kCandRecord]) { c.node = kPairLeaf + (int)cand[bank * kCandRecord + 1 + k]; depth += 1 << 16; phase = PH_SPECIAL; }
                            else { c.node = kPairEnd; depth &= 0xFFFF; phase = PH_SHADE; }
                        } else {
                            const int next = top;
                            sp -= BLOCK; top = (int)stack_lane[sp];
                            c.node = next;
                            if (next < 0) phase = next == kPairEnd ? PH_SHADE : PH_SPECIAL;
                        }
                    }
                }
                n = fetch_pair(S, c.node);
            } else {
            if (phase == PH_SPECIAL) {
                special_entry<FEAT>(S, n, c, o, d, g, key);
                phase = PH_TRAV;
                if (FEAT == 0 && (depth & kPrimaryFlag)) {  // next candidate of the tile's list, or done
                    const int k = ((depth >> 16) & 0xFF) + 1;
                    const int bank = pix >> kPixBankShift;
                    if (k < cand[bank * kCandRecord]) { c.node = cand[bank * kCandRecord + 1 + k]; depth += 1 << 16; phase = PH_SPECIAL; }
                    else { c.node = n_nodes; depth &= 0xFFFF; phase = PH_SHADE; }
                }
            }
            n = fetch_entry(S, c.node, n_nodes);
            }
        }
        if (MLRT_STATS_ON(P)) { const unsigned long long t = __builtin_readcyclecounter(); st_c_special += t - st_c0; st_c0 = t; }
        // (C) batched shading, path termination and regeneration
        // (measured and not kept, sphere-only variants: ONE Philox evaluation per batch - paths that end by leaving the scene take their
        // next sample first, then one call serves block 0 of the fresh samples and the scatter blocks of the hits, with a second round for
        // the rare paths that end in the scatter.  Philox is a tenth of the kernel's instructions and runs twice per batch, but the
        // restructured batch needs 12-20 bytes of scratch and a loop: C2 f64 52.4 -> 54.1 ms, f32 36.6 -> 37.6.)
        const unsigned long long m_shade = __ballot(phase == PH_SHADE);
        // (the end of the loop, below, decided here where the masks are at hand)
        const bool nobody_runs = (m_trav | m_special | m_shade) == 0ull;
        const bool batch = m_shade != 0ull && (__popcll(m_shade) >= (FEAT == 0 ? MLRT_T_SHADE_LEAN : (big_world_variant(FEAT) ? MLRT_T_SHADE : MLRT_T_SHADE_SMALL)) || (m_trav == 0ull && m_special == 0ull));  // wave-uniform
        if (MLRT_RARELY(batch)) {
            if (MLRT_STATS_ON(P)) { st_hb++; st_hl += (unsigned int)__popcll(m_shade); }
            const bool mine = phase == PH_SHADE;
            bool new_ray = false;
            // hit attributes first, then - wave-uniform - the turbulence of the few lanes that hit a NoiseTexture
            const bool hit = mine && has_path && c.hit_node >= 0;
            constexpr bool kWaveNoise = (FEAT & FEAT_TEX_NOISE) && kFast;
            HitAttr h;
            bool have_turbulence = false;
            real turbulence = (real)0;
            if (kWaveNoise) {
                if (hit) h = hit_attributes<FEAT, PAIR>(S, c, o, d);
                bool marble = false;
                int table = 0;
                float qx = 0.0f, qy = 0.0f, qz = 0.0f;
                if (hit && (h.m.kind & MATFLAG_NOISE)) {
                    const DTexture<real> t = S.textures[h.m.tex];
                    marble = true; table = t.a;
                    qx = (float)t.scale * (float)h.p.x; qy = (float)t.scale * (float)h.p.y; qz = (float)t.scale * (float)h.p.z;  // scale * p, textures.rs:164
                }
                for (unsigned long long pending = __ballot(marble); pending != 0ull; pending &= pending - 1ull) {
                    const int src = (int)__builtin_ctzll(pending);
                    const float value = wave_turbulence(S, __builtin_amdgcn_readlane(table, src), __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qx), src)),
                                                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qy), src)),
                                                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qz), src)), lane);
                    if (lane == src) { turbulence = (real)value; have_turbulence = true; }
                }
            }
            if (mine && has_path) {
                if (MLRT_DEBUG_ON(P) && (int)g.pixel == P.debug_pixel && (int)g.sample == P.debug_sample)
                    printf("[mlrt dbg] depth %d hit_node %d tag %d xf %d t %.17g draw %u o %.17g %.17g %.17g d %.17g %.17g %.17g thr %g %g %g\n", depth, c.hit_node,
                           (c.hit_node >= 0 && !PAIR) ? S.nodes[c.hit_node].tag : -1, c.hit_xf, c.closest, g.draw, o.x, o.y, o.z, d.x, d.y, d.z, (double)thr.x, (double)thr.y, (double)thr.z);
                V3 radiance = mk(0, 0, 0);
                bool done = true;
                if (hit) {
                    if (!kWaveNoise) h = hit_attributes<FEAT, PAIR>(S, c, o, d);
                    V3 emitted = mk(0, 0, 0);
                    const V3 texture_colour = material_texture<FEAT>(S, h, have_turbulence, turbulence);
                    // the Renderer's other RayTracer (MLRT_TRACER_SINGLE_LIGHT; all-features variant only, wave-uniform): one bounce
                    const bool single_light = FEAT == FEAT_ALL && !PAIR && (P.has_lens & 4) != 0;
                    V3 view = mk(0, 0, 0);
                    if (MLRT_RARELY(single_light)) view = -unit(to_v3(d));
                    if (scatter<FEAT>(S, h, texture_colour, o, d, thr, emitted, g, key, P.max_depth - depth)) {
                        if (MLRT_RARELY(single_light)) {
                            radiance = single_light_colour(thr, h, view);  // (a primary ray's throughput is 1: thr is the attenuation)
                        } else {
                        depth--;
                        if (depth > 0) { done = false; new_ray = true; }  // depth <= 0: raytrace.rs:87-89, radiance 0
                        }
                    } else {
                        radiance = thr * emitted;
                    }
                } else {
                    radiance = thr * background_color(P, F, d);
                }
                if (done) {
                    const int sum0 = pix & ((1 << kPixBankShift) - 1);
                    atomicAdd(&acc[sum0 + 0], to_fixed(radiance.x));
                    atomicAdd(&acc[sum0 + 1], to_fixed(radiance.y));
                    atomicAdd(&acc[sum0 + 2], to_fixed(radiance.z));
                    has_path = false;
                    atomicSub(&bank_remaining[pix >> kPixBankShift], 1);  // (no return value: the wave looks at the counts below)
                }
            }
            const bool freed = flush_finished_banks();
            if (MLRT_STATS_ON(P)) { st_c1 = __builtin_readcyclecounter(); st_c_end += st_c1 - st_c0; }
            if (freed && phase == PH_WAIT) phase = PH_SHADE;  // starved lanes join the next shading batch
            // Path regeneration, in wave-uniform control flow so the tile state stays uniform: the batch's lanes without
            // a path take the next queue entries in lane order, opening the next tile when the current one runs out.
            bool want = mine && !has_path;
            for (;;) {
                const unsigned long long need = __ballot(want);
                if (need == 0ull) break;
#ifdef MLRT_DEBUG_BUILD
                stall = F.stall_cap;  // a path of this wave has ended: the wave makes progress (watchdog (2) below)
#endif
                if (MLRT_RARELY(t_next >= t_total)) {
                    // open the next tile into ANY bank whose previous tile has fully drained (not the next of a ring: one
                    // 50-bounce path would hold its bank, and with it the whole ring, while the other lanes run out of samples)
                    if (queue_done) break;
                    if (P.max_depth == 0) flush_finished_banks();  // (samples that end inside this loop)
                    const unsigned int free_banks = ~open_banks & (n_banks >= 32 ? 0xFFFFFFFFu : (1u << n_banks) - 1u);
                    if (free_banks == 0u) break;
                    // (LDS / global loads below are wave-uniform by construction; readfirstlane tells the compiler, which
                    // keeps the tile state in scalar registers)
                    const int nb = (int)__builtin_ctz(free_banks);
                    const int tile = queue_pop();
                    if (tile >= P.n_tiles) { queue_done = true; break; }
                    open_banks |= 1u << nb;
                    // tile -> (pass, pixel group): chunk-major queue, see KParams
                    const int pass = magic_div(tile, P.n_groups_magic, P.n_groups_shift);
                    const int group = tile - pass * P.n_groups;
                    const int row_k = magic_div(group, P.tiles_per_row_magic, P.tiles_per_row_shift);
                    t_x0 = (group - row_k * P.tiles_per_row) * P.pix_per_tile;
                    const int npix = min(P.pix_per_tile, P.width - t_x0);
                    if (pass < P.n_big) {
                        t_s0 = pass * P.big;
                        t_ns = P.big;
                    } else {
                        t_s0 = P.tail_s0[pass - P.n_big];
                        t_ns = P.tail_s0[pass - P.n_big + 1] - t_s0;  // samples of every pixel handled by this tile
                    }
                    uint32_t rec = 0u;  // lanes 0-7: the group's record (pre-pass), with its image row in the last half word
                    if (lists) { if (lane < kCandRecord / 2) rec = reinterpret_cast<const uint32_t*>(P.primary_lists + (size_t)group * kCandRecord)[lane]; }
                    else if (P.rows) rec = (uint32_t)P.rows[row_k];
                    t_j = lists ? (int)(__builtin_amdgcn_readlane(rec, kCandRecord / 2 - 1) >> 16) : (P.rows ? (int)__builtin_amdgcn_readfirstlane(rec) : row_k);
                    t_total = npix * t_ns;
                    t_next = 0;
                    t_bank = nb;
                    if (lane < npix * 3) acc[nb * bank_vals + lane] = 0ull;
                    if (lane == 0) {
                        bank_remaining[nb] = t_total;
                        bank_out[nb] = (row_k * P.width + t_x0) * 3;
                        bank_values[nb] = npix * 3;
                    }
                    if (lists && lane < kCandRecord / 2) reinterpret_cast<uint32_t*>(cand + nb * kCandRecord)[lane] = rec;  // the group's record (pre-pass)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (MLRT_STATS_ON(P)) { st_tiles++; st_samples += (unsigned int)t_total; }
                }
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)need, 0u));
                const int avail = t_total - t_next;
                if (want && rank < avail) {
                    const int q = t_next + rank;
                    const int p_in_tile = P.pix_per_tile == 1 ? 0 : q / t_ns;  // (uniform branch: no per-lane integer division at high spp)
                    const int smp = t_s0 + (q - p_in_tile * t_ns);
                    pix = (t_bank << kPixBankShift) | (t_bank * bank_vals + p_in_tile * 3);
                    g.pixel = (uint32_t)(t_j * P.width + t_x0 + p_in_tile);
                    g.sample = (uint32_t)smp;
                    gen_camera_ray(P, F, g, key, t_x0 + p_in_tile, t_j, o, d);
                    if (MLRT_RARELY(P.has_lens & 2)) {  // a ray frame (mlrt_debug_ray_frame): the sample's ray comes from the table
                        const double* q = ray_table_pointer() + ((size_t)g.pixel * (size_t)P.spp + g.sample) * 6;
                        o = mkp(q[0], q[1], q[2]); d = mkp(q[3], q[4], q[5]);
                    }
                    thr = mk(1, 1, 1);
                    depth = P.max_depth;
                    has_path = true;
                    new_ray = depth > 0;
                    want = false;
                    if (!new_ray) {  // max_depth == 0: the sample contributes 0 (raytrace.rs:87-89)
                        has_path = false;
                        atomicSub(&bank_remaining[t_bank], 1);
                    }
                }
                t_next += min((int)__popcll(need), avail);
            }
            if (MLRT_STATS_ON(P)) st_c_regen += __builtin_readcyclecounter() - st_c1;
            if (mine) {
                if (new_ray) {
                    c.node = 0; c.xf = -1; c.closest = (double)INFINITY; c.hit_node = -1; c.hit_aux = 0; c.hit_xf = -1;
                    if (PAIR) { sp = BLOCK; top = kPairEnd; }  // the root node, an empty stack
                    Ray wr;
                    wr.o = o; wr.d = d;
                    c.sr = slab_ray(wr);
                    phase = PH_TRAV;
                    if (lists && depth == P.max_depth) {  // a fresh sample: its primary ray
                        const int bank = pix >> kPixBankShift;
                        const int count = cand[bank * kCandRecord];
                        if (count != kCandInvalid) {
                            if (count > 0) { c.node = (PAIR ? (int)kPairLeaf : 0) + (int)cand[bank * kCandRecord + 1]; depth |= kPrimaryFlag; phase = PH_SPECIAL; }
                            else { c.node = PAIR ? (int)kPairEnd : n_nodes; phase = PH_SHADE; }  // nothing inside the tile's frustum: background
                        }
                    }
                } else if (want) {
                    // no sample to be had right now: the stream is finished, or every bank is taken (woken when one drains)
                    phase = (queue_done && t_next >= t_total) ? PH_IDLE : PH_WAIT;
                }  // else (max_depth == 0): the lane's sample is already finished; it stays in PH_SHADE and asks again
            }
            if constexpr (PAIR) n = fetch_pair(S, c.node); else n = fetch_entry(S, c.node, n_nodes);
        }
        if (MLRT_STATS_ON(P)) { const unsigned long long t = __builtin_readcyclecounter(); st_c_shade += t - st_c0; }
        // The end of the loop, and watchdog (1).  No lane was traversing, parked for a batch or shading in this iteration:
        // either every lane is idle - the wave is done - or the rest wait for an accumulator bank that only one of those could
        // free, and nothing can ever change again (round 1 had such a slip in the lane state machine: it span forever and took
        // the GPU with it).  Such a wave records what its lanes were doing and leaves: the launch ends and the host reports
        // MLRT_ERR_HIP (mlrt_context_check) instead of a wedged device.  One scalar flag instead of the all-lanes-idle ballot
        // that used to close every iteration.
        if (MLRT_RARELY(nobody_runs)) {
            if (__ballot(phase != PH_IDLE) != 0ull) MLRT_GIVE_UP();
            break;
        }
#ifdef MLRT_DEBUG_BUILD
        // Watchdog (2), debug build (build.py --debug) only: a countdown of iterations since a path of this wave last ended
        // (KParams::stall_cap, default 2^24: seconds; a legitimate path takes a few hundred), reloaded from LDS when one does.
        // Kept out of the release kernels: its scalar register and two instructions cost 1.5-2 % (C2 39.1 vs 38.5 ms).
        if (MLRT_RARELY(--stall == 0u)) { MLRT_GIVE_UP(); break; }
#endif
    }
    flush_finished_banks();  // (max_depth == 0 finishes its last samples inside the regeneration loop)
    if (MLRT_STATS_ON(P) && lane == 0) {
        atomicAdd(&P.stats[0], (unsigned long long)st_iter); atomicAdd(&P.stats[1], (unsigned long long)st_steps);
        atomicAdd(&P.stats[2], (unsigned long long)st_sb); atomicAdd(&P.stats[3], (unsigned long long)st_sl);
        atomicAdd(&P.stats[4], (unsigned long long)st_hb); atomicAdd(&P.stats[5], (unsigned long long)st_hl);
        atomicAdd(&P.stats[6], (unsigned long long)st_samples); atomicAdd(&P.stats[7], (unsigned long long)st_tiles);
        atomicAdd(&P.stats[11], (unsigned long long)st_psteps); atomicAdd(&P.stats[12], (unsigned long long)st_pspecial);
        atomicAdd(&P.stats[8], st_c_trav); atomicAdd(&P.stats[9], st_c_special); atomicAdd(&P.stats[10], st_c_shade);
        atomicAdd(&P.stats[13], st_c_end); atomicAdd(&P.stats[14], st_c_regen);
        atomicAdd(&P.stats[15], st_wait);
    }
}

#if MLRT_HAVE_SPLIT
#include "mlrt_split.inc"
#endif

// to_rgb (raytrace.rs:59-68) in f64 over the finished accumulation buffer.  The kernel also leaves the frame's scratch the
// way the next frame wants to find it - the sums it has read and the tile counters zero - so a frame is three launches
// and no memset (a 23 MB hipMemsetAsync per C2 frame does not shrink when a GPU renders 1/8 of the rows).
__global__ void mlrt_resolve_kernel(unsigned long long* acc, int n, int spp, int32_t* rgb_out, double* accum_out, unsigned int* tile_counter) {
    if (blockIdx.x == 0 && threadIdx.x < kMaxQueues) tile_counter[threadIdx.x * 16] = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double sum = (double)acc[i] * (1.0 / 4294967296.0);
        acc[i] = 0ull;
        if (accum_out) accum_out[i] = sum;
        const double scale = 1.0 / (double)spp;
        double cc = sqrt(sum * scale);
        cc = cc < 0.0 ? 0.0 : (cc > 0.99999999 ? 0.99999999 : cc);
        rgb_out[i] = (int32_t)(255.999 * cc);
    }
}


// ---------------------------------------------------------------------------------------------- directed probes (debug build)
// Test instruments (build.py --debug only; exported through mlrt_debug_probe_* in mlrt_api.cpp): the device functions above
// evaluated on CHOSEN inputs - one thread per input, a plain per-thread loop - so that tests/test_gpu_directed.py can compare
// each of them with the oracle on the edge cases an image almost never produces (axis-parallel rays, origins inside boxes,
// tangent spheres, closed rect edges, media entered from inside, total internal reflection, NaN sums ...).  The probes walk
// the REAL packed scene of a context with the REAL step arithmetic; only the lane scheduling of the megakernel (phases,
// batches) is replaced by a loop.
#ifdef MLRT_DEBUG_BUILD
constexpr int kProbeRayOut = 24;  // doubles per ray, see mlrt_debug_probe_rays
template <uint32_t FEAT>
__global__ void mlrt_probe_rays_kernel(const KParams<real> P, int n, const double* rays, const uint32_t* psd, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Scene S;
    S.nodes = reinterpret_cast<const DNode<creal>*>(P.blob + P.off_nodes);
    S.materials = reinterpret_cast<const DMaterial<real>*>(P.blob + P.off_materials);
    S.textures = reinterpret_cast<const DTexture<real>*>(P.blob + P.off_textures);
    S.perlin_vec = reinterpret_cast<const real*>(P.blob + P.off_perlin_vec);
    S.perlin_perm = reinterpret_cast<const uint16_t*>(P.blob + P.off_perlin_perm);
    S.images = P.images;
    const RngKey key = {P.seed_lo, P.seed_hi};
    P3 o = mkp(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]), d = mkp(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    Rng g = {psd[i * 3 + 0], psd[i * 3 + 1], psd[i * 3 + 2], 0u};
    if ((g.draw & 3u) == 3u) {  // (word 3 of the current block is lane state in the kernel)
        uint32_t w0, w1, w2, w3;
        philox4x32_10(key.k0, key.k1, g.draw >> 2, g.sample, g.pixel, 0u, w0, w1, w2, w3);
        g.b3 = w3;
    }
    Cursor c;
    c.node = 0; c.xf = -1; c.closest = (double)INFINITY; c.hit_node = -1; c.hit_aux = 0; c.hit_xf = -1;
    Ray wr;
    wr.o = o; wr.d = d;
    c.sr = slab_ray(wr);
    long steps = 0;
    for (;;) {  // the megakernel's step, one lane, no phases
        const DNode<creal> e = S.nodes[c.node];
        const int kind = e.tag & TAG_KIND_MASK;
        if (kind == TAG_END || ++steps > 1000000) break;
        const creal far_bound = slab_far_bound(c.closest);
        if (kind == TAG_INNER) { c.node = slab_hit_any(e, c.sr, far_bound) ? c.node + 1 : e.link; continue; }
        if ((FEAT & FEAT_BOX) && kind == TAG_BOX) {
            if (kProduct64) {
                if (!slab_hit_any(e, c.sr, far_bound)) { c.node += slots_of(e.tag); continue; }
            } else {
                double t; int aux = 0;
                bool hit_box;
                if (kFast) hit_box = box_hit_fast(e, c.sr, 0.001, c.closest, t, aux);
                else {
                    const double b[6] = {(double)e.v[0], (double)e.v[1], (double)e.v[2], (double)e.v[3], (double)e.v[4], (double)e.v[5]};
                    hit_box = box_hit_exact(b, mkp((double)c.sr.ofx, (double)c.sr.ofy, (double)c.sr.ofz), mkp((double)c.sr.ix, (double)c.sr.iy, (double)c.sr.iz), 0.001, c.closest, t, aux);
                }
                if (hit_box) { c.closest = t; c.hit_node = c.node; c.hit_aux = aux; c.hit_xf = c.xf; }
                c.node += 1;
                continue;
            }
        }
        special_entry<FEAT>(S, e, c, o, d, g, key);
    }
    double* r = out + (size_t)i * kProbeRayOut;
    for (int k = 0; k < kProbeRayOut; k++) r[k] = 0.0;
    r[21] = (double)g.draw;  // draws consumed by the traversal (media)
    if (c.hit_node < 0) return;
    const HitAttr h = hit_attributes<FEAT>(S, c, o, d);
    r[0] = 1.0; r[1] = c.closest; r[2] = h.p.x; r[3] = h.p.y; r[4] = h.p.z;
    r[5] = (double)h.normal.x; r[6] = (double)h.normal.y; r[7] = (double)h.normal.z;
    r[8] = (double)h.u; r[9] = (double)h.v; r[10] = h.front_face ? 1.0 : 0.0;
    V3 thr = mk(1, 1, 1), emitted = mk(0, 0, 0);
    const int bounce = (int)((g.draw + 3u) >> 2) - 1;  // (the block rule of scenes without media: block = 1 + bounce)
    const bool scattered = scatter<FEAT>(S, h, material_texture<FEAT>(S, h, false, (real)0), o, d, thr, emitted, g, key, bounce);
    r[11] = scattered ? 1.0 : 0.0;
    if (scattered) {
        r[12] = o.x; r[13] = o.y; r[14] = o.z; r[15] = d.x; r[16] = d.y; r[17] = d.z;
        r[18] = (double)thr.x; r[19] = (double)thr.y; r[20] = (double)thr.z;
    } else {
        r[18] = (double)emitted.x; r[19] = (double)emitted.y; r[20] = (double)emitted.z;
    }
    r[22] = (double)(S.materials[S.nodes[c.hit_node].link].kind & 0xFF);
    r[23] = (double)c.hit_node;
}
// Camera::get_ray through gen_camera_ray: pixel (i, j) of a W x H frame, sample `smp` -> origin, direction
__global__ void mlrt_probe_camera_kernel(const KParams<real> P, int n, const int32_t* ijs, double* out) {
    __shared__ FrameParams F;
    if (threadIdx.x < 21) (&F.cam_origin[0])[threadIdx.x] = (&P.cam_origin[0])[threadIdx.x];
    frame_reciprocals(P, F);
    if (threadIdx.x < 3) {
        F.bg_top[threadIdx.x] = P.bg_top[threadIdx.x]; F.bg_bottom[threadIdx.x] = P.bg_bottom[threadIdx.x];
        F.through0[threadIdx.x] = (float)(P.cam_llc[threadIdx.x] - P.cam_origin[threadIdx.x]);
        F.hor_f[threadIdx.x] = (float)P.cam_horizontal[threadIdx.x]; F.ver_f[threadIdx.x] = (float)P.cam_vertical[threadIdx.x];
        F.u_f[threadIdx.x] = (float)P.cam_u[threadIdx.x]; F.v_f[threadIdx.x] = (float)P.cam_v[threadIdx.x];
    }
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const RngKey key = {P.seed_lo, P.seed_hi};
    const int i = ijs[k * 3 + 0], j = ijs[k * 3 + 1];
    Rng g = {(uint32_t)(j * P.width + i), (uint32_t)ijs[k * 3 + 2], 0u, 0u};
    P3 o, d;
    gen_camera_ray(P, F, g, key, i, j, o, d);
    double* r = out + (size_t)k * 6;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
}
// the build's culling test (slab_hit_any over an entry made by the packer's rounding rule) on explicit boxes and rays
__global__ void mlrt_probe_slab_kernel(int n, const DNode<creal>* boxes, const double* rays, const double* tmax, int32_t* hit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = mkp(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]); r.d = mkp(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    r.a = 0.0; r.inv_a = 0.0;
    hit[i] = slab_hit_any(boxes[i], slab_ray(r), slab_far_bound(tmax[i])) ? 1 : 0;
}
// Divisor::over against the division itself on pseudo-random operand pairs made on the device: mantissas from Philox words,
// exponents uniform in [exp_lo, exp_hi] (numerator and divisor separately), every sign, a zero numerator now and then.
// out[0] += quotients that differ in any bit; out[1..3] = the bits of one such x, a and the shared quotient.
__global__ void mlrt_probe_divide_kernel(int per_thread, uint32_t seed, int ax_lo, int ax_hi, int xx_lo, int xx_hi, unsigned long long* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0ull;
    for (int k = 0; k < per_thread; k++) {
        uint32_t w[8];
        philox4x32_10(seed, 0x5eedu, tid, (uint32_t)k, 0u, 0u, w[0], w[1], w[2], w[3]);
        philox4x32_10(seed, 0x5eedu, tid, (uint32_t)k, 1u, 0u, w[4], w[5], w[6], w[7]);
        auto make = [](uint32_t lo, uint32_t hi, uint32_t e_word, int e_lo, int e_hi) {
            const uint64_t exponent = (uint64_t)(1023 + e_lo + (int)(e_word % (uint32_t)(e_hi - e_lo + 1)));
            const uint64_t bits = ((uint64_t)(hi >> 31) << 63) | (exponent << 52) | (((uint64_t)(hi & 0xFFFFFu) << 32) | lo);
            return __builtin_bit_cast(double, bits);
        };
        const double a = make(w[0], w[1], w[2], ax_lo, ax_hi);
        double x[3] = {make(w[3], w[4], w[5], xx_lo, xx_hi), make(w[6], w[7], w[2] >> 11, xx_lo, xx_hi), make(w[1], w[3], w[6] >> 7, xx_lo, xx_hi)};
        if ((w[7] & 63u) == 0u) x[1] = (w[7] & 64u) ? -0.0 : 0.0;
        if ((w[7] & 0xFC0u) == 0u) x[2] = x[0];  // (the low mantissa bits of a are often zero in practice: radii, squared lengths)
        double q[3];
        Divisor(a).over(x[0], x[1], x[2], q[0], q[1], q[2]);
        for (int c = 0; c < 3; c++) {
            const double plain = x[c] / a;
            if (__builtin_bit_cast(uint64_t, plain) != __builtin_bit_cast(uint64_t, q[c])) {
                bad++;
                out[1] = __builtin_bit_cast(uint64_t, x[c]); out[2] = __builtin_bit_cast(uint64_t, a); out[3] = __builtin_bit_cast(uint64_t, q[c]);
            }
        }
        double q2[2];
        Divisor(a).over(x[2], x[0], q2[0], q2[1]);
        if (__builtin_bit_cast(uint64_t, q2[0]) != __builtin_bit_cast(uint64_t, x[2] / a) || __builtin_bit_cast(uint64_t, q2[1]) != __builtin_bit_cast(uint64_t, x[0] / a)) bad++;
    }
    if (bad) atomicAdd(&out[0], bad);
}
// Perlin::turbulence (textures.rs:76-88) at scale * p: the serial evaluation, and - fast build - the wave-level one of the
// shading batch (one point per wave: 64 threads per point)
__global__ void mlrt_probe_turbulence_kernel(const KParams<real> P, int table, int n, const double* points, double* serial, double* wave) {
    Scene S;
    S.nodes = nullptr; S.materials = nullptr; S.textures = nullptr; S.images = nullptr;
    S.perlin_vec = reinterpret_cast<const real*>(P.blob + P.off_perlin_vec);
    S.perlin_perm = reinterpret_cast<const uint16_t*>(P.blob + P.off_perlin_perm);
    const int k = blockIdx.x;
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    const V3 p = mk((real)points[k * 3 + 0], (real)points[k * 3 + 1], (real)points[k * 3 + 2]);
    if (lane == 0) serial[k] = (double)perlin_turbulence(S, table, p);
    const float w = wave_turbulence(S, table, (float)p.x, (float)p.y, (float)p.z, lane);
    if (lane == 0) wave[k] = (double)w;
}
// log_uniform of the 24-bit uniforms (k0 + i) 2^-24, i < n: the medium's logarithm over its whole domain
__global__ void mlrt_probe_log_kernel(int k0, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log_uniform(u01d((uint32_t)(k0 + i) << 8));
}
// Texture::value (textures.rs, image_texture.rs) of texture `tex` at (u, v, p)
__global__ void mlrt_probe_texture_kernel(const KParams<real> P, int tex, int n, const double* uvp, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Scene S;
    S.nodes = nullptr;
    S.materials = reinterpret_cast<const DMaterial<real>*>(P.blob + P.off_materials);
    S.textures = reinterpret_cast<const DTexture<real>*>(P.blob + P.off_textures);
    S.perlin_vec = reinterpret_cast<const real*>(P.blob + P.off_perlin_vec);
    S.perlin_perm = reinterpret_cast<const uint16_t*>(P.blob + P.off_perlin_perm);
    S.images = P.images;
    const V3 c = texture_value<FEAT_ALL>(S, tex, mk(0, 0, 0), (real)uvp[i * 5 + 0], (real)uvp[i * 5 + 1], mk((real)uvp[i * 5 + 2], (real)uvp[i * 5 + 3], (real)uvp[i * 5 + 4]));
    out[i * 3 + 0] = (double)c.x; out[i * 3 + 1] = (double)c.y; out[i * 3 + 2] = (double)c.z;
}
// to_fixed (the kernel's accumulation of one finished sample) followed by the resolve kernel's to_rgb, on explicit radiance values
__global__ void mlrt_probe_accumulate_kernel(int n, const double* radiance, unsigned long long* acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    acc[i] = to_fixed((real)radiance[i]);
}
#endif  // MLRT_DEBUG_BUILD

}  // namespace MLRT_NS

// ---------------------------------------------------------------------------------------------- launcher

// Kernel variants: FULL supports every scene; LEAN is for scenes of spheres with solid-colour materials only (the
// `random` / `simple` worlds): 127 VGPRs instead of 168, so 1024-thread workgroups run at 4 waves per SIMD.
#ifndef MLRT_LEAN_BLOCK
#define MLRT_LEAN_BLOCK 768
#endif
namespace MLRT_NS {
constexpr uint32_t kLeanFeatures = 0;
// Variants 3 .. 3 + kSubsets - 1: the FULL kernel compiled for a subset of the scene features (mlrt_device.hpp FEAT_*) - the dead branches
// of the others leave the loop and the registers they held with them.  One per reference world that is not sphere-only, by what its
// builder creates (worlds.rs): cornell_box (:244-290), cornell_smoke (:292-338), final_scene (:368-469: everything but a Checker),
// earth (:165-189), simple_light (:214-242), two_spheres and debug_perlin (:191-212, :339-366: Noise), random_chk (:115-163: Checker).  The host picks the
// smallest compiled superset of a scene's features (variant 0 = all of them).
constexpr int kSubsets = 7;
constexpr uint32_t kSubsetFeatures[kSubsets] = {
    FEAT_BOX | FEAT_RECT | FEAT_XFORM, FEAT_BOX | FEAT_RECT | FEAT_XFORM | FEAT_MEDIUM, FEAT_ALL & ~FEAT_TEX_CHECKER,
    FEAT_TEX_IMAGE, FEAT_RECT | FEAT_TEX_NOISE, FEAT_TEX_NOISE, FEAT_TEX_CHECKER};
template <bool LDS, uint32_t FEAT, int BLOCK, bool PAIR = false>
int launch_variant(const KParams<real>& P, int grid, size_t dyn_lds, hipStream_t stream) {
    if (LDS || PAIR) {  // (the pair walk's stacks are dynamic LDS even when the scene stays in HBM)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlrt_render_kernel<LDS, FEAT, BLOCK, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((mlrt_render_kernel<LDS, FEAT, BLOCK, PAIR>), dim3((unsigned)grid), dim3(BLOCK), dyn_lds, stream, P);
    return (int)hipGetLastError();
}
template <bool LDS, uint32_t FEAT, int BLOCK, bool PAIR = false>
int occupancy_variant(size_t dyn_lds) {
    int blocks = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, mlrt_render_kernel<LDS, FEAT, BLOCK, PAIR>, BLOCK, dyn_lds);
    return e == hipSuccess ? blocks : -1;
}
// the pair-walk variant exists in the builds with f32 culling entries
template <bool LDS>
int launch_pair(const KParams<real>& P, int grid, size_t dyn_lds, hipStream_t stream) {
    if constexpr (kCull32) return launch_variant<LDS, kLeanFeatures, MLRT_LEAN_BLOCK, true>(P, grid, dyn_lds, stream);
    else return (int)hipErrorInvalidValue;
}
template <bool LDS>
int occupancy_pair(size_t dyn_lds) {
    if constexpr (kCull32) return occupancy_variant<LDS, kLeanFeatures, MLRT_LEAN_BLOCK, true>(dyn_lds);
    else return -1;
}
// the split-role kernel over the pair structure (KParams::pair_walk == 2)
template <bool LDS>
int launch_split(const KParams<real>& P, int grid, size_t dyn_lds, hipStream_t stream) {
#if MLRT_HAVE_SPLIT
    if constexpr (kCull32) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlrt_split_kernel<LDS, MLRT_LEAN_BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((mlrt_split_kernel<LDS, MLRT_LEAN_BLOCK>), dim3((unsigned)grid), dim3(MLRT_LEAN_BLOCK), dyn_lds, stream, P);
        return (int)hipGetLastError();
    }
#endif
    return (int)hipErrorInvalidValue;
}
}  // namespace MLRT_NS

// variant: 0 = FULL, 1 = LEAN (threaded list), 2 = LEAN over the pair structure (KParams::pair_walk)
extern "C" int MLRT_CAT(mlrt_launch_, MLRT_SUFFIX)(const void* kparams, int variant, int lds_scene, int grid, size_t dyn_lds, hipStream_t stream) {
    const mlrt::KParams<MLRT_NS::real>& P = *static_cast<const mlrt::KParams<MLRT_NS::real>*>(kparams);
    int rc;
    if (P.primary_lists) {
        const dim3 pgrid((unsigned)(P.n_rows * ((P.tiles_per_row + 63) / 64)));
        if (variant == 2) hipLaunchKernelGGL(MLRT_NS::mlrt_primary_lists_kernel<true>, pgrid, dim3(64), 0, stream, P, const_cast<uint16_t*>(P.primary_lists));
        else hipLaunchKernelGGL(MLRT_NS::mlrt_primary_lists_kernel<false>, pgrid, dim3(64), 0, stream, P, const_cast<uint16_t*>(P.primary_lists));
        rc = (int)hipGetLastError();
        if (rc != 0) return rc;
    }
    if (variant == 2 && P.pair_walk == 2) rc = lds_scene ? MLRT_NS::launch_split<true>(P, grid, dyn_lds, stream) : MLRT_NS::launch_split<false>(P, grid, dyn_lds, stream);
    else if (variant == 2) rc = lds_scene ? MLRT_NS::launch_pair<true>(P, grid, dyn_lds, stream) : MLRT_NS::launch_pair<false>(P, grid, dyn_lds, stream);
    else if (variant == 1) rc = lds_scene ? MLRT_NS::launch_variant<true, MLRT_NS::kLeanFeatures, MLRT_LEAN_BLOCK>(P, grid, dyn_lds, stream)
                                     : MLRT_NS::launch_variant<false, MLRT_NS::kLeanFeatures, MLRT_LEAN_BLOCK>(P, grid, dyn_lds, stream);
#define MLRT_SUBSET_LAUNCH(k) else if (variant == 3 + k) rc = lds_scene ? MLRT_NS::launch_variant<true, MLRT_NS::kSubsetFeatures[k], MLRT_NS::full_block(MLRT_NS::kSubsetFeatures[k])>(P, grid, dyn_lds, stream) \
                                                                         : MLRT_NS::launch_variant<false, MLRT_NS::kSubsetFeatures[k], MLRT_NS::full_block(MLRT_NS::kSubsetFeatures[k])>(P, grid, dyn_lds, stream);
    MLRT_SUBSET_LAUNCH(0) MLRT_SUBSET_LAUNCH(1) MLRT_SUBSET_LAUNCH(2) MLRT_SUBSET_LAUNCH(3) MLRT_SUBSET_LAUNCH(4) MLRT_SUBSET_LAUNCH(5) MLRT_SUBSET_LAUNCH(6)
#undef MLRT_SUBSET_LAUNCH
    else rc = lds_scene ? MLRT_NS::launch_variant<true, mlrt::FEAT_ALL, MLRT_NS::full_block(mlrt::FEAT_ALL)>(P, grid, dyn_lds, stream)
                        : MLRT_NS::launch_variant<false, mlrt::FEAT_ALL, MLRT_NS::full_block(mlrt::FEAT_ALL)>(P, grid, dyn_lds, stream);
    if (rc != 0) return rc;
    {
        const int n = P.n_rows * P.width * 3;
        const int rgrid = (n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048;
        hipLaunchKernelGGL(MLRT_NS::mlrt_resolve_kernel, dim3((unsigned)rgrid), dim3(256), 0, stream, P.acc_global, n, P.spp, P.rgb_out, P.accum_out, P.tile_counter);
    }
    return (int)hipGetLastError();
}
extern "C" int MLRT_CAT(mlrt_perlin_in_hbm_, MLRT_SUFFIX)(void) { return MLRT_NS::kPerlinInHbm ? 1 : 0; }
extern "C" int MLRT_CAT(mlrt_lean_materials_in_hbm_, MLRT_SUFFIX)(void) { return MLRT_NS::kLeanMaterialsInHbm ? 1 : 0; }
extern "C" int MLRT_CAT(mlrt_instrumented_, MLRT_SUFFIX)(void) {
#ifdef MLRT_DEBUG_BUILD
    return 1;
#else
    return 0;
#endif
}
extern "C" int MLRT_CAT(mlrt_variant_count_, MLRT_SUFFIX)(void) { return 3 + MLRT_NS::kSubsets; }
extern "C" uint32_t MLRT_CAT(mlrt_variant_features_, MLRT_SUFFIX)(int variant) {
    if (variant == 1 || variant == 2) return MLRT_NS::kLeanFeatures;
    return (variant >= 3 && variant < 3 + MLRT_NS::kSubsets) ? MLRT_NS::kSubsetFeatures[variant - 3] : mlrt::FEAT_ALL;
}
extern "C" int MLRT_CAT(mlrt_block_threads_, MLRT_SUFFIX)(int variant) {
    return (variant == 1 || variant == 2) ? MLRT_LEAN_BLOCK : MLRT_NS::full_block(MLRT_CAT(mlrt_variant_features_, MLRT_SUFFIX)(variant));
}
extern "C" int MLRT_CAT(mlrt_has_pair_walk_, MLRT_SUFFIX)(void) { return MLRT_NS::kCull32 ? 1 : 0; }
// split-role kernel: bytes of LDS behind the frame parameters for a scene whose stacks need `stack_levels` rows (0: the build has no such kernel)
extern "C" size_t MLRT_CAT(mlrt_split_lds_bytes_, MLRT_SUFFIX)(int stack_levels) {
#if MLRT_HAVE_SPLIT
    if (MLRT_NS::kCull32) {
        const size_t stacks = (size_t)(stack_levels + 1) * (size_t)(MLRT_NS::split_t_waves(MLRT_LEAN_BLOCK) * 64) * sizeof(int16_t);
        return ((stacks + 15) & ~(size_t)15) + MLRT_NS::split_queue_bytes();
    }
#endif
    return 0;
}
extern "C" int MLRT_CAT(mlrt_acc_bytes_per_wave_, MLRT_SUFFIX)(int variant) { return MLRT_NS::acc_bytes_per_wave(MLRT_CAT(mlrt_variant_features_, MLRT_SUFFIX)(variant)); }
extern "C" int MLRT_CAT(mlrt_lds_frame_bytes_, MLRT_SUFFIX)(void) { return MLRT_NS::kFrameParamBytes; }
extern "C" int MLRT_CAT(mlrt_occupancy_, MLRT_SUFFIX)(int variant, int lds_scene, size_t dyn_lds) {
    if (variant == 2) return lds_scene ? MLRT_NS::occupancy_pair<true>(dyn_lds) : MLRT_NS::occupancy_pair<false>(dyn_lds);
    if (variant == 1) return lds_scene ? MLRT_NS::occupancy_variant<true, MLRT_NS::kLeanFeatures, MLRT_LEAN_BLOCK>(dyn_lds) : MLRT_NS::occupancy_variant<false, MLRT_NS::kLeanFeatures, MLRT_LEAN_BLOCK>(dyn_lds);
#define MLRT_SUBSET_OCC(k) if (variant == 3 + k) return lds_scene ? MLRT_NS::occupancy_variant<true, MLRT_NS::kSubsetFeatures[k], MLRT_NS::full_block(MLRT_NS::kSubsetFeatures[k])>(dyn_lds) : MLRT_NS::occupancy_variant<false, MLRT_NS::kSubsetFeatures[k], MLRT_NS::full_block(MLRT_NS::kSubsetFeatures[k])>(dyn_lds);
    MLRT_SUBSET_OCC(0) MLRT_SUBSET_OCC(1) MLRT_SUBSET_OCC(2) MLRT_SUBSET_OCC(3) MLRT_SUBSET_OCC(4) MLRT_SUBSET_OCC(5) MLRT_SUBSET_OCC(6)
#undef MLRT_SUBSET_OCC
    return lds_scene ? MLRT_NS::occupancy_variant<true, mlrt::FEAT_ALL, MLRT_NS::full_block(mlrt::FEAT_ALL)>(dyn_lds) : MLRT_NS::occupancy_variant<false, mlrt::FEAT_ALL, MLRT_NS::full_block(mlrt::FEAT_ALL)>(dyn_lds);
}

#ifdef MLRT_DEBUG_BUILD
// Probe launchers (see "directed probes" above).  `kind`: 0 rays, 1 camera, 2 slab, 3 turbulence, 4 texture, 5 accumulate + resolve.
// All pointers are device pointers; returns a hipError_t.
extern "C" int MLRT_CAT(mlrt_probe_launch_, MLRT_SUFFIX)(int kind, const void* kparams, int n, const void* a, const void* b, void* c, void* d2, int iarg,
                                                          hipStream_t stream) {
    using namespace MLRT_NS;
    const mlrt::KParams<real>& P = *static_cast<const mlrt::KParams<real>*>(kparams);
    const unsigned grid = (unsigned)((n + 63) / 64);
    if (n <= 0) return 0;
    if (kind == 0) hipLaunchKernelGGL((mlrt_probe_rays_kernel<mlrt::FEAT_ALL>), dim3(grid), dim3(64), 0, stream, P, n, static_cast<const double*>(a), static_cast<const uint32_t*>(b), static_cast<double*>(c));
    else if (kind == 1) hipLaunchKernelGGL(mlrt_probe_camera_kernel, dim3(grid), dim3(64), 0, stream, P, n, static_cast<const int32_t*>(a), static_cast<double*>(c));
    else if (kind == 2) hipLaunchKernelGGL(mlrt_probe_slab_kernel, dim3(grid), dim3(64), 0, stream, n, static_cast<const mlrt::DNode<creal>*>(kparams), static_cast<const double*>(a), static_cast<const double*>(b), static_cast<int32_t*>(c));
    else if (kind == 3) hipLaunchKernelGGL(mlrt_probe_turbulence_kernel, dim3((unsigned)n), dim3(64), 0, stream, P, iarg, n, static_cast<const double*>(a), static_cast<double*>(c), static_cast<double*>(d2));
    else if (kind == 4) hipLaunchKernelGGL(mlrt_probe_texture_kernel, dim3(grid), dim3(64), 0, stream, P, iarg, n, static_cast<const double*>(a), static_cast<double*>(c));
    else if (kind == 5) {
        hipLaunchKernelGGL(mlrt_probe_accumulate_kernel, dim3(grid), dim3(64), 0, stream, n, static_cast<const double*>(a), static_cast<unsigned long long*>(d2));
        hipLaunchKernelGGL(mlrt_resolve_kernel, dim3(grid), dim3(64), 0, stream, static_cast<unsigned long long*>(d2), n, iarg, static_cast<int32_t*>(c), static_cast<double*>(nullptr), static_cast<unsigned int*>(const_cast<void*>(b)));
    } else if (kind == 6) {
        const int* e = static_cast<const int*>(a);  // host pointer: per_thread, seed, divisor exponents lo / hi, numerator exponents lo / hi
        hipLaunchKernelGGL(mlrt_probe_divide_kernel, dim3((unsigned)n), dim3(256), 0, stream, e[0], (uint32_t)e[1], e[2], e[3], e[4], e[5], static_cast<unsigned long long*>(c));
    } else if (kind == 7) {
        hipLaunchKernelGGL(mlrt_probe_log_kernel, dim3(grid), dim3(64), 0, stream, iarg, n, static_cast<double*>(c));
    } else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
extern "C" int MLRT_CAT(mlrt_probe_entry_bytes_, MLRT_SUFFIX)(void) { return (int)sizeof(mlrt::DNode<MLRT_NS::creal>); }
#endif

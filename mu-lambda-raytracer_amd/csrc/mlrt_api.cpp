// csrc/mlrt_api.cpp - the C ABI of include/mlrt.h: scene validation, packing into the device layout
// (threaded BVH blob, see mlrt_device.hpp), upload to HBM and kernel launch.  Host code only; the kernels
// live in mlrt_kernel.hip (built twice: f32 and f64).  There is no CPU render path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <functional>
#include <vector>

#include "../../include/mlrt.h"
#include "mlrt_device.hpp"

// One set of entry points per kernel build (mlrt_kernel.hip compiled with MLRT_SUFFIX = f32 | f64 | f64s).
// variant: 0 = FULL (every feature), 1 = LEAN over the threaded list, 2 = LEAN over the pair structure, 3 .. = FULL compiled for a subset
// of the features (see mlrt_kernel.hip)
#define MLRT_DECLARE_BUILD(sfx)                                                                                                    \
    int mlrt_launch_##sfx(const void* kparams, int variant, int lds_scene, int grid, size_t dyn_lds, hipStream_t stream);                \
    int mlrt_instrumented_##sfx(void);                                                                                             \
    uint32_t mlrt_variant_features_##sfx(int variant);                                                                             \
    int mlrt_variant_count_##sfx(void);                                                                                            \
    int mlrt_block_threads_##sfx(int variant);                                                                                     \
    int mlrt_acc_bytes_per_wave_##sfx(int variant);                                                                                       \
    int mlrt_lds_frame_bytes_##sfx(void);                                                                                          \
    int mlrt_occupancy_##sfx(int variant, int lds_scene, size_t dyn_lds);                                                          \
    int mlrt_perlin_in_hbm_##sfx(void);                                                                                           \
    int mlrt_lean_materials_in_hbm_##sfx(void);                                                                                    \
    int mlrt_has_pair_walk_##sfx(void);                                                                                            \
    size_t mlrt_split_lds_bytes_##sfx(int stack_levels);
extern "C" {
MLRT_DECLARE_BUILD(f32)
MLRT_DECLARE_BUILD(f64)
MLRT_DECLARE_BUILD(f64s)
MLRT_DECLARE_BUILD(f64c)
#ifdef MLRT_DEBUG_BUILD
#define MLRT_DECLARE_PROBE(sfx) \
    int mlrt_probe_launch_##sfx(int kind, const void* kparams, int n, const void* a, const void* b, void* c, void* d2, int iarg, hipStream_t stream);
MLRT_DECLARE_PROBE(f32)
MLRT_DECLARE_PROBE(f64)
MLRT_DECLARE_PROBE(f64s)
MLRT_DECLARE_PROBE(f64c)
#endif
}

namespace mlrt {
void set_last_error(const std::string& msg);  // host/host_api.cpp (same library)
const char* last_error_cstr();

namespace {

struct KernelBuild {
    int (*launch)(const void*, int, int, int, size_t, hipStream_t);
    int (*instrumented)(void);
    uint32_t (*variant_features)(int);
    int (*block_threads)(int);
    int (*acc_bytes_per_wave)(int);
    int (*lds_frame_bytes)(void);
    int (*occupancy)(int, int, size_t);
    int (*probe)(int, const void*, int, const void*, const void*, void*, void*, int, hipStream_t);  // debug library only, else nullptr
    int (*perlin_in_hbm)(void);
    int (*lean_materials_in_hbm)(void);
    int (*has_pair_walk)(void);
    int (*variant_count)(void);
    size_t (*split_lds_bytes)(int);
};
#ifdef MLRT_DEBUG_BUILD
#define MLRT_PROBE_OF(sfx) mlrt_probe_launch_##sfx
#else
#define MLRT_PROBE_OF(sfx) nullptr
#endif
#define MLRT_BUILD_ROW(sfx) {mlrt_launch_##sfx, mlrt_instrumented_##sfx, mlrt_variant_features_##sfx, mlrt_block_threads_##sfx, mlrt_acc_bytes_per_wave_##sfx, mlrt_lds_frame_bytes_##sfx, mlrt_occupancy_##sfx, MLRT_PROBE_OF(sfx), mlrt_perlin_in_hbm_##sfx, mlrt_lean_materials_in_hbm_##sfx, mlrt_has_pair_walk_##sfx, mlrt_variant_count_##sfx, mlrt_split_lds_bytes_##sfx}
const KernelBuild kBuilds[4] = {MLRT_BUILD_ROW(f32), MLRT_BUILD_ROW(f64), MLRT_BUILD_ROW(f64s), MLRT_BUILD_ROW(f64c)};  // MLRT_PRECISION_*; [3]: FP64 with options.cull_f64

struct Fail {
    int32_t code;
    std::string msg;
};
#define MLRT_REQUIRE(cond, text) \
    do { if (!(cond)) throw Fail{MLRT_ERR_INVALID, std::string(text)}; } while (0)
#define MLRT_HIP(call) \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) throw Fail{MLRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)}; } while (0)

void validate(const mlrt_scene& s) {
    MLRT_REQUIRE(s.n_objects > 0 && s.objects, "scene has no objects");
    MLRT_REQUIRE(s.n_spheres >= 0 && s.n_boxes >= 0 && s.n_rects >= 0 && s.n_prim_refs >= 0 && s.n_bvh_nodes >= 0 && s.n_bvhs >= 0 &&
                     s.n_transforms >= 0 && s.n_materials >= 0 && s.n_textures >= 0 && s.n_images >= 0 && s.n_perlins >= 0,
                 "negative element count");
    MLRT_REQUIRE((s.n_spheres == 0 || s.spheres) && (s.n_boxes == 0 || s.boxes) && (s.n_rects == 0 || s.rects) && (s.n_prim_refs == 0 || s.prim_refs) &&
                     (s.n_bvh_nodes == 0 || s.bvh_nodes) && (s.n_bvhs == 0 || s.bvhs) && (s.n_transforms == 0 || s.transforms) && (s.n_materials == 0 || s.materials) &&
                     (s.n_textures == 0 || s.textures) && (s.n_images == 0 || s.images) && (s.n_perlins == 0 || s.perlins),
                 "an array with a positive element count is NULL");
    // Geometry must be finite: the reference would render NaNs, the device structure (bounds, SAH sort) cannot be built from them.
    // ... and of a magnitude a ray tracer can work with: below 2^100 (1.3e30).  (The f64 kernels' shared-divisor quotients, mlrt_kernel.hip
    // Divisor, are the compiler's divisions bit for bit for numerators below 2^600: products of two such lengths stay far below.)
    auto finite = [](const double* v, int n) { for (int k = 0; k < n; k++) if (!(std::fabs(v[k]) < 0x1p100)) return false; return true; };
    auto mat_ok = [&](int32_t m) { return m >= 0 && m < s.n_materials; };
    for (int i = 0; i < s.n_spheres; i++) {
        MLRT_REQUIRE(mat_ok(s.spheres[i].material), "sphere material out of range");
        MLRT_REQUIRE(finite(s.spheres[i].center, 3) && finite(&s.spheres[i].radius, 1), "sphere centre / radius is not finite (or is 2^100 or more)");
    }
    for (int i = 0; i < s.n_boxes; i++) {
        MLRT_REQUIRE(mat_ok(s.boxes[i].material), "box material out of range");
        MLRT_REQUIRE(finite(s.boxes[i].min, 3) && finite(s.boxes[i].max, 3), "box corner is not finite (or is 2^100 or more)");
        // Block::new(p0, p1) with a corner pair that is not (min, max) builds degenerate rects in the reference (AARect::new keeps
        // a1_v0 as given, aarects.rs:39 - quirk Q2 - so a side can come out with a1_v0 > a1_v1 and never be hit): not a shape
        // anyone means, and not what the device's slab form of a Block would render
        for (int k = 0; k < 3; k++) MLRT_REQUIRE(s.boxes[i].min[k] <= s.boxes[i].max[k], "box min must not exceed max (Block::new with reversed corners is degenerate in the reference)");
    }
    for (int i = 0; i < s.n_rects; i++) {
        const mlrt_rect& r = s.rects[i];
        MLRT_REQUIRE(mat_ok(r.material), "rect material out of range");
        // aarects.rs:9-17 other(): only (X,Y), (X,Z), (Y,Z) are legal, anything else panics
        MLRT_REQUIRE((r.a0 == 0 && r.a1 == 1) || (r.a0 == 0 && r.a1 == 2) || (r.a0 == 1 && r.a1 == 2), "rect axes must be (0,1), (0,2) or (1,2)");
        MLRT_REQUIRE(finite(&r.a0_v0, 1) && finite(&r.a0_v1, 1) && finite(&r.a1_v0, 1) && finite(&r.a1_v1, 1) && finite(&r.plane, 1), "rect coordinate is not finite (or is 2^100 or more)");
    }
    auto prim_ok = [&](const mlrt_prim_ref& p) {
        if (p.kind == MLRT_PRIM_SPHERE) return p.index >= 0 && p.index < s.n_spheres;
        if (p.kind == MLRT_PRIM_BOX) return p.index >= 0 && p.index < s.n_boxes;
        if (p.kind == MLRT_PRIM_RECT) return p.index >= 0 && p.index < s.n_rects;
        return false;
    };
    for (int i = 0; i < s.n_prim_refs; i++) MLRT_REQUIRE(prim_ok(s.prim_refs[i]), "prim_ref out of range");
    auto child_ok = [&](int32_t c) { return c >= 0 ? c < s.n_bvh_nodes : (~c) < s.n_prim_refs; };
    for (int i = 0; i < s.n_bvh_nodes; i++) {
        MLRT_REQUIRE(child_ok(s.bvh_nodes[i].left) && child_ok(s.bvh_nodes[i].right), "bvh child out of range");
        MLRT_REQUIRE(finite(s.bvh_nodes[i].bounds_min, 3) && finite(s.bvh_nodes[i].bounds_max, 3), "bvh node bounds are not finite (or are 2^100 or more)");
    }
    for (int i = 0; i < s.n_bvhs; i++) {
        MLRT_REQUIRE(s.bvhs[i].root == MLRT_BVH_EMPTY || child_ok(s.bvhs[i].root), "bvh root out of range");
        // a tree (bhv.rs:109-120: every Node owns its children): a walk that visits more than every node and leaf once has met a cycle
        int64_t budget = 2 * (int64_t)s.n_bvh_nodes + s.n_prim_refs + 2;
        std::vector<int32_t> stack;
        if (s.bvhs[i].root >= 0) stack.push_back(s.bvhs[i].root);
        while (!stack.empty()) {
            const int32_t node = stack.back();
            stack.pop_back();
            MLRT_REQUIRE(--budget >= 0, "bvh is cyclic or larger than its node count allows");
            if (s.bvh_nodes[node].left >= 0) stack.push_back(s.bvh_nodes[node].left);
            if (s.bvh_nodes[node].right >= 0) stack.push_back(s.bvh_nodes[node].right);
        }
    }
    for (int i = 0; i < s.n_transforms; i++) {
        const mlrt_transform& t = s.transforms[i];
        MLRT_REQUIRE(!t.has_rotate || (t.axis >= 0 && t.axis <= 2), "rotation axis out of range");
        MLRT_REQUIRE((!t.has_rotate || (std::isfinite(t.sin_theta) && std::isfinite(t.cos_theta))) && (!t.has_translate || finite(t.offset, 3)), "transform is not finite");
    }
    for (int i = 0; i < s.n_textures; i++) {
        const mlrt_texture& t = s.textures[i];
        if (t.kind == MLRT_TEX_IMAGE) MLRT_REQUIRE(t.index >= 0 && t.index < s.n_images, "image texture index out of range");
        else if (t.kind == MLRT_TEX_NOISE) MLRT_REQUIRE(t.index >= 0 && t.index < s.n_perlins, "noise texture index out of range");
        else if (t.kind == MLRT_TEX_CHECKER) MLRT_REQUIRE(t.index >= 0 && t.index < s.n_textures && t.index2 >= 0 && t.index2 < s.n_textures, "checker texture index out of range");
        else MLRT_REQUIRE(t.kind == MLRT_TEX_SOLID, "unknown texture kind");
    }
    // Checker(odd, even) owns its two textures in the reference (textures.rs:28-38): a tree.  The device follows the indices for
    // at most 8 levels, so deeper nesting - and any cycle - is refused rather than rendered differently.
    std::function<int(int32_t, int)> checker_depth = [&](int32_t tex, int depth) -> int {
        if (depth > 8 || s.textures[tex].kind != MLRT_TEX_CHECKER) return depth;
        return std::max(checker_depth(s.textures[tex].index, depth + 1), checker_depth(s.textures[tex].index2, depth + 1));
    };
    for (int i = 0; i < s.n_textures; i++) MLRT_REQUIRE(checker_depth(i, 0) <= 8, "checker textures nest deeper than 8 levels (or refer to themselves)");
    for (int i = 0; i < s.n_images; i++) {
        MLRT_REQUIRE(s.images[i].width > 0 && s.images[i].height > 0 && s.images[i].rgb, "empty image");
        MLRT_REQUIRE((int64_t)s.images[i].width * s.images[i].height <= (int64_t)1 << 28, "image texture larger than 2^28 texels");
    }
    for (int i = 0; i < s.n_perlins; i++) {
        MLRT_REQUIRE(s.perlins[i].ranvec && s.perlins[i].perm, "perlin table is NULL");
        // the three permutations index the 1024 gradient vectors (textures.rs:113-117): anything else would read outside the table
        for (int k = 0; k < 3072; k++) MLRT_REQUIRE(s.perlins[i].perm[k] >= 0 && s.perlins[i].perm[k] < 1024, "perlin permutation entry outside 0..1023");
    }
    for (int i = 0; i < s.n_materials; i++) {
        const mlrt_material& m = s.materials[i];
        MLRT_REQUIRE(m.kind >= MLRT_MAT_LAMBERTIAN && m.kind <= MLRT_MAT_ISOTROPIC, "unknown material kind");
        if (m.kind == MLRT_MAT_LAMBERTIAN || m.kind == MLRT_MAT_DIFFUSE_LIGHT || m.kind == MLRT_MAT_ISOTROPIC)
            MLRT_REQUIRE(m.texture >= 0 && m.texture < s.n_textures, "material texture out of range");
    }
    for (int i = 0; i < s.n_objects; i++) {
        const mlrt_object& o = s.objects[i];
        MLRT_REQUIRE(o.transform >= -1 && o.transform < s.n_transforms, "object transform out of range");
        if (o.kind == MLRT_OBJ_BVH) MLRT_REQUIRE(o.bvh >= 0 && o.bvh < s.n_bvhs, "object bvh out of range");
        else if (o.kind == MLRT_OBJ_PRIM) MLRT_REQUIRE(prim_ok(o.prim), "object primitive out of range");
        else if (o.kind == MLRT_OBJ_MEDIUM) MLRT_REQUIRE(prim_ok(o.prim) && mat_ok(o.phase_material) && std::isfinite(o.neg_inv_density), "medium boundary/material out of range or density not finite");
        else MLRT_REQUIRE(false, "unknown object kind");
    }
}

// Which kernel features the scene needs (mlrt_device.hpp FEAT_*).
uint32_t scene_features(const mlrt_scene& s) {
    uint32_t f = 0;
    if (s.n_boxes > 0) f |= FEAT_BOX;
    if (s.n_rects > 0) f |= FEAT_RECT;
    for (int i = 0; i < s.n_objects; i++) {
        if (s.objects[i].kind == MLRT_OBJ_MEDIUM) f |= FEAT_MEDIUM;
        if (s.objects[i].transform >= 0) f |= FEAT_XFORM;
    }
    for (int i = 0; i < s.n_materials; i++) if (s.materials[i].kind == MLRT_MAT_ISOTROPIC) f |= FEAT_MEDIUM;
    for (int i = 0; i < s.n_textures; i++) {
        if (s.textures[i].kind == MLRT_TEX_IMAGE) f |= FEAT_TEX_IMAGE;
        if (s.textures[i].kind == MLRT_TEX_NOISE) f |= FEAT_TEX_NOISE;
        if (s.textures[i].kind == MLRT_TEX_CHECKER) f |= FEAT_TEX_CHECKER;
    }
    return f;
}

bool texture_needs_uv(const mlrt_scene& s, int32_t tex, int depth) {
    if (tex < 0 || depth > 8) return false;
    const mlrt_texture& t = s.textures[tex];
    if (t.kind == MLRT_TEX_IMAGE) return true;
    if (t.kind == MLRT_TEX_CHECKER) return texture_needs_uv(s, t.index, depth + 1) || texture_needs_uv(s, t.index2, depth + 1);
    return false;
}

// ---- packing -----------------------------------------------------------------------------------------------------
// R = the entries' type (mlrt_device.hpp: f32 entries with payload slots, or f64 entries), M = the type of materials,
// textures and Perlin tables (= the arithmetic of the build).
enum class Round { EXACT, OUTWARD, RIGOROUS };
// Box bounds in the entry's precision, moved OUTWARD so the culling test stays conservative.  f32 entries: rounded away
// from the box (RIGOROUS - the product f64 build, whose culling may never reject what an f64 primitive test accepts - one
// more ulp, so that a bound that happens to be exact in f32 still clears the f64 rounding of the primitive's own arithmetic).
// f64 entries (where the conversion is exact): 1e-12 relative - the cull_f64 product build tests them by reciprocal + fma, a few
// 1e-16 away from the division the primitive tests use.  EXACT (strict build): the reference's own bounds, untouched.
template <typename R>
R round_down(double x, Round mode) {
    R f = (R)x;
    if (mode == Round::EXACT) return f;
    if (sizeof(R) == 8) return (R)(x - 1e-12 * (1.0 + std::fabs(x)));
    if ((double)f > x) f = std::nextafter(f, -std::numeric_limits<R>::infinity());
    if (mode == Round::RIGOROUS) f = std::nextafter(f, -std::numeric_limits<R>::infinity());
    return f;
}
template <typename R>
R round_up(double x, Round mode) {
    R f = (R)x;
    if (mode == Round::EXACT) return f;
    if (sizeof(R) == 8) return (R)(x + 1e-12 * (1.0 + std::fabs(x)));
    if ((double)f < x) f = std::nextafter(f, std::numeric_limits<R>::infinity());
    if (mode == Round::RIGOROUS) f = std::nextafter(f, std::numeric_limits<R>::infinity());
    return f;
}

struct Bounds {
    double lo[3], hi[3];
    void grow(const Bounds& b) { for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], b.lo[k]); hi[k] = std::fmax(hi[k], b.hi[k]); } }
    double half_area() const { const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2]; return x * y + y * z + z * x; }
};
Bounds empty_bounds() { Bounds b; for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; } return b; }
bool prim_bounds(const mlrt_scene& s, const mlrt_prim_ref& p, Bounds& b) {
    if (p.kind == MLRT_PRIM_SPHERE) {
        const mlrt_sphere& sp = s.spheres[p.index];
        for (int k = 0; k < 3; k++) { b.lo[k] = std::fmin(sp.center[k] - sp.radius, sp.center[k] + sp.radius); b.hi[k] = std::fmax(sp.center[k] - sp.radius, sp.center[k] + sp.radius); }
        return true;
    }
    if (p.kind == MLRT_PRIM_BOX) {
        const mlrt_box& bx = s.boxes[p.index];
        for (int k = 0; k < 3; k++) { b.lo[k] = std::fmin(bx.min[k], bx.max[k]); b.hi[k] = std::fmax(bx.min[k], bx.max[k]); }
        return true;
    }
    const mlrt_rect& r = s.rects[p.index];  // zero thickness: padded so the slab test (which rejects tmax <= tmin) can pass
    const int ap = 3 - r.a0 - r.a1;
    const double pad = 1e-4 * (1.0 + std::fabs(r.plane));
    b.lo[r.a0] = r.a0_v0; b.hi[r.a0] = r.a0_v1; b.lo[r.a1] = r.a1_v0; b.hi[r.a1] = r.a1_v1; b.lo[ap] = r.plane - pad; b.hi[ap] = r.plane + pad;
    return true;
}

// The entry array under construction.  Links of entries marked `reloc` are indices relative to the start of the array they
// were emitted into (a sub-array is appended to the world with append()); payload slots are raw bytes and never touched.
template <typename R>
struct Entries {
    static constexpr bool kCull32 = sizeof(R) == 4;
    std::vector<DNode<R>> nodes;
    std::vector<uint8_t> reloc;
    std::vector<int32_t> spheres;  // indices of the sphere entries (the candidate pre-pass scans these)
    bool payload_all = false;      // product f64 build over f32 entries
    Round round = Round::OUTWARD;

    int32_t size() const { return (int32_t)nodes.size(); }
    int32_t push(const DNode<R>& n, bool relocatable) { nodes.push_back(n); reloc.push_back(relocatable ? 1 : 0); return size() - 1; }
    void payload(const double* v, int count) {  // f32 entries only: `count` doubles into ceil(count / 4) raw slots
        static_assert(sizeof(DNode<float>) == 4 * sizeof(double), "a payload slot holds four doubles");
        for (int k0 = 0; k0 < count; k0 += 4) {
            double four[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4 && k0 + k < count; k++) four[k] = v[k0 + k];
            DNode<R> raw;
            std::memcpy(static_cast<void*>(&raw), four, sizeof raw <= sizeof four ? sizeof raw : sizeof four);
            push(raw, false);
        }
    }
    DNode<R> box_entry(const Bounds& b, int32_t skip) const {
        DNode<R> n{};
        for (int k = 0; k < 3; k++) { n.v[k] = round_down<R>(b.lo[k], round); n.v[3 + k] = round_up<R>(b.hi[k], round); }
        n.tag = TAG_INNER;
        n.link = skip;
        return n;
    }
    // One primitive.  `wrap`: with a culling box (an inner entry) in front; Block entries are never wrapped: the step tests
    // their own bounds.
    void leaf(const mlrt_scene& s, const mlrt_prim_ref& p, bool wrap) {
        Bounds pb = empty_bounds();
        prim_bounds(s, p, pb);
        if (p.kind == MLRT_PRIM_SPHERE) {
            const mlrt_sphere& sp = s.spheres[p.index];
            const double v[4] = {sp.center[0], sp.center[1], sp.center[2], sp.radius};
            if (payload_all) {
                // (a sphere leaf fused with its box into one entry was tried: one entry less to step over, ~4 instructions more
                // in every step: -2 % on C2 in the product f64 build, -3.5 % in the fast build)
                if (wrap) push(box_entry(pb, size() + 3), true);
                DNode<R> n{};
                n.link = sp.material; n.tag = TAG_SPHERE;
                spheres.push_back(push(n, false));
                payload(v, 4);
            } else {
                if (wrap) push(box_entry(pb, size() + 2), true);
                DNode<R> n{};
                for (int k = 0; k < 4; k++) n.v[k] = (R)v[k];
                n.link = sp.material; n.tag = TAG_SPHERE;
                spheres.push_back(push(n, false));
            }
        } else if (p.kind == MLRT_PRIM_BOX) {
            const mlrt_box& b = s.boxes[p.index];
            const double v[6] = {b.min[0], b.min[1], b.min[2], b.max[0], b.max[1], b.max[2]};
            DNode<R> n{};
            if (payload_all) n = box_entry(pb, 0);  // outward bounds for the step's culling test; the faces are in the payload
            else for (int k = 0; k < 6; k++) n.v[k] = (R)v[k];
            n.link = b.material; n.tag = TAG_BOX;
            push(n, false);
            if (payload_all) payload(v, 6);
        } else {
            const mlrt_rect& r = s.rects[p.index];
            const double v[5] = {r.a0_v0, r.a0_v1, r.a1_v0, r.a1_v1, r.plane};
            if (wrap) push(box_entry(pb, size() + 1 + entry_slots(TAG_RECT, kCull32, payload_all)), true);
            DNode<R> n{};
            if (!payload_all) for (int k = 0; k < 5; k++) n.v[k] = (R)v[k];
            n.link = r.material;
            n.tag = TAG_RECT | (r.a0 << 4) | (r.a1 << 6) | ((3 - r.a0 - r.a1) << 8);
            push(n, false);
            if (payload_all) payload(v, 5);
        }
    }
    void append(const Entries<R>& body) {
        const int32_t base = size();
        for (size_t i = 0; i < body.nodes.size(); i++) {
            DNode<R> e = body.nodes[i];
            if (body.reloc[i]) e.link += base;
            push(e, false);
        }
        for (int32_t sidx : body.spheres) spheres.push_back(sidx + base);
    }
};

template <typename R, typename M>
struct Packed {
    Entries<R> entries;
    std::vector<DMaterial<M>> materials;
    std::vector<DTexture<M>> textures;
    std::vector<M> perlin_vec;
    std::vector<uint16_t> perlin_perm;
    std::vector<uint16_t> sphere_index;
    std::vector<unsigned char> blob;
    int32_t off_nodes = 0, off_materials = 0, off_textures = 0, off_perlin_vec = 0, off_perlin_perm = 0, off_sphere_index = 0;
};

// Depth-first emission of the REFERENCE topology with skip links.  `budget` guards against cyclic child references.
template <typename R>
void emit_subtree(const mlrt_scene& s, int32_t child, Entries<R>& out, int& budget, bool wrap_leaves) {
    MLRT_REQUIRE(--budget >= 0, "bvh is cyclic or larger than its node count allows");
    if (child < 0) {
        const mlrt_prim_ref& p = s.prim_refs[~child];
        out.leaf(s, p, wrap_leaves && p.kind == MLRT_PRIM_SPHERE);
        return;
    }
    const mlrt_bvh_node& src = s.bvh_nodes[child];
    Bounds b;
    for (int k = 0; k < 3; k++) { b.lo[k] = src.bounds_min[k]; b.hi[k] = src.bounds_max[k]; }
    const int32_t me = out.push(out.box_entry(b, 0), true);
    emit_subtree<R>(s, src.left, out, budget, wrap_leaves);
    emit_subtree<R>(s, src.right, out, budget, wrap_leaves);
    out.nodes[(size_t)me].link = out.size();
}

// ---- device-side BVH quality -----------------------------------------------------------------------------------------
// The closest hit does not depend on the BVH's topology, so every build but the strict one is free to replace the
// reference's random-axis median split (bhv.rs:122-145, kept for the scene RNG replay on the host and used as-is by the strict
// build) with a surface-area-heuristic tree over the same primitives: fewer box tests per ray, same image.
void collect_leaves(const mlrt_scene& s, int32_t child, std::vector<int32_t>& refs, int& budget) {
    MLRT_REQUIRE(--budget >= 0, "bvh is cyclic or larger than its node count allows");
    if (child < 0) { refs.push_back(~child); return; }
    collect_leaves(s, s.bvh_nodes[child].left, refs, budget);
    collect_leaves(s, s.bvh_nodes[child].right, refs, budget);
}
struct SahItem {
    int32_t ref;
    Bounds b;
    double c[3];
};
// Emits the SAH subtree over items[begin, end) in threaded order; returns its bounds.
template <typename R>
Bounds emit_sah(const mlrt_scene& s, std::vector<SahItem>& items, size_t begin, size_t end, Entries<R>& out, bool wrap_leaves) {
    const size_t n = end - begin;
    if (n == 1) {
        const mlrt_prim_ref& p = s.prim_refs[items[begin].ref];
        out.leaf(s, p, wrap_leaves && p.kind != MLRT_PRIM_BOX);
        return items[begin].b;
    }
    int best_axis = 0;
    size_t best_split = begin + n / 2;
    double best_cost = INFINITY;
    std::vector<double> right_area(n);
    for (int axis = 0; axis < 3; axis++) {
        std::sort(items.begin() + begin, items.begin() + end, [axis](const SahItem& a, const SahItem& b) { return a.c[axis] < b.c[axis]; });
        Bounds acc = empty_bounds();
        for (size_t i = n; i-- > 1;) { acc.grow(items[begin + i].b); right_area[i] = acc.half_area(); }
        acc = empty_bounds();
        for (size_t i = 1; i < n; i++) {
            acc.grow(items[begin + i - 1].b);
            const double cost = acc.half_area() * (double)i + right_area[i] * (double)(n - i);
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = begin + i; }
        }
    }
    std::sort(items.begin() + begin, items.begin() + end, [best_axis](const SahItem& a, const SahItem& b) { return a.c[best_axis] < b.c[best_axis]; });
    const int32_t me = out.push(DNode<R>{}, true);
    // Child order of the threaded walk: the child with the larger surface first (the likelier one to shrink the hit
    // distance early; measured against input order, smaller-first and nearer-to-the-eye-first in round 1).
    Bounds lb = empty_bounds(), rb = empty_bounds();
    for (size_t i = begin; i < best_split; i++) lb.grow(items[i].b);
    for (size_t i = best_split; i < end; i++) rb.grow(items[i].b);
    const bool swap_children = rb.half_area() > lb.half_area();
    if (swap_children) {
        std::rotate(items.begin() + begin, items.begin() + best_split, items.begin() + end);
        best_split = begin + (end - best_split);
    }
    Bounds b = emit_sah<R>(s, items, begin, best_split, out, wrap_leaves);
    b.grow(emit_sah<R>(s, items, best_split, end, out, wrap_leaves));
    out.nodes[(size_t)me] = out.box_entry(b, out.size());
    return b;
}
// World-space bounds of `b` under Translate(offset, Rotate(axis, angle, .)): the 8 rotated corners, then the offset.
Bounds transform_bounds(const mlrt_transform& t, const Bounds& b) {
    Bounds r = empty_bounds();
    for (int corner = 0; corner < 8; corner++) {
        double p[3];
        for (int k = 0; k < 3; k++) p[k] = (corner >> k) & 1 ? b.hi[k] : b.lo[k];
        if (t.has_rotate) {  // transforms.rs:116-124 rotate()
            const int a1 = t.axis, a2 = (a1 + 1) % 3, a0 = (a1 + 2) % 3;
            const double v0 = p[a0], v2 = p[a2];
            p[a0] = t.cos_theta * v0 + t.sin_theta * v2;
            p[a2] = -t.sin_theta * v0 + t.cos_theta * v2;
        }
        if (t.has_translate) for (int k = 0; k < 3; k++) p[k] += t.offset[k];
        for (int k = 0; k < 3; k++) { r.lo[k] = std::fmin(r.lo[k], p[k]); r.hi[k] = std::fmax(r.hi[k], p[k]); }
    }
    return r;
}
void pad_bounds(Bounds& b, double rel) {
    for (int k = 0; k < 3; k++) { const double pad = rel * (1.0 + std::fmax(std::fabs(b.lo[k]), std::fabs(b.hi[k]))); b.lo[k] -= pad; b.hi[k] += pad; }
}

template <typename T>
int32_t append(std::vector<unsigned char>& blob, const std::vector<T>& v) {
    while (blob.size() % 16) blob.push_back(0);
    const int32_t off = (int32_t)blob.size();
    const unsigned char* p = reinterpret_cast<const unsigned char*>(v.data());
    blob.insert(blob.end(), p, p + v.size() * sizeof(T));
    return off;
}

template <typename M>
void pack_materials(const mlrt_scene& s, bool solid_only, std::vector<DMaterial<M>>& out) {
    for (int i = 0; i < s.n_materials; i++) {
        const mlrt_material& m = s.materials[i];
        const bool textured = m.kind == MLRT_MAT_LAMBERTIAN || m.kind == MLRT_MAT_DIFFUSE_LIGHT || m.kind == MLRT_MAT_ISOTROPIC;
        const int32_t tex = textured ? m.texture : -1;  // (Metal / Dielectric never sample a texture, whatever the caller left in the field)
        DMaterial<M> d{};
        d.kind = m.kind | (texture_needs_uv(s, tex, 0) ? MATFLAG_NEEDS_UV : 0);
        if (tex >= 0 && s.textures[tex].kind == MLRT_TEX_NOISE) d.kind |= MATFLAG_NOISE;
        d.tex = tex;
        for (int k = 0; k < 3; k++) d.albedo[k] = (M)m.albedo[k];
        // a scene whose textures are all SolidColor (the LEAN kernel variant) carries the colour in the material and no
        // texture table: 32 bytes per texture less in LDS and no dependent texture load in the shading code
        if (solid_only && tex >= 0)
            for (int k = 0; k < 3; k++) d.albedo[k] = (M)s.textures[tex].color[k];
        d.param = (M)m.param;
        // Dielectric (attenuation 1: its albedo is never read): albedo[0] = 1.0 / ir, the refraction ratio of a front-face hit
        // (materials.rs:92) - the same IEEE division, done here once instead of in every shading batch
        if (m.kind == MLRT_MAT_DIELECTRIC) d.albedo[0] = (M)1 / (M)m.param;
        out.push_back(d);
    }
}

// `wraps`: bounding-box entries in front of sphere leaves (`wrap_leaves`) and in front of the top-level
// entries whose miss a box can prove (primitives, medium boundaries, whole transform brackets); `sah`: SAH device BVHs.
// None of it changes a closest hit or the conditions under which a medium draws (a ray that misses the boundary's box
// misses the boundary: volumes.rs:27-34 returns None before drawing).  The strict build packs the reference structure
// verbatim (wraps = sah = false, Round::EXACT).
template <typename R, typename M>
Packed<R, M> pack(const mlrt_scene& s, bool wraps, bool wrap_leaves, bool sah, bool solid_only, bool payload_all, Round round) {
    Packed<R, M> P;
    constexpr bool kCull32 = sizeof(R) == 4;
    P.entries.payload_all = payload_all; P.entries.round = round;
    for (int i = 0; i < s.n_objects; i++) {
        const mlrt_object& src = s.objects[i];
        const bool xf = src.transform >= 0 && (s.transforms[src.transform].has_translate || s.transforms[src.transform].has_rotate);
        // contents in object space, emitted into a scratch array first so a wrapper can be put in front
        Entries<R> body;
        body.payload_all = payload_all; body.round = round;
        Bounds bounds = empty_bounds();
        bool bounded = false;  // `bounds` encloses everything a ray can interact with inside this object
        if (src.kind == MLRT_OBJ_BVH) {
            const int32_t root = s.bvhs[src.bvh].root;
            if (root == MLRT_BVH_EMPTY) {
                DNode<R> e{};
                e.tag = TAG_EMPTY;
                body.push(e, false);
            } else if (sah) {
                std::vector<int32_t> refs;
                int budget = 2 * s.n_bvh_nodes + s.n_prim_refs + 2;
                collect_leaves(s, root, refs, budget);
                std::vector<SahItem> items(refs.size());
                for (size_t k = 0; k < refs.size(); k++) {
                    items[k].ref = refs[k];
                    prim_bounds(s, s.prim_refs[refs[k]], items[k].b);
                    for (int a = 0; a < 3; a++) items[k].c[a] = 0.5 * (items[k].b.lo[a] + items[k].b.hi[a]);
                }
                bounds = emit_sah<R>(s, items, 0, items.size(), body, wraps && wrap_leaves);
                bounded = true;
            } else {
                int budget = 2 * s.n_bvh_nodes + s.n_prim_refs + 2;
                emit_subtree<R>(s, root, body, budget, wraps && wrap_leaves);
                if (root >= 0) { for (int k = 0; k < 3; k++) { bounds.lo[k] = s.bvh_nodes[root].bounds_min[k]; bounds.hi[k] = s.bvh_nodes[root].bounds_max[k]; } bounded = true; }
                else bounded = prim_bounds(s, s.prim_refs[~root], bounds);
            }
        } else if (src.kind == MLRT_OBJ_PRIM) {
            body.leaf(s, src.prim, false);
            bounded = prim_bounds(s, src.prim, bounds);
        } else {  // ConstantMedium: the medium entry, then its boundary primitive
            DNode<R> e{};
            if (!payload_all) e.v[0] = (R)src.neg_inv_density;
            e.link = src.phase_material;
            e.tag = TAG_MEDIUM;
            body.push(e, false);
            if (payload_all) body.payload(&src.neg_inv_density, 1);
            body.leaf(s, src.prim, false);
            bounded = prim_bounds(s, src.prim, bounds);
        }
        // (a Block tests its own bounds in the step, and an untransformed BVH whose root is an inner entry already opens with the
        // box of everything in it and the same skip target: a wrapper would be the same test twice)
        const bool root_is_box = src.kind == MLRT_OBJ_BVH && body.size() > 1 && body.nodes[0].tag == TAG_INNER && body.reloc[0] && body.nodes[0].link == body.size();
        const bool wrap = wraps && bounded && !(src.kind == MLRT_OBJ_PRIM && src.prim.kind == MLRT_PRIM_BOX && !xf) && !(root_is_box && !xf);
        const int32_t xf_slots = xf ? entry_slots(TAG_XFORM, kCull32, payload_all) : 0;
        const int32_t after = P.entries.size() + (wrap ? 1 : 0) + xf_slots + body.size() + (xf ? 1 : 0);
        if (wrap) {
            Bounds wb = xf ? transform_bounds(s.transforms[src.transform], bounds) : bounds;
            pad_bounds(wb, 1e-6);
            P.entries.push(P.entries.box_entry(wb, after), false);  // world-space box in front of everything this object emits
        }
        if (xf) {  // Translate(offset, Rotate(axis, angle, .)) opens: the cursor's ray moves to object space
            const mlrt_transform& t = s.transforms[src.transform];
            const double v[5] = {t.has_rotate ? t.sin_theta : 0.0, t.has_rotate ? t.cos_theta : 1.0, t.has_translate ? t.offset[0] : 0.0,
                                 t.has_translate ? t.offset[1] : 0.0, t.has_translate ? t.offset[2] : 0.0};
            DNode<R> e{};
            if (!payload_all) for (int k = 0; k < 5; k++) e.v[k] = (R)v[k];
            e.tag = TAG_XFORM | ((t.has_translate ? XF_TRANSLATE : 0) << 4) | ((t.has_rotate ? XF_ROTATE : 0) << 4) | ((t.has_rotate ? t.axis : 0) << 8);
            P.entries.push(e, false);
            if (payload_all) P.entries.payload(v, 5);
        }
        P.entries.append(body);
        if (xf) {
            DNode<R> e{};
            e.tag = TAG_XFORM_END;
            P.entries.push(e, false);
        }
        MLRT_REQUIRE(P.entries.size() == after, "internal error: entry count of an object");
    }
    pack_materials<M>(s, solid_only, P.materials);
    for (int i = 0; i < s.n_textures && !solid_only; i++) {
        const mlrt_texture& t = s.textures[i];
        DTexture<M> d{};
        d.kind = t.kind; d.a = t.index; d.b = t.index2;
        for (int k = 0; k < 3; k++) d.color[k] = (M)t.color[k];
        d.scale = (M)t.scale;
        P.textures.push_back(d);
    }
    for (int i = 0; i < s.n_perlins; i++) {
        for (int k = 0; k < 3072; k++) P.perlin_vec.push_back((M)s.perlins[i].ranvec[k]);
        for (int k = 0; k < 3072; k++) {
            const int32_t v = s.perlins[i].perm[k];
            MLRT_REQUIRE(v >= 0 && v < 1024, "perlin permutation entry out of range");
            P.perlin_perm.push_back((uint16_t)v);
        }
    }
    {   // sentinel: skip links that leave the array point here (index = number of real entries)
        DNode<R> end{};
        end.tag = TAG_END;
        end.link = P.entries.size();
        P.entries.push(end, false);
    }
    // (the candidate pre-pass exists for sphere-only scenes, primary_lists_wanted(): no list otherwise - final_scene's blob just fits the LDS)
    if (solid_only && P.entries.size() <= 65536) for (int32_t e : P.entries.spheres) P.sphere_index.push_back((uint16_t)e);
    P.off_nodes = append(P.blob, P.entries.nodes);
    P.off_materials = append(P.blob, P.materials);
    P.off_textures = append(P.blob, P.textures);
    P.off_sphere_index = append(P.blob, P.sphere_index);
    while (P.blob.size() % 16) P.blob.push_back(0);
    P.off_perlin_vec = append(P.blob, P.perlin_vec);  // (the Perlin tables end the blob: the f64 builds stage everything before them in LDS)
    P.off_perlin_perm = append(P.blob, P.perlin_perm);
    while (P.blob.size() % 16) P.blob.push_back(0);
    return P;
}


// ---- the pair walk's structure (mlrt_device.hpp DPair): sphere-only scenes --------------------------------------------------
// Every sphere of the world - top-level primitives and the leaves of every BHV, numbered in the order the reference visits them
// (hittable.rs:57-66 list order, bhv.rs:152-163 left before right) - under ONE surface-area-heuristic binary tree whose nodes hold
// both children's boxes.  Legal because a closest hit over spheres does not depend on how they are grouped (the same argument as
// for the SAH threaded list above), with the tie rule of the sphere test (mlrt_kernel.hip) keeping equal distances the
// reference's way.  M = the primitives' and materials' type (float: fast build, double: product f64 build).
template <typename M>
struct PackedPairs {
    std::vector<DPair> pairs;
    std::vector<M> prims;  // [n_prims][4]: centre, radius
    std::vector<uint16_t> prim_mat;
    std::vector<DMaterial<M>> materials;
    std::vector<unsigned char> blob;
    int32_t off_pairs = 0, off_prims = 0, off_prim_mat = 0, off_materials = 0;
    int32_t n_prims = 0, stack_levels = 0;
};
struct PairItem {
    int32_t prim;
    Bounds b;
    double c[3];
};
constexpr int kPairLevelCap = 24;  // inner nodes on any root-to-leaf path (= rows of a lane's LDS stack): below it the builder splits by count
inline int ceil_log2(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }
// Builds the subtree over items[begin, end) (>= 1 item); returns its reference and bounds.  `level` = inner nodes above it.
template <typename M>
int32_t build_pairs(std::vector<PairItem>& items, size_t begin, size_t end, int level, Round round, PackedPairs<M>& out, Bounds& bounds) {
    const size_t n = end - begin;
    if (n == 1) { bounds = items[begin].b; return kPairLeaf + items[begin].prim; }
    size_t split = begin + n / 2;
    int axis = 0;
    if (kPairLevelCap - level <= ceil_log2(n)) {
        // out of levels for anything but a balanced tree: median split along the widest extent of the centroids
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = begin; i < end; i++) for (int a = 0; a < 3; a++) { lo[a] = std::fmin(lo[a], items[i].c[a]); hi[a] = std::fmax(hi[a], items[i].c[a]); }
        for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
    } else {
        double best_cost = INFINITY;
        std::vector<double> right_area(n);
        for (int a = 0; a < 3; a++) {
            std::stable_sort(items.begin() + begin, items.begin() + end, [a](const PairItem& x, const PairItem& y) { return x.c[a] < y.c[a]; });
            Bounds acc = empty_bounds();
            for (size_t i = n; i-- > 1;) { acc.grow(items[begin + i].b); right_area[i] = acc.half_area(); }
            acc = empty_bounds();
            for (size_t i = 1; i < n; i++) {
                acc.grow(items[begin + i - 1].b);
                const double cost = acc.half_area() * (double)i + right_area[i] * (double)(n - i);
                if (cost < best_cost) { best_cost = cost; axis = a; split = begin + i; }
            }
        }
    }
    std::stable_sort(items.begin() + begin, items.begin() + end, [axis](const PairItem& x, const PairItem& y) { return x.c[axis] < y.c[axis]; });
    const int32_t me = (int32_t)out.pairs.size();
    out.pairs.push_back(DPair{});
    out.stack_levels = std::max(out.stack_levels, level + 1);
    Bounds ba, bb;
    const int32_t ra = build_pairs<M>(items, begin, split, level + 1, round, out, ba);
    const int32_t rb = build_pairs<M>(items, split, end, level + 1, round, out, bb);
    DPair& p = out.pairs[(size_t)me];
    for (int k = 0; k < 3; k++) {
        p.a[k] = round_down<float>(ba.lo[k], round); p.a[3 + k] = round_up<float>(ba.hi[k], round);
        p.b[k] = round_down<float>(bb.lo[k], round); p.b[3 + k] = round_up<float>(bb.hi[k], round);
    }
    p.ref_a = ra; p.ref_b = rb;
    bounds = ba;
    bounds.grow(bb);
    return me;
}
// The spheres of a sphere-only world in the reference's visiting order (as indices into s.spheres).
void spheres_in_reference_order(const mlrt_scene& s, std::vector<int32_t>& order) {
    for (int i = 0; i < s.n_objects; i++) {
        const mlrt_object& src = s.objects[i];
        if (src.kind == MLRT_OBJ_PRIM) { order.push_back(src.prim.index); continue; }
        const int32_t root = s.bvhs[src.bvh].root;  // (MLRT_OBJ_BVH: a sphere-only world has no media)
        if (root == MLRT_BVH_EMPTY) continue;
        std::vector<int32_t> refs;
        int budget = 2 * s.n_bvh_nodes + s.n_prim_refs + 2;
        collect_leaves(s, root, refs, budget);
        for (int32_t r : refs) order.push_back(s.prim_refs[r].index);
    }
}
// false: the world is outside what the pair walk covers (no sphere at all, or too many for 16-bit references)
template <typename M>
bool pack_pairs(const mlrt_scene& s, Round round, PackedPairs<M>& P) {
    std::vector<int32_t> order;
    spheres_in_reference_order(s, order);
    if (order.empty() || order.size() > (size_t)kPairMaxPrims) return false;
    P.n_prims = (int32_t)order.size();
    std::vector<PairItem> items(order.size());
    for (size_t k = 0; k < order.size(); k++) {
        const mlrt_sphere& sp = s.spheres[order[k]];
        P.prims.push_back((M)sp.center[0]); P.prims.push_back((M)sp.center[1]); P.prims.push_back((M)sp.center[2]); P.prims.push_back((M)sp.radius);
        P.prim_mat.push_back((uint16_t)sp.material);
        items[k].prim = (int32_t)k;
        const mlrt_prim_ref ref{MLRT_PRIM_SPHERE, order[k]};
        prim_bounds(s, ref, items[k].b);
        for (int a = 0; a < 3; a++) items[k].c[a] = 0.5 * (items[k].b.lo[a] + items[k].b.hi[a]);
    }
    if (s.n_materials > 65535) return false;
    Bounds all;
    const int32_t root = build_pairs<M>(items, 0, items.size(), 0, round, P, all);
    if (root < 0) {
        // a single sphere: one node whose two children are that leaf (the second test finds the same distance from the same
        // primitive and changes nothing)
        DPair p{};
        for (int k = 0; k < 3; k++) { p.a[k] = p.b[k] = round_down<float>(all.lo[k], round); p.a[3 + k] = p.b[3 + k] = round_up<float>(all.hi[k], round); }
        p.ref_a = p.ref_b = root;
        P.pairs.push_back(p);
        P.stack_levels = 1;
    }
    if (P.pairs.size() > (size_t)kPairMaxNodes) return false;
    pack_materials<M>(s, true, P.materials);
    P.off_pairs = append(P.blob, P.pairs);
    P.off_prims = append(P.blob, P.prims);
    P.off_prim_mat = append(P.blob, P.prim_mat);
    while (P.blob.size() % 16) P.blob.push_back(0);
    P.off_materials = append(P.blob, P.materials);
    while (P.blob.size() % 16) P.blob.push_back(0);
    return true;
}
}  // namespace
}  // namespace mlrt

using namespace mlrt;

struct mlrt_context {
    int32_t device = 0, precision = MLRT_PRECISION_FP32;
    mlrt_options opt{};  // read once at creation: nothing below looks at the environment
    const KernelBuild* build = nullptr;
    int32_t n_cus = 0;
    unsigned char* d_blob = nullptr;
    int32_t blob_bytes = 0, lds_blob_bytes = 0;
    int32_t off[6] = {0, 0, 0, 0, 0, 0};
    int32_t n_nodes = 0, n_sphere_index = 0;
    bool pair_walk = false;  // the blob is the pair structure (mlrt_device.hpp DPair), variant == 2
    bool split_roles = false;  // ... rendered by the split-role kernel (mlrt_split.inc)
    int32_t off_prims = 0, off_prim_mat = 0, n_prims = 0, stack_levels = 0;
    std::vector<unsigned char*> d_image_pixels;
    DImage* d_images = nullptr;
    unsigned int* d_tile_counter = nullptr;
    unsigned int* d_status = nullptr;  // watchdog record of the render kernel (KParams::status)
    uint16_t* d_cand = nullptr;  // primary candidate records, [cand_groups][16]
    size_t cand_groups = 0;
    unsigned long long* d_stats = nullptr;
    unsigned long long* d_acc = nullptr;  // frame accumulation buffer, grown on demand
    size_t acc_elems = 0;
    int32_t variant = 0;  // kernel variant (0 FULL, 1 LEAN, 2 LEAN over the pair structure)
    uint32_t features = 0;
    bool lds_scene = false;
    size_t dyn_lds = 0;
    int32_t block_threads = 0, blocks_per_cu = 1, last_grid = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    // mlrt_context_render's device-side row list and frame, kept between calls (a movie renders one frame per call)
    int32_t* d_host_rows = nullptr;
    int32_t* d_host_rgb = nullptr;
    size_t host_rows_cap = 0, host_rgb_cap = 0;
    const double* d_ray_table = nullptr;  // set for the duration of a ray frame (mlrt_debug_ray_frame)
};

namespace {

void destroy(mlrt_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_blob) (void)hipFree(c->d_blob);
    for (unsigned char* p : c->d_image_pixels) (void)hipFree(p);
    if (c->d_images) (void)hipFree(c->d_images);
    if (c->d_tile_counter) (void)hipFree(c->d_tile_counter);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_cand) (void)hipFree(c->d_cand);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_acc) (void)hipFree(c->d_acc);
    if (c->d_host_rows) (void)hipFree(c->d_host_rows);
    if (c->d_host_rgb) (void)hipFree(c->d_host_rgb);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    delete c;
}

// The frame's tile queue (KParams), planned from the frame size and the number of resident waves alone: a plain function so
// that the CPU tests can check it (mlrt_debug_tile_plan): every sample of every pixel group in exactly one tile.
struct TilePlan {
    int32_t pix_per_tile, tiles_per_row, n_groups, n_big, big, n_tail;
    int32_t tail_s0[mlrt::kMaxTailPasses + 1];
    int64_t n_tiles;
    uint32_t n_groups_magic, tiles_per_row_magic;
    int32_t n_groups_shift, tiles_per_row_shift;
};
void plan_tiles(int width, int n_rows, int spp, double waves, bool lean, const mlrt_options& opt, TilePlan& T) {
    // Tile geometry (see KParams).  Big chunks of ~MLRT_TILE_SAMPLES samples (default 512): enough to keep 64 lanes busy
    // through a tile's ramp-up, with per-tile work (queue atomics, the candidate record) small against it.  The last
    // samples of every pixel go out in tapering passes (sizes from the END of the sample range; MLRT_TILE_TAIL overrides):
    // the integer accumulation makes the split invisible in the image.  How deep the taper goes follows the number of
    // big tiles per wave - the fewer, the larger the share of the frame time one late heavy tile is (measured, kernel ms
    // for 1/1, 1/2, 1/4, 1/8 of C2's rows: no taper 47.6 25.0 14.1 8.6; "128" 48.4 24.7 13.1 7.8; "64,128" 49.5 25.2 13.1
    // 7.5; "64,64,128" 50.6 25.7 13.4 7.1) - and a small tile costs the FULL variant's expensive samples relatively less.
    // Low spp: several pixels per tile and a single pass.
    const int target = std::max(64, opt.tile_samples);
    int pix = 1;
    T.n_big = 1; T.big = spp; T.n_tail = 0;
    for (int k = 0; k <= mlrt::kMaxTailPasses; k++) T.tail_s0[k] = spp;
    if (spp >= 128) {
        std::vector<int> tail;  // from the end
        if (opt.n_tail >= 0) {
            for (int k = 0; k < opt.n_tail && k < mlrt::kMaxTailPasses; k++) if (opt.tail[k] > 0) tail.push_back(opt.tail[k]);
        } else {
            const double big_tiles_per_wave = (double)width * n_rows * ((spp + target - 1) / target) / waves / (lean ? 1.0 : 2.0);
            static const int kTaper[5] = {64, 64, 64, 64, 128};
            const int depth = big_tiles_per_wave >= 128 ? 0 : big_tiles_per_wave >= 64 ? 1 : big_tiles_per_wave >= 32 ? 2 : big_tiles_per_wave >= 16 ? 3 : 5;
            for (int k = 0; k < depth; k++) tail.push_back(kTaper[5 - depth + k]);  // "128" | "64,128" | "64,64,128" | "64,64,64,64,128"
        }
        int tail_total = 0, n_tail = 0;
        while (n_tail < (int)tail.size() && tail_total + tail[n_tail] <= spp - 64) tail_total += tail[n_tail++];  // the big passes keep >= 64 samples
        const int rest = spp - tail_total;
        T.n_big = (rest + target - 1) / target;
        T.big = rest / T.n_big;  // the remainder (< n_big samples) joins the first tail pass, or is one more small pass
        // queue order: the big passes, then tail[n_tail-1] ... tail[0]; the first pass after the big ones starts where they end
        T.n_tail = n_tail;
        if (n_tail == 0 && T.n_big * T.big < spp) T.n_tail = 1;
        T.tail_s0[0] = T.n_big * T.big;
        for (int k = 1, end = tail_total; k <= n_tail; k++) {
            end -= tail[n_tail - k];
            T.tail_s0[k] = spp - end;
        }
    } else {
        pix = std::min(4, std::max(1, (target + spp / 2) / spp));  // nearest whole number of pixels (kMaxPixPerTile)
        pix = std::min(pix, width);
    }
    T.pix_per_tile = pix;
    T.tiles_per_row = (width + pix - 1) / pix;
    T.n_groups = T.tiles_per_row * n_rows;
    auto magic = [](int32_t d, uint32_t& m, int32_t& shift) {  // floor(n / d) = mulhi(n, m) >> shift for 0 <= n < 2^31 (tests/test_host_scene.py checks the rule)
        if (d <= 1) { m = 0u; shift = -1; return; }
        int L = 0;
        while (((int64_t)1 << L) < d) L++;
        m = (uint32_t)((((uint64_t)1 << (31 + L)) / (uint64_t)d) + 1u);
        shift = L - 1;
    };
    magic(T.n_groups, T.n_groups_magic, T.n_groups_shift);
    magic(T.tiles_per_row, T.tiles_per_row_magic, T.tiles_per_row_shift);
    const int64_t n_tiles64 = (int64_t)T.n_groups * (T.n_big + T.n_tail);
    T.n_tiles = n_tiles64;
}

// Per-pixel-group primary candidate lists (mlrt_kernel.hip): the sphere-only variant, few enough entries that the pre-pass
// stays small against the frame, and max_depth small enough for the list cursor the kernel keeps in the upper bits of a
// lane's depth word (max_depth is any i32 in the reference: above 65535 the lanes simply walk the BVH).
bool primary_lists_wanted(const mlrt_context* c, const mlrt_params* p) {
    return (c->variant == 1 || c->variant == 2) && !c->d_ray_table && c->precision != MLRT_PRECISION_FP64_STRICT && c->opt.primary_lists != 0 && (c->pair_walk ? c->n_prims : c->n_nodes) <= 4096 &&
           p->image_height < 65536 && p->max_depth <= 0xFFFF;
}

template <typename R>
void fill_params(const mlrt_context* c, const mlrt_camera* cam, const mlrt_params* p, const int32_t* d_rows, int32_t n_rows, int32_t* d_rgb,
                 double* d_accum, KParams<R>& K) {
    K.blob = c->d_blob; K.blob_bytes = c->blob_bytes; K.lds_blob_bytes = c->lds_blob_bytes;
    K.off_nodes = c->off[0]; K.off_materials = c->off[1]; K.off_textures = c->off[2];
    K.off_perlin_vec = c->off[3]; K.off_perlin_perm = c->off[4];
    K.off_sphere_index = c->off[5]; K.n_sphere_index = c->n_sphere_index;
    K.n_nodes = c->n_nodes;
    K.pair_walk = c->pair_walk ? (c->split_roles ? 2 : 1) : 0; K.off_prims = c->off_prims; K.off_prim_mat = c->off_prim_mat; K.n_prims = c->n_prims; K.stack_levels = c->stack_levels;
    K.images = c->d_images;
    for (int k = 0; k < 3; k++) {
        K.cam_origin[k] = cam->origin[k]; K.cam_llc[k] = cam->lower_left_corner[k];
        K.cam_horizontal[k] = cam->horizontal[k]; K.cam_vertical[k] = cam->vertical[k];
        K.cam_u[k] = cam->u[k]; K.cam_v[k] = cam->v[k];
        K.bg_top[k] = (R)p->background_top[k]; K.bg_bottom[k] = (R)p->background_bottom[k];
    }
    K.lens_radius = cam->lens_radius;
    const bool single_light = p->tracer == MLRT_TRACER_SINGLE_LIGHT;
    K.has_lens = (cam->lens_radius != 0.0 ? 1 : 0) | (c->d_ray_table ? 2 : 0) | (single_light ? 4 : 0);
    K.ray_table = c->d_ray_table;
    for (int k = 0; k < 3; k++) K.single_light[k] = p->light_source[k];
    K.single_light[3] = p->light_intensity;
    K.inv_wm1 = 1.0 / ((double)p->image_width - 1.0); K.inv_hm1 = 1.0 / ((double)p->image_height - 1.0);
    K.width = p->image_width; K.height = p->image_height; K.spp = p->samples_per_pixel; K.max_depth = single_light ? 1 : p->max_depth;  // (SingleLightSourceRayTracer has no depth: one bounce)
    K.seed_lo = (uint32_t)p->seed; K.seed_hi = (uint32_t)(p->seed >> 32);
    K.bg_kind = p->background_kind;
    K.rows = d_rows; K.n_rows = n_rows;
    TilePlan T;
    plan_tiles(p->image_width, n_rows, p->samples_per_pixel, (double)c->n_cus * c->blocks_per_cu * (c->block_threads / 64), c->variant == 1 || c->variant == 2, c->opt, T);
    MLRT_REQUIRE(T.n_tiles < ((int64_t)1 << 31), "too many tiles for the 32-bit tile counter");
    K.pix_per_tile = T.pix_per_tile; K.tiles_per_row = T.tiles_per_row; K.n_groups = T.n_groups; K.n_big = T.n_big; K.big = T.big; K.n_tail = T.n_tail;
    for (int k = 0; k <= mlrt::kMaxTailPasses; k++) K.tail_s0[k] = T.tail_s0[k];
    K.n_tiles = (int32_t)T.n_tiles;
    K.n_groups_magic = T.n_groups_magic; K.n_groups_shift = T.n_groups_shift;
    K.tiles_per_row_magic = T.tiles_per_row_magic; K.tiles_per_row_shift = T.tiles_per_row_shift;
    const int pix = T.pix_per_tile;
    {   // cut the wave's accumulator area (mlrt_kernel.hip kAccBytesPerWave) into banks: sums + remaining/out/values + candidate list
        const bool lean = c->variant == 1 || c->variant == 2;
        const int area = c->build->acc_bytes_per_wave(c->variant);
        K.bank_values = 3 * pix;
        K.n_banks = std::max(1, std::min(16, (area - 16 * (12 + (lean ? 32 : 0))) / (K.bank_values * 8)));  // kMaxBanks x (3 ints + candidate record: sphere-only variants)
    }
    K.acc_global = c->d_acc;
    K.rgb_out = d_rgb; K.accum_out = d_accum;
    K.tile_counter = c->d_tile_counter;
    K.status = c->d_status;
    // 2^24 iterations (seconds) without a finished path; a path may legitimately bounce max_depth times (a handful of iterations
    // per bounce), so absurd depths switch the watchdog off rather than risk a false alarm
    K.stall_cap = p->max_depth > (1 << 20) ? 0xFFFFFFFFu : (1u << 24);
    if (c->opt.stall_cap > 0) K.stall_cap = (uint32_t)c->opt.stall_cap;
    K.n_queues = c->opt.tile_queues;
    K.stats = c->d_stats;
    K.primary_lists = primary_lists_wanted(c, p) ? c->d_cand : nullptr;
    K.debug_pixel = c->opt.debug_pixel; K.debug_sample = c->opt.debug_sample;
}

void check_params(const mlrt_camera* cam, const mlrt_params* p, int32_t n_rows) {
    MLRT_REQUIRE(cam && p, "camera/params is NULL");
    MLRT_REQUIRE(p->image_width >= 2 && p->image_height >= 2, "image must be at least 2x2 (u = (i+x)/(W-1), raytrace.rs:191-192)");
    MLRT_REQUIRE(p->samples_per_pixel >= 1 && p->samples_per_pixel <= (1 << 24), "samples_per_pixel out of range");
    MLRT_REQUIRE(p->max_depth >= 0, "max_depth must be >= 0");
    MLRT_REQUIRE((int64_t)p->image_width * p->image_height < (int64_t)1 << 31, "image too large for the 32-bit pixel counter of the Philox stream");
    MLRT_REQUIRE(n_rows >= 0 && n_rows <= p->image_height, "n_rows out of range");
    MLRT_REQUIRE((int64_t)n_rows * p->image_width * 3 < (int64_t)1 << 31, "frame too large: n_rows * width * 3 must stay below 2^31 (32-bit indices of the accumulation buffer)");
    MLRT_REQUIRE(p->background_kind == MLRT_BG_BLACK || p->background_kind == MLRT_BG_GRADIENT, "unknown background kind");
    MLRT_REQUIRE(p->tracer == MLRT_TRACER_RECURSIVE || p->tracer == MLRT_TRACER_SINGLE_LIGHT, "unknown tracer");
    if (p->tracer == MLRT_TRACER_SINGLE_LIGHT)
        for (int k = 0; k < 4; k++) MLRT_REQUIRE(std::fabs(k < 3 ? p->light_source[k] : p->light_intensity) < 0x1p100, "light_source / light_intensity is not finite (or is 2^100 or more)");
    // (a camera of NaNs renders NaNs in the reference too; refused here like non-finite geometry, with the same bound on magnitudes)
    const double* c = cam->origin;
    static_assert(sizeof(mlrt_camera) == 19 * sizeof(double), "mlrt_camera is 19 doubles");
    for (int k = 0; k < 19; k++) MLRT_REQUIRE(std::fabs(c[k]) < 0x1p100, "camera field is not finite (or is 2^100 or more)");
}

// One frame: candidate pre-pass (sphere-only scenes), render kernel, resolve kernel - nothing else.  The accumulation
// buffer and the tile counters are zero on entry: they are zeroed when allocated, and the resolve kernel leaves them
// zeroed again after reading them (before: two hipMemsetAsync per frame, 23 MB for C2, which do not shrink with 1/N of
// the rows per GPU).  A launch that fails leaves them undefined: the context must then be destroyed.
void launch(mlrt_context* c, const mlrt_camera* cam, const mlrt_params* p, const int32_t* d_rows, int32_t n_rows, int32_t* d_rgb, double* d_accum,
            hipStream_t stream) {
    if (n_rows == 0) return;
    if (p->tracer == MLRT_TRACER_SINGLE_LIGHT && c->variant != 0)
        throw Fail{MLRT_ERR_UNSUPPORTED, "MLRT_TRACER_SINGLE_LIGHT needs a context created with mlrt_options.single_light_tracer = 1 (the all-features kernel variant carries that tracer)"};
    MLRT_HIP(hipSetDevice(c->device));
    const size_t need = (size_t)n_rows * (size_t)p->image_width * 3;
    if (need > c->acc_elems) {
        // grows only when a larger frame is requested (not in the steady state of a render loop)
        MLRT_HIP(hipStreamSynchronize(stream));
        if (c->d_acc) MLRT_HIP(hipFree(c->d_acc));
        c->d_acc = nullptr; c->acc_elems = 0;
        MLRT_HIP(hipMalloc(&c->d_acc, need * sizeof(unsigned long long)));
        MLRT_HIP(hipMemset(c->d_acc, 0, need * sizeof(unsigned long long)));
        c->acc_elems = need;
    }
    if (primary_lists_wanted(c, p)) {  // one 32-byte record per pixel group; at most one group per pixel
        const size_t groups = (size_t)n_rows * (size_t)p->image_width;
        if (groups > c->cand_groups) {
            MLRT_HIP(hipStreamSynchronize(stream));
            if (c->d_cand) MLRT_HIP(hipFree(c->d_cand));
            c->d_cand = nullptr; c->cand_groups = 0;
            MLRT_HIP(hipMalloc(&c->d_cand, groups * 32));
            c->cand_groups = groups;
        }
    }
    const int waves_per_block = c->block_threads / 64;
    int rc;
    int grid;
    MLRT_HIP(hipEventRecord(c->ev_start, stream));
    if (c->precision == MLRT_PRECISION_FP32) {
        KParams<float> K{};
        fill_params<float>(c, cam, p, d_rows, n_rows, d_rgb, d_accum, K);
        grid = std::max(1, std::min(c->n_cus * c->blocks_per_cu, (K.n_tiles + waves_per_block - 1) / waves_per_block));
        rc = c->build->launch(&K, c->variant, c->lds_scene ? 1 : 0, grid, c->dyn_lds, stream);
    } else {
        KParams<double> K{};
        fill_params<double>(c, cam, p, d_rows, n_rows, d_rgb, d_accum, K);
        grid = std::max(1, std::min(c->n_cus * c->blocks_per_cu, (K.n_tiles + waves_per_block - 1) / waves_per_block));
        rc = c->build->launch(&K, c->variant, c->lds_scene ? 1 : 0, grid, c->dyn_lds, stream);
    }
    if (rc != 0) throw Fail{MLRT_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc)};
    MLRT_HIP(hipEventRecord(c->ev_stop, stream));
    c->last_grid = grid;
    c->timed = true;
}

// Runs an entry point's body: no exception crosses the ABI, and the caller's current HIP device is what it was on every way out (the
// body switches to the context's device; an embedding application or a torch process does not expect another device left current).
template <typename F>
int32_t guarded(F&& f) {
    int saved = -1;
    if (hipGetDevice(&saved) != hipSuccess) { saved = -1; (void)hipGetLastError(); }
    int32_t rc = MLRT_OK;
    try {
        f();
    } catch (const Fail& e) {
        mlrt::set_last_error(e.msg);
        rc = e.code;
    } catch (const std::exception& e) {
        mlrt::set_last_error(std::string("internal error: ") + e.what());
        rc = MLRT_ERR_INVALID;
    }
    if (saved >= 0) (void)hipSetDevice(saved);
    return rc;
}

}  // namespace

extern "C" {

// Test hook (not part of include/mlrt.h): the tile plan of a frame.  out = pix_per_tile, tiles_per_row, n_groups, n_big, big, n_tail,
// tail_s0[0..8], n_tiles, n_groups_magic, n_groups_shift, tiles_per_row_magic, tiles_per_row_shift  (20 values)
void mlrt_debug_tile_plan(int32_t width, int32_t n_rows, int32_t spp, int32_t waves, int32_t lean, int64_t* out) {
    TilePlan T;
    mlrt_options o;
    mlrt_options_from_env(&o);
    plan_tiles(width, n_rows, spp, (double)waves, lean != 0, o, T);
    int k = 0;
    out[k++] = T.pix_per_tile; out[k++] = T.tiles_per_row; out[k++] = T.n_groups; out[k++] = T.n_big; out[k++] = T.big; out[k++] = T.n_tail;
    for (int i = 0; i <= mlrt::kMaxTailPasses; i++) out[k++] = T.tail_s0[i];
    out[k++] = T.n_tiles; out[k++] = T.n_groups_magic; out[k++] = T.n_groups_shift; out[k++] = T.tiles_per_row_magic; out[k++] = T.tiles_per_row_shift;
}
// Test hook (not part of include/mlrt.h): the pair structure the packer builds for a sphere-only scene (mlrt_device.hpp DPair), as
// the blob a context would upload.  precision: MLRT_PRECISION_FP32 (f32 primitives) or MLRT_PRECISION_FP64 (f64 primitives, rigorous
// outward rounding).  info = blob bytes, off_pairs, n_pairs, off_prims, off_prim_mat, n_prims, stack_levels, off_materials.
// Returns MLRT_ERR_UNSUPPORTED when the scene is outside what the pair walk covers.  Free *blob_out with mlrt_free.  No device call.
int32_t mlrt_debug_pack_pairs(const mlrt_scene* scene, int32_t precision, unsigned char** blob_out, int64_t info[8]) {
    return guarded([&] {
        MLRT_REQUIRE(scene && blob_out && info, "NULL argument");
        validate(*scene);
        if (scene_features(*scene) != 0) throw Fail{MLRT_ERR_UNSUPPORTED, "the pair walk serves sphere-only scenes with solid-colour materials"};
        auto give = [&](auto& Q) {
            *blob_out = static_cast<unsigned char*>(std::malloc(Q.blob.size()));
            MLRT_REQUIRE(*blob_out, "out of memory");
            std::memcpy(*blob_out, Q.blob.data(), Q.blob.size());
            const int64_t v[8] = {(int64_t)Q.blob.size(), Q.off_pairs, (int64_t)Q.pairs.size(), Q.off_prims, Q.off_prim_mat, Q.n_prims, Q.stack_levels, Q.off_materials};
            std::memcpy(info, v, sizeof v);
        };
        if (precision == MLRT_PRECISION_FP32) {
            PackedPairs<float> Q;
            if (!pack_pairs<float>(*scene, Round::OUTWARD, Q)) throw Fail{MLRT_ERR_UNSUPPORTED, "scene outside the pair walk's limits"};
            give(Q);
        } else {
            PackedPairs<double> Q;
            if (!pack_pairs<double>(*scene, Round::RIGOROUS, Q)) throw Fail{MLRT_ERR_UNSUPPORTED, "scene outside the pair walk's limits"};
            give(Q);
        }
    });
}
void mlrt_version(int32_t* major, int32_t* minor) {
    if (major) *major = MLRT_VERSION_MAJOR;
    if (minor) *minor = MLRT_VERSION_MINOR;
}
const char* mlrt_last_error(void) { return mlrt::last_error_cstr(); }
int32_t mlrt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t mlrt_device_warm_up(int32_t device) {
    return guarded([&] {
        MLRT_REQUIRE(device >= 0 && device < mlrt_device_count(), "device ordinal out of range");
        MLRT_HIP(hipSetDevice(device));
        void* p = nullptr;
        MLRT_HIP(hipMalloc(&p, 256));
        const hipError_t e = hipMemset(p, 0, 256);  // (the first dispatch of the process: queues, the runtime's own kernels)
        const hipError_t s = hipDeviceSynchronize();
        (void)hipFree(p);
        MLRT_HIP(e);
        MLRT_HIP(s);
    });
}

void mlrt_options_default(mlrt_options* o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->sah = 1; o->leaf_wrap = 1; o->primary_lists = 1; o->lds_scene = 1; o->force_full = 0; o->cull_f64 = 0; o->pair_walk = 1; o->split_roles = 0; o->single_light_tracer = 0;
    o->tile_samples = 512; o->tile_queues = 8; o->n_tail = -1;
    o->stats = 0; o->debug_pixel = -1; o->debug_sample = 0; o->stall_cap = 0;
}
// The MLRT_* environment variables (A/B scripts under tools/): read here and nowhere else, once per context.
void mlrt_options_from_env(mlrt_options* o) {
    if (!o) return;
    mlrt_options_default(o);
    if (std::getenv("MLRT_NO_SAH")) o->sah = 0;
    if (std::getenv("MLRT_NO_LEAF_WRAP")) o->leaf_wrap = 0;
    if (std::getenv("MLRT_NO_PRIMARY_LISTS")) o->primary_lists = 0;
    if (std::getenv("MLRT_NO_LDS")) o->lds_scene = 0;
    if (std::getenv("MLRT_FORCE_FULL")) o->force_full = 1;
    if (std::getenv("MLRT_CULL_F64")) o->cull_f64 = 1;
    if (std::getenv("MLRT_NO_PAIR_WALK")) o->pair_walk = 0;
    if (const char* sr = std::getenv("MLRT_SPLIT_ROLES")) o->split_roles = std::atoi(sr);
    if (const char* ts = std::getenv("MLRT_TILE_SAMPLES")) o->tile_samples = std::atoi(ts);
    if (const char* nq = std::getenv("MLRT_TILE_QUEUES")) o->tile_queues = std::atoi(nq);
    if (const char* spec = std::getenv("MLRT_TILE_TAIL")) {
        o->n_tail = 0;
        for (const char* q = spec; *q && o->n_tail < 8;) {
            const int v = std::atoi(q);
            if (v > 0) o->tail[o->n_tail++] = v;
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
    }
    if (std::getenv("MLRT_STATS")) o->stats = 1;
    if (const char* sc = std::getenv("MLRT_STALL_CAP")) o->stall_cap = std::atoi(sc);
    if (const char* dbg = std::getenv("MLRT_DEBUG_PIXEL")) { int a = -1, b = 0; if (std::sscanf(dbg, "%d,%d", &a, &b) >= 1) { o->debug_pixel = a; o->debug_sample = b; } }
}

int32_t mlrt_context_create(const mlrt_scene* scene, int32_t device, int32_t precision, mlrt_context** out) {
    mlrt_options o;
    mlrt_options_from_env(&o);
    return mlrt_context_create_opts(scene, device, precision, &o, out);
}

int32_t mlrt_context_create_opts(const mlrt_scene* scene, int32_t device, int32_t precision, const mlrt_options* options, mlrt_context** out) {
    mlrt_context* c = nullptr;
    const int32_t rc = guarded([&] {
        MLRT_REQUIRE(scene && out && options, "scene/options/out is NULL");
        MLRT_REQUIRE(precision == MLRT_PRECISION_FP32 || precision == MLRT_PRECISION_FP64 || precision == MLRT_PRECISION_FP64_STRICT, "unknown precision");
        *out = nullptr;
        validate(*scene);
        MLRT_REQUIRE(options->tile_samples >= 64 && options->tile_samples <= (1 << 20), "options: tile_samples out of range");
        MLRT_REQUIRE(options->tile_queues >= 1 && options->tile_queues <= mlrt::kMaxQueues && (options->tile_queues & (options->tile_queues - 1)) == 0,
                     "options: tile_queues must be a power of two <= 64");
        MLRT_REQUIRE(options->n_tail >= -1 && options->n_tail <= 8, "options: n_tail out of range");
        const int n_dev = mlrt_device_count();
        if (n_dev <= 0) throw Fail{MLRT_ERR_NO_DEVICE, "no HIP device is visible: libmlrt_hip has no CPU fallback"};
        MLRT_REQUIRE(device >= 0 && device < n_dev, "device ordinal out of range");
        c = new mlrt_context();
        c->device = device; c->precision = precision; c->opt = *options;
        c->build = &kBuilds[(precision == MLRT_PRECISION_FP64 && options->cull_f64) ? 3 : precision];
        const bool strict = precision == MLRT_PRECISION_FP64_STRICT;
        MLRT_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        MLRT_HIP(hipGetDeviceProperties(&prop, device));
        c->n_cus = prop.multiProcessorCount;
        // the leanest compiled kernel variant that covers the scene's features
        c->features = scene_features(*scene);
        const uint32_t lean = c->build->variant_features(1);
        const bool all_features = options->force_full != 0 || options->single_light_tracer != 0;
        c->variant = ((c->features & ~lean) == 0 && !all_features) ? 1 : 0;
        if (c->variant == 0 && !all_features) {
            // the FULL kernel compiled for the smallest superset of the scene's features (variant 0 has them all)
            auto bits = [](uint32_t f) { int n = 0; for (; f; f &= f - 1) n++; return n; };
            for (int v = 3; v < c->build->variant_count(); v++) {
                const uint32_t f = c->build->variant_features(v);
                if ((c->features & ~f) == 0 && bits(f) < bits(c->build->variant_features(c->variant))) c->variant = v;
            }
        }
        std::vector<unsigned char> blob;
        // a kernel variant without texture features reads no texture table (its scenes have SolidColor textures only: the colour rides in the material)
        const bool solid_only = (c->build->variant_features(c->variant) & (FEAT_TEX_IMAGE | FEAT_TEX_NOISE | FEAT_TEX_CHECKER)) == 0;
        c->block_threads = c->build->block_threads(c->variant);
        auto take = [&](auto& P) {
            blob.swap(P.blob);
            const int32_t o[6] = {P.off_nodes, P.off_materials, P.off_textures, P.off_perlin_vec, P.off_perlin_perm, P.off_sphere_index};
            std::memcpy(c->off, o, sizeof o);
            c->n_nodes = P.entries.size() - 1;  // without the sentinel
            c->n_sphere_index = (int32_t)P.sphere_index.size();
        };
        const bool wraps = !strict && options->leaf_wrap != 0, sah = !strict && options->sah != 0;
        // what of a blob of `bytes` (Perlin tables at `off_perlin`) must fit the LDS next to the waves' accumulators
        const size_t lds_limit = 160 * 1024;
        const size_t lds_other = (size_t)(c->block_threads / 64) * (size_t)c->build->acc_bytes_per_wave(c->variant) + (size_t)c->build->lds_frame_bytes();
        auto fits_lds = [&](const auto& P) { return (size_t)(c->build->perlin_in_hbm() ? P.off_perlin_vec : (int32_t)P.blob.size()) + lds_other <= lds_limit; };
        // Sphere-only scenes: the pair structure (mlrt_device.hpp DPair) when the build has the walk, the spheres fit its 16-bit
        // references and the lanes' stacks fit the LDS; otherwise the threaded list below.
        size_t pair_stack_bytes = 0;
        auto take_pairs = [&](auto& Q) {
            size_t stack = (size_t)(Q.stack_levels + 1) * (size_t)c->block_threads * sizeof(int16_t);  // (mlrt_kernel.hip: one guard row below the stack)
            // the split-role kernel: stacks for its traversal waves only, plus the control block and the two queues; used when that and
            // the scene fit the LDS (its queues are what it is made of: no HBM form)
            const size_t split_bytes = options->split_roles != 0 ? c->build->split_lds_bytes(Q.stack_levels) : 0;
            const size_t lds_scene_part = c->build->lean_materials_in_hbm() ? (size_t)Q.off_materials : Q.blob.size();
            if (split_bytes > 0 && lds_scene_part + lds_other + split_bytes <= lds_limit && options->lds_scene != 0) { stack = split_bytes; c->split_roles = true; }
            if (stack + lds_other + 1024 > lds_limit) return;  // (even with the scene in HBM the stacks must fit)
            blob.swap(Q.blob);
            const int32_t o[6] = {Q.off_pairs, Q.off_materials, (int32_t)blob.size(), (int32_t)blob.size(), (int32_t)blob.size(), (int32_t)blob.size()};
            std::memcpy(c->off, o, sizeof o);
            c->n_nodes = (int32_t)Q.pairs.size();
            c->n_sphere_index = 0;
            c->pair_walk = true; c->variant = 2;
            c->off_prims = Q.off_prims; c->off_prim_mat = Q.off_prim_mat; c->n_prims = Q.n_prims; c->stack_levels = Q.stack_levels;
            pair_stack_bytes = stack;
        };
        if (c->variant == 1 && options->pair_walk != 0 && !strict && !(precision == MLRT_PRECISION_FP64 && options->cull_f64) && c->build->has_pair_walk()) {
            if (precision == MLRT_PRECISION_FP32) {
                PackedPairs<float> Q;
                if (pack_pairs<float>(*scene, Round::OUTWARD, Q)) take_pairs(Q);
            } else {
                PackedPairs<double> Q;
                if (pack_pairs<double>(*scene, Round::RIGOROUS, Q)) take_pairs(Q);
            }
        }
        if (c->pair_walk) {
            // (packed above)
        } else if (precision == MLRT_PRECISION_FP32) {
            auto P = pack<float, float>(*scene, wraps, true, sah, solid_only, false, Round::OUTWARD);
            take(P);
        } else if (precision == MLRT_PRECISION_FP64 && !options->cull_f64) {
            // product f64: f32 culling entries (rigorously outward), every primitive's f64 parameters in payload slots
            auto P = pack<float, double>(*scene, wraps, true, sah, solid_only, true, Round::RIGOROUS);
            if (wraps && options->lds_scene != 0 && !fits_lds(P)) {
                // The boxes in front of the sphere leaves are a third of a sphere-heavy scene's entries (final_scene: 1000 of
                // 5638 slots): without them its f64 blob fits the LDS.  A walk over LDS-resident entries with twice the
                // sphere tests beats a walk through L2 with the cheap culls (C4 f64, measured) - and neither changes a colour sum.
                auto Q = pack<float, double>(*scene, wraps, false, sah, solid_only, true, Round::RIGOROUS);
                if (fits_lds(Q)) take(Q); else take(P);
            } else {
                take(P);
            }
        } else {
            // cull_f64: the same culling structure over f64 entries; strict: the reference's structure verbatim
            auto P = pack<double, double>(*scene, wraps, true, sah, solid_only, false, strict ? Round::EXACT : Round::OUTWARD);
            take(P);
        }
        c->blob_bytes = (int32_t)blob.size();
        MLRT_HIP(hipMalloc(&c->d_blob, blob.size()));
        MLRT_HIP(hipMemcpy(c->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
        std::vector<DImage> images;
        for (int i = 0; i < scene->n_images; i++) {
            const size_t bytes = (size_t)scene->images[i].width * scene->images[i].height * 3;
            unsigned char* d = nullptr;
            MLRT_HIP(hipMalloc(&d, bytes));
            c->d_image_pixels.push_back(d);
            MLRT_HIP(hipMemcpy(d, scene->images[i].rgb, bytes, hipMemcpyHostToDevice));
            images.push_back(DImage{d, scene->images[i].width, scene->images[i].height});
        }
        if (!images.empty()) {
            MLRT_HIP(hipMalloc(&c->d_images, images.size() * sizeof(DImage)));
            MLRT_HIP(hipMemcpy(c->d_images, images.data(), images.size() * sizeof(DImage), hipMemcpyHostToDevice));
        }
        MLRT_HIP(hipMalloc(&c->d_tile_counter, mlrt::kMaxQueues * 64));
        MLRT_HIP(hipMemset(c->d_tile_counter, 0, mlrt::kMaxQueues * 64));  // (every frame's resolve kernel leaves them zero again)
        MLRT_HIP(hipMalloc(&c->d_status, 68 * sizeof(unsigned int)));
        MLRT_HIP(hipMemset(c->d_status, 0, 68 * sizeof(unsigned int)));
        if (options->stats || options->debug_pixel >= 0) {
            const bool instrumented = c->build->instrumented() != 0;
            if (!instrumented) {
                std::fprintf(stderr, "mlrt: stats / debug_pixel need the debug build of the library (python mu-lambda-raytracer_amd/build.py --debug; MLRT_LIB=.../libmlrt_hip_debug.so): ignored\n");
                c->opt.stats = 0; c->opt.debug_pixel = -1;
            } else if (options->stats) {
                MLRT_HIP(hipMalloc(&c->d_stats, 16 * sizeof(unsigned long long)));
                MLRT_HIP(hipMemset(c->d_stats, 0, 16 * sizeof(unsigned long long)));
            }
        }
        MLRT_HIP(hipEventCreate(&c->ev_start));
        MLRT_HIP(hipEventCreate(&c->ev_stop));
        // LDS plan: the blob is staged in LDS when it fits next to the per-wave accumulators and the frame parameters
        // (the strict build is a test instrument: small workgroups reading the blob through L2)
        const int waves = c->block_threads / 64;
        const size_t acc = (size_t)waves * (size_t)c->build->acc_bytes_per_wave(c->variant) + (size_t)c->build->lds_frame_bytes() + pair_stack_bytes;
        // (f64 builds read the Perlin tables - the blob's tail - from HBM: they alone would keep final_scene's blob out of LDS)
        c->lds_blob_bytes = c->build->perlin_in_hbm() ? c->off[3] : (int32_t)blob.size();
        if ((c->variant == 1 || c->variant == 2) && c->build->lean_materials_in_hbm()) c->lds_blob_bytes = c->off[1];  // entries only (the blob starts with them)
        c->lds_scene = !strict && (size_t)c->lds_blob_bytes + acc <= lds_limit && options->lds_scene != 0;
        c->dyn_lds = (c->lds_scene ? (size_t)c->lds_blob_bytes : 0) + acc;
        const int occ = c->build->occupancy(c->variant, c->lds_scene, c->dyn_lds);
        c->blocks_per_cu = occ > 0 ? occ : 1;
        *out = c;
    });
    if (rc != MLRT_OK && c) destroy(c);
    return rc;
}

void mlrt_context_destroy(mlrt_context* ctx) {
    (void)guarded([&] { destroy(ctx); });  // (guarded: leaves the caller's current device as it was)
}

int32_t mlrt_context_render_device(mlrt_context* ctx, const mlrt_camera* camera, const mlrt_params* params, const int32_t* d_rows, int32_t n_rows,
                                   int32_t* d_rgb_out, double* d_accum_out, void* hip_stream) {
    return guarded([&] {
        MLRT_REQUIRE(ctx, "context is NULL");
        check_params(camera, params, n_rows);
        MLRT_REQUIRE(d_rgb_out || n_rows == 0, "d_rgb_out is NULL");
        launch(ctx, camera, params, d_rows, n_rows, d_rgb_out, d_accum_out, static_cast<hipStream_t>(hip_stream));
    });
}

int32_t mlrt_context_render(mlrt_context* ctx, const mlrt_camera* camera, const mlrt_params* params, const int32_t* rows, int32_t n_rows,
                            int32_t* rgb_out, mlrt_progress_fn progress, void* user) {
    return guarded([&] {
        MLRT_REQUIRE(ctx, "context is NULL");
        MLRT_REQUIRE(camera && params, "camera/params is NULL");
        if (!rows) n_rows = params->image_height;
        check_params(camera, params, n_rows);
        MLRT_REQUIRE(rgb_out || n_rows == 0, "rgb_out is NULL");
        if (n_rows == 0) return;
        MLRT_HIP(hipSetDevice(ctx->device));
        std::vector<int32_t> host_rows((size_t)n_rows);
        for (int32_t k = 0; k < n_rows; k++) {
            host_rows[(size_t)k] = rows ? rows[k] : k;
            MLRT_REQUIRE(host_rows[(size_t)k] >= 0 && host_rows[(size_t)k] < params->image_height, "row index out of range");
        }
        const size_t row_elems = (size_t)params->image_width * 3;
        // the row list and the frame on the device: the context's own buffers, grown on demand (not allocated and freed per call)
        if ((size_t)n_rows > ctx->host_rows_cap) {
            if (ctx->d_host_rows) MLRT_HIP(hipFree(ctx->d_host_rows));
            ctx->d_host_rows = nullptr; ctx->host_rows_cap = 0;
            MLRT_HIP(hipMalloc(&ctx->d_host_rows, (size_t)n_rows * sizeof(int32_t)));
            ctx->host_rows_cap = (size_t)n_rows;
        }
        if ((size_t)n_rows * row_elems > ctx->host_rgb_cap) {
            if (ctx->d_host_rgb) MLRT_HIP(hipFree(ctx->d_host_rgb));
            ctx->d_host_rgb = nullptr; ctx->host_rgb_cap = 0;
            MLRT_HIP(hipMalloc(&ctx->d_host_rgb, (size_t)n_rows * row_elems * sizeof(int32_t)));
            ctx->host_rgb_cap = (size_t)n_rows * row_elems;
        }
        int32_t* const d_rows = ctx->d_host_rows;
        int32_t* const d_rgb = ctx->d_host_rgb;
        MLRT_HIP(hipMemcpy(d_rows, host_rows.data(), (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice));
        // with a progress callback the image is rendered in a few bands so the caller sees rows complete (each band is a
        // synchronised launch with its own queue tail: four cost ~3 % of a C2 frame, sixteen cost ~15 %); without one: one launch
        const int32_t bands = progress ? std::min<int32_t>(n_rows, 4) : 1;
        int32_t done = 0;
        for (int32_t b = 0; b < bands; b++) {
            const int32_t begin = (int32_t)((int64_t)n_rows * b / bands), end = (int32_t)((int64_t)n_rows * (b + 1) / bands);
            if (end == begin) continue;
            launch(ctx, camera, params, d_rows + begin, end - begin, d_rgb + (size_t)begin * row_elems, nullptr, nullptr);
            // a band's rows go home as soon as the band is done (in stream order behind its kernels)
            MLRT_HIP(hipMemcpyAsync(rgb_out + (size_t)begin * row_elems, d_rgb + (size_t)begin * row_elems, (size_t)(end - begin) * row_elems * sizeof(int32_t), hipMemcpyDeviceToHost, nullptr));
            if (progress) {
                MLRT_HIP(hipStreamSynchronize(nullptr));
                done = end;
                progress(done, n_rows, user);
            }
        }
        MLRT_HIP(hipStreamSynchronize(nullptr));
        if (mlrt_context_check(ctx) != MLRT_OK) throw Fail{MLRT_ERR_HIP, std::string(mlrt_last_error())};
    });
}

int32_t mlrt_context_check(mlrt_context* ctx) {
    return guarded([&] {
        MLRT_REQUIRE(ctx, "context is NULL");
        MLRT_HIP(hipSetDevice(ctx->device));
        if (ctx->timed) MLRT_HIP(hipEventSynchronize(ctx->ev_stop));
        unsigned int st[68];
        MLRT_HIP(hipMemcpy(st, ctx->d_status, sizeof st, hipMemcpyDeviceToHost));
        if (st[0] == 0u) return;
        MLRT_HIP(hipMemset(ctx->d_status, 0, sizeof st));
        static const char* kPhase[] = {"trav", "special", "shade", "idle", "wait"};
        std::string lanes;
        for (int l = 0; l < 64; l++) {
            const unsigned int ph = st[4 + l] & 0xFFu;
            lanes += std::string(l ? " " : "") + (ph < 5 ? kPhase[ph] : "?") + ((st[4 + l] >> 8) & 1u ? "*" : "") + "@" + std::to_string(st[4 + l] >> 16);
        }
        throw Fail{MLRT_ERR_HIP, "render kernel watchdog: " + std::to_string(st[0]) + " wave(s) made no progress and gave up (last: workgroup " + std::to_string(st[1]) +
                                     ", wave " + std::to_string(st[2]) + "); the frame is incomplete.  Lane phases (* = holds a path, @entry): " + lanes};
    });
}

double mlrt_context_last_kernel_ms(mlrt_context* ctx) {
    if (!ctx || !ctx->timed) return -1.0;
    if (hipEventSynchronize(ctx->ev_stop) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop) != hipSuccess) return -1.0;
    return (double)ms;
}

int32_t mlrt_context_info(mlrt_context* ctx, int64_t out[4]) {
    if (!ctx || !out) { mlrt::set_last_error("mlrt_context_info: NULL argument"); return MLRT_ERR_INVALID; }
    out[0] = ctx->blob_bytes;
    out[1] = ctx->lds_scene ? 1 : 0;
    out[2] = ctx->block_threads;
    out[3] = ctx->last_grid > 0 ? ctx->last_grid : (int64_t)ctx->n_cus * ctx->blocks_per_cu;
    return MLRT_OK;
}

int32_t mlrt_context_kernel_variant(mlrt_context* ctx) { return ctx ? ctx->variant : -1; }

// Test hook (not part of include/mlrt.h): a "ray frame" through the RELEASE render kernel - tile queue, lane state machine, batches and
// all.  Sample s of pixel p of a W x H frame starts on rays[p * spp + s] (origin xyz, direction xyz; p = j * W + i, row 0 = bottom)
// instead of the camera's ray; everything after the primary ray is the product path (the sample's Philox stream is positioned behind its
// four camera draws as always).  accum_out [H][W][3]: the f64 colour sums of the pixels; rgb_out [H][W][3] or NULL.  The primary
// candidate lists, which are derived from the camera, are off for the frame.  tests/test_gpu_ray_frames.py compares with the oracle's
// trace of the same rays.
int32_t mlrt_debug_ray_frame(mlrt_context* ctx, const mlrt_params* params, const double* rays, double* accum_out, int32_t* rgb_out) {
    double* d_table = nullptr;
    double* d_accum = nullptr;
    int32_t* d_rgb = nullptr;
    const int32_t rc = guarded([&] {
        MLRT_REQUIRE(ctx && params && rays && accum_out, "NULL argument");
        mlrt_camera cam{};  // unused by a ray frame, but checked like any camera
        cam.origin[2] = 1.0; cam.lower_left_corner[0] = -1.0; cam.lower_left_corner[1] = -1.0; cam.horizontal[0] = 2.0; cam.vertical[1] = 2.0; cam.u[0] = 1.0; cam.v[1] = 1.0;
        check_params(&cam, params, params->image_height);
        MLRT_HIP(hipSetDevice(ctx->device));
        const size_t pixels = (size_t)params->image_width * (size_t)params->image_height;
        const size_t n_rays = pixels * (size_t)params->samples_per_pixel;
        MLRT_HIP(hipMalloc(&d_table, n_rays * 6 * sizeof(double)));
        MLRT_HIP(hipMemcpy(d_table, rays, n_rays * 6 * sizeof(double), hipMemcpyHostToDevice));
        MLRT_HIP(hipMalloc(&d_accum, pixels * 3 * sizeof(double)));
        MLRT_HIP(hipMalloc(&d_rgb, pixels * 3 * sizeof(int32_t)));
        ctx->d_ray_table = d_table;
        try {
            launch(ctx, &cam, params, nullptr, params->image_height, d_rgb, d_accum, nullptr);
        } catch (...) {
            ctx->d_ray_table = nullptr;
            throw;
        }
        ctx->d_ray_table = nullptr;
        MLRT_HIP(hipStreamSynchronize(nullptr));
        MLRT_HIP(hipMemcpy(accum_out, d_accum, pixels * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (rgb_out) MLRT_HIP(hipMemcpy(rgb_out, d_rgb, pixels * 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (mlrt_context_check(ctx) != MLRT_OK) throw Fail{MLRT_ERR_HIP, std::string(mlrt_last_error())};
    });
    if (d_table) (void)hipFree(d_table);
    if (d_accum) (void)hipFree(d_accum);
    if (d_rgb) (void)hipFree(d_rgb);
    return rc;
}

// Debug only (MLRT_STATS=1; not part of the ABI headers): copies and clears the in-kernel counters.
int32_t mlrt_debug_stats(mlrt_context* ctx, unsigned long long out[16]) {
    if (!ctx || !ctx->d_stats) return MLRT_ERR_UNSUPPORTED;
    if (hipMemcpy(out, ctx->d_stats, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return MLRT_ERR_HIP;
    (void)hipMemset(ctx->d_stats, 0, 16 * sizeof(unsigned long long));
    return MLRT_OK;
}

int32_t mlrt_render(const mlrt_scene* scene, const mlrt_camera* camera, const mlrt_params* params, int32_t precision, int32_t* rgb_out,
                    mlrt_progress_fn progress, void* user) {
    mlrt_context* ctx = nullptr;
    mlrt_options o;
    mlrt_options_from_env(&o);
    if (params && params->tracer == MLRT_TRACER_SINGLE_LIGHT) o.single_light_tracer = 1;
    int32_t rc = mlrt_context_create_opts(scene, 0, precision, &o, &ctx);
    if (rc != MLRT_OK) return rc;
    rc = mlrt_context_render(ctx, camera, params, nullptr, 0, rgb_out, progress, user);
    mlrt_context_destroy(ctx);
    return rc;
}

}  // extern "C"

// ---- directed probes (debug library only; see "directed probes" in mlrt_kernel.hip and tests/test_gpu_directed.py) ----------
namespace {
struct DeviceBuffers {  // scratch for one probe call
    std::vector<void*> ptrs;
    ~DeviceBuffers() { for (void* p : ptrs) (void)hipFree(p); }
    void* alloc(size_t bytes, const void* host = nullptr) {
        void* p = nullptr;
        MLRT_HIP(hipMalloc(&p, bytes ? bytes : 16));
        ptrs.push_back(p);
        if (host && bytes) MLRT_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
        return p;
    }
};
template <typename R>
void scene_params(const mlrt_context* c, uint64_t seed, KParams<R>& K) {
    K.blob = c->d_blob; K.blob_bytes = c->blob_bytes;
    K.off_nodes = c->off[0]; K.off_materials = c->off[1]; K.off_textures = c->off[2];
    K.off_perlin_vec = c->off[3]; K.off_perlin_perm = c->off[4];
    K.off_sphere_index = c->off[5]; K.n_sphere_index = c->n_sphere_index;
    K.n_nodes = c->n_nodes;
    K.images = c->d_images;
    K.seed_lo = (uint32_t)seed; K.seed_hi = (uint32_t)(seed >> 32);
}
// kind, and the launch on the context's build; `with` fills the KParams of the right type
template <typename F>
void probe_launch(mlrt_context* c, int kind, uint64_t seed, int n, const void* a, const void* b, void* out, void* out2, int iarg, F&& with) {
    if (!c->build->probe) throw Fail{MLRT_ERR_UNSUPPORTED, "the directed probes exist in the debug build of the library only (build.py --debug)"};
    MLRT_REQUIRE(c->variant == 0 || c->variant >= 3, "probes need the FULL packing of the scene: create the context with options.force_full = 1");
    MLRT_HIP(hipSetDevice(c->device));
    int rc;
    if (c->precision == MLRT_PRECISION_FP32) {
        KParams<float> K{};
        scene_params(c, seed, K);
        with(K);
        rc = c->build->probe(kind, &K, n, a, b, out, out2, iarg, nullptr);
    } else {
        KParams<double> K{};
        scene_params(c, seed, K);
        with(K);
        rc = c->build->probe(kind, &K, n, a, b, out, out2, iarg, nullptr);
    }
    if (rc != 0) throw Fail{MLRT_ERR_HIP, std::string("probe launch failed: ") + hipGetErrorString((hipError_t)rc)};
    MLRT_HIP(hipDeviceSynchronize());
}
}  // namespace

extern "C" {
// One bounce of each ray through the context's packed scene: out[n][24] = hit, t, p xyz, normal xyz, u, v, front_face, scattered,
// scattered origin xyz, scattered direction xyz, attenuation (scattered) | emitted (not) rgb, stream draw index after the
// traversal, material kind, entry index.  psd[n][3] = Philox (pixel, sample, draw index) the ray starts with.
int32_t mlrt_debug_probe_rays(mlrt_context* ctx, uint64_t seed, int32_t n, const double* rays, const uint32_t* psd, double* out) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && rays && psd && out && n >= 0, "bad argument");
        DeviceBuffers d;
        void* d_rays = d.alloc((size_t)n * 6 * sizeof(double), rays);
        void* d_psd = d.alloc((size_t)n * 3 * sizeof(uint32_t), psd);
        void* d_out = d.alloc((size_t)n * 24 * sizeof(double));
        probe_launch(ctx, 0, seed, n, d_rays, d_psd, d_out, nullptr, 0, [](auto&) {});
        MLRT_HIP(hipMemcpy(out, d_out, (size_t)n * 24 * sizeof(double), hipMemcpyDeviceToHost));
    });
}
// Camera::get_ray for (i, j, sample) triples of a frame: out[n][6] = origin, direction
int32_t mlrt_debug_probe_camera(mlrt_context* ctx, const mlrt_camera* cam, const mlrt_params* p, int32_t n, const int32_t* ijs, double* out) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && cam && p && ijs && out && n >= 0, "bad argument");
        DeviceBuffers d;
        void* d_ijs = d.alloc((size_t)n * 3 * sizeof(int32_t), ijs);
        void* d_out = d.alloc((size_t)n * 6 * sizeof(double));
        probe_launch(ctx, 1, p->seed, n, d_ijs, nullptr, d_out, nullptr, 0, [&](auto& K) {
            for (int k = 0; k < 3; k++) {
                K.cam_origin[k] = cam->origin[k]; K.cam_llc[k] = cam->lower_left_corner[k];
                K.cam_horizontal[k] = cam->horizontal[k]; K.cam_vertical[k] = cam->vertical[k];
                K.cam_u[k] = cam->u[k]; K.cam_v[k] = cam->v[k];
            }
            K.lens_radius = cam->lens_radius; K.has_lens = cam->lens_radius != 0.0 ? 1 : 0;
            K.inv_wm1 = 1.0 / ((double)p->image_width - 1.0); K.inv_hm1 = 1.0 / ((double)p->image_height - 1.0);
            K.width = p->image_width; K.height = p->image_height;
        });
        MLRT_HIP(hipMemcpy(out, d_out, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost));
    });
}
// The build's culling test on explicit boxes (entries made by the packer's rounding rule for this build) and rays, t in [0.001, tmax]
int32_t mlrt_debug_probe_slab(mlrt_context* ctx, int32_t n, const double* boxes, const double* rays, const double* tmax, int32_t* hit) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && boxes && rays && tmax && hit && n >= 0, "bad argument");
        if (!ctx->build->probe) throw Fail{MLRT_ERR_UNSUPPORTED, "the directed probes exist in the debug build of the library only (build.py --debug)"};
        MLRT_HIP(hipSetDevice(ctx->device));
        DeviceBuffers d;
        const bool cull32 = ctx->precision == MLRT_PRECISION_FP32 || (ctx->precision == MLRT_PRECISION_FP64 && !ctx->opt.cull_f64);
        const Round round = ctx->precision == MLRT_PRECISION_FP64_STRICT ? Round::EXACT : (ctx->precision == MLRT_PRECISION_FP64 && cull32 ? Round::RIGOROUS : Round::OUTWARD);
        void* d_boxes;
        auto bounds_of = [&](int i) { Bounds b; for (int k = 0; k < 3; k++) { b.lo[k] = std::fmin(boxes[i * 6 + k], boxes[i * 6 + 3 + k]); b.hi[k] = std::fmax(boxes[i * 6 + k], boxes[i * 6 + 3 + k]); } return b; };
        if (cull32) {
            Entries<float> E; E.round = round;
            std::vector<DNode<float>> e((size_t)n);
            for (int i = 0; i < n; i++) e[(size_t)i] = E.box_entry(bounds_of(i), 0);
            d_boxes = d.alloc(e.size() * sizeof(DNode<float>), e.data());
        } else {
            Entries<double> E; E.round = round;
            std::vector<DNode<double>> e((size_t)n);
            for (int i = 0; i < n; i++) {
                e[(size_t)i] = E.box_entry(bounds_of(i), 0);
                if (round == Round::EXACT) for (int k = 0; k < 6; k++) e[(size_t)i].v[k] = boxes[i * 6 + k];  // (the reference's AABB as given, min/max not sorted)
            }
            d_boxes = d.alloc(e.size() * sizeof(DNode<double>), e.data());
        }
        void* d_rays = d.alloc((size_t)n * 6 * sizeof(double), rays);
        void* d_tmax = d.alloc((size_t)n * sizeof(double), tmax);
        void* d_hit = d.alloc((size_t)n * sizeof(int32_t));
        const int rc = ctx->build->probe(2, d_boxes, n, d_rays, d_tmax, d_hit, nullptr, 0, nullptr);
        if (rc != 0) throw Fail{MLRT_ERR_HIP, std::string("probe launch failed: ") + hipGetErrorString((hipError_t)rc)};
        MLRT_HIP(hipDeviceSynchronize());
        MLRT_HIP(hipMemcpy(hit, d_hit, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    });
}
// Perlin::turbulence of table `table` at points (already multiplied by the texture's scale): serial evaluation and the wave-level one
int32_t mlrt_debug_probe_turbulence(mlrt_context* ctx, int32_t table, int32_t n, const double* points, double* serial, double* wave) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && points && serial && wave && n >= 0 && table >= 0, "bad argument");
        DeviceBuffers d;
        void* d_p = d.alloc((size_t)n * 3 * sizeof(double), points);
        void* d_s = d.alloc((size_t)n * sizeof(double));
        void* d_w = d.alloc((size_t)n * sizeof(double));
        probe_launch(ctx, 3, 0, n, d_p, nullptr, d_s, d_w, table, [](auto&) {});
        MLRT_HIP(hipMemcpy(serial, d_s, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        MLRT_HIP(hipMemcpy(wave, d_w, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    });
}
// Texture::value of texture `tex` at uvp[n][5] = u, v, p xyz: out[n][3]
int32_t mlrt_debug_probe_texture(mlrt_context* ctx, int32_t tex, int32_t n, const double* uvp, double* out) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && uvp && out && n >= 0 && tex >= 0, "bad argument");
        DeviceBuffers d;
        void* d_in = d.alloc((size_t)n * 5 * sizeof(double), uvp);
        void* d_out = d.alloc((size_t)n * 3 * sizeof(double));
        probe_launch(ctx, 4, 0, n, d_in, nullptr, d_out, nullptr, tex, [](auto&) {});
        MLRT_HIP(hipMemcpy(out, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    });
}
// One finished sample of radiance[i] accumulated (to_fixed) and resolved with spp (to_rgb): rgb[n]
// Divisor::over (mlrt_kernel.hip) against x / a on blocks * 256 * per_thread * 5 operand pairs made on the device; exponents of the
// divisor / the numerators uniform in [a_lo, a_hi] / [x_lo, x_hi].  out[0] = quotients that differ, out[1..3] = bits of one x, a, q.
int32_t mlrt_debug_probe_divide(mlrt_context* ctx, int32_t blocks, int32_t per_thread, uint32_t seed, int32_t a_lo, int32_t a_hi, int32_t x_lo, int32_t x_hi,
                                unsigned long long out[4]) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && out && blocks >= 1 && per_thread >= 1 && a_lo <= a_hi && x_lo <= x_hi && a_lo >= -1022 && a_hi <= 1023 && x_lo >= -1022 && x_hi <= 1023, "bad argument");
        DeviceBuffers d;
        const unsigned long long zero[4] = {0, 0, 0, 0};
        void* d_out = d.alloc(sizeof zero, zero);
        const int args[6] = {per_thread, (int)seed, a_lo, a_hi, x_lo, x_hi};
        probe_launch(ctx, 6, 0, blocks, args, nullptr, d_out, nullptr, 0, [](auto&) {});
        MLRT_HIP(hipMemcpy(out, d_out, sizeof zero, hipMemcpyDeviceToHost));
    });
}
// log_uniform (mlrt_kernel.hip: the logarithm of a medium's draw, volumes.rs:48) of the uniforms (k0 + i) 2^-24, i < n
int32_t mlrt_debug_probe_log(mlrt_context* ctx, int32_t k0, int32_t n, double* out) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && out && n >= 0 && k0 >= 0 && (int64_t)k0 + n <= (1 << 24), "bad argument");
        DeviceBuffers d;
        void* d_out = d.alloc((size_t)n * sizeof(double));
        probe_launch(ctx, 7, 0, n, nullptr, nullptr, d_out, nullptr, k0, [](auto&) {});
        MLRT_HIP(hipMemcpy(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    });
}
int32_t mlrt_debug_probe_to_rgb(mlrt_context* ctx, int32_t spp, int32_t n, const double* radiance, int32_t* rgb) {
    return guarded([&] {
        MLRT_REQUIRE(ctx && radiance && rgb && n >= 0 && spp >= 1, "bad argument");
        DeviceBuffers d;
        void* d_in = d.alloc((size_t)n * sizeof(double), radiance);
        void* d_rgb = d.alloc((size_t)n * sizeof(int32_t));
        void* d_acc = d.alloc((size_t)n * sizeof(unsigned long long));
        probe_launch(ctx, 5, 0, n, d_in, ctx->d_tile_counter, d_rgb, d_acc, spp, [](auto&) {});
        MLRT_HIP(hipMemcpy(rgb, d_rgb, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    });
}
}  // extern "C"

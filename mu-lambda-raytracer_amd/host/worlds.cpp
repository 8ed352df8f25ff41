// host/worlds.cpp - host-side mirror of the reference's world builders, emitting the flattened mlrt_scene.
//
// Follows /root/reference/src/worlds.rs (trait World :14-25, registry :471-484) and the pieces of
// src/bhv.rs (SceneBuilder/BHV::new/Node::new :65-145), src/textures.rs (Perlin::new :62-74,136-148) that consume
// the scene RNG stream, replaying every draw in the reference's order so the scene at --seed is the reference's.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "host_internal.hpp"

namespace mlrt {

namespace {

struct Box3 {
    double lo[3], hi[3];
};
Box3 box_of(const double a[3], const double b[3]) {  // AABB::new, bhv.rs:16-20
    Box3 r;
    for (int k = 0; k < 3; k++) { r.lo[k] = std::fmin(a[k], b[k]); r.hi[k] = std::fmax(a[k], b[k]); }
    return r;
}
Box3 surround(const Box3& a, const Box3& b) {  // AABB::surround, bhv.rs:44-53
    Box3 r;
    for (int k = 0; k < 3; k++) { r.lo[k] = std::fmin(a.lo[k], b.lo[k]); r.hi[k] = std::fmax(a.hi[k], b.hi[k]); }
    return r;
}

// SceneBuilder (bhv.rs:65-82): the shapes handed to BHV::new, as prim_refs indices with their bounding boxes.
struct Pending {
    int32_t ref;
    Box3 box;
};

class Flattener {
  public:
    explicit Flattener(WorldStore& store) : s_(store) {}

    int32_t solid(double r, double g, double b) {
        mlrt_texture t{};
        t.kind = MLRT_TEX_SOLID; t.index = -1; t.index2 = -1;
        t.color[0] = r; t.color[1] = g; t.color[2] = b;
        s_.textures.push_back(t);
        return (int32_t)s_.textures.size() - 1;
    }
    int32_t checker(int32_t odd, int32_t even) {  // textures.rs:28-49
        mlrt_texture t{};
        t.kind = MLRT_TEX_CHECKER; t.index = odd; t.index2 = even;
        s_.textures.push_back(t);
        return (int32_t)s_.textures.size() - 1;
    }
    int32_t image_texture(int32_t image) {
        mlrt_texture t{};
        t.kind = MLRT_TEX_IMAGE; t.index = image; t.index2 = -1;
        s_.textures.push_back(t);
        return (int32_t)s_.textures.size() - 1;
    }
    // NoiseTexture::new(scale, rng) -> Perlin::new(rng), textures.rs:62-74,157-161
    int32_t noise_texture(double scale, Pcg64& rng) {
        auto tables = std::make_unique<PerlinTables>();
        for (int i = 0; i < 1024; i++) {
            double v[3];
            for (double& c : v) c = rng.gen_range(-1.0, 1.0);  // Vec3::random(-1,1): x, y, z in order
            const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            for (int k = 0; k < 3; k++) tables->ranvec[i * 3 + k] = v[k] / len;  // .unit()
        }
        for (int axis = 0; axis < 3; axis++) {  // Perlin::permute, textures.rs:136-148
            int32_t* p = &tables->perm[axis * 1024];
            for (int i = 0; i < 1024; i++) p[i] = i;
            for (int i = 1023; i >= 1; i--) {
                const uint64_t j = rng.gen_range_usize(0, (uint64_t)i);
                std::swap(p[i], p[j]);
            }
        }
        s_.perlin_tables.push_back(std::move(tables));
        mlrt_texture t{};
        t.kind = MLRT_TEX_NOISE; t.index = (int32_t)s_.perlin_tables.size() - 1; t.index2 = -1; t.scale = scale;
        s_.textures.push_back(t);
        return (int32_t)s_.textures.size() - 1;
    }
    int32_t material(int32_t kind, int32_t texture, double a0 = 0, double a1 = 0, double a2 = 0, double param = 0) {
        mlrt_material m{};
        m.kind = kind; m.texture = texture; m.albedo[0] = a0; m.albedo[1] = a1; m.albedo[2] = a2; m.param = param;
        s_.materials.push_back(m);
        return (int32_t)s_.materials.size() - 1;
    }
    int32_t lambertian(int32_t texture) { return material(MLRT_MAT_LAMBERTIAN, texture); }
    int32_t lambertian_rgb(double r, double g, double b) { return lambertian(solid(r, g, b)); }
    int32_t metal(double r, double g, double b, double fuzz) { return material(MLRT_MAT_METAL, -1, r, g, b, fuzz); }
    int32_t dielectric(double ior) { return material(MLRT_MAT_DIELECTRIC, -1, 0, 0, 0, ior); }
    int32_t diffuse_light_rgb(double r, double g, double b) { return material(MLRT_MAT_DIFFUSE_LIGHT, solid(r, g, b)); }
    int32_t isotropic_rgb(double r, double g, double b) { return material(MLRT_MAT_ISOTROPIC, solid(r, g, b)); }

    mlrt_prim_ref sphere(double cx, double cy, double cz, double radius, int32_t mat) {  // Sphere::new, shapes.rs:32-35
        mlrt_sphere sp{};
        sp.center[0] = cx; sp.center[1] = cy; sp.center[2] = cz; sp.radius = radius; sp.material = mat;
        s_.spheres.push_back(sp);
        return mlrt_prim_ref{MLRT_PRIM_SPHERE, (int32_t)s_.spheres.size() - 1};
    }
    mlrt_prim_ref block(const double p0[3], const double p1[3], int32_t mat) {  // Block::new, shapes.rs:172-192
        mlrt_box b{};
        for (int k = 0; k < 3; k++) { b.min[k] = p0[k]; b.max[k] = p1[k]; }
        b.material = mat;
        s_.boxes.push_back(b);
        return mlrt_prim_ref{MLRT_PRIM_BOX, (int32_t)s_.boxes.size() - 1};
    }
    // AARect::new, aarects.rs:31-43 (with its quirk: a1_v0 = a1_v0.min(a1_v0))
    mlrt_prim_ref rect(int a0, double v00, double v01, int a1, double v10, double v11, double plane, int32_t mat) {
        mlrt_rect r{};
        r.a0 = a0; r.a1 = a1;
        r.a0_v0 = std::fmin(v00, v01); r.a0_v1 = std::fmax(v01, v00);
        r.a1_v0 = v10; r.a1_v1 = std::fmax(v11, v10);
        r.plane = plane; r.material = mat;
        s_.rects.push_back(r);
        return mlrt_prim_ref{MLRT_PRIM_RECT, (int32_t)s_.rects.size() - 1};
    }
    mlrt_prim_ref xy_rect(double x0, double x1, double y0, double y1, double z, int32_t m) { return rect(0, x0, x1, 1, y0, y1, z, m); }
    mlrt_prim_ref xz_rect(double x0, double x1, double z0, double z1, double y, int32_t m) { return rect(0, x0, x1, 2, z0, z1, y, m); }
    mlrt_prim_ref yz_rect(double y0, double y1, double z0, double z1, double x, int32_t m) { return rect(1, y0, y1, 2, z0, z1, x, m); }

    // Bounded::bounding_box of a primitive (shapes.rs:84-89, :193-197).  Bare rects are never given to a
    // SceneBuilder by the reference (AARect::bounding_box is degenerate, aarects.rs:66-77), so they are refused.
    bool bounds(const mlrt_prim_ref& p, Box3& out) const {
        if (p.kind == MLRT_PRIM_SPHERE) {
            const mlrt_sphere& sp = s_.spheres[(size_t)p.index];
            double a[3], b[3];
            for (int k = 0; k < 3; k++) { a[k] = sp.center[k] - sp.radius; b[k] = sp.center[k] + sp.radius; }
            out = box_of(a, b);
            return true;
        }
        if (p.kind == MLRT_PRIM_BOX) {
            const mlrt_box& bx = s_.boxes[(size_t)p.index];
            out = box_of(bx.min, bx.max);
            return true;
        }
        return false;
    }
    void add(std::vector<Pending>& builder, const mlrt_prim_ref& p) {  // SceneBuilder::add
        Pending e;
        s_.prim_refs.push_back(p);
        e.ref = (int32_t)s_.prim_refs.size() - 1;
        bounds(p, e.box);
        builder.push_back(e);
    }

    // BHV::new -> Node::new (bhv.rs:85-95, 122-145).  Returns the bvhs index.
    int32_t bvh(std::vector<Pending>& shapes, Pcg64& rng) {
        mlrt_bvh b{};
        Box3 box;
        b.root = shapes.empty() ? MLRT_BVH_EMPTY : build_node(shapes.data(), shapes.size(), rng, box);
        shapes.clear();  // scene.contents.clear()
        s_.bvhs.push_back(b);
        return (int32_t)s_.bvhs.size() - 1;
    }

    void object_prim(const mlrt_prim_ref& p, int32_t transform = -1) {
        mlrt_object o{};
        o.kind = MLRT_OBJ_PRIM; o.bvh = -1; o.prim = p; o.transform = transform; o.phase_material = -1;
        s_.objects.push_back(o);
    }
    void object_bvh(int32_t bvh_index, int32_t transform = -1) {
        mlrt_object o{};
        o.kind = MLRT_OBJ_BVH; o.bvh = bvh_index; o.prim = mlrt_prim_ref{-1, -1}; o.transform = transform; o.phase_material = -1;
        s_.objects.push_back(o);
    }
    // ConstantMedium::from_color(boundary, d, color), volumes.rs:13-23
    void object_medium(const mlrt_prim_ref& boundary, double density, double r, double g, double b, int32_t transform = -1) {
        mlrt_object o{};
        o.kind = MLRT_OBJ_MEDIUM; o.bvh = -1; o.prim = boundary; o.transform = transform;
        o.phase_material = isotropic_rgb(r, g, b);
        o.neg_inv_density = -1.0 / density;
        s_.objects.push_back(o);
    }
    // Translate::new(offset, Rotate::new(axis, angle_degrees, x)), transforms.rs:26-28, 60-67
    int32_t translate_rotate(int axis, double angle_degrees, double ox, double oy, double oz) {
        mlrt_transform t{};
        t.has_rotate = 1; t.axis = axis;
        const double theta = angle_degrees * 3.14159265358979323846264338327950288 / 180.0;
        t.sin_theta = std::sin(theta); t.cos_theta = std::cos(theta);
        t.has_translate = 1; t.offset[0] = ox; t.offset[1] = oy; t.offset[2] = oz;
        s_.transforms.push_back(t);
        return (int32_t)s_.transforms.size() - 1;
    }

  private:
    // Node::new: `[v] => Leaf`, else random axis, STABLE sort by bounding_box().minimum[axis], split at len/2,
    // left subtree first.  Returns the child code (>=0 inner node index, <0 ~prim_ref) and the node's box.
    int32_t build_node(Pending* shapes, size_t n, Pcg64& rng, Box3& box_out) {
        if (n == 1) { box_out = shapes[0].box; return ~shapes[0].ref; }
        const size_t axis = (size_t)rng.gen_range_usize(0, 3);
        std::stable_sort(shapes, shapes + n, [axis](const Pending& a, const Pending& b) { return a.box.lo[axis] < b.box.lo[axis]; });
        const size_t half = n / 2;
        const int32_t me = (int32_t)s_.bvh_nodes.size();
        s_.bvh_nodes.push_back(mlrt_bvh_node{});
        Box3 lb, rb;
        const int32_t l = build_node(shapes, half, rng, lb);
        const int32_t r = build_node(shapes + half, n - half, rng, rb);
        box_out = surround(lb, rb);
        mlrt_bvh_node& node = s_.bvh_nodes[(size_t)me];
        for (int k = 0; k < 3; k++) { node.bounds_min[k] = box_out.lo[k]; node.bounds_max[k] = box_out.hi[k]; }
        node.left = l; node.right = r;
        return me;
    }
    WorldStore& s_;
};

void set_camera(WorldStore& s, double fx, double fy, double fz, double ax, double ay, double az, double fov) {
    s.lookfrom[0] = fx; s.lookfrom[1] = fy; s.lookfrom[2] = fz;
    s.lookat[0] = ax; s.lookat[1] = ay; s.lookat[2] = az;
    s.field_of_view = fov;
}
uint64_t peek(const Pcg64& rng) { Pcg64 c = rng; return c.next_u64(); }

// worlds.rs:79-112 Random / :129-162 RandomChk
void build_random(WorldStore& s, Flattener& f, Pcg64& rng, bool chk) {
    std::vector<Pending> world;
    const int32_t ground = chk ? f.lambertian(f.checker(f.solid(0.2, 0.3, 0.1), f.solid(0.9, 0.9, 0.9))) : f.lambertian_rgb(0.5, 0.5, 0.5);
    f.add(world, f.sphere(0.0, -1000.0, 0.0, 1000.0, ground));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            const double choose_mat = rng.gen_range(0.0, 1.0);
            const double cx = (double)a + 0.9 * rng.gen_range(0.0, 1.0);
            const double cz = (double)b + 0.9 * rng.gen_range(0.0, 1.0);
            const double dx = cx - 4.0, dy = 0.2 - 0.2, dz = cz - 0.0;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9) {
                if (choose_mat < 0.8) {
                    double c1[3], c2[3];
                    for (double& c : c1) c = rng.gen_range(0.0, 1.0);
                    for (double& c : c2) c = rng.gen_range(0.0, 1.0);
                    f.add(world, f.sphere(cx, 0.2, cz, 0.2, f.lambertian_rgb(c1[0] * c2[0], c1[1] * c2[1], c1[2] * c2[2])));
                } else if (choose_mat < 0.95) {
                    double al[3];
                    for (double& c : al) c = rng.gen_range(0.5, 1.0);
                    const double fuzz = rng.gen_range(0.0, 0.5);
                    f.add(world, f.sphere(cx, 0.2, cz, 0.2, f.metal(al[0], al[1], al[2], fuzz)));
                } else {
                    f.add(world, f.sphere(cx, 0.2, cz, 0.2, f.dielectric(1.5)));
                }
            }
        }
    }
    f.add(world, f.sphere(0.0, 1.0, 0.0, 1.0, f.dielectric(1.5)));
    f.add(world, f.sphere(-4.0, 1.0, 0.0, 1.0, f.lambertian_rgb(0.4, 0.2, 0.1)));
    f.add(world, f.sphere(4.0, 1.0, 0.0, 1.0, f.metal(0.7, 0.6, 0.5, 0.0)));
    s.probe_before_last_bvh = peek(rng);
    f.object_bvh(f.bvh(world, rng));
    s.background_kind = MLRT_BG_GRADIENT;
    set_camera(s, 13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 20.0);
}

// worlds.rs:260-287 CornellBox / :308-335 CornellSmoke
void build_cornell(WorldStore& s, Flattener& f, bool smoke) {
    const int32_t red = f.lambertian_rgb(0.65, 0.05, 0.05), white = f.lambertian_rgb(0.73, 0.73, 0.73), green = f.lambertian_rgb(0.12, 0.45, 0.15);
    const int32_t light = f.diffuse_light_rgb(7.0, 7.0, 7.0);
    f.object_prim(f.yz_rect(0.0, 555.0, 0.0, 555.0, 555.0, green));
    f.object_prim(f.yz_rect(0.0, 555.0, 0.0, 555.0, 0.0, red));
    f.object_prim(f.xz_rect(113.0, 443.0, 127.0, 432.0, 554.0, light));
    f.object_prim(f.xz_rect(0.0, 555.0, 0.0, 555.0, 0.0, white));
    f.object_prim(f.xz_rect(0.0, 555.0, 0.0, 555.0, 555.0, white));
    f.object_prim(f.xy_rect(0.0, 555.0, 0.0, 555.0, 555.0, white));
    const double zero[3] = {0, 0, 0}, big[3] = {165.0, 330.0, 165.0}, small_b[3] = {165.0, 165.0, 165.0};
    const mlrt_prim_ref large = f.block(zero, big, white);
    const int32_t large_t = f.translate_rotate(1, 15.0, 265.0, 0.0, 295.0);
    const mlrt_prim_ref small_p = f.block(zero, small_b, white);
    const int32_t small_t = f.translate_rotate(1, -18.0, 130.0, 0.0, 65.0);
    if (smoke) {
        f.object_medium(large, 0.01, 0.0, 0.0, 0.0, large_t);
        f.object_medium(small_p, 0.01, 1.0, 1.0, 1.0, small_t);
    } else {
        f.object_prim(large, large_t);
        f.object_prim(small_p, small_t);
    }
    s.background_kind = MLRT_BG_BLACK;
    set_camera(s, 278.0, 278.0, -800.0, 278.0, 278.0, 0.0, 40.0);
}

// worlds.rs:386-468 FinalScene
void build_final_scene(WorldStore& s, Flattener& f, Pcg64& rng) {
    f.object_prim(f.xz_rect(123.0, 423.0, 147.0, 412.0, 554.0, f.diffuse_light_rgb(9.0, 9.0, 9.0)));  // Light.
    {  // Ground.
        const int32_t ground = f.lambertian_rgb(0.48, 0.83, 0.53);
        std::vector<Pending> ground_blocks;
        for (int i = 0; i < 20; i++) {
            for (int j = 0; j < 20; j++) {
                const double w = 100.0;
                const double x0 = -1000.0 + (double)i * w, z0 = -1000.0 + (double)j * w, y0 = 0.0;
                const double x1 = x0 + w;
                const double y1 = rng.gen_range(1.0, 70.0);
                const double z1 = z0 + w;
                const double p0[3] = {x0, y0, z0}, p1[3] = {x1, y1, z1};
                f.add(ground_blocks, f.block(p0, p1, ground));
            }
        }
        f.object_bvh(f.bvh(ground_blocks, rng));
    }
    f.object_prim(f.sphere(400.0, 400.0, 400.0, 50.0, f.lambertian_rgb(0.7, 0.3, 0.1)));  // gold sphere
    f.object_prim(f.sphere(260.0, 150.0, 45.0, 50.0, f.dielectric(1.5)));                  // glass sphere
    f.object_prim(f.sphere(0.0, 150.0, 145.0, 50.0, f.metal(0.8, 0.8, 0.9, 1.0)));         // metallic sphere
    {  // Glass sphere filled with blue-ish smoke
        const mlrt_prim_ref boundary = f.sphere(360.0, 150.0, 145.0, 70.0, f.dielectric(1.5));
        f.object_prim(boundary);
        f.object_medium(boundary, 0.2, 0.2, 0.4, 0.9);
    }
    f.object_medium(f.sphere(0.0, 0.0, 0.0, 1000.0, f.dielectric(1.5)), 0.0001, 1.0, 1.0, 1.0);  // fog
    f.object_prim(f.sphere(400.0, 200.0, 400.0, 100.0, f.lambertian(f.image_texture(0))));        // earth
    f.object_prim(f.sphere(220.0, 280.0, 300.0, 80.0, f.lambertian(f.noise_texture(0.1, rng))));  // marble
    {  // Foam.
        std::vector<Pending> foam;
        const int32_t white = f.lambertian_rgb(0.73, 0.73, 0.73);
        for (int i = 0; i < 1000; i++) {
            double c[3];
            for (double& x : c) x = rng.gen_range(0.0, 165.0);
            f.add(foam, f.sphere(c[0], c[1], c[2], 10.0, white));
        }
        s.probe_before_last_bvh = peek(rng);
        const int32_t foam_bvh = f.bvh(foam, rng);
        f.object_bvh(foam_bvh, f.translate_rotate(1, 15.0, -100.0, 270.0, 395.0));
    }
    s.background_kind = MLRT_BG_BLACK;
    set_camera(s, 478.0, 278.0, -600.0, 278.0, 278.0, 0.0, 40.0);
}

}  // namespace

const std::vector<std::string>& world_registry() {  // worlds.rs:471-484
    static const std::vector<std::string> names = {"simple", "random", "random_chk", "two_spheres", "simple_light",
                                                   "cornell_box", "cornell_smoke", "earth", "debug_perlin", "final_scene"};
    return names;
}

std::unique_ptr<WorldStore> build_world_store(const std::string& name, uint64_t seed, const uint8_t* earth_rgb, int32_t ew, int32_t eh,
                                              std::string& error) {
    auto store = std::make_unique<WorldStore>();
    WorldStore& s = *store;
    s.name = name;
    Flattener f(s);
    Pcg64 rng = Pcg64::seed_from_u64(seed + 0);  // main.rs:185 rngator.rng(0)
    const bool wants_earth = name == "earth" || name == "final_scene";
    if (wants_earth) {
        if (!earth_rgb || ew <= 0 || eh <= 0) { error = "world '" + name + "' needs the decoded earth map (worlds.rs:180,441)"; return nullptr; }
        s.image_pixels.emplace_back(earth_rgb, earth_rgb + (size_t)ew * eh * 3);
        s.images.push_back(mlrt_image{ew, eh, nullptr});
    }
    if (name == "simple") {  // worlds.rs:42-59
        std::vector<Pending> world;
        const int32_t left = f.dielectric(1.5);
        f.add(world, f.sphere(0.0, -100.5, -1.0, 100.0, f.lambertian_rgb(0.8, 0.8, 0.0)));
        f.add(world, f.sphere(0.0, 0.0, -1.0, 0.5, f.lambertian_rgb(0.1, 0.3, 0.5)));
        f.add(world, f.sphere(-1.0, 0.0, -1.0, 0.5, left));
        f.add(world, f.sphere(-1.0, 0.0, -1.0, -0.4, left));
        f.add(world, f.sphere(1.0, 0.0, -1.0, 0.5, f.metal(0.8, 0.6, 0.2, 0.0)));
        s.probe_before_last_bvh = peek(rng);
        f.object_bvh(f.bvh(world, rng));
        s.background_kind = MLRT_BG_GRADIENT;
        set_camera(s, -2.0, 2.0, 1.0, 0.0, 0.0, -1.0, 20.0);
    } else if (name == "random" || name == "random_chk") {
        build_random(s, f, rng, name == "random_chk");
    } else if (name == "earth") {  // worlds.rs:179-186
        f.object_prim(f.sphere(0.0, 0.0, 0.0, 2.0, f.lambertian(f.image_texture(0))));
        s.background_kind = MLRT_BG_GRADIENT;
        set_camera(s, 13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 20.0);
    } else if (name == "two_spheres" || name == "simple_light") {  // worlds.rs:203-210, 227-239
        const int32_t pertext = f.lambertian(f.noise_texture(4.0, rng));
        f.object_prim(f.sphere(0.0, -1000.0, 0.0, 1000.0, pertext));
        f.object_prim(f.sphere(0.0, 2.0, 0.0, 2.0, pertext));
        if (name == "simple_light") {
            f.object_prim(f.xy_rect(3.0, 5.0, 1.0, 3.0, -2.0, f.diffuse_light_rgb(0.0, 7.0, 0.0)));
            f.object_prim(f.sphere(0.0, 6.0, 0.0, 1.5, f.diffuse_light_rgb(7.0, 0.0, 0.0)));
            s.background_kind = MLRT_BG_BLACK;
            set_camera(s, 20.0, 3.0, 6.0, 0.0, 2.0, 0.0, 20.0);
        } else {
            s.background_kind = MLRT_BG_GRADIENT;
            set_camera(s, 13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 20.0);
        }
    } else if (name == "cornell_box" || name == "cornell_smoke") {
        build_cornell(s, f, name == "cornell_smoke");
    } else if (name == "debug_perlin") {  // worlds.rs:355-365
        f.object_prim(f.sphere(278.0, 278.0, 0.0, 80.0, f.lambertian(f.noise_texture(0.1, rng))));
        s.background_kind = MLRT_BG_GRADIENT;
        set_camera(s, 278.0, 278.0, -600.0, 278.0, 278.0, 0.0, 40.0);
    } else if (name == "final_scene") {
        build_final_scene(s, f, rng);
    } else {
        error = "unknown world '" + name + "'";
        return nullptr;
    }
    s.probe_after_build = rng.next_u64();
    s.finalize();
    return store;
}

void WorldStore::finalize() {
    for (size_t i = 0; i < images.size(); i++) images[i].rgb = image_pixels[i].data();
    perlins.clear();
    for (const auto& t : perlin_tables) perlins.push_back(mlrt_perlin{t->ranvec, t->perm});
    scene = mlrt_scene{};
    scene.n_spheres = (int32_t)spheres.size(); scene.spheres = spheres.data();
    scene.n_boxes = (int32_t)boxes.size(); scene.boxes = boxes.data();
    scene.n_rects = (int32_t)rects.size(); scene.rects = rects.data();
    scene.n_prim_refs = (int32_t)prim_refs.size(); scene.prim_refs = prim_refs.data();
    scene.n_bvh_nodes = (int32_t)bvh_nodes.size(); scene.bvh_nodes = bvh_nodes.data();
    scene.n_bvhs = (int32_t)bvhs.size(); scene.bvhs = bvhs.data();
    scene.n_transforms = (int32_t)transforms.size(); scene.transforms = transforms.data();
    scene.n_objects = (int32_t)objects.size(); scene.objects = objects.data();
    scene.n_materials = (int32_t)materials.size(); scene.materials = materials.data();
    scene.n_textures = (int32_t)textures.size(); scene.textures = textures.data();
    scene.n_images = (int32_t)images.size(); scene.images = images.data();
    scene.n_perlins = (int32_t)perlins.size(); scene.perlins = perlins.data();
}

}  // namespace mlrt

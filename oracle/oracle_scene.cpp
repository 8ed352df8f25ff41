// oracle/oracle_scene.cpp - the oracle over an ARBITRARY flattened scene, and its per-function probes.
// TEST INFRASTRUCTURE ONLY (see oracle_core.hpp for the full header and the parity-pinning statement).
//
// oracle.cpp builds the reference's ten worlds by name.  The directed device tests (tests/test_gpu_directed.py) need
// scenes the worlds do not contain - one primitive, one material, an edge case - so this file builds the oracle's object
// tree (the trait objects of /root/reference/src/{hittable,bhv,shapes,aarects,transforms,volumes,materials,textures}.rs as
// restated in oracle_core.hpp) from the same plain-old-data description the product's C ABI takes (include/mlrt.h,
// mlrt_scene: a public header, nothing of the product is linked), and evaluates single calls on it:
//   orc_scene_probe_rays     world.hit(ray, 0.001, inf) + material.scatter / emit   (raytrace.rs:90-97)
//   orc_scene_texture_value  Texture::value                                          (textures.rs, image_texture.rs)
//   orc_scene_turbulence     Perlin::turbulence                                      (textures.rs:76-88)
//   orc_probe_get_ray        the two pixel draws + Camera::get_ray                   (raytrace.rs:191-193, camera.rs:40-48)
// with the device's Philox stream (PhiloxStream) positioned where the caller says.
#include "../include/mlrt.h"
#include "oracle_core.hpp"

namespace orc {
namespace {

struct SceneWorld {
    HittableList root;
    std::vector<std::shared_ptr<Texture>> textures;
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<std::shared_ptr<Perlin>> perlins;
    std::vector<std::shared_ptr<ImageData>> images;
};

struct BvhShape : Bounded {  // a BHV whose topology is GIVEN (bhv.rs:109-112, 147-165): no RNG, no sort
    std::unique_ptr<BvhNode> root;
    bool hit(const Ray& r, double tmin, double tmax, Rng& rng, Hit& out) const override { return root->hit(r, tmin, tmax, rng, out); }
    AABB bounding_box() const override { return root->bounding_box(); }
};

std::shared_ptr<Bounded> make_prim(const mlrt_scene& s, const SceneWorld& w, const mlrt_prim_ref& p) {
    if (p.kind == MLRT_PRIM_SPHERE) {
        const mlrt_sphere& sp = s.spheres[p.index];
        return std::make_shared<Sphere>(Point3(sp.center[0], sp.center[1], sp.center[2]), sp.radius, w.materials[(size_t)sp.material]);
    }
    if (p.kind == MLRT_PRIM_BOX) {
        const mlrt_box& b = s.boxes[p.index];
        return std::make_shared<Block>(Point3(b.min[0], b.min[1], b.min[2]), Point3(b.max[0], b.max[1], b.max[2]), w.materials[(size_t)b.material]);
    }
    const mlrt_rect& r = s.rects[p.index];
    AARect a;  // the fields as the caller computed them (AARect::new's min/max quirk Q2 is the scene builder's business)
    a.a0 = r.a0; a.a1 = r.a1; a.aplane = 3 - r.a0 - r.a1;
    a.a0_v0 = r.a0_v0; a.a0_v1 = r.a0_v1; a.a1_v0 = r.a1_v0; a.a1_v1 = r.a1_v1; a.aplane_v = r.plane;
    return std::make_shared<RectShape>(a, w.materials[(size_t)r.material]);
}

std::unique_ptr<BvhNode> make_bvh(const mlrt_scene& s, const SceneWorld& w, int32_t child) {
    std::unique_ptr<BvhNode> node(new BvhNode());
    if (child == MLRT_BVH_EMPTY) { node->leaf = true; node->shape = std::make_shared<Empty>(); return node; }
    if (child < 0) { node->leaf = true; node->shape = make_prim(s, w, s.prim_refs[~child]); return node; }
    const mlrt_bvh_node& src = s.bvh_nodes[child];
    node->leaf = false;
    node->bounds.minimum = Point3(src.bounds_min[0], src.bounds_min[1], src.bounds_min[2]);
    node->bounds.maximum = Point3(src.bounds_max[0], src.bounds_max[1], src.bounds_max[2]);
    node->left = make_bvh(s, w, src.left);
    node->right = make_bvh(s, w, src.right);
    return node;
}

std::shared_ptr<Bounded> wrap_transform(const mlrt_scene& s, int32_t t, std::shared_ptr<Bounded> inner) {
    if (t < 0) return inner;
    const mlrt_transform& x = s.transforms[t];
    std::shared_ptr<Bounded> out = inner;
    if (x.has_rotate) {
        auto rot = std::make_shared<Rotate>(x.axis, 0.0, out);  // (the bounding box Rotate::new computes is never used by hit)
        rot->sin_theta = x.sin_theta; rot->cos_theta = x.cos_theta;
        out = rot;
    }
    if (x.has_translate) out = std::make_shared<Translate>(Vec3(x.offset[0], x.offset[1], x.offset[2]), out);
    return out;
}

SceneWorld* build(const mlrt_scene& s) {
    std::unique_ptr<SceneWorld> w(new SceneWorld());
    for (int i = 0; i < s.n_images; i++) {
        auto im = std::make_shared<ImageData>();
        im->width = s.images[i].width; im->height = s.images[i].height;
        im->rgb.assign(s.images[i].rgb, s.images[i].rgb + (size_t)im->width * im->height * 3);
        w->images.push_back(im);
    }
    for (int i = 0; i < s.n_perlins; i++) {
        auto p = std::make_shared<Perlin>();
        for (int k = 0; k < 1024; k++) {
            p->ranvec[k] = Vec3(s.perlins[i].ranvec[k * 3], s.perlins[i].ranvec[k * 3 + 1], s.perlins[i].ranvec[k * 3 + 2]);
            p->perm_x[k] = (uint64_t)s.perlins[i].perm[k]; p->perm_y[k] = (uint64_t)s.perlins[i].perm[1024 + k]; p->perm_z[k] = (uint64_t)s.perlins[i].perm[2048 + k];
        }
        w->perlins.push_back(p);
    }
    w->textures.resize((size_t)s.n_textures);
    // checkers may refer to later textures: two passes
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < s.n_textures; i++) {
            const mlrt_texture& t = s.textures[i];
            if (t.kind == MLRT_TEX_SOLID && pass == 0) w->textures[(size_t)i] = std::make_shared<SolidColor>(t.color[0], t.color[1], t.color[2]);
            if (t.kind == MLRT_TEX_IMAGE && pass == 0) w->textures[(size_t)i] = std::make_shared<ImageTexture>(w->images[(size_t)t.index]);
            if (t.kind == MLRT_TEX_NOISE && pass == 0) w->textures[(size_t)i] = std::make_shared<NoiseTexture>(w->perlins[(size_t)t.index], t.scale);
            if (t.kind == MLRT_TEX_CHECKER && pass == 1) w->textures[(size_t)i] = std::make_shared<Checker>(w->textures[(size_t)t.index], w->textures[(size_t)t.index2]);
        }
    for (int i = 0; i < s.n_materials; i++) {
        const mlrt_material& m = s.materials[i];
        std::shared_ptr<Material> out;
        if (m.kind == MLRT_MAT_LAMBERTIAN) out = std::make_shared<Lambertian>(w->textures[(size_t)m.texture]);
        else if (m.kind == MLRT_MAT_METAL) out = std::make_shared<Metal>(Color(m.albedo[0], m.albedo[1], m.albedo[2]), m.param);
        else if (m.kind == MLRT_MAT_DIELECTRIC) out = std::make_shared<Dielectric>(m.param);
        else if (m.kind == MLRT_MAT_DIFFUSE_LIGHT) out = std::make_shared<DiffuseLight>(w->textures[(size_t)m.texture]);
        else out = std::make_shared<Isotropic>(w->textures[(size_t)m.texture]);
        w->materials.push_back(out);
    }
    for (int i = 0; i < s.n_objects; i++) {
        const mlrt_object& o = s.objects[i];
        if (o.kind == MLRT_OBJ_MEDIUM) {
            auto boundary = wrap_transform(s, o.transform, make_prim(s, *w, o.prim));
            auto medium = std::make_shared<ConstantMedium>(boundary, 1.0, std::make_shared<SolidColor>(0.0, 0.0, 0.0));
            medium->neg_inv_density = o.neg_inv_density;
            medium->phase_function = std::dynamic_pointer_cast<Isotropic>(w->materials[(size_t)o.phase_material]);
            w->root.add(medium);
            continue;
        }
        std::shared_ptr<Bounded> shape;
        if (o.kind == MLRT_OBJ_BVH) {
            auto b = std::make_shared<BvhShape>();
            b->root = make_bvh(s, *w, s.bvhs[o.bvh].root);
            shape = b;
        } else {
            shape = make_prim(s, *w, o.prim);
        }
        w->root.add(wrap_transform(s, o.transform, shape));
    }
    return w.release();
}

int material_kind(const Material* m) {
    if (dynamic_cast<const Lambertian*>(m)) return MLRT_MAT_LAMBERTIAN;
    if (dynamic_cast<const Metal*>(m)) return MLRT_MAT_METAL;
    if (dynamic_cast<const Dielectric*>(m)) return MLRT_MAT_DIELECTRIC;
    if (dynamic_cast<const DiffuseLight*>(m)) return MLRT_MAT_DIFFUSE_LIGHT;
    return MLRT_MAT_ISOTROPIC;
}

// a Philox stream standing at draw index `draw` (its current block loaded, as if the earlier draws had happened)
PhiloxStream stream_at(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw) {
    PhiloxStream rng(seed, pixel, sample);
    rng.draw = draw;
    if ((draw & 3u) != 0u) {
        uint32_t ctr[4] = {draw >> 2, sample, pixel, 0};
        philox4x32_10(rng.key, ctr, rng.buf);
    }
    return rng;
}

}  // namespace
}  // namespace orc

namespace orc { const Hittable* orc_scene_root(void* scene) { return &((SceneWorld*)scene)->root; } }  // for orc_scene_render (oracle.cpp)

using namespace orc;

extern "C" {

void* orc_scene_new(const mlrt_scene* scene) { return scene ? build(*scene) : nullptr; }
void orc_scene_free(void* h) { delete (SceneWorld*)h; }

// One bounce per ray.  rays[n][6] = origin, direction; psd[n][3] = Philox (pixel, sample, draw index) at the start of the ray.
// out[n][24] = hit, t, p xyz, normal xyz, u, v, front_face, scattered, scattered origin xyz, scattered direction xyz,
// attenuation (scattered) | emitted (not) rgb, stream draw index after the traversal, material kind, 0.
void orc_scene_probe_rays(void* h, uint64_t seed, int n, const double* rays, const uint32_t* psd, double* out) {
    const SceneWorld& w = *(SceneWorld*)h;
    for (int i = 0; i < n; i++) {
        double* r = out + (size_t)i * 24;
        for (int k = 0; k < 24; k++) r[k] = 0.0;
        const Ray ray(Point3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), Vec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
        PhiloxStream rng = stream_at(seed, psd[i * 3], psd[i * 3 + 1], psd[i * 3 + 2]);
        Hit hit;
        const bool got = w.root.hit(ray, 0.001, INF, rng, hit);  // raytrace.rs:90
        r[21] = (double)rng.draw;
        if (!got) continue;
        r[0] = 1.0; r[1] = hit.t;
        for (int k = 0; k < 3; k++) { r[2 + k] = hit.p.e[k]; r[5 + k] = hit.normal.e[k]; }
        r[8] = hit.u; r[9] = hit.v; r[10] = hit.front_face ? 1.0 : 0.0;
        Color att;
        Ray scattered;
        rng.begin_scatter();
        const bool sc = hit.material->scatter(ray, hit, rng, att, scattered);  // raytrace.rs:91
        rng.end_scatter();
        r[11] = sc ? 1.0 : 0.0;
        if (sc) {
            for (int k = 0; k < 3; k++) { r[12 + k] = scattered.orig.e[k]; r[15 + k] = scattered.dir.e[k]; r[18 + k] = att.e[k]; }
        } else {
            const Color e = hit.material->emit(hit.u, hit.v, hit.p);  // raytrace.rs:96
            for (int k = 0; k < 3; k++) r[18 + k] = e.e[k];
        }
        r[22] = (double)material_kind(hit.material);
    }
}

// RecursiveRayTracer::trace (raytrace.rs:79-107) of GIVEN rays: the radiance of ray i with the device's Philox stream of (pixel, sample) =
// ps[i] positioned behind a sample's four camera draws (draw index 4: raytrace.rs:191-192 + camera.rs:41).  out[n][3].  The checker of the
// "ray frames" the release kernel renders (tests/test_gpu_ray_frames.py).
void orc_scene_trace_rays(void* h, int bg_kind, const double* top, const double* bottom, int max_depth, uint64_t seed, int n, const double* rays,
                          const uint32_t* ps, double* out) {
    const SceneWorld& w = *(SceneWorld*)h;
    Background bg;
    bg.kind = bg_kind;
    bg.top = Color(top[0], top[1], top[2]); bg.bottom = Color(bottom[0], bottom[1], bottom[2]);
    for (int i = 0; i < n; i++) {
        const Ray ray(Point3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), Vec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
        PhiloxStream rng = stream_at(seed, ps[i * 2], ps[i * 2 + 1], 4u);
        const Color c = trace_internal(ray, w.root, bg, max_depth, rng);
        for (int k = 0; k < 3; k++) out[i * 3 + k] = c.e[k];
    }
}

// Texture::value of texture `tex` at uvp[n][5] = u, v, p xyz
void orc_scene_texture_value(void* h, int tex, int n, const double* uvp, double* out) {
    const SceneWorld& w = *(SceneWorld*)h;
    for (int i = 0; i < n; i++) {
        const Color c = w.textures[(size_t)tex]->value(uvp[i * 5], uvp[i * 5 + 1], Point3(uvp[i * 5 + 2], uvp[i * 5 + 3], uvp[i * 5 + 4]));
        for (int k = 0; k < 3; k++) out[i * 3 + k] = c.e[k];
    }
}

// Perlin::turbulence of table `table` at points[n][3]
void orc_scene_turbulence(void* h, int table, int n, const double* points, double* out) {
    const SceneWorld& w = *(SceneWorld*)h;
    for (int i = 0; i < n; i++) out[i] = w.perlins[(size_t)table]->turbulence(Point3(points[i * 3], points[i * 3 + 1], points[i * 3 + 2]));
}

// The primary ray of (pixel i, j; sample): raytrace.rs:191-193 + camera.rs:40-48 with the device's stream.  ijs[n][3]; out[n][6].
void orc_probe_get_ray(const double c[19], int width, int height, uint64_t seed, int n, const int32_t* ijs, double* out) {
    Camera cam;
    cam.origin = Point3(c[0], c[1], c[2]); cam.lower_left_corner = Point3(c[3], c[4], c[5]);
    cam.horizontal = Vec3(c[6], c[7], c[8]); cam.vertical = Vec3(c[9], c[10], c[11]);
    cam.u = Vec3(c[12], c[13], c[14]); cam.v = Vec3(c[15], c[16], c[17]); cam.lens_radius = c[18];
    const double wm1 = (double)(width - 1), hm1 = (double)(height - 1);
    for (int k = 0; k < n; k++) {
        const int i = ijs[k * 3], j = ijs[k * 3 + 1];
        PhiloxStream rng(seed, (uint32_t)(j * width + i), (uint32_t)ijs[k * 3 + 2]);
        const double u = ((double)i + rng.gen_range(0.0, 1.0)) / wm1;
        const double v = ((double)j + rng.gen_range(0.0, 1.0)) / hm1;
        const Ray r = cam.get_ray(u, v, rng);
        for (int q = 0; q < 3; q++) { out[k * 6 + q] = r.orig.e[q]; out[k * 6 + 3 + q] = r.dir.e[q]; }
    }
}

// to_rgb (raytrace.rs:59-68) of n single-channel colour sums
void orc_probe_to_rgb(int spp, int n, const double* sums, int32_t* out) {
    for (int i = 0; i < n; i++) {
        int32_t rgb[3];
        to_rgb(Color(sums[i], sums[i], sums[i]), spp, rgb);
        out[i] = rgb[0];
    }
}

}  // extern "C"

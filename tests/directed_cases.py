"""Directed cases for the per-function device tests: small scenes (flattened exactly as the C ABI wants them, include/mlrt.h)
and chosen inputs - the edge cases an image almost never produces.  Shared by tests/golden/make_directed_vectors.py (which
evaluates them with the ORACLE and commits the results to tests/golden/directed_vectors.npz), tests/test_oracle_directed.py
(CPU: the oracle still reproduces the committed vectors, and the reference's own known answers are among them) and
tests/test_gpu_directed.py (GPU: every precision build of the device functions against the committed vectors).

Reference behaviour each group probes (file:line in /root/reference/src):
  aabb        bhv.rs:29-42 + its five unit tests bhv.rs:168-221 (both boxes), zero direction components, origins inside
  spheres     shapes.rs:58-82 (tangent, behind the origin, origin inside, negative radius), sphere_uv shapes.rs:44-56, 200-214
  rects       aarects.rs:45-64 (closed edges, parallel rays), blocks shapes.rs:172-192 (origin inside, edges, corners)
  transforms  transforms.rs:31-42, 127-142 (quirk Q10)
  media       volumes.rs:26-65 (origin inside: t1 < 0; hidden behind a closer hit: t1 >= t2; no draw then)
  materials   materials.rs:25-34, 51-61, 88-106 (total internal reflection boundary), 119-127, volumes.rs:77-83
  textures    textures.rs:40-49, 76-134 (negative lattice coordinates), 163-167, image_texture.rs:16-28 (u = 1, v = 0 clamps)
  camera      raytrace.rs:191-193, camera.rs:40-48          to_rgb  raytrace.rs:59-68 (NaN, negative, saturating)
"""
import ctypes as C

import numpy as np

SEED = 20240607


class MiniScene:
    """A flattened scene under construction; to_ctypes() gives the mlrt_scene (and keeps its arrays alive)."""

    def __init__(self, mlrt):
        self.m = mlrt
        self.spheres, self.boxes, self.rects, self.prim_refs, self.bvh_nodes, self.bvhs = [], [], [], [], [], []
        self.transforms, self.objects, self.materials, self.textures, self.images, self.perlins = [], [], [], [], [], []
        self._keep = []

    # textures / materials -------------------------------------------------------------------------------------
    def solid(self, r, g, b):
        self.textures.append((self.m.Texture, dict(kind=0, index=0, index2=0, color=(r, g, b), scale=0.0)))
        return len(self.textures) - 1

    def checker(self, odd, even):
        self.textures.append((self.m.Texture, dict(kind=3, index=odd, index2=even, color=(0, 0, 0), scale=0.0)))
        return len(self.textures) - 1

    def noise(self, perlin, scale):
        self.textures.append((self.m.Texture, dict(kind=2, index=perlin, index2=0, color=(0, 0, 0), scale=scale)))
        return len(self.textures) - 1

    def image_texture(self, image):
        self.textures.append((self.m.Texture, dict(kind=1, index=image, index2=0, color=(0, 0, 0), scale=0.0)))
        return len(self.textures) - 1

    def image(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        self._keep.append(rgb)
        self.images.append(rgb)
        return len(self.images) - 1

    def perlin(self, rng):
        vec = rng.uniform(-1.0, 1.0, size=(1024, 3))
        vec /= np.linalg.norm(vec, axis=1)[:, None]
        perm = np.stack([rng.permutation(1024) for _ in range(3)]).astype(np.int32)
        self._keep += [np.ascontiguousarray(vec), np.ascontiguousarray(perm)]
        self.perlins.append((self._keep[-2], self._keep[-1]))
        return len(self.perlins) - 1

    def material(self, kind, texture=-1, albedo=(0, 0, 0), param=0.0):
        self.materials.append(dict(kind=kind, texture=texture, albedo=albedo, param=param))
        return len(self.materials) - 1

    def lambertian(self, tex):
        return self.material(0, texture=tex)

    def metal(self, albedo, fuzz):
        return self.material(1, albedo=albedo, param=fuzz)

    def dielectric(self, ior):
        return self.material(2, param=ior)

    def light(self, tex):
        return self.material(3, texture=tex)

    def isotropic(self, tex):
        return self.material(4, texture=tex)

    # primitives ------------------------------------------------------------------------------------------------
    def sphere(self, c, r, mat):
        self.spheres.append(dict(center=c, radius=r, material=mat))
        return (0, len(self.spheres) - 1)

    def box(self, p0, p1, mat):
        self.boxes.append(dict(min=p0, max=p1, material=mat))
        return (1, len(self.boxes) - 1)

    def rect(self, a0, a1, v00, v01, v10, v11, plane, mat):
        self.rects.append(dict(a0=a0, a1=a1, a0_v0=v00, a0_v1=v01, a1_v0=v10, a1_v1=v11, plane=plane, material=mat))
        return (2, len(self.rects) - 1)

    def transform(self, rotate_axis=None, degrees=0.0, offset=None):
        th = np.deg2rad(degrees)
        self.transforms.append(dict(has_rotate=int(rotate_axis is not None), axis=rotate_axis or 0, sin_theta=float(np.sin(th)), cos_theta=float(np.cos(th)),
                                    has_translate=int(offset is not None), offset=offset or (0, 0, 0)))
        return len(self.transforms) - 1

    # objects ---------------------------------------------------------------------------------------------------
    def add_prim(self, prim, transform=-1):
        self.objects.append(dict(kind=0, bvh=0, prim=prim, transform=transform, phase_material=0, neg_inv_density=0.0))

    def add_medium(self, boundary, density, phase_material, transform=-1):
        self.objects.append(dict(kind=2, bvh=0, prim=boundary, transform=transform, phase_material=phase_material, neg_inv_density=-1.0 / density))

    def add_bvh(self, prims, transform=-1):
        """A median-split tree over `prims` in the given order (the topology is the caller's: no RNG here)."""
        first = len(self.prim_refs)
        self.prim_refs += list(prims)

        def bounds(p):
            kind, idx = p
            if kind == 0:
                s = self.spheres[idx]
                c, r = np.array(s["center"], dtype=np.float64), abs(s["radius"])
                return c - r, c + r
            if kind == 1:
                b = self.boxes[idx]
                return np.minimum(b["min"], b["max"]).astype(np.float64), np.maximum(b["min"], b["max"]).astype(np.float64)
            # AARect::bounding_box (aarects.rs:66-76) as the reference computes it: BOTH corners at (a0_v0, a1_v0) - a point in the rect's
            # plane, 0.002 thick - so an inner node above a rect leaf encloses one corner of it only (no reference world does this;
            # include/mlrt.h mlrt_rect says what the product modes do instead)
            r = self.rects[idx]
            ap = 3 - r["a0"] - r["a1"]
            lo, hi = np.zeros(3), np.zeros(3)
            lo[r["a0"]] = hi[r["a0"]] = r["a0_v0"]
            lo[r["a1"]] = hi[r["a1"]] = r["a1_v0"]
            lo[ap], hi[ap] = r["plane"] - 0.001, r["plane"] + 0.001
            return lo, hi

        def build(lo, hi):
            if hi - lo == 1:
                return ~(first + lo), bounds(prims[lo])  # a leaf child: ~index into prim_refs
            mid = (lo + hi) // 2
            (l, lb), (r, rb) = build(lo, mid), build(mid, hi)
            mn, mx = np.minimum(lb[0], rb[0]), np.maximum(lb[1], rb[1])
            self.bvh_nodes.append(dict(bounds_min=tuple(mn), bounds_max=tuple(mx), left=l, right=r))
            return len(self.bvh_nodes) - 1, (mn, mx)

        root, _ = build(0, len(prims)) if prims else (-(2 ** 31), None)
        self.bvhs.append(dict(root=root))
        self.objects.append(dict(kind=1, bvh=len(self.bvhs) - 1, prim=(0, 0), transform=transform, phase_material=0, neg_inv_density=0.0))

    # -----------------------------------------------------------------------------------------------------------
    def to_ctypes(self):
        m = self.m

        def arr(ctype, rows, fill):
            a = (ctype * max(1, len(rows)))()
            for i, row in enumerate(rows):
                fill(a[i], row)
            self._keep.append(a)
            return a

        def f_sphere(s, r):
            s.center[:] = r["center"]; s.radius = r["radius"]; s.material = r["material"]

        def f_box(b, r):
            b.min[:] = r["min"]; b.max[:] = r["max"]; b.material = r["material"]

        def f_rect(x, r):
            for k in ("a0", "a1", "a0_v0", "a0_v1", "a1_v0", "a1_v1", "plane", "material"):
                setattr(x, k, r[k])

        def f_ref(x, r):
            x.kind, x.index = r

        def f_node(x, r):
            x.bounds_min[:] = r["bounds_min"]; x.bounds_max[:] = r["bounds_max"]; x.left = r["left"]; x.right = r["right"]

        def f_bvh(x, r):
            x.root = r["root"]

        def f_xf(x, r):
            x.has_rotate, x.axis, x.sin_theta, x.cos_theta, x.has_translate = r["has_rotate"], r["axis"], r["sin_theta"], r["cos_theta"], r["has_translate"]
            x.offset[:] = r["offset"]

        def f_obj(x, r):
            x.kind, x.bvh, x.transform, x.phase_material, x.neg_inv_density = r["kind"], r["bvh"], r["transform"], r["phase_material"], r["neg_inv_density"]
            x.prim.kind, x.prim.index = r["prim"]

        def f_mat(x, r):
            x.kind, x.texture, x.param = r["kind"], r["texture"], r["param"]
            x.albedo[:] = r["albedo"]

        def f_tex(x, r):
            d = r[1]
            x.kind, x.index, x.index2, x.scale = d["kind"], d["index"], d["index2"], d["scale"]
            x.color[:] = d["color"]

        def f_img(x, r):
            x.height, x.width = r.shape[:2]
            x.rgb = r.ctypes.data_as(C.POINTER(C.c_uint8))

        def f_perlin(x, r):
            x.ranvec = r[0].ctypes.data_as(C.POINTER(C.c_double))
            x.perm = r[1].ctypes.data_as(C.POINTER(C.c_int32))

        s = m.Scene()
        s.spheres = arr(m.Sphere, self.spheres, f_sphere); s.n_spheres = len(self.spheres)
        s.boxes = arr(m.Box, self.boxes, f_box); s.n_boxes = len(self.boxes)
        s.rects = arr(m.Rect, self.rects, f_rect); s.n_rects = len(self.rects)
        s.prim_refs = arr(m.PrimRef, self.prim_refs, f_ref); s.n_prim_refs = len(self.prim_refs)
        s.bvh_nodes = arr(m.BvhNode, self.bvh_nodes, f_node); s.n_bvh_nodes = len(self.bvh_nodes)
        s.bvhs = arr(m.Bvh, self.bvhs, f_bvh); s.n_bvhs = len(self.bvhs)
        s.transforms = arr(m.Transform, self.transforms, f_xf); s.n_transforms = len(self.transforms)
        s.objects = arr(m.Object, self.objects, f_obj); s.n_objects = len(self.objects)
        s.materials = arr(m.Material, self.materials, f_mat); s.n_materials = len(self.materials)
        s.textures = arr(m.Texture, self.textures, f_tex); s.n_textures = len(self.textures)
        s.images = arr(m.Image, self.images, f_img); s.n_images = len(self.images)
        s.perlins = arr(m.Perlin, self.perlins, f_perlin); s.n_perlins = len(self.perlins)
        self._keep.append(s)
        return s


# ======================================================================================================================
# Scenes.  Every scene carries one image texture, one noise texture and one box so that it packs as a FULL scene with every
# table present; the probes address textures by index.
def base(mlrt):
    rng = np.random.default_rng(7)
    sc = MiniScene(mlrt)
    img = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)  # 7 wide, 5 high: neither a power of two
    sc.t_white = sc.solid(0.73, 0.73, 0.73)
    sc.t_red = sc.solid(0.65, 0.05, 0.05)
    sc.t_image = sc.image_texture(sc.image(img))
    sc.t_noise = sc.noise(sc.perlin(rng), 0.1)
    sc.t_checker = sc.checker(sc.t_red, sc.t_noise)
    sc.t_light = sc.solid(7.0, 6.0, 5.0)
    return sc


def prim_centre(k):
    """Where object k of scene_primitives sits: 10 units apart in x, and shifted by k-specific binary fractions in y and z.  A ray
    lying EXACTLY in the plane of an AARect (origin on the plane, direction component 0) gives t = 0/0 = NaN in the reference, NaN
    passes every comparison of aarects.rs:47-54 and the rect reports a hit at t = NaN (quirk Q16, measure zero for jittered camera
    rays).  Rays built from one object's features would lie in planes of OTHER objects if all shared y = 0 / z = -5; the shifts
    keep that from happening by accident, and the cases that do it on purpose are labelled "inplane:" and carry no parity."""
    return np.array([10.0 * k, k / 128.0, -5.0 - k / 64.0])


def scene_primitives(mlrt):
    """One of everything at the top level, well separated along x (object k around prim_centre(k))."""
    sc = base(mlrt)
    m_img = sc.lambertian(sc.t_image)
    m_white = sc.lambertian(sc.t_white)
    m_chk = sc.lambertian(sc.t_checker)
    m_metal = sc.metal((0.8, 0.85, 0.88), 0.0)
    m_fuzz = sc.metal((0.7, 0.6, 0.5), 1.0)
    m_glass = sc.dielectric(1.5)
    m_light = sc.light(sc.t_light)
    c = [prim_centre(k) for k in range(12)]
    t = tuple
    sc.add_prim(sc.sphere(t(c[0]), 1.0, m_img))            # 0: uv sphere
    sc.add_prim(sc.sphere(t(c[1]), 2.0, m_glass))          # 1: glass
    sc.add_prim(sc.sphere(t(c[1]), -1.5, m_glass))         # 2: hollow (negative radius, worlds.rs:54), concentric with 1
    sc.add_prim(sc.sphere(t(c[3]), 1.0, m_metal))          # 3
    sc.add_prim(sc.sphere(t(c[4]), 1.0, m_fuzz))           # 4
    sc.add_prim(sc.rect(0, 2, c[5][0], c[5][0] + 1.0, c[5][2] - 1.0, c[5][2], c[5][1], m_light))         # 5: XZ rect at y = c.y
    sc.add_prim(sc.rect(0, 1, c[6][0], c[6][0] + 2.0, c[6][1] - 1.0, c[6][1] + 1.0, c[6][2], m_chk))     # 6: XY rect at z = c.z
    sc.add_prim(sc.rect(1, 2, c[7][1] - 1.0, c[7][1] + 1.0, c[7][2] - 1.0, c[7][2] + 1.0, c[7][0], m_white))  # 7: YZ rect at x = c.x
    sc.add_prim(sc.box(t(c[8] - 1.0), t(c[8] + 1.0), m_white))   # 8: block
    sc.add_prim(sc.box(t(c[9] - (0.5, 1.0, 1.0)), t(c[9] + (0.5, 1.0, 1.0)), m_metal))  # 9: metal block
    sc.add_prim(sc.sphere(t(c[10]), 2.0, m_glass))         # 10: solid glass (total internal reflection from inside)
    sc.add_prim(sc.box(t(c[11] - (1.0, 0.5, 2.0)), t(c[11] + (1.0, 0.5, 2.0)), m_img))  # 11: image-textured block (u, v of the face rect)
    return sc


def scene_bvh(mlrt):
    """A BVH of 24 spheres and 8 blocks on a lattice (axis-parallel rays run exactly along box faces), inside a transform, plus
    a plain BVH."""
    sc = base(mlrt)
    mats = [sc.lambertian(sc.t_white), sc.metal((0.9, 0.9, 0.9), 0.25), sc.dielectric(1.3), sc.lambertian(sc.t_noise)]
    prims = []
    k = 0
    for ix in range(4):
        for iy in range(2):
            for iz in range(3):
                prims.append(sc.sphere((2.0 * ix, 2.0 * iy, -4.0 - 2.0 * iz), 0.5 + 0.25 * (k % 3), mats[k % 4]))
                k += 1
    for ix in range(4):
        for iy in range(2):
            prims.append(sc.box((2.0 * ix - 0.5, 2.0 * iy + 4.0, -5.0), (2.0 * ix + 0.5, 2.0 * iy + 5.0, -4.0), mats[(ix + iy) % 2]))
    sc.add_bvh(prims)
    sc.add_bvh(prims[:7], transform=sc.transform(rotate_axis=1, degrees=15.0, offset=(20.0, 0.5, -3.0)))
    sc.add_prim(sc.box((0.0, 0.0, 0.0), (1.65, 3.3, 1.65), mats[0]), transform=sc.transform(rotate_axis=1, degrees=-18.0, offset=(30.0, 0.03125, -6.0)))
    sc.add_prim(sc.sphere((0.0, 0.0, 0.0), 1.0, mats[1]), transform=sc.transform(offset=(40.0, 0.0, -5.0)))
    sc.add_prim(sc.sphere((0.0, 0.0, 0.0), 1.0, mats[2]), transform=sc.transform(rotate_axis=0, degrees=30.0))
    return sc


def scene_media(mlrt):
    """An opaque sphere, then two media (a sphere of smoke and a block of fog inside a transform), then a sphere behind them."""
    sc = base(mlrt)
    m_white = sc.lambertian(sc.t_white)
    iso_blue = sc.isotropic(sc.solid(0.2, 0.4, 0.9))
    iso_img = sc.isotropic(sc.t_checker)
    sc.add_prim(sc.sphere((0.0, 0.0, -3.0), 0.5, m_white))                       # in front of the smoke for rays near the axis
    sc.add_medium(sc.sphere((0.0, 0.0, -8.0), 2.0, m_white), 0.8, iso_blue)      # dense: most rays scatter inside
    sc.add_medium(sc.box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), m_white), 0.3, iso_img, transform=sc.transform(rotate_axis=1, degrees=20.0, offset=(10.0, 0.0, -8.0)))
    sc.add_medium(sc.sphere((0.0, 0.0, 0.0), 50.0, m_white), 0.01, iso_blue)     # thin global fog: the origin of every ray is inside
    sc.add_prim(sc.sphere((0.0, 0.0, -14.0), 1.5, m_white))
    return sc


SCENES = {"primitives": scene_primitives, "bvh": scene_bvh, "media": scene_media}


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def rays_for(name):
    """rays [n][6] (origin, direction - not normalised, as the reference's rays) and labels."""
    R, L = [], []

    def add(o, d, label):
        R.append(list(map(float, o)) + list(map(float, d))); L.append(label)

    if name == "primitives":
        C = [prim_centre(k) for k in range(12)]
        X, Y, Z = np.eye(3)
        c = C[0]
        # sphere_uv known answers (shapes.rs:200-214): hit the unit sphere exactly at +-x, +-y, +-z (outside, towards the centre)
        for n in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1)):
            n = np.array(n, dtype=np.float64)
            add(c + 4.0 * n, -2.0 * n, "sphere_uv %s" % (tuple(int(x) for x in n),))
        add(c + X + 5 * Z, -Z, "sphere tangent (discriminant exactly 0)")
        add(c + (1.0 + 2.0 ** -40) * X + 5 * Z, -Z, "sphere just missed")
        add(c - 4 * Z, -Z, "sphere behind the origin")
        add(c + 0.25 * Y, (0.3, 0.1, 1.0), "origin inside the sphere: far root")
        add(c + 1.0005 * Z, -Z, "near root below t_min = 0.001: far root")
        add(c + 10 * Z, -0.5 * Z, "direction of length 0.5")
        add(c + 10 * Z, -300.0 * Z, "direction of length 300")
        # glass: front face, inside the shell (between the spheres), inside the hollow; total internal reflection boundary
        add(C[1] + 10 * Z, (0.1, 0.05, -1.0), "glass front")
        add(C[1] + 1.75 * Y, (0.0, -1.0, 0.2), "inside the glass shell towards the hollow (negative radius: normal flips)")
        add(C[1], (0.3, 0.9, 0.1), "inside the hollow outwards")
        for s in (0.60, 0.66, 0.6666, 0.6667, 0.67, 0.75, 0.999):  # sin(theta) at the exit; 1 / 1.5 = 0.66667 is the boundary (materials.rs:95-98)
            th = np.arcsin(s)
            p = C[10] + 2.0 * Y  # top of the solid glass sphere, reached from inside
            d = np.array([np.sin(th), np.cos(th), 0.0])
            add(p - 0.05 * d, d, "glass from inside, sin = %g" % s)
        add(C[3] + 10 * Z, (0.05, 0.02, -1.0), "mirror")
        add(C[3] + 0.999 * Y + 10 * Z, -Z, "mirror grazing")
        for k in range(6):
            add(C[4] + (0.15 * k - 0.375) * X + 0.9 * Y + 10 * Z, -Z, "fuzzy metal near the rim (absorbed when the fuzzed ray points inwards) %d" % k)
        # rects: closed edges (aarects.rs:54), corners, parallel rays, both sides
        c = C[5]
        add(c + 3 * Y - 0.5 * Z, -Y, "xz rect: edge a0 = a0_v0")
        add(c + X + 3 * Y, -Y, "xz rect: corner (a0_v1, a1_v1)")
        add(c - 2.0 ** -30 * X + 3 * Y - 0.5 * Z, -Y, "xz rect: just outside the edge")
        add(c + 0.5 * X - 3 * Y - 0.5 * Z, Y, "xz rect from below (light: emits, no scatter)")
        add(c + 0.5 * X + Y - 0.5 * Z, X, "ray parallel to the rect")
        add(c + 0.5 * X - 0.5 * Z, X, "inplane: ray inside the plane of the rect (0/0: the reference reports a hit at t = NaN)")
        c = C[6]
        add(c + X + 0.5 * Y + 7 * Z, (0.1, -0.05, -1.0), "xy rect, checker of solid and noise")
        add(c + 0.3 * X - 0.7 * Y + 7 * Z, -Z, "xy rect, checker, other parity")
        add(c + 2 * X + Y + 5 * Z, -Z, "xy rect corner")
        c = C[7]
        add(c - 5 * X + 0.3 * Y - 0.2 * Z, (1.0, 0.01, 0.02), "yz rect")
        add(c + 5 * X + 0.3 * Y - 0.2 * Z, (-1.0, 0.01, 0.02), "yz rect from the other side")
        # blocks: origin inside, edge, corner, face-parallel (bhv.rs:183-220 shapes on a Block)
        c = C[8]
        add(c + 10 * Z, -Z, "block front face")
        add(c, (0.3, 0.2, 1.0), "origin inside the block: exit face")
        add(c - 0.9999 * X + 10 * Z, -Z, "block: just inside the face a0 = min")
        add(c - 1.0001 * X + 10 * Z, -Z, "block: just outside the face")
        add(c - 0.9999 * X + 0.9999 * Y + 10 * Z, -Z, "block: just inside an edge")
        add(c - 2 * X + 2 * Y + 2 * Z, (1.0, -1.0, -1.0), "block: through the corner")
        add(c + 10 * Z, Z, "block behind the origin")
        add(c - X + 10 * Z, -Z, "inplane: block: along a face plane")
        add(c - X + Y + 10 * Z, -Z, "inplane: block: along an edge")
        add(C[9] + 0.2 * Y + 10 * Z, (0.01, 0.0, -1.0), "metal block")
        add(C[9] + 5 * Y, -Y, "metal block from above")
        # an image texture on a Block reads the u, v of the face's rect (shapes.rs:172-192, aarects.rs:56-57): one ray per face pair
        add(C[11] + 0.3 * X - 0.2 * Y + 10 * Z, (0.01, 0.02, -1.0), "textured block: xy face (u along x, v along y)")
        add(C[11] - 0.6 * X + 5 * Y + 1.1 * Z, (0.02, -1.0, 0.03), "textured block: xz face (u along x, v along z)")
        add(C[11] - 6 * X + 0.1 * Y - 1.3 * Z, (1.0, 0.02, 0.05), "textured block: yz face (u along y, v along z)")
        add(C[11] + 0.1 * X + 0.1 * Y, (0.2, 0.1, 1.0), "textured block: from inside (exit face, back face)")
        add((300.0, 100.0, 100.0), (1.0, 1.0, 1.0), "nothing")
    elif name == "bvh":
        # The primitives of the lattice (as scene_bvh builds them), to recognise a ray that lies EXACTLY in a face plane of a
        # primitive's bounding box.  There the reference's AABB::hit computes 0/0 = NaN on that axis and its NaN-ignoring min / max
        # turn the box into a miss (bhv.rs:33-38) - even when Sphere::hit would report a tangent hit - so WHICH boxes a BVH
        # happens to test decides the answer (quirk Q17; measure zero for jittered rays; the reference's own tests step
        # 1e-4 off the plane, bhv.rs:183-220).  Such rays are labelled "boxplane:": the strict build (reference topology and
        # arithmetic) must still equal the oracle on them, the builds with their own culling structure are not held to it.
        bounds = []
        k = 0
        for ix in range(4):
            for iy in range(2):
                for iz in range(3):
                    r = 0.5 + 0.25 * (k % 3)
                    c = np.array([2.0 * ix, 2.0 * iy, -4.0 - 2.0 * iz])
                    bounds.append((c - r, c + r))
                    k += 1
        for ix in range(4):
            for iy in range(2):
                bounds.append((np.array([2.0 * ix - 0.5, 2.0 * iy + 4.0, -5.0]), np.array([2.0 * ix + 0.5, 2.0 * iy + 5.0, -4.0])))
        plain_add = add

        def add(o, d, label):  # noqa: F811
            o, d = np.asarray(o, dtype=np.float64), np.asarray(d, dtype=np.float64)
            inplane = any(d[a] == 0.0 and (o[a] == lo[a] or o[a] == hi[a]) for lo, hi in bounds for a in range(3))
            plain_add(o, d, ("boxplane: " if inplane else "") + label)
        # axis-parallel rays (zero direction components, both signs of zero) along lattice lines, faces and edges of the boxes
        for x in (0.0, 0.4999, 0.5001, 1.0, 2.0, 6.0, 6.75, 7.0):  # (boxes end at x = 2 ix +- 0.5: just inside / outside, as bhv.rs:183-200)
            for y in (0.0, 0.5, 2.0, 4.0001, 4.5, 4.9999):
                add((x, y, 5.0), (0.0, 0.0, -1.0), "axis -z at (%g, %g)" % (x, y))
        for y in (0.0, 2.0, 4.5):
            for z in (-4.0001, -4.5, -6.0, -8.0):
                add((-5.0, y, z), (1.0, 0.0, -0.0), "axis +x, d.z = -0 at (%g, %g)" % (y, z))
                add((12.0, y, z), (-2.0, -0.0, 0.0), "axis -x, d.y = -0 at (%g, %g)" % (y, z))
        for x in (0.0, 2.0, 3.0):
            add((x, 9.0, -4.5), (0.0, -1.0, 0.0), "axis -y through a box and a sphere column at x = %g" % x)
        # the same families off the lattice planes: zero direction components without a coincidence
        for x in (0.1234, 0.8125, 1.3, 2.2, 5.7, 6.6):
            for y in (0.3, 0.61, 2.2, 4.4, 6.3):
                add((x, y, 5.0), (0.0, 0.0, -1.0), "axis -z off the planes at (%g, %g)" % (x, y))
                add((x, y, -12.0), (-0.0, 0.0, 2.5), "axis +z, d.x = -0, from behind at (%g, %g)" % (x, y))
        for y in (0.3, 2.2, 4.4):
            for z in (-4.3, -6.2, -7.9):
                add((-5.0, y, z), (1.0, 0.0, -0.0), "axis +x off the planes at (%g, %g)" % (y, z))
        for x in (0.2, 2.3, 6.4):
            add((x, 9.0, -4.3), (0.0, -1.0, 0.0), "axis -y off the planes at x = %g" % x)
        rng = np.random.default_rng(11)
        for k in range(40):
            o = rng.uniform([-2.0, -1.0, 2.0], [8.0, 7.0, 6.0])
            t = rng.uniform([0.0, 0.0, -9.0], [6.0, 5.0, -3.0])
            add(o, t - o, "random %d" % k)
        add((3.0, 1.0, -6.0), (0.2, 0.3, -0.5), "origin inside the BVH's bounds, outside every primitive")
        add((2.0, 0.0, -4.0), (0.0, 0.1, -1.0), "origin inside a sphere of the BVH")
        # the transformed copies (quirk Q10: Rotate re-faces the normal against the object-space ray, Translate against the moved one)
        for k in range(12):
            o = rng.uniform([16.0, -1.0, 3.0], [26.0, 4.0, 6.0])
            t = rng.uniform([19.0, 0.0, -9.0], [25.0, 3.0, -5.0])
            add(o, t - o, "rotated + translated BVH %d" % k)
        for k in range(8):
            o = rng.uniform([27.0, -1.0, 0.0], [34.0, 5.0, 4.0])
            add(o, np.array([30.8, 1.6, -5.2]) + rng.uniform(-0.9, 0.9, 3) - o, "rotated + translated block %d" % k)
        add((30.8, 1.6, -5.2), (0.3, 0.1, 1.0), "origin inside the rotated block")
        for k in range(4):
            add((40.0 + 0.3 * k, 0.2 * k, 3.0), (0.0, 0.0, -1.0), "translated sphere %d" % k)
            add((0.3 * k, 0.2 * k, 3.0), (0.0, 0.0, -1.0), "rotated sphere (glass) %d" % k)
    elif name == "media":
        add((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), "opaque sphere hides the smoke: t1 >= t2, no draw for it")
        for k in range(12):
            add((0.9 + 0.05 * k, 0.1, 0.0), (0.0, 0.0, -1.0), "through the smoke %d" % k)
        add((0.0, 0.9, -8.0), (0.0, 0.1, -1.0), "origin inside the smoke: t1 clamped to t_min")
        add((0.0, 0.9, -8.0), (0.0, 0.0, -0.001), "origin inside the smoke, very short direction (ray_scale)")
        add((2.0, 0.0, 0.0), (0.0, 0.0, -1.0), "tangent to the smoke's boundary")
        add((2.5, 0.0, 0.0), (0.0, 0.0, -1.0), "past the smoke")
        for k in range(10):
            add((10.0 + 0.17 * k - 0.8, 0.3, 0.0), (0.0, 0.01, -1.0), "through the fog block (rotated, translated) %d" % k)
        add((10.0, 0.0, -8.0), (0.3, 0.2, 1.0), "origin inside the fog block")
        add((0.0, 0.0, -20.0), (0.0, 0.0, 1.0), "from behind: far sphere first in space, smoke first in the list")
        add((0.0, 60.0, 0.0), (0.0, 1.0, 0.0), "outside the global fog, leaving")
        add((0.0, 60.0, 0.0), (0.0, -1.0, 0.0), "outside the global fog, entering")
    return np.array(R, dtype=np.float64), L


def psd_for(n):
    """Philox (pixel, sample, draw index) per ray: draw 4 is where a primary ray stands after the camera's block; a few rays start
    mid-block (draw 5..7), as a path does after a medium draw."""
    psd = np.zeros((n, 3), dtype=np.uint32)
    psd[:, 0] = 1000 + 7 * np.arange(n)
    psd[:, 1] = np.arange(n) % 5
    psd[:, 2] = 4
    psd[::7, 2] = 8
    return psd


# ---- explicit boxes for the culling test (bhv.rs:168-221 first) -------------------------------------------------------
def aabb_cases():
    boxes, rays, tmax, labels = [], [], [], []

    def add(b, o, d, tm, label):
        boxes.append(b); rays.append(list(o) + list(d)); tmax.append(tm); labels.append(label)

    inf = float("inf")
    for b, tag in (((1, 1, 1, 2, 2, 2), "minmax"), ((2, 2, 2, 1, 1, 1), "maxmin")):
        add(b, (0, 0, 0), (1, 1, 1), inf, "bhv.rs test_minmax %s" % tag)                              # hit
        add(b, (1.0001, 0, 1.0001), (0, 1, 0), inf, "bhv.rs test_edge_parallel %s" % tag)             # hit
        add(b, (0.99999, 0, 0.9999), (0, 1, 0), inf, "bhv.rs test_edge_parallel_outside %s" % tag)    # miss
        add(b, (1.5, 0, 1.0001), (0, 3, 0), inf, "bhv.rs test_face_parallel %s" % tag)                # hit
        add(b, (1.5, 0, 0.9999), (0, 3, 0), inf, "bhv.rs test_face_parallel_outside %s" % tag)        # miss
    b = (1, 1, 1, 2, 2, 2)
    add(b, (1.5, 1.5, 1.5), (0.3, -0.2, 0.9), inf, "origin inside")
    add(b, (1.5, 1.5, 1.5), (0, 0, 1), inf, "origin inside, axis-parallel")
    add(b, (1.5, 1.5, 1.5), (0.0, -0.0, -1.0), inf, "origin inside, negative zeros")
    add(b, (1.5, 1.5, 5.0), (0, 0, 1), inf, "behind the origin")
    add(b, (1.5, 1.5, -5.0), (0, 0, 1), 5.9, "closest-so-far in front of the box")
    add(b, (1.5, 1.5, -5.0), (0, 0, 1), 6.0, "closest-so-far exactly at the entry face (tmax <= tmin rejects)")
    add(b, (1.5, 1.5, -5.0), (0, 0, 1), 6.5, "closest-so-far inside the box")
    add(b, (1.0, 1.5, -5.0), (0, 0, 1), inf, "along the face plane x = 1 (0/0 on that axis)")
    add(b, (1.0, 1.0, -5.0), (0, 0, 1), inf, "along the edge x = 1, y = 1")
    add(b, (0, 0, 0), (1, 1, 1.0000001), inf, "through the corner, almost")
    add((1, 1, 1, 2, 2, 1), (1.5, 1.5, -5.0), (0, 0, 1), inf, "zero thickness never hits (quirk Q5)")
    add(b, (1000.0, 1000.5, 0.0), (-998.5, -999.0, 1.5), inf, "origin 1000 units away (the absolute error term of an f32 slab)")
    add(b, (1e6 + 1.5, 1.5, 1.5), (-1.0, 0.0, 0.0), inf, "origin 1e6 units away, axis-parallel")
    rng = np.random.default_rng(5)
    for k in range(200):  # rays aimed at or just past the silhouette of random boxes from far and near
        lo = rng.uniform(-50, 50, 3); size = rng.uniform(0.1, 20, 3)
        box = tuple(lo) + tuple(lo + size)
        o = lo + size * 0.5 + unit(rng.normal(size=3)) * rng.choice([3.0, 30.0, 1000.0]) * np.linalg.norm(size)
        target = lo + size * rng.choice([0.0, 1.0, rng.uniform()], size=3) + rng.choice([0.0, 1e-9, -1e-9, 1e-4, -1e-4], size=3)
        add(box, o, target - o, inf, "silhouette %d" % k)
    return np.array(boxes, dtype=np.float64), np.array(rays, dtype=np.float64), np.array(tmax, dtype=np.float64), labels


def texture_cases():
    """(u, v, p) for the image texture (clamps of image_texture.rs:17-23: u = 1 and v = 0 land on the last column / row) and points
    for the checker / noise textures, negative lattice coordinates included."""
    uv = []
    for u in (-0.5, 0.0, 1e-9, 1.0 / 7.0, 0.5, 6.0 / 7.0, 1.0 - 1e-12, 1.0, 1.5):
        for v in (-0.5, 0.0, 0.2, 0.5, 0.8, 1.0 - 1e-12, 1.0, 2.0):
            uv.append((u, v, 0.0, 0.0, 0.0))
    pts = []
    rng = np.random.default_rng(3)
    for p in ((0, 0, 0), (-0.5, -0.5, -0.5), (-1.0, -2.0, -3.0), (1023.5, 1024.0, 1024.5), (-1023.5, -1024.0, -1025.5), (1e-9, -1e-9, 0.5),
              (3.999999999, -3.999999999, 7.0), (12345.678, -9876.543, 0.001)):
        pts.append((0.0, 0.0) + tuple(float(x) for x in p))
    for _ in range(40):
        pts.append((0.0, 0.0) + tuple(rng.uniform(-300, 300, 3)))
    return np.array(uv, dtype=np.float64), np.array(pts, dtype=np.float64)


def camera_cases(mlrt):
    """Two cameras (pinhole, thin lens) over a 37 x 23 frame: corners, centre, a few interior pixels, several samples."""
    cams = [mlrt.camera_new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, 37.0 / 23.0, 0.0, 10.0),
            mlrt.camera_new((478.0, 278.0, -600.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 37.0 / 23.0, 2.5, 500.0)]
    ijs = []
    for (i, j) in ((0, 0), (36, 0), (0, 22), (36, 22), (18, 11), (5, 17), (30, 3)):
        for s in (0, 1, 499):
            ijs.append((i, j, s))
    return cams, np.array(ijs, dtype=np.int32), 37, 23


def to_rgb_cases():
    vals = [0.0, -0.0, 1e-300, 1e-12, 1e-9, 0.001, 0.25, 0.5, 0.99, 0.99999998, 1.0, 1.0000001, 2.0, 100.0, 1048575.0, 1048576.0, 2e6, 1e30,
            float("inf"), -1e-9, -1.0, -float("inf"), float("nan")]
    return np.array(vals, dtype=np.float64)

// csrc/mlrt_device.hpp - device data layout shared by the host-side packer (mlrt_api.cpp) and the HIP kernel.
//
// The packed scene is ONE position-independent blob (indices, no pointers) so that it can be uploaded to HBM
// once and staged into LDS by every workgroup with coalesced 16-byte loads:
//
//   [ entries | materials | textures | sphere index | perlin ranvec | perlin perm ]
//
// entries: the WHOLE world as one linear, threaded array that a ray walks front to back with a single cursor and
// no stack.  The top-level HittableList (hittable.rs:37-69) becomes the order of the entries; every BVH
// (bhv.rs:109-112) is linearised in place in depth-first (pre-order) order with "skip" links: an inner node whose
// box is hit continues at entry+1, otherwise at its skip link; a leaf carries its primitive INLINE (sphere / box /
// rect parameters in the 6 reals of the entry), so a primitive test needs no second dependent load.  Because the
// order is the reference's own left-then-right recursion (bhv.rs:147-165) and list order, with the hit distance
// shrinking as primitives are accepted, the walk performs exactly the box and primitive tests the reference
// performs, in the same order (so ConstantMedium draws happen under the same conditions, quirk Q9).  Leaves are
// not box-tested, as in the reference (quirk Q4), unless the packer wraps them (fast build option, same result).
// Translate/Rotate wrappers (transforms.rs) become XFORM ... XFORM_END brackets that switch the cursor's ray
// between world and object space; a ConstantMedium (volumes.rs) is a MEDIUM entry followed by its boundary.
#pragma once
#include <cstdint>

namespace mlrt {

enum : int32_t {
    TAG_INNER = 0, TAG_SPHERE = 1, TAG_BOX = 2, TAG_RECT = 3, TAG_EMPTY = 4, TAG_MEDIUM = 5, TAG_XFORM = 6, TAG_XFORM_END = 7,
    TAG_END = 8,  // the sentinel after the last entry: every walk ends on it (its index is KParams::n_nodes)
    TAG_KIND_MASK = 0xF
};
// TAG_RECT entries carry their axes in the tag: (a0 << 4) | (a1 << 6) | (aplane << 8)
// TAG_XFORM entries carry XF_TRANSLATE << 4 | XF_ROTATE << 4 | axis << 8
enum : int32_t { XF_TRANSLATE = 1, XF_ROTATE = 2 };

// Two entry layouts.
//  * f64 entries (64 B; the strict build and the cull_f64 product build): everything inline, as listed below.
//  * f32 entries (32 B; the fast build and the product f64 build): the bounds of inner entries are f32, moved OUTWARD -
//    they only decide what is skipped.  Fast build: the primitives' parameters are inline f32 values, as in the list below.
//    Product f64 build: the f64 parameters of a primitive live in PAYLOAD slots (raw 32-byte slots of doubles) right behind
//    its entry (a Block entry's own v[] holds outward f32 bounds: the step culls with them):
//        sphere 1 slot: cx, cy, cz, radius | box 2 slots: min xyz, max xyz |
//        rect 2 slots: a0_v0, a0_v1, a1_v0, a1_v1, plane | medium 1 slot: neg_inv_density | xform 2 slots: sin, cos, offset xyz
//    Skip links are absolute entry indices and already account for payload slots; entry_slots() is the one rule both the
//    packer and the kernel use to step over an entry.
template <typename R>
struct alignas(sizeof(R) == 4 ? 16 : 64) DNode {
    // inner    : v = bounds min xyz, max xyz              link = skip (absolute entry index)
    // sphere   : v = cx, cy, cz, radius (f64 entries)     link = material
    // box      : v = min xyz, max xyz (shapes.rs Block)   link = material
    // rect     : v = a0_v0, a0_v1, a1_v0, a1_v1, plane    link = material
    // medium   : v = neg_inv_density                      link = phase (Isotropic) material; boundary = next entry
    // xform    : v = sin, cos, offset xyz                 (Translate(offset, Rotate(axis, angle, .)))
    R v[6];
    int32_t link;
    int32_t tag;
};
static_assert(sizeof(DNode<float>) == 32, "f32 entry must be 32 bytes (two ds_read_b128)");
static_assert(sizeof(DNode<double>) == 64, "f64 entry is one 64-byte line");

// ---- the pair walk (sphere-only scenes, builds with f32 culling) ---------------------------------------------------
// The second device structure: a binary BVH whose node holds the boxes of BOTH its children (bhv.rs:147-165: the
// reference's Node::hit tests its own box, then recurses left, then right with tmax shrunk; here a visit of a node
// decides both children at once, enters the nearer one and pushes the other on a short per-lane stack in LDS).  One
// visit = one LDS round trip (four ds_read_b128) where the threaded list pays one round trip per box.  A child
// reference is a pair index (>= 0), a leaf (kPairLeaf + primitive index, negative) or - only ever at the bottom of a
// lane's stack - kPairEnd.  References are 16-bit quantities (the stack stores them as such), sign-extended.
// Leaves are primitive indices into `prims` (4 reals: centre, radius; f64 in the product build) in the REFERENCE's
// visiting order (top-level list order, then each BVH's leaves left to right): the sphere test accepts a hit at the
// same distance as the closest so far only from a primitive of higher index - what the reference's `t <= closest_so_far`
// (shapes.rs:66-73 with hittable.rs:58-66) does when it visits them in that order - so the result does not depend on the
// order the device visits them in.
struct alignas(16) DPair {
    float a[6];  // child A: min xyz, max xyz (f32, moved outward like the threaded entries)
    float b[6];  // child B
    int32_t ref_a, ref_b;
    int32_t pad[2];
};
static_assert(sizeof(DPair) == 64, "a pair node is four ds_read_b128");
enum : int32_t { kPairEnd = -1, kPairLeaf = -32768, kPairMaxPrims = 32767, kPairMaxNodes = 32768 };

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define MLRT_HD __host__ __device__
#else
#define MLRT_HD
#endif
// Slots an entry occupies.  cull32 && payload_all: the product f64 build (every primitive's parameters in payload slots).
MLRT_HD inline int32_t entry_slots(int32_t kind, bool cull32, bool payload_all) {
    if (!cull32 || !payload_all) return 1;
    if (kind == TAG_SPHERE) return 2;
    if (kind == TAG_BOX || kind == TAG_RECT || kind == TAG_XFORM) return 3;
    if (kind == TAG_MEDIUM) return 2;
    return 1;
}

enum : int32_t { MATFLAG_NEEDS_UV = 1 << 8, MATFLAG_NOISE = 1 << 9 };  // NOISE: the material's own texture is a NoiseTexture

// Scene features a kernel variant must support (the packer records which ones a scene uses; the launcher picks the
// leanest compiled variant that covers them: fewer registers -> more waves per SIMD, less dead code in the hot loop).
enum : uint32_t {
    FEAT_BOX = 1, FEAT_RECT = 2, FEAT_XFORM = 4, FEAT_MEDIUM = 8, FEAT_TEX_IMAGE = 16, FEAT_TEX_NOISE = 32, FEAT_TEX_CHECKER = 64,
    FEAT_ALL = 127
};  // set on the material kind when its texture chain samples an image

template <typename R>
struct DMaterial {
    int32_t kind;  // MLRT_MAT_* | MATFLAG_*
    int32_t tex;
    R albedo[3];
    R param;
};

template <typename R>
struct DTexture {
    int32_t kind, a, b, _pad;  // IMAGE: a = image; NOISE: a = perlin table; CHECKER: a = odd, b = even
    R color[3];
    R scale;
};

struct DImage {
    const uint8_t* rgb;
    int32_t width, height;
};

constexpr int kMaxTailPasses = 8;
constexpr int kMaxQueues = 64;

template <typename R>
struct KParams {
    const unsigned char* blob;  // packed scene in HBM
    int32_t blob_bytes;
    int32_t lds_blob_bytes;  // bytes of the blob a workgroup stages in LDS (the whole blob, or up to the Perlin tables: f64 builds)
    int32_t off_nodes, off_materials, off_textures, off_perlin_vec, off_perlin_perm;
    int32_t off_sphere_index, n_sphere_index;  // u16 entry indices of the sphere entries, in entry order (the candidate pre-pass scans these)
    int32_t n_nodes;
    // pair walk (DPair): off_nodes is then the offset of the pair nodes, n_nodes their number
    int32_t pair_walk;                 // 1: the blob holds [pairs | prims | prim_mat | materials], the kernel is the PAIR variant
    int32_t off_prims, off_prim_mat;   // prims: [n_prims][4] reals (centre, radius); prim_mat: [n_prims] u16 material indices
    int32_t n_prims;
    int32_t stack_levels;              // rows of the per-lane LDS stack (the longest chain of inner nodes from the root)
    const DImage* images;
    // Camera (camera.rs:3-12): f64 in both builds, primary rays are generated exactly as the reference does
    double cam_origin[3], cam_llc[3], cam_horizontal[3], cam_vertical[3], cam_u[3], cam_v[3];
    double lens_radius;
    double inv_wm1, inv_hm1;  // 1 / (width - 1), 1 / (height - 1)  (cam_origin ... inv_hm1 stay contiguous: the kernel copies them as 21 doubles)
    int32_t has_lens;    // bit 0: lens_radius != 0; bit 1: ray_table is set (a ray frame); bit 2: MLRT_TRACER_SINGLE_LIGHT
    int32_t width, height, spp, max_depth;
    uint32_t seed_lo, seed_hi;
    int32_t bg_kind;
    R bg_top[3], bg_bottom[3];
    // Work decomposition: a tile = one pixel group (pix_per_tile consecutive pixels of one row) x one chunk of its
    // samples.  The queue is chunk-major - pass p hands out chunk p of EVERY pixel group before pass p + 1 starts - and
    // the chunks taper: n_big passes of `big` samples, then the passes of tail_s0 (sample offsets; tail_s0[n_tail] = spp),
    // shrinking to one path per lane.  Tile costs are heavy-tailed (one 500-sample tile on a glass sphere of `random`
    // takes 2.8 ms, the average 0.05 ms), so the launch must END on small tiles: its tail is the longest tile taken late.
    const int32_t* rows;  // device array of row indices or nullptr (= identity)
    int32_t n_rows, pix_per_tile, tiles_per_row, n_groups, n_big, big, n_tail, n_tiles;
    int32_t tail_s0[kMaxTailPasses + 1];
    // n / n_groups and n / tiles_per_row for n < 2^31 as mulhi(n, magic) >> shift (shift < 0: divisor 1): a scalar integer
    // division is ~35 instructions, and a wave opening a tile executes them with all 64 lanes waiting
    uint32_t n_groups_magic, tiles_per_row_magic;
    int32_t n_groups_shift, tiles_per_row_shift;
    int32_t n_banks, bank_values;  // the wave's LDS accumulator area: n_banks banks of bank_values (= 3 * pix_per_tile) sums
    unsigned long long* acc_global;  // [n_rows][W][3] 2^-32 fixed-point colour sums (integer adds: order independent)
    int32_t* rgb_out;     // [n_rows][W][3], written by the resolve kernel
    double* accum_out;    // optional [n_rows][W][3]
    unsigned int* tile_counter;  // [n_queues] counters, 64 bytes apart
    int32_t n_queues;            // power of two <= kMaxQueues
    unsigned int* status;        // [4 + 64]: watchdog record (waves that gave up, workgroup, wave, -, then the lanes' phases), zero when all is well
    uint32_t stall_cap;          // kernel-loop iterations a wave may spend without finishing a path before it gives up
    unsigned long long* stats;  // optional [8]: iterations, trav lane-steps, special batches, special lanes, shade batches, shade lanes, samples, tiles
    const uint16_t* primary_lists;  // or nullptr; sphere-only scene + pinhole camera: [n_groups][16] per-pixel-group candidate records (mlrt_primary_lists_kernel):
                                    // a group's primary rays test those entries instead of walking the BVH
    int32_t debug_pixel, debug_sample;  // >= 0: printf the path of that (pixel, sample) from the kernel (MLRT_DEBUG_PIXEL=pixel,sample)
    // Test instrument ("ray frames", mlrt_debug_ray_frame): when not null, sample s of pixel p starts on ray_table[p * spp + s] (origin xyz,
    // direction xyz) instead of the camera's ray - the sample's Philox block 0 is consumed as always.  Read from the kernel-argument
    // segment at its one point of use (gen_camera_ray); kept last so that no other field moves.
    const double* ray_table;
    // MLRT_TRACER_SINGLE_LIGHT (has_lens bit 2, all-features variant only): light_source xyz and intensity, read from the kernel-argument
    // segment where a hit is shaded
    double single_light[4];
};

}  // namespace mlrt

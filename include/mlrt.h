/* include/mlrt.h - C ABI of libmlrt_hip.so: the MI355X (gfx950) render path.
 *
 * This is the drop-in boundary for ONE hot path of the reference CPU path tracer
 * (mu-lambda-raytracer): the call pair
 *     Renderer::new_with_rng(camera, world, background, params, RecursiveRayTracer{max_depth}, rngator)
 *         (/root/reference/src/main.rs:147-155, signature src/raytrace.rs:151-160)
 *     rt.render(logger) -> Vec<Vec<RGB>>        (src/main.rs:157, src/raytrace.rs:172-186)
 * The reference has no FFI surface of its own (the seams are Rust traits and `world` is an opaque
 * Box<dyn Hittable>, src/worlds.rs:18), so the host side passes a FLATTENED, plain-old-data scene description
 * (the same objects the world builders of src/worlds.rs create, in the same order) plus the 7 Camera fields
 * (src/camera.rs:3-12) and RenderingParams (src/raytrace.rs:50-55).  INTEGRATION.md shows the Rust
 * `extern "C"` binding a maintainer would add at src/main.rs:157.
 *
 * Conventions (mirroring the seam): every pointer is owned by the caller before and after a call; the
 * library copies what it needs to the device.  Errors: non-zero return + mlrt_last_error() (thread-local
 * message); no exception crosses the ABI.  All geometry is f64 at the ABI, as in the reference (src/vec.rs:5-8).
 * There is NO CPU fallback: every compute entry point fails with MLRT_ERR_NO_DEVICE when no HIP device exists.
 * Threads: one call at a time per context (the reference's render() is called once, from the main thread, and blocks:
 * src/main.rs:157); different contexts may be used from different threads at the same time.
 */
#ifndef MLRT_H
#define MLRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLRT_VERSION_MAJOR 0
#define MLRT_VERSION_MINOR 1

enum {
    MLRT_OK = 0,
    MLRT_ERR_INVALID = 1,   /* bad argument / malformed scene (the reference would panic) */
    MLRT_ERR_NO_DEVICE = 2, /* no HIP device: the product path never falls back to the CPU */
    MLRT_ERR_HIP = 3,       /* a HIP runtime call failed */
    MLRT_ERR_UNSUPPORTED = 4
};

/* src/materials.rs, src/volumes.rs:67-83 */
enum { MLRT_MAT_LAMBERTIAN = 0, MLRT_MAT_METAL = 1, MLRT_MAT_DIELECTRIC = 2, MLRT_MAT_DIFFUSE_LIGHT = 3, MLRT_MAT_ISOTROPIC = 4 };
/* src/textures.rs, src/image_texture.rs */
enum { MLRT_TEX_SOLID = 0, MLRT_TEX_IMAGE = 1, MLRT_TEX_NOISE = 2, MLRT_TEX_CHECKER = 3 };
/* src/shapes.rs: Sphere, Block (box), XY/XZ/YZRect */
enum { MLRT_PRIM_SPHERE = 0, MLRT_PRIM_BOX = 1, MLRT_PRIM_RECT = 2 };
/* entries of the top-level HittableList (src/hittable.rs:37-69), in reference order */
enum { MLRT_OBJ_PRIM = 0, MLRT_OBJ_BVH = 1, MLRT_OBJ_MEDIUM = 2 };
/* src/raytrace.rs:8-48 */
enum { MLRT_BG_BLACK = 0, MLRT_BG_GRADIENT = 1 };
/* arithmetic of the device path */
enum { MLRT_PRECISION_FP32 = 0,       /* fast path: f32 boxes/rects/shading, f64 ray state and sphere quadratic */
       MLRT_PRECISION_FP64 = 1,       /* reference precision (src/vec.rs:5-8): every operation that decides a sample's value is f64
                                       * in the reference's operation order; the culling structure (SAH BVH, conservative
                                       * bounding-box tests, primary candidate lists) is the device's own and cannot change a
                                       * closest hit: colour sums bit-equal to MLRT_PRECISION_FP64_STRICT (tested) */
       MLRT_PRECISION_FP64_STRICT = 2 /* as FP64, plus the reference's BVH topology and AABB::hit arithmetic (bhv.rs:29-42,
                                       * 122-165): exactly the tests the reference performs.  A test instrument (slow). */ };

typedef struct mlrt_material {
    int32_t kind;     /* MLRT_MAT_* */
    int32_t texture;  /* Lambertian albedo / DiffuseLight emit / Isotropic albedo: index into textures; else -1 */
    double albedo[3]; /* Metal albedo (materials.rs:37-40) */
    double param;     /* Metal fuzz | Dielectric index_of_refraction */
} mlrt_material;

typedef struct mlrt_texture {
    int32_t kind;    /* MLRT_TEX_* */
    int32_t index;   /* IMAGE: index into images; NOISE: index into perlins; CHECKER: odd texture */
    int32_t index2;  /* CHECKER: even texture */
    int32_t _pad;
    double color[3]; /* SOLID (textures.rs:8-26) */
    double scale;    /* NOISE (textures.rs:151-167) */
} mlrt_texture;

typedef struct mlrt_image {  /* image_texture.rs: decoded RGB8, row 0 = top */
    int32_t width, height;
    const uint8_t* rgb;
} mlrt_image;

typedef struct mlrt_perlin {  /* textures.rs:51-74: tables as drawn by Perlin::new */
    const double* ranvec;     /* [1024][3] */
    const int32_t* perm;      /* [3][1024]: perm_x, perm_y, perm_z */
} mlrt_perlin;

typedef struct mlrt_sphere {  /* shapes.rs:25-42 */
    double center[3];
    double radius;
    int32_t material;
    int32_t _pad;
} mlrt_sphere;

typedef struct mlrt_box {  /* shapes.rs:166-198 Block: six AARects sharing one material.  min <= max componentwise (the
                            * reference's Block::new(p0, p1) with reversed corners yields degenerate sides, aarects.rs:39: rejected).
                            * A hit carries the u, v of the face's AARect - its two in-plane axes in axis order, each normalised over
                            * the face (shapes.rs:172-192, aarects.rs:56-57) - as the reference's does. */
    double min[3], max[3];
    int32_t material;
    int32_t _pad;
} mlrt_box;

typedef struct mlrt_rect {  /* aarects.rs:19-43.  A rect that is a LEAF OF A BVH (no reference world builds one) is the one place the product
                             * modes follow the intended geometry and not the reference: AARect::bounding_box (aarects.rs:66-76) sets the
                             * box's maximum to (a0_v0, a1_v0) - a point in the rect's plane - so the reference's inner node above it
                             * encloses one corner only and culls most rays that would hit the rect.  MLRT_PRECISION_FP32 / _FP64 bound the
                             * rect by its true extent (their BVHs are rebuilt from the primitives); MLRT_PRECISION_FP64_STRICT walks the
                             * caller's bvh_nodes with the caller's bounds and so does whatever those say. */
    int32_t a0, a1;         /* in-plane axes (0=X,1=Y,2=Z); the plane axis is 3-a0-a1 */
    double a0_v0, a0_v1, a1_v0, a1_v1, plane;
    int32_t material;
    int32_t _pad;
} mlrt_rect;

typedef struct mlrt_prim_ref {
    int32_t kind;  /* MLRT_PRIM_* */
    int32_t index; /* into spheres / boxes / rects */
} mlrt_prim_ref;

/* One inner node of a BHV (bhv.rs:109-112) in the topology the reference's Node::new builds
 * (random axis, stable sort, median split; bhv.rs:122-145).  A child >= 0 is an inner-node index into
 * bvh_nodes; a child < 0 is a leaf: ~child indexes prim_refs. */
typedef struct mlrt_bvh_node {
    double bounds_min[3], bounds_max[3];
    int32_t left, right;
} mlrt_bvh_node;

#define MLRT_BVH_EMPTY INT32_MIN /* BHV over zero shapes: Leaf(Empty), bhv.rs:124 */
typedef struct mlrt_bvh {
    int32_t root; /* child encoding as above, or MLRT_BVH_EMPTY */
    int32_t _pad;
} mlrt_bvh;

/* Translate(offset, Rotate(axis, angle, x)) (transforms.rs:20-49, 51-148); either part may be absent. */
typedef struct mlrt_transform {
    int32_t has_rotate, axis;
    double sin_theta, cos_theta;
    int32_t has_translate, _pad;
    double offset[3];
} mlrt_transform;

typedef struct mlrt_object {
    int32_t kind;           /* MLRT_OBJ_* */
    int32_t bvh;            /* BVH: index into bvhs */
    mlrt_prim_ref prim;     /* PRIM: the primitive; MEDIUM: the boundary (volumes.rs:7-11) */
    int32_t transform;      /* -1 or index into transforms */
    int32_t phase_material; /* MEDIUM: Isotropic material */
    double neg_inv_density; /* MEDIUM: -1/density (volumes.rs:15) */
} mlrt_object;

typedef struct mlrt_scene {
    int32_t n_spheres, n_boxes, n_rects, n_prim_refs, n_bvh_nodes, n_bvhs, n_transforms, n_objects, n_materials, n_textures,
        n_images, n_perlins;
    const mlrt_sphere* spheres;
    const mlrt_box* boxes;
    const mlrt_rect* rects;
    const mlrt_prim_ref* prim_refs;
    const mlrt_bvh_node* bvh_nodes;
    const mlrt_bvh* bvhs;
    const mlrt_transform* transforms;
    const mlrt_object* objects; /* the world: hit like a HittableList, in this order */
    const mlrt_material* materials;
    const mlrt_texture* textures;
    const mlrt_image* images;
    const mlrt_perlin* perlins;
} mlrt_scene;

typedef struct mlrt_camera { /* the private fields of src/camera.rs:3-12, as Camera::new computes them */
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    double lens_radius;
} mlrt_camera;

typedef struct mlrt_params {
    int32_t image_width, image_height, samples_per_pixel; /* RenderingParams, raytrace.rs:50-55 */
    int32_t max_depth;                                    /* RecursiveRayTracer, raytrace.rs:74-76 */
    uint64_t seed;                                        /* SeedableRngator, rngator.rs:17-31.  The device does not replay the
                                                           * reference's per-row Pcg64 streams (a row's stream is consumed pixel by
                                                           * pixel: inherently serial): every (pixel, sample) owns a Philox4x32-10
                                                           * stream keyed by this seed - same distributions, image deterministic and
                                                           * independent of how rows are spread over GPUs ("RNG contract") */
    int32_t background_kind;                              /* MLRT_BG_* */
    int32_t tracer;                                       /* MLRT_TRACER_*: the Renderer's RayTracer (raytrace.rs:70-72, 137-141).  0 = the
                                                           * path tracer, as main.rs:150-152 constructs it */
    double background_top[3], background_bottom[3]; /* GradientBackground, raytrace.rs:12-35 */
    double light_source[3];                         /* MLRT_TRACER_SINGLE_LIGHT: SingleLightSourceRayTracer::light_source, */
    double light_intensity;                         /*   ::intensity (raytrace.rs:110-113; main.rs:153 has (14, 3, 3) and 1.0) */
} mlrt_params;
/* RecursiveRayTracer { max_depth } (raytrace.rs:74-108) */
#define MLRT_TRACER_RECURSIVE 0
/* SingleLightSourceRayTracer { light_source, intensity } (raytrace.rs:110-135): one bounce.  A hit whose material scatters is shaded
 * attenuation * intensity * max(l.n, 0) + 0.5 * intensity * max(h.n, 0)^100 towards the point light (the scattered ray is dropped, its
 * RNG draws are made); one that does not scatter emits; a miss is the background.  max_depth is not used.  Needs a context created with
 * mlrt_options.single_light_tracer = 1 (mlrt_render / mlrt_render_multi set it themselves); MLRT_ERR_UNSUPPORTED otherwise. */
#define MLRT_TRACER_SINGLE_LIGHT 1

/* Called on the calling thread after each finished band of rows (coarser than the reference's per-row
 * logger, raytrace.rs:182; main.rs:157-173 only derives a throttled percentage from it). */
typedef void (*mlrt_progress_fn)(int32_t rows_done, int32_t rows_total, void* user);

/* Tuning and diagnostic switches of the device path.  None of them may change an image in the f64 modes (the
 * culling structure and the tile geometry are result-neutral by construction; tests/test_gpu_properties.py checks
 * each switch).  MLRT_PRECISION_FP64_STRICT ignores sah / leaf_wrap / primary_lists / lds_scene. */
typedef struct mlrt_options {
    int32_t sah;           /* 1: surface-area-heuristic device BVH; 0: the reference's topology (bhv.rs:122-145) */
    int32_t leaf_wrap;     /* 1: bounding-box entries in front of sphere leaves and of top-level objects */
    int32_t primary_lists; /* 1: per-pixel-group candidate lists for primary rays (sphere-only scenes) */
    int32_t lds_scene;     /* 1: stage the packed scene in LDS when it fits */
    int32_t force_full;    /* 1: always the FULL kernel variant */
    int32_t cull_f64;      /* MLRT_PRECISION_FP64 only. 0 (default): the culling entries are f32, moved outward with a proven error
                            * bound, the primitives' f64 parameters sit behind them; 1: f64 entries, culling arithmetic in f64 too
                            * (twice the LDS traffic per traversal step; same colour sums, tested) */
    int32_t tile_samples;  /* samples per big tile, >= 64 (default 512) */
    int32_t tile_queues;   /* tile sub-queues, a power of two <= 64 (default 8) */
    int32_t n_tail;        /* -1: adaptive tapering passes; else the number of entries of tail[] */
    int32_t tail[8];       /* samples of the tapering passes, counted from the end of a pixel's sample range */
    int32_t stats;         /* debug library only: in-kernel counters */
    int32_t debug_pixel, debug_sample; /* debug library only: print the path of one (pixel, sample); -1 = off */
    int32_t stall_cap;     /* debug library only (libmlrt_hip_debug.so; ignored by the release library): kernel-loop iterations a wave may
                            * spend without finishing a path before the countdown watchdog makes it give up; 0 = default (2^24) */
    int32_t pair_walk;     /* 1 (default): sphere-only scenes are walked through a binary BVH whose nodes hold both children's boxes, nearer
                            * child first, with a per-lane stack in LDS (FP32 and FP64 modes); 0: the threaded list for them too */
    int32_t split_roles;   /* 1: sphere-only scenes on the pair structure are rendered by the split-role kernel - traversal waves and shading
                            * waves of a workgroup handing paths to each other through two queues in LDS - when its queues fit the LDS next to
                            * the scene; 0: every wave does everything.  Same image either way. */
    int32_t single_light_tracer; /* 1: the context can also render with MLRT_TRACER_SINGLE_LIGHT (it then runs the all-features kernel
                            * variant, the one that carries that tracer's shading, whatever the scene holds); 0 (default): path tracer only */
} mlrt_options;
void mlrt_options_default(mlrt_options* out);
/* The defaults overridden by the MLRT_* environment variables (MLRT_NO_SAH, MLRT_NO_LEAF_WRAP, MLRT_NO_PRIMARY_LISTS,
 * MLRT_NO_LDS, MLRT_FORCE_FULL, MLRT_CULL_F64, MLRT_NO_PAIR_WALK, MLRT_TILE_SAMPLES, MLRT_TILE_QUEUES, MLRT_TILE_TAIL, MLRT_STATS, MLRT_DEBUG_PIXEL, MLRT_STALL_CAP):
 * the only place the library reads its environment. */
void mlrt_options_from_env(mlrt_options* out);

typedef struct mlrt_context mlrt_context;

void mlrt_version(int32_t* major, int32_t* minor);
const char* mlrt_last_error(void);
int32_t mlrt_device_count(void);
/* Optional.  A process's first operation on a GPU brings the HIP runtime and the device's queues up - about 0.2 s on this stack, whatever
 * the operation is - and would otherwise be spent inside the first mlrt_context_create / render of the process.  A caller that times
 * "renderer construction + render" (main.rs:145-174) the way the reference does, with the process already running, calls this first: one
 * tiny allocation, fill and wait on `device`.  Leaves the caller's current device as it was. */
int32_t mlrt_device_warm_up(int32_t device);

/* Replaces Renderer::new_with_rng: validates the scene, packs it into the device layout (threaded BVH,
 * see DESIGN.md) and uploads it to HBM of `device`.  `precision` is MLRT_PRECISION_*.
 * Validation happens before any device call and returns MLRT_ERR_INVALID (message in mlrt_last_error) for: no objects; a NULL
 * array with a positive count; any index out of range (materials, textures, images, Perlin tables, primitives, BVH children,
 * transforms); a BVH that is not a tree; geometry that is not finite or of magnitude 2^100 and more (spheres, boxes, rects, BVH bounds,
 * transform offsets; the camera of a render call likewise), a non-finite transform angle or medium density;
 * a box with min > max; rect axes other than (0,1), (0,2), (1,2); Checker textures nested deeper than 8 levels or cyclic;
 * a Perlin permutation entry outside 0..1023; unknown kinds. */
int32_t mlrt_context_create(const mlrt_scene* scene, int32_t device, int32_t precision, mlrt_context** out); /* options from the environment */
int32_t mlrt_context_create_opts(const mlrt_scene* scene, int32_t device, int32_t precision, const mlrt_options* options, mlrt_context** out);
void mlrt_context_destroy(mlrt_context* ctx);

/* Replaces Renderer::render.  Renders the rows listed in `rows` (row j = 0 is the BOTTOM row, quirk Q13;
 * rows == NULL means all rows 0..H-1) into rgb_out, a caller-allocated host array [n_rows][W][3] of int32 in
 * 0..255 - exactly Vec<Vec<(i32,i32,i32)>> flattened.  Blocks until done. */
int32_t mlrt_context_render(mlrt_context* ctx, const mlrt_camera* camera, const mlrt_params* params, const int32_t* rows,
                            int32_t n_rows, int32_t* rgb_out, mlrt_progress_fn progress, void* user);

/* Asynchronous device-resident form used for benchmarking and multi-GPU runs: `d_rows` (n_rows row indices; NULL
 * = the rows 0..n_rows-1, so pass n_rows = image_height for the whole frame) and `d_rgb_out` [n_rows][W][3] int32 are
 * DEVICE pointers on the context's device; `d_accum_out` is an optional device array [n_rows][W][3] of f64 colour sums
 * (NULL to skip).  n_rows == 0 renders nothing.  The kernels are enqueued on `hip_stream` (a hipStream_t, NULL = default
 * stream) and the call returns without synchronising.
 * ONE frame in flight per context: the accumulation buffer, the tile counters and the timing events are per-context
 * scratch, so successive calls on one context must use the same stream (or be synchronised by the caller); use one
 * context per stream for concurrent frames (mlrt_multi does).  A frame is limited to n_rows * W * 3 < 2^31 values. */
int32_t mlrt_context_render_device(mlrt_context* ctx, const mlrt_camera* camera, const mlrt_params* params, const int32_t* d_rows,
                                   int32_t n_rows, int32_t* d_rgb_out, double* d_accum_out, void* hip_stream);

/* Waits for the context's last launch and reports its health.  The render kernel's persistent waves leave their loop when every
 * lane is idle; the release kernels also leave it - and record their lanes' states - in the one state from which nothing can change
 * any more (no lane traversing, parked for a batch or shading, but some lane still waiting for an accumulator bank), instead of
 * spinning forever.  MLRT_ERR_HIP + a message naming the lanes' states if that happened (the frame is then incomplete), MLRT_OK
 * otherwise.  The debug library (libmlrt_hip_debug.so) additionally counts iterations since a path of the wave last ended and gives
 * up after mlrt_options.stall_cap of them.  mlrt_context_render and mlrt_multi_render call this function. */
int32_t mlrt_context_check(mlrt_context* ctx);

/* Time of the most recent render kernel of this context in milliseconds, measured with HIP events on the
 * launch stream (synchronises on the stop event). Negative if nothing was launched. */
double mlrt_context_last_kernel_ms(mlrt_context* ctx);

/* Introspection for tests and DESIGN.md: bytes of the packed device scene, whether it is staged in LDS,
 * threads per workgroup, workgroups launched. out[4]. */
int32_t mlrt_context_info(mlrt_context* ctx, int64_t out[4]);
/* Which render-kernel variant the context's scene runs on: 0 = FULL (every feature), 1 = sphere-only scene over the threaded entry
 * list, 2 = sphere-only scene over the pair structure (mlrt_options.pair_walk), 3 .. = FULL compiled for a subset of the scene
 * features (the smallest compiled superset of what the scene uses).  Negative: NULL context. */
int32_t mlrt_context_kernel_variant(mlrt_context* ctx);

/* ---- several GPUs of one node, one process --------------------------------------------------------
 * The rows of the frame are dealt to the devices round-robin (the analogue of the reference's one-task-per-row
 * into_par_iter, /root/reference/src/raytrace.rs:172-186, with the devices as workers), every device renders its rows
 * from its own copy of the scene on its own stream, and ONE gather of the finished rows (one byte per channel) to
 * devices[0] closes the frame: RCCL over xGMI (grouped ncclSend / ncclRecv; librccl.so is loaded on first use) or, for
 * device lists with repeats - a rehearsal of the N-device path on fewer GPUs, which cannot form a communicator -
 * hipMemcpyPeerAsync.  The frame is bit-identical for any device list (Philox keyed per (pixel, sample), integer
 * accumulation). */
enum { MLRT_GATHER_AUTO = 0, /* RCCL when the devices are distinct and more than one, else copies */
       MLRT_GATHER_RCCL = 1, /* always RCCL (distinct devices; on ONE device the root's rows are sent to itself: a test of the wire) */
       MLRT_GATHER_COPY = 2  /* hipMemcpyPeerAsync */ };
typedef struct mlrt_multi mlrt_multi;
/* The row plan: the rows device k of n renders (j = k, k + n, ...; row 0 = bottom), written to rows_out when not NULL.
 * Returns their number.  Pure host arithmetic. */
int32_t mlrt_rows_for_device(int32_t image_height, int32_t k, int32_t n_devices, int32_t* rows_out);
/* One context and one stream per entry of `devices`.  options == NULL: mlrt_options_from_env(). */
int32_t mlrt_multi_create(const mlrt_scene* scene, const int32_t* devices, int32_t n_devices, int32_t precision, const mlrt_options* options,
                          int32_t gather_mode, mlrt_multi** out);
void mlrt_multi_destroy(mlrt_multi* multi);
/* Replaces Renderer::render: the whole frame into rgb_out, host [H][W][3] int32, row 0 = bottom.  Blocks until done;
 * `progress` is called on the calling thread as devices finish. */
int32_t mlrt_multi_render(mlrt_multi* multi, const mlrt_camera* camera, const mlrt_params* params, int32_t* rgb_out, mlrt_progress_fn progress,
                          void* user);
/* Of the last frame: wall milliseconds of mlrt_multi_render, the render-kernel milliseconds of every device (HIP events;
 * array of n_devices) and the gather actually used (MLRT_GATHER_RCCL or MLRT_GATHER_COPY).  Any pointer may be NULL. */
int32_t mlrt_multi_info(mlrt_multi* multi, double* frame_ms, double* kernel_ms_per_device, int32_t* gather_used);
/* One-shot: create (options from the environment, MLRT_GATHER_AUTO) + render + destroy. */
int32_t mlrt_render_multi(const mlrt_scene* scene, const mlrt_camera* camera, const mlrt_params* params, int32_t precision,
                          const int32_t* devices, int32_t n_devices, int32_t* rgb_out, mlrt_progress_fn progress, void* user);

/* One-shot drop-in for main.rs:147-157: create + render all rows + destroy. */
int32_t mlrt_render(const mlrt_scene* scene, const mlrt_camera* camera, const mlrt_params* params, int32_t precision,
                    int32_t* rgb_out, mlrt_progress_fn progress, void* user);

#ifdef __cplusplus
}
#endif
#endif /* MLRT_H */

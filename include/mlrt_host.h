/* include/mlrt_host.h - C ABI of the host-side mirror of the reference's driver code, exported by the same
 * libmlrt_hip.so.  These functions run on the CPU and need no GPU: they feed the render path of mlrt.h.
 *
 *   mlrt_world_*        <- trait World {name, camera, background, build}  (/root/reference/src/worlds.rs:14-25)
 *                          and the registry worlds()                     (src/worlds.rs:471-484)
 *   mlrt_camera_new     <- Camera::new                                   (src/camera.rs:15-38)
 *   mlrt_image_height   <- (image_width as f64 / aspect_ratio) as usize  (src/main.rs:121)
 *   mlrt_write_p3       <- the PPM printer                               (src/main.rs:144,175-179)
 *   mlrt_pcg64_*        <- SeedableRngator::rng = Pcg64::seed_from_u64(seed + site)  (src/rngator.rs:27-31)
 *   mlrt_decode_jpeg*   <- image::open("earthmap.jpg") + to_rgb8()                   (src/worlds.rs:180-181, 441-442)
 *
 * The world builders replay the reference's scene RNG stream draw for draw (rng(0), main.rs:185), so the
 * scene at a given --seed is the reference's scene; the BVH topology is the reference's too (bhv.rs:122-145).
 */
#ifndef MLRT_HOST_H
#define MLRT_HOST_H

#include <stdint.h>
#include <stdio.h>

#include "mlrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mlrt_world mlrt_world;

int32_t mlrt_world_count(void);
const char* mlrt_world_name(int32_t i);

/* World::build with rng = SeedableRngator::new(seed).rng(0).  earth_rgb (decoded RGB8 of earthmap.jpg,
 * row 0 = top) is needed by the worlds "earth" and "final_scene" only (worlds.rs:180,441); NULL otherwise.
 * Returns NULL on error (see mlrt_last_error). */
mlrt_world* mlrt_world_build(const char* name, uint64_t seed, const uint8_t* earth_rgb, int32_t earth_w, int32_t earth_h);
void mlrt_world_free(mlrt_world* w);
const mlrt_scene* mlrt_world_scene(const mlrt_world* w);
/* World::camera (worlds.rs:21-25): lookfrom[3], lookat[3], field_of_view */
void mlrt_world_camera(const mlrt_world* w, double lookfrom[3], double lookat[3], double* field_of_view);
/* World::background: fills params->background_* */
void mlrt_world_background(const mlrt_world* w, mlrt_params* params);
/* Stream probes for known-answer tests: the next raw u64 of the build stream taken just before the last
 * BHV::new, and after the whole build. */
void mlrt_world_stream_probes(const mlrt_world* w, uint64_t out[2]);

void mlrt_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov_degrees, double aspect_ratio,
                     double aperture, double focus_dist, mlrt_camera* out);
int32_t mlrt_image_height(int32_t image_width, double aspect_ratio);

/* rgb: [H][W][3] int32 with row j = 0 the bottom row; writes "P3\nW H\n255\n" then rows top to bottom. */
int32_t mlrt_write_p3(FILE* f, const int32_t* rgb, int32_t width, int32_t height);
int32_t mlrt_write_p3_path(const char* path, const int32_t* rgb, int32_t width, int32_t height);

void mlrt_pcg64_stream(uint64_t seed, uint64_t site_id, int32_t n, uint64_t* out);

/* Baseline (sequential Huffman, 8-bit, 1 or 3 components) JPEG -> RGB8, row 0 = top.  *rgb_out is malloc'ed by the
 * library: release it with mlrt_free.  Progressive / arithmetic files return MLRT_ERR_UNSUPPORTED. */
int32_t mlrt_decode_jpeg(const uint8_t* data, int64_t size, uint8_t** rgb_out, int32_t* width, int32_t* height);
int32_t mlrt_decode_jpeg_file(const char* path, uint8_t** rgb_out, int32_t* width, int32_t* height);
void mlrt_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* MLRT_HOST_H */

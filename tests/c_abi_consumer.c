/* tests/c_abi_consumer.c - a plain C99 consumer of include/mlrt.h + include/mlrt_host.h: what a non-C++ host (or a
 * Rust/Go FFI layer) sees.  Builds the `random` world, renders a small frame through the one-shot entry point and
 * prints the P3 image.  `gcc -std=c99 -pedantic -fsyntax-only` of this file is part of the CPU test tier; the GPU tier
 * links and runs it.  The call sequence is the one integration/mlrt_ffi.rs (the uncompiled Rust binding) makes: the reference's
 * precision, one context for one device, the multi-device entry point for several.
 * usage: c_abi_consumer <width> <spp> <seed> [n_devices (entries of a device list that repeats device 0; 0 = single-context path)] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mlrt.h"
#include "mlrt_host.h"

static void progress(int32_t done, int32_t total, void* user) {
    int* calls = (int*)user;
    (void)done; (void)total;
    (*calls)++;
}

int main(int argc, char** argv) {
    const int32_t width = argc > 1 ? atoi(argv[1]) : 120;
    const int32_t spp = argc > 2 ? atoi(argv[2]) : 4;
    const uint64_t seed = argc > 3 ? (uint64_t)strtoull(argv[3], NULL, 10) : 42u;
    const int32_t n_devices = argc > 4 ? atoi(argv[4]) : 0;
    int32_t devices[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double aspect = 1.5;
    mlrt_params params;
    mlrt_camera camera;
    double lookfrom[3], lookat[3], up[3] = {0.0, 1.0, 0.0}, fov, dx, dy, dz, focus;
    int32_t* rgb;
    int calls = 0, rc;
    mlrt_world* world = mlrt_world_build("random", seed, NULL, 0, 0);
    if (!world) { fprintf(stderr, "world: %s\n", mlrt_last_error()); return 1; }
    memset(&params, 0, sizeof params);
    params.image_width = width;
    params.image_height = mlrt_image_height(width, aspect);
    params.samples_per_pixel = spp;
    params.max_depth = 50;
    params.seed = seed;
    mlrt_world_background(world, &params);
    mlrt_world_camera(world, lookfrom, lookat, &fov);
    dx = lookat[0] - lookfrom[0]; dy = lookat[1] - lookfrom[1]; dz = lookat[2] - lookfrom[2];
    focus = sqrt(dx * dx + dy * dy + dz * dz); /* main.rs:109-112 */
    mlrt_camera_new(lookfrom, lookat, up, fov, aspect, 0.0, focus, &camera);
    rgb = (int32_t*)malloc((size_t)params.image_width * params.image_height * 3 * sizeof(int32_t));
    if (!rgb) return 1;
    if (n_devices > 0) rc = mlrt_render_multi(mlrt_world_scene(world), &camera, &params, MLRT_PRECISION_FP64, devices, n_devices > 8 ? 8 : n_devices, rgb, progress, &calls);
    else rc = mlrt_render(mlrt_world_scene(world), &camera, &params, MLRT_PRECISION_FP64, rgb, progress, &calls);
    if (rc != MLRT_OK) { fprintf(stderr, "render failed (%d): %s\n", rc, mlrt_last_error()); return rc == MLRT_ERR_NO_DEVICE ? 2 : 1; }
    if (calls < 1) { fprintf(stderr, "progress callback was never called\n"); return 1; }
    rc = mlrt_write_p3(stdout, rgb, params.image_width, params.image_height);
    free(rgb);
    mlrt_world_free(world);
    return rc;
}

// host/host_internal.hpp - storage behind mlrt_world (see include/mlrt_host.h).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mlrt.h"
#include "../../include/mlrt_host.h"
#include "pcg64.hpp"

namespace mlrt {

struct PerlinTables {  // textures.rs:53-59
    double ranvec[1024 * 3];
    int32_t perm[3 * 1024];
};

// Owns every array an mlrt_scene points into.
struct WorldStore {
    std::string name;
    std::vector<mlrt_sphere> spheres;
    std::vector<mlrt_box> boxes;
    std::vector<mlrt_rect> rects;
    std::vector<mlrt_prim_ref> prim_refs;
    std::vector<mlrt_bvh_node> bvh_nodes;
    std::vector<mlrt_bvh> bvhs;
    std::vector<mlrt_transform> transforms;
    std::vector<mlrt_object> objects;
    std::vector<mlrt_material> materials;
    std::vector<mlrt_texture> textures;
    std::vector<mlrt_image> images;
    std::vector<std::vector<uint8_t>> image_pixels;
    std::vector<mlrt_perlin> perlins;
    std::vector<std::unique_ptr<PerlinTables>> perlin_tables;
    mlrt_scene scene{};
    // World::camera / World::background (worlds.rs:14-25)
    double lookfrom[3] = {0, 0, 0}, lookat[3] = {0, 0, 0}, field_of_view = 20.0;
    int32_t background_kind = MLRT_BG_GRADIENT;
    uint64_t probe_before_last_bvh = 0, probe_after_build = 0;
    void finalize();
};

const std::vector<std::string>& world_registry();
std::unique_ptr<WorldStore> build_world_store(const std::string& name, uint64_t seed, const uint8_t* earth_rgb, int32_t ew, int32_t eh,
                                              std::string& error);
void set_last_error(const std::string& msg);

}  // namespace mlrt

struct mlrt_world {
    std::unique_ptr<mlrt::WorldStore> store;
};

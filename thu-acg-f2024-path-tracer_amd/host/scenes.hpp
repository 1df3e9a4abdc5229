// Scene scripts of the reference's main.rs (see scenes.cpp).
#pragma once
#include <string>

#include "pt.hpp"

struct SceneSetup {
    path_tracer::World world;
    path_tracer::Camera camera;
    std::string output;   // the file name the reference script renders to
};

SceneSetup balls_scene(size_t width, size_t spp, uint64_t scene_seed);           // -s 1
SceneSetup earth_scene(size_t width, size_t spp);                                // -s 2
SceneSetup cornell_box_scene(size_t width, size_t spp);                          // -s 3
SceneSetup environment_map_scene(size_t width, size_t spp);                      // -s 4
SceneSetup bsdf_demo_scene(size_t width, size_t spp);                            // -s 5
SceneSetup everything_scene(size_t width, size_t spp, const std::string& asset_dir);   // -s 6
SceneSetup normal_demo_scene(size_t width, size_t spp);                          // -s 7
SceneSetup make_scene(int scene_id, size_t width, size_t spp, const std::string& asset_dir, uint64_t scene_seed);

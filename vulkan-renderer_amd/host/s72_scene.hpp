// Finds the environment cube maps a scene (.s72) refers to, the way the renderer's
// scene loader resolves them (src/scene_loader.cpp:59-62,113-116,361-379,509-535):
// environments are numbered in the order their ENVIRONMENT objects appear in the
// scene array; one with a "radiance" cube texture names <scene dir>/<src>; one with a
// "local" description names <scene>.localIBL_<number>.png (written by the local-IBL
// capture tool, src/utilities/local_ibl.cpp:289-295).  The loader then expects
// X.lambertian.png and X.ggx<m>.png next to each (scene_loader.cpp:559-583), which is
// what the cube tool writes.
#pragma once
#include <filesystem>
#include <string>
#include <vector>

struct SceneEnvironment {
    std::string name;
    std::filesystem::path file;   // the RGBE cube strip to prefilter
    bool local;                   // captured local environment rather than a radiance texture
};

// Throws std::runtime_error with the loader's wording where it has one.
std::vector<SceneEnvironment> scene_environments(const std::filesystem::path& scene_file);

// CLI with the reference's two flags (src/main.rs:620-645): -q/--quality toggles
// 1920 px @ 4000 spp vs 600 px @ 100 spp, -s/--scene N picks the scene script. Extra,
// explicit overrides (not in the reference): --width, --spp, --seed, --out, --assets, --device, --float-hdr
// (.hdr environments keep their f32 samples instead of the reference's .to_rgb8() squash, texture.rs:67).
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "scenes.hpp"

using namespace path_tracer;

int main(int argc, char** argv) {
    bool quality = false, float_hdr = false;
    int scene = 1, device = 0;
    long width = -1, spp = -1;
    uint64_t seed = 1;
    std::string out, assets = "assets";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; exit(2); }
            return argv[++i];
        };
        if (a == "-q" || a == "--quality") quality = true;
        else if (a == "-s" || a == "--scene") scene = atoi(next());
        else if (a == "--width") width = atol(next());
        else if (a == "--spp") spp = atol(next());
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--out") out = next();
        else if (a == "--assets") assets = next();
        else if (a == "--device") device = atoi(next());
        else if (a == "--float-hdr") float_hdr = true;
        else if (a == "-h" || a == "--help") {
            std::cout << "usage: pt_render [-q] [-s N] [--width W] [--spp S] [--seed K] [--out file.png] [--assets DIR] [--device D] [--float-hdr]\n";
            return 0;
        } else { std::cerr << "unknown argument " << a << "\n"; return 2; }
    }
    size_t w = quality ? 1920 : 600, s = quality ? 4000 : 100;   // main.rs:633
    if (width > 0) w = (size_t)width;
    if (spp > 0) s = (size_t)spp;
    if (scene < 1 || scene > 7) return 0;   // `_ => ()` main.rs:643
    pt_ctx* ctx = nullptr;
    if (pt_ctx_create(device, &ctx) != 0) {
        std::cerr << "fatal: " << pt_last_error() << "\n";
        return 1;
    }
    try {
        SceneSetup setup = make_scene(scene, w, s, assets, 1);
        setup.world.float_hdr = float_hdr;
        setup.world.build_bvh(ctx, setup.camera.environment.is_map ? setup.camera.environment.map : nullptr);
        setup.camera.init();
        std::cerr << "rendering production\n";   // camera.rs:101
        setup.camera.render(setup.world, out.empty() ? setup.output : out, seed);
        setup.world.release();
    } catch (const std::exception& e) {
        std::cerr << "panic: " << e.what() << "\n";   // the reference unwrap()s asset errors
        pt_ctx_destroy(ctx);
        return 101;
    }
    pt_ctx_destroy(ctx);
    return 0;
}

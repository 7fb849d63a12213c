// `raytrace <scene.json> -o out.png [--no-progress] [--passes N]` -- the reference
// CLI's flags and output lines (src/main.rs:9-78) on top of the C ABI.  The window
// preview (no -o) is out of scope: -o is required here; --passes N renders the frame
// progressively (rt_render_progressive, the counterpart of raytrace_to_buffer's
// progressively filled buffer) and reports pixels done + rays after every pass, which
// is what the reference's progress bar shows (raytracing_scene.rs:624-643).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/mi355rt.h"
#include "../../../include/mi355rt_host.h"

static double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
    const char *scene_path = nullptr, *output = nullptr;
    unsigned long long seed = 0;
    unsigned passes = 0;
    bool progress = true;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "-o") || !std::strcmp(argv[i], "--output")) {
            if (++i >= argc) {
                std::fprintf(stderr, "error: --output needs a value\n");
                return 2;
            }
            output = argv[i];
        } else if (!std::strcmp(argv[i], "--no-progress")) {
            progress = false;
        } else if (!std::strcmp(argv[i], "--passes")) {
            if (++i >= argc) return 2;
            passes = static_cast<unsigned>(std::strtoul(argv[i], nullptr, 10));
        } else if (!std::strcmp(argv[i], "--seed")) {
            if (++i >= argc) return 2;
            seed = std::strtoull(argv[i], nullptr, 10);
        } else if (!std::strcmp(argv[i], "-h") || !std::strcmp(argv[i], "--help")) {
            std::printf("ray tracer\nA ray tracer (MI355X hot path)\n\nUSAGE:\n    raytrace [OPTIONS] <scene>\n\n"
                        "ARGS:\n    <scene>    input scene as a json file\n\nOPTIONS:\n"
                        "    -o, --output <output>    Output rendered image to file\n"
                        "        --no-progress        Hide progress bar\n"
                        "        --passes <n>         Render progressively in n passes\n"
                        "        --seed <n>           Sample-stream seed (default 0)\n");
            return 0;
        } else if (!scene_path) {
            scene_path = argv[i];
        } else {
            std::fprintf(stderr, "error: unexpected argument '%s'\n", argv[i]);
            return 2;
        }
    }
    if (!scene_path) {
        std::fprintf(stderr, "error: the following required arguments were not provided:\n    <scene>\n");
        return 2;
    }
    if (!output) {
        std::fprintf(stderr, "error: rendering to a window is not supported by this build; pass -o <file.png>\n");
        return 2;
    }
    double total = 0.0;
    auto t0 = std::chrono::steady_clock::now();
    rth_scene *hs = nullptr;
    if (rth_scene_load(scene_path, &hs) != RT_OK) {
        std::fprintf(stderr, "failed to parse scene: %s\n", rth_last_error());
        return 1;
    }
    double dt = seconds_since(t0);
    total += dt;
    std::printf("Took %.6fs to load assets.\n", dt);

    t0 = std::chrono::steady_clock::now();
    const rt_scene_desc *desc = rth_scene_desc(hs);
    rt_scene *scene = nullptr;
    if (rt_scene_create(desc, &scene) != RT_OK) {
        std::fprintf(stderr, "failed to build scene: %s\n", rt_last_error());
        return 1;
    }
    dt = seconds_since(t0);
    total += dt;
    std::printf("Took %.6fs to pre-process scene and construct bounding boxes for %u primitives.\n", dt, desc->n_prims);

    std::vector<uint8_t> rgba(static_cast<size_t>(desc->opts.width) * desc->opts.height * 4);
    rt_render_params params{};
    params.seed = seed;
    rt_render_stats stats{};
    uint64_t rays = 0;
    t0 = std::chrono::steady_clock::now();
    int rc;
    if (passes > 1) {
        const size_t npx = static_cast<size_t>(desc->opts.width) * desc->opts.height;
        std::vector<uint32_t> argb(npx);
        struct Ctx {
            bool progress;
            size_t npx;
        } ctx{progress, npx};
        auto on_pass = [](void *user, uint32_t done, uint32_t total, uint64_t rays_so_far) -> int {
            const Ctx *c = static_cast<const Ctx *>(user);
            if (c->progress)
                std::printf("pass %u/%u: %zu/%zu pixels, %llu rays\n", done, total, c->npx * done / total, c->npx,
                            static_cast<unsigned long long>(rays_so_far));
            return 0;
        };
        rc = rt_render_progressive(scene, &params, passes, argb.data(), on_pass, &ctx, nullptr, &rays);
        for (size_t i = 0; i < npx; ++i) {  // ARGB words -> RGBA bytes
            rgba[4 * i] = static_cast<uint8_t>(argb[i] >> 16);
            rgba[4 * i + 1] = static_cast<uint8_t>(argb[i] >> 8);
            rgba[4 * i + 2] = static_cast<uint8_t>(argb[i]);
            rgba[4 * i + 3] = static_cast<uint8_t>(argb[i] >> 24);
        }
    } else {
        rc = rt_render(scene, &params, nullptr, rgba.data(), &rays, &stats);
    }
    if (rc != RT_OK) {
        std::fprintf(stderr, "render failed: %s\n", rt_last_error());
        return 1;
    }
    dt = seconds_since(t0);
    total += dt;
    std::printf("Took %.6fs to render the scene.\n", dt);
    std::printf("%llu rays, %.1f Mrays/s (kernels %.3f ms)\n", static_cast<unsigned long long>(rays),
                static_cast<double>(rays) / dt / 1e6, stats.ms_total);
    t0 = std::chrono::steady_clock::now();
    if (rth_write_png(output, desc->opts.width, desc->opts.height, rgba.data()) != RT_OK) {
        std::fprintf(stderr, "unable to write image\n");
        return 1;
    }
    std::printf("Output written to %s in %.3fs (+ %.1f ms to encode and write the PNG)\n", output, total, 1e3 * seconds_since(t0));
    rt_scene_destroy(scene);
    rth_scene_free(hs);
    return 0;
}

// pt_render — non-interactive counterpart of main.zig:21-192: same parameters (scene, width,
// rays per subsample, seed), same presets (subpixels 4, t in (0.1, 1000), rebounds 6 by
// default, 16:9, max value 255), same output file name, rendered on one MI355X.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "pt_host.h"

using namespace pt;

static void usage() {
    fprintf(stderr, "usage: pt_render [--scene 18|19|20|21] [--width W] [--spp N] [--depth D] [--seed S] [--prng xoroshiro128|xoshiro256]\n"
                    "                 [--gpus N [--split tiles|samples]] [--device G] [--deterministic] [--chunk C] [--aov radiance|normal|albedo] [--sampler rejection|direct] [--out FILE]\n");
}

int main(int argc, char **argv) {
    int num_scene = 18, device = 0, gpus = 0, split = PT_SPLIT_TILES; // --gpus: the node's GPUs instead of main.zig's threads               // prompt defaults: main.zig:50,58,63,66
    unsigned width = 256, spp = 8, depth = 6, chunk = 0, illumination = PT_ILLUM_RADIANCE, sampler = PT_SAMPLER_REJECTION;
    unsigned long long seed = 42;
    bool deterministic = false;
    Prng prng_kind = Prng::Default;
    std::string out;
    for (int i = 1; i < argc; ++i) {
        auto need = [&](const char *flag) { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", flag); exit(2); } return argv[++i]; };
        if (!strcmp(argv[i], "--scene")) num_scene = atoi(need("--scene"));
        else if (!strcmp(argv[i], "--width")) width = (unsigned)strtoul(need("--width"), nullptr, 10);
        else if (!strcmp(argv[i], "--spp")) spp = (unsigned)strtoul(need("--spp"), nullptr, 10);
        else if (!strcmp(argv[i], "--depth")) depth = (unsigned)strtoul(need("--depth"), nullptr, 10);
        else if (!strcmp(argv[i], "--seed")) seed = strtoull(need("--seed"), nullptr, 10);
        else if (!strcmp(argv[i], "--device")) device = atoi(need("--device"));
        else if (!strcmp(argv[i], "--gpus")) gpus = atoi(need("--gpus"));
        else if (!strcmp(argv[i], "--split")) {
            const char *a = need("--split");
            if (!strcmp(a, "tiles")) split = PT_SPLIT_TILES;
            else if (!strcmp(a, "samples")) split = PT_SPLIT_SAMPLES;
            else { usage(); return 2; }
        }
        else if (!strcmp(argv[i], "--chunk")) chunk = (unsigned)strtoul(need("--chunk"), nullptr, 10);
        else if (!strcmp(argv[i], "--out")) out = need("--out");
        else if (!strcmp(argv[i], "--deterministic")) deterministic = true;
        else if (!strcmp(argv[i], "--prng")) { // which std.rand.DefaultPrng generation builds scene 21 (pt_host.h)
            const char *a = need("--prng");
            if (!strcmp(a, "xoroshiro128")) prng_kind = Prng::Xoroshiro128;
            else if (!strcmp(a, "xoshiro256")) prng_kind = Prng::Xoshiro256;
            else { usage(); return 2; }
        }
        else if (!strcmp(argv[i], "--sampler")) { // the reference's rejection loops, or their direct equivalents (include/ptgpu.h)
            const char *a = need("--sampler");
            if (!strcmp(a, "rejection")) sampler = PT_SAMPLER_REJECTION;
            else if (!strcmp(a, "direct")) sampler = PT_SAMPLER_DIRECT;
            else { usage(); return 2; }
        }
        else if (!strcmp(argv[i], "--aov")) { // the reference's commented-out colour functions (renderers/ppm_image.zig:64-65)
            const char *a = need("--aov");
            if (!strcmp(a, "normal")) illumination = PT_ILLUM_NORMAL;
            else if (!strcmp(a, "albedo")) illumination = PT_ILLUM_ALBEDO;
            else if (!strcmp(a, "radiance")) illumination = PT_ILLUM_RADIANCE;
            else { usage(); return 2; }
        }
        else { usage(); return 2; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    const Image img = Image::make(width, 16.0f / 9.0f, 255);              // main.zig:78
    const RayTracerConfig cfg{4, 0.1f, 1000.0f, depth, spp};               // main.zig:85-91
    DefaultPrng prng(seed, prng_kind);                                    // main.zig:100
    Scene scene;
    int rc = scene.setupScene(prng, num_scene);                           // main.zig:105
    if (rc != Ok) { fprintf(stderr, "error: SceneNotAvailable\n"); return 1; }
    Camera camera;
    rc = Camera::preset(num_scene, img.aspect_ratio, &camera);            // main.zig:107-136
    if (rc != Ok) { fprintf(stderr, "error: SceneNotAvailable\n"); return 1; }
    std::vector<uint8_t> slice(img.size);                                 // main.zig:139
    ptgpu_ctx *gpu = nullptr;
    ptgpu_multi *multi = nullptr;
    ppm_renderer::Config rcfg{nullptr, &camera, &img, 0, img.num_pixels, seed, &cfg, slice.data(), slice.size(), &scene};
    rcfg.deterministic = deterministic;
    rcfg.chunk = chunk;
    rcfg.illumination = illumination;
    rcfg.sampler = sampler;
    pt_stats st;
    double reduce_ms = 0.0;
    if (gpus > 0) { // main.zig:142-165: "use the multi-threaded version" -> the GPUs of the node
        rc = ptgpu_multi_create(nullptr, gpus, &multi);
        if (rc != PT_OK) { fprintf(stderr, "error: ptgpu_multi_create(%d GPUs): %s\n", gpus, ptgpu_strerror(rc)); return 1; }
        rc = ppm_renderer::renderMultiGpu(rcfg, multi, split);
        if (rc != PT_OK) { fprintf(stderr, "error: render: %s\n", ptgpu_strerror(rc)); ptgpu_multi_destroy(multi); return 1; }
        ptgpu_multi_get_stats(multi, -1, &st);
        reduce_ms = ptgpu_multi_reduce_ms(multi);
    } else {
        rc = ptgpu_create(device, &gpu);
        if (rc != PT_OK) { fprintf(stderr, "error: ptgpu_create: %s\n", ptgpu_strerror(rc)); return 1; }
        rcfg.gpu = gpu;
        rc = ppm_renderer::render(rcfg);                                  // main.zig:167-177
        if (rc != PT_OK) { fprintf(stderr, "error: render: %s\n", ptgpu_strerror(rc)); ptgpu_destroy(gpu); return 1; }
        ptgpu_get_stats(gpu, &st);
    }
    const std::string header = img.header();                              // main.zig:180-182
    memcpy(slice.data(), header.data(), header.size());
    const std::string path = out.empty() ? filepath((size_t)num_scene, cfg, img, (size_t)seed) : out;
    FILE *f = fopen(path.c_str(), "wb");                                  // main.zig:187
    if (!f) { fprintf(stderr, "error: cannot open %s\n", path.c_str()); if (gpu) ptgpu_destroy(gpu); if (multi) ptgpu_multi_destroy(multi); return 1; }
    fwrite(slice.data(), 1, slice.size(), f);
    fclose(f);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "info: wrote %s\n", path.c_str());
    fprintf(stderr, "info: %llu paths, %llu rays, kernel %.2f ms: %.3e paths/s, %.1f Mrays/s\n", (unsigned long long)st.paths,
            (unsigned long long)st.rays, st.kernel_ms, st.paths / (st.kernel_ms * 1e-3), st.rays / (st.kernel_ms * 1e-3) * 1e-6);
    if (multi) fprintf(stderr, "info: %d GPUs, %s split, slowest kernel %.2f ms, RCCL reduce %.3f ms\n", gpus,
                       split == PT_SPLIT_TILES ? "tile" : "sample-range", st.kernel_ms, reduce_ms);
    fprintf(stderr, "info: Program took %.2f seconds\n", s);             // main.zig:191
    if (gpu) ptgpu_destroy(gpu);
    if (multi) ptgpu_multi_destroy(multi);
    return 0;
}

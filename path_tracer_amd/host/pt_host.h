// pt_host.h — C++ host side above the C ABI: the reference's renderer interface
// (Scene, Camera, Image, RayTracerConfig, render) with the same names, argument meaning
// and error behaviour, driving libptgpu instead of CPU threads.
//
// The reference is Zig; no Zig toolchain exists in this image, so the host mirror is
// C++ (INTEGRATION.md shows the Zig `extern fn` binding a maintainer would add).
// Citations are file:line under src/ of jackdbd/path-tracer.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/ptgpu.h"

namespace pt {

// errors.zig + the error sets of the path, as status codes shared with the C ABI
enum : int { Ok = PT_OK, SceneNotAvailable = PT_ERR_SCENE_NOT_AVAILABLE, ValueExceedsMaxValue = PT_ERR_VALUE_EXCEEDS_MAX,
             NoSpaceLeft = PT_ERR_NO_SPACE, DirectionIsNotUnitVector = PT_ERR_DIRECTION_NOT_UNIT };

// vec3.zig:7-91 (f32)
struct Vec3f {
    float x, y, z;
    static Vec3f make(float x, float y, float z) { return Vec3f{x, y, z}; }
    float lengthSquared() const;
    float length() const;
    Vec3f add(Vec3f o) const;
    Vec3f sub(Vec3f o) const;
    Vec3f mul(float k) const;
    Vec3f unitVector() const;
    float dot(Vec3f o) const;
    Vec3f cross(Vec3f o) const;
    Vec3f elementwiseMul(Vec3f o) const;
};

// std.rand.DefaultPrng + Random.float(f32), restated from the published algorithms.  Zig's std had two
// DefaultPrng generations: Xoroshiro128+ (older std) and Xoshiro256++ (0.9.1, as recalled).  The only image the
// reference holds (images/demo.png: scene 21, seed 456) was made with Xoroshiro128+ — scene 21 built from it
// reproduces the picture's small spheres, Xoshiro256++ does not (tests/test_demo_pin.py) — so that is the default.
enum class Prng : uint32_t { Xoroshiro128 = 0, Xoshiro256 = 1, Default = Xoroshiro128 };
class DefaultPrng {
public:
    explicit DefaultPrng(uint64_t seed, Prng kind = Prng::Default);
    uint64_t next();
    float float32();
private:
    uint64_t s_[4];
    Prng kind_;
};

// material.zig:174-189
struct Material {
    uint32_t kind; // PT_LAMBERTIAN / PT_METAL / PT_DIELECTRIC
    Vec3f albedo;
    float param;   // fuzziness | refraction_index
    static Material lambertian(Vec3f albedo);
    static Material metal(Vec3f albedo, float fuzziness);
    static Material dielectric(float refraction_index);
};

// sphere.zig:36-50
struct Sphere {
    Vec3f center;
    float radius;
    Material material;
};

// scene.zig:19-94
class Scene {
public:
    std::vector<Sphere> spheres;
    void sceneImage18_19_20();                 // scene.zig:32-50
    void sceneImage21(DefaultPrng &r);         // scene.zig:52-84
    int setupScene(DefaultPrng &r, int n);     // scene.zig:86-94 (SceneNotAvailable otherwise)
    // BASELINE config 5 (not in the reference): ground + n-1 random spheres.
    void sceneSynthetic(uint32_t n, uint64_t seed);
    std::vector<pt_sphere> flatten() const;    // the PODs that cross the C ABI
};

// camera.zig:18-107
struct Camera {
    float aperture, aspect_ratio, focus_dist;
    Vec3f horizontal;
    float lens_radius;
    Vec3f lookat, lookfrom, lower_left_corner, u, v, vertical;
    float viewport_height, viewport_width;
    Vec3f vup;
    static Camera make(Vec3f lookfrom, Vec3f lookat, Vec3f vup, float vfov, float aspect_ratio, float aperture,
                       float focus_dist);                    // Camera.new, camera.zig:48-97
    static int preset(int num_scene, float aspect_ratio, Camera *out); // main.zig:107-136
    pt_camera flatten() const;
};

// render_targets/ppm_image.zig:28-104
struct Image {
    float aspect_ratio;
    Vec3f blend_start, blend_stop;
    size_t data_size, header_size, height, max_px_value, num_pixels, px_size, size, width;
    static Image make(size_t width, float aspect_ratio, size_t max_px_value); // Image.new, :59-81
    std::string header() const;                                              // :84-88
    int rgbToAscii(Vec3f col, char *out, size_t cap) const;                  // :93-103
};

// utils.zig
uint8_t numDigits(size_t num);                                               // utils.zig:12-20
int colNumToString(size_t max_value, uint32_t value, char *out, size_t cap); // utils.zig:62-84
float degreesToRadians(float deg);                                           // utils.zig:113-115
size_t chunkSize(size_t num_pixels, size_t num_threads);                     // multithreading.zig:40-44

// raytracer.zig:17-23 (counters widened from u8, see include/ptgpu.h)
struct RayTracerConfig {
    uint32_t subpixels;
    float t_min, t_max;
    uint32_t rebounds;
    uint32_t rays_per_subsample;
};

namespace ppm_renderer {

// renderers/ppm_image.zig:16-26.  `r: *Random` becomes `seed` (the Philox key) and
// `allocator` disappears (the caller still owns `slice`); `gpu` is the device context
// that replaces the CPU thread the reference would run this range on.
struct Config {
    ptgpu_ctx *gpu;
    const Camera *camera;
    const Image *img;
    size_t istart, istop;
    uint64_t seed;
    const RayTracerConfig *ray_tracer_config;
    uint8_t *slice;
    size_t slice_len;
    const Scene *scene;
    bool deterministic = false;
    uint32_t chunk = 0; // samples per GPU work item (0: one sequential sum per pixel, as the reference)
    uint32_t illumination = PT_ILLUM_RADIANCE; // renderers/ppm_image.zig:64-66: radiance | colorNormal | colorAlbedo
    uint32_t sampler = PT_SAMPLER_REJECTION;   // camera.zig:113-120 / material.zig:27-34 as the reference loops them, or PT_SAMPLER_DIRECT (not in the reference)
};

// render(), renderers/ppm_image.zig:30-87: traces pixels [istart, istop) and writes
// their ASCII records into slice[header_size + idx*px_size ..].
int render(const Config &cfg);

// The multi-GPU counterpart of renderMultiThread (renderers/ppm_image.zig:91-129) and its thread
// block (main.zig:142-165): the same Config (cfg.gpu is ignored), the GPUs of `multi` share the range
// by interleaved tiles (PT_SPLIT_TILES) or by sample ranges (PT_SPLIT_SAMPLES) and the fp32 sums are
// combined by one RCCL reduce before the single resolve.
int renderMultiGpu(const Config &cfg, ptgpu_multi *multi, int split);

} // namespace ppm_renderer

// main.zig:194-203
std::string filepath(size_t num_scene, const RayTracerConfig &cfg, const Image &img, size_t seed);

} // namespace pt

// ---- C view of the same host objects (for ctypes / tests) ------------------------------
extern "C" {
int pthost_setup_scene(int num_scene, uint64_t seed, pt_sphere *out, uint32_t cap);            // Prng::Default
int pthost_setup_scene_prng(int num_scene, uint64_t seed, uint32_t prng, pt_sphere *out, uint32_t cap);
int pthost_scene_synthetic(uint32_t n, uint64_t seed, pt_sphere *out, uint32_t cap);
int pthost_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov, float aspect_ratio,
                      float aperture, float focus_dist, pt_camera *out);
int pthost_camera_preset(int num_scene, float aspect_ratio, pt_camera *out);
typedef struct pthost_image {
    float aspect_ratio;
    uint32_t width, height, max_px_value, px_size, header_size;
    uint64_t num_pixels, data_size, size;
    float blend_start[3], blend_stop[3];
} pthost_image;
int pthost_image_new(uint32_t width, float aspect_ratio, uint32_t max_px_value, pthost_image *out);
int pthost_image_header(const pthost_image *img, char *buf, size_t cap);
int pthost_rgb_to_ascii(const pthost_image *img, const float col[3], char *out, size_t cap);
uint32_t pthost_num_digits(uint64_t num);
int pthost_col_num_to_string(uint64_t max_value, uint32_t value, char *out, size_t cap);
uint64_t pthost_chunk_size(uint64_t num_pixels, uint64_t num_threads);
// Whole-frame render of a preset scene through the renderer mirror; writes header + records.
int pthost_render_scene_ppm(ptgpu_ctx *gpu, int num_scene, uint32_t width, uint32_t subpixels, uint32_t rays_per_subsample,
                            uint32_t rebounds, uint64_t seed, int deterministic, uint8_t *slice, size_t slice_len);
// The same on all GPUs of `multi` (split: PT_SPLIT_TILES | PT_SPLIT_SAMPLES).
int pthost_render_scene_ppm_multi(ptgpu_multi *multi, int split, int num_scene, uint32_t width, uint32_t subpixels,
                                  uint32_t rays_per_subsample, uint32_t rebounds, uint64_t seed, int deterministic, uint32_t chunk,
                                  uint8_t *slice, size_t slice_len);
int pthost_filepath(uint64_t num_scene, uint32_t subpixels, uint32_t rays_per_subsample, uint32_t rebounds, uint32_t width,
                    uint64_t seed, char *buf, size_t cap);
}

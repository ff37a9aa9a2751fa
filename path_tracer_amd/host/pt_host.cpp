// pt_host.cpp — host side above the C ABI (see pt_host.h).  Compiled with
// -ffp-contract=off so that scene/camera/image arithmetic rounds as the reference's f32.
#include "pt_host.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

namespace pt {

// ---- vec3.zig ---------------------------------------------------------------------------
float Vec3f::lengthSquared() const { return x * x + y * y + z * z; } // pow(x,2)+pow(y,2)+pow(z,2), vec3.zig:37-39
float Vec3f::length() const { return sqrtf(lengthSquared()); }
Vec3f Vec3f::add(Vec3f o) const { return make(x + o.x, y + o.y, z + o.z); }
Vec3f Vec3f::sub(Vec3f o) const { return make(x - o.x, y - o.y, z - o.z); }
Vec3f Vec3f::mul(float k) const { return make(k * x, k * y, k * z); }
Vec3f Vec3f::unitVector() const {
    const float inv_n = 1.0f / length();
    return make(inv_n * x, inv_n * y, inv_n * z);
}
float Vec3f::dot(Vec3f o) const { return x * o.x + y * o.y + z * o.z; }
Vec3f Vec3f::cross(Vec3f o) const { return make(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
Vec3f Vec3f::elementwiseMul(Vec3f o) const { return make(x * o.x, y * o.y, z * o.z); }

// ---- std.rand.DefaultPrng ---------------------------------------------------------------------
static uint64_t splitmix64(uint64_t &s) {
    s += 0x9e3779b97f4a7c15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
DefaultPrng::DefaultPrng(uint64_t seed, Prng kind) : kind_(kind) {
    uint64_t s = seed;
    const int words = (kind == Prng::Xoshiro256) ? 4 : 2; // Xoshiro256.seed / Xoroshiro128.seed
    for (int i = 0; i < 4; ++i) s_[i] = (i < words) ? splitmix64(s) : 0;
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t DefaultPrng::next() {
    if (kind_ == Prng::Xoshiro256) { // xoshiro256++
        const uint64_t r = rotl(s_[0] + s_[3], 23) + s_[0];
        const uint64_t t = s_[1] << 17;
        s_[2] ^= s_[0]; s_[3] ^= s_[1]; s_[1] ^= s_[2]; s_[0] ^= s_[3];
        s_[2] ^= t;
        s_[3] = rotl(s_[3], 45);
        return r;
    }
    const uint64_t s0 = s_[0]; // xoroshiro128+ (55, 14, 36)
    uint64_t s1 = s_[1];
    const uint64_t r = s0 + s1;
    s1 ^= s0;
    s_[0] = rotl(s0, 55) ^ s1 ^ (s1 << 14);
    s_[1] = rotl(s1, 36);
    return r;
}
float DefaultPrng::float32() { // Random.float(f32): [1,2) from the top 23 bits of a u32, minus 1
    const uint32_t s = (uint32_t)next();
    const uint32_t repr = (0x7fu << 23) | (s >> 9);
    float f;
    memcpy(&f, &repr, 4);
    return f - 1.0f;
}

// ---- material.zig / sphere.zig / scene.zig -------------------------------------------------
Material Material::lambertian(Vec3f albedo) { return Material{PT_LAMBERTIAN, albedo, 0.0f}; }
Material Material::metal(Vec3f albedo, float fuzziness) { return Material{PT_METAL, albedo, fuzziness}; }
Material Material::dielectric(float ri) { return Material{PT_DIELECTRIC, Vec3f::make(1.0f, 1.0f, 1.0f), ri}; }

void Scene::sceneImage18_19_20() {
    const Material ground = Material::lambertian(Vec3f::make(0.8f, 0.8f, 0.0f));
    const Material center = Material::lambertian(Vec3f::make(0.1f, 0.2f, 0.5f));
    const Material left = Material::dielectric(1.5f);
    const Material right = Material::metal(Vec3f::make(0.8f, 0.6f, 0.2f), 0.0f);
    spheres.push_back(Sphere{Vec3f::make(0, -100.5f, -1), 100.5f, ground});
    spheres.push_back(Sphere{Vec3f::make(0, 0, -1), 0.5f, center});
    spheres.push_back(Sphere{Vec3f::make(-1, 0, -1), 0.5f, left});
    spheres.push_back(Sphere{Vec3f::make(-1, 0, -1), -0.45f, left});
    spheres.push_back(Sphere{Vec3f::make(1, 0, -1), 0.5f, right});
}

void Scene::sceneImage21(DefaultPrng &r) {
    spheres.push_back(Sphere{Vec3f::make(0, -1000, 0), 1000, Material::lambertian(Vec3f::make(0.5f, 0.5f, 0.5f))});
    for (float a = -11; a < 11; a += 1) {
        for (float b = -11; b < 11; b += 1) {
            const float choose_mat = r.float32();
            const float cx = a + 0.9f * r.float32();
            const float cz = b + 0.9f * r.float32();
            const Vec3f center = Vec3f::make(cx, 0.2f, cz);
            if (choose_mat < 0.8f) {
                const float ar = r.float32(), ag = r.float32(), ab = r.float32();
                spheres.push_back(Sphere{center, 0.2f, Material::lambertian(Vec3f::make(ar, ag, ab))});
            } else if (choose_mat < 0.95f) {
                const float ar = r.float32(), ag = r.float32(), ab = r.float32();
                const float fuzziness = 0.5f * r.float32();
                spheres.push_back(Sphere{center, 0.2f, Material::metal(Vec3f::make(ar, ag, ab), fuzziness)});
            } else {
                spheres.push_back(Sphere{center, 0.2f, Material::dielectric(1.5f)});
            }
        }
    }
    spheres.push_back(Sphere{Vec3f::make(0, 1, 0), 1.0f, Material::dielectric(1.5f)});
    spheres.push_back(Sphere{Vec3f::make(-4, 1, 0), 1.0f, Material::lambertian(Vec3f::make(0.4f, 0.2f, 0.1f))});
    spheres.push_back(Sphere{Vec3f::make(4, 1, 0), 1.0f, Material::metal(Vec3f::make(0.7f, 0.6f, 0.5f), 0.0f)});
}

int Scene::setupScene(DefaultPrng &r, int n) {
    if (n == 18 || n == 19 || n == 20) sceneImage18_19_20();
    else if (n == 21) sceneImage21(r);
    else return SceneNotAvailable;
    return Ok;
}

void Scene::sceneSynthetic(uint32_t n, uint64_t seed) {
    DefaultPrng r(seed, Prng::Xoshiro256); // this scene's own definition (not a reference scene)
    spheres.push_back(Sphere{Vec3f::make(0, -1000, 0), 1000, Material::lambertian(Vec3f::make(0.5f, 0.5f, 0.5f))});
    while (spheres.size() < n) {
        const float choose_mat = r.float32();
        const float radius = 0.05f + 0.25f * r.float32();
        const float cx = -50.0f + 100.0f * r.float32();
        const float cz = -50.0f + 100.0f * r.float32();
        const Vec3f center = Vec3f::make(cx, radius, cz);
        if (choose_mat < 0.8f) {
            const float ar = r.float32(), ag = r.float32(), ab = r.float32();
            spheres.push_back(Sphere{center, radius, Material::lambertian(Vec3f::make(ar, ag, ab))});
        } else if (choose_mat < 0.95f) {
            const float ar = r.float32(), ag = r.float32(), ab = r.float32();
            const float fuzziness = 0.5f * r.float32();
            spheres.push_back(Sphere{center, radius, Material::metal(Vec3f::make(ar, ag, ab), fuzziness)});
        } else {
            spheres.push_back(Sphere{center, radius, Material::dielectric(1.5f)});
        }
    }
}

std::vector<pt_sphere> Scene::flatten() const {
    std::vector<pt_sphere> out(spheres.size());
    for (size_t i = 0; i < spheres.size(); ++i) {
        const Sphere &s = spheres[i];
        out[i] = pt_sphere{s.center.x, s.center.y, s.center.z, s.radius, s.material.kind,
                           s.material.albedo.x, s.material.albedo.y, s.material.albedo.z, s.material.param};
    }
    return out;
}

// ---- utils.zig / multithreading.zig ----------------------------------------------------------
uint8_t numDigits(size_t num) {
    size_t x = num;
    uint8_t count = 0;
    while (x != 0) { x /= 10; count += 1; }
    return count;
}
float degreesToRadians(float deg) { return deg * 3.14159265358979323846f / 180.0f; }
int colNumToString(size_t max_value, uint32_t value, char *out, size_t cap) {
    if (value > max_value) return ValueExceedsMaxValue;
    if (cap < 4) return NoSpaceLeft;
    if (value <= 9) snprintf(out, cap, "  %u", value);
    else if (value <= 99) snprintf(out, cap, " %u", value);
    else snprintf(out, cap, "%u", value);
    return Ok;
}
size_t chunkSize(size_t num_pixels, size_t num_threads) {
    const size_t n = num_pixels / num_threads, rem = num_pixels % num_threads;
    return rem > 0 ? n + 1 : n;
}

// ---- camera.zig -------------------------------------------------------------------------------
Camera Camera::make(Vec3f lookfrom, Vec3f lookat, Vec3f vup, float vfov, float aspect_ratio, float aperture,
                    float focus_dist) {
    const float theta = degreesToRadians(vfov);
    // Zig's f32 tan is unavailable here: tan in double, rounded to f32 (may differ from
    // Zig's tan32 in the last bit; unpinned by the reference's tests).
    const float h = (float)tan((double)(theta / 2));
    const float vh = 2 * h;
    const float vw = vh * aspect_ratio;
    const Vec3f w = lookfrom.sub(lookat).unitVector();
    const Vec3f u = vup.cross(w).unitVector();
    const Vec3f v = w.cross(u);
    const Vec3f horizontal = u.mul(vw).mul(focus_dist);
    const Vec3f vertical = v.mul(vh).mul(focus_dist);
    const Vec3f half_vw = horizontal.mul(0.5f);
    const Vec3f half_vh = vertical.mul(0.5f);
    const Vec3f llc = lookfrom.sub(half_vw).sub(half_vh).sub(w.mul(focus_dist));
    Camera c;
    c.aperture = aperture; c.aspect_ratio = aspect_ratio; c.focus_dist = focus_dist;
    c.horizontal = horizontal; c.lens_radius = aperture / 2.0f;
    c.lookat = lookat; c.lookfrom = lookfrom; c.lower_left_corner = llc;
    c.u = u; c.v = v; c.vertical = vertical;
    c.viewport_height = vh; c.viewport_width = vw; c.vup = vup;
    return c;
}

int Camera::preset(int num_scene, float aspect_ratio, Camera *out) {
    Vec3f lookfrom = Vec3f::make(0.0f, 0.0f, 0.0f);
    const Vec3f lookat = Vec3f::make(0.0f, 0.0f, -1.0f);
    const Vec3f up = Vec3f::make(0.0f, 1.0f, 0.0f);
    float vfov = 20.0f, focal_dist, aperture = 1.0f;
    if (num_scene == 18) {
        lookfrom = Vec3f::make(-2.0f, 2.0f, 1.0f); vfov = 90.0f;
        focal_dist = lookfrom.sub(lookat).length(); aperture = 0.2f;
    } else if (num_scene == 19) {
        lookfrom = Vec3f::make(-2.0f, 2.0f, 1.0f);
        focal_dist = lookfrom.sub(lookat).length(); aperture = 0.2f;
    } else if (num_scene == 20) {
        lookfrom = Vec3f::make(3.0f, 3.0f, 2.0f);
        focal_dist = lookfrom.sub(lookat).length(); aperture = 2.0f;
    } else if (num_scene == 21) {
        lookfrom = Vec3f::make(13.0f, 2.0f, 3.0f);
        focal_dist = 10.0f; aperture = 0.1f;
    } else {
        return SceneNotAvailable;
    }
    *out = make(lookfrom, lookat, up, vfov, aspect_ratio, aperture, focal_dist);
    return Ok;
}

pt_camera Camera::flatten() const {
    pt_camera c;
    const Vec3f *src[6] = {&lookfrom, &u, &v, &lower_left_corner, &horizontal, &vertical};
    float *dst[6] = {c.lookfrom, c.u, c.v, c.lower_left_corner, c.horizontal, c.vertical};
    for (int i = 0; i < 6; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    c.lens_radius = lens_radius;
    return c;
}

// ---- render_targets/ppm_image.zig ----------------------------------------------------------------
Image Image::make(size_t width, float aspect_ratio, size_t max_px_value) {
    Image im;
    const size_t height = (size_t)((float)width / aspect_ratio);
    im.aspect_ratio = aspect_ratio;
    im.blend_start = Vec3f::make(1.0f, 1.0f, 1.0f);
    im.blend_stop = Vec3f::make(0.5f, 0.7f, 1.0f);
    im.px_size = 3 * numDigits(max_px_value) + 2 + 1;
    im.header_size = 2 + 1 + numDigits(width) + 1 + numDigits(height) + 1 + numDigits(max_px_value) + 1;
    im.data_size = im.px_size * width * height;
    im.height = height; im.max_px_value = max_px_value; im.num_pixels = width * height;
    im.size = im.header_size + im.data_size; im.width = width;
    return im;
}
std::string Image::header() const {
    char buf[96];
    snprintf(buf, sizeof buf, "P3\n%zu %zu\n%zu\n", width, height, max_px_value);
    return std::string(buf);
}
static uint32_t floatToU8(float x) { // @floatToInt(u8, x); out of range clamps (UB in the reference's ReleaseFast)
    if (!(x > 0.0f)) return 0;
    if (x >= 255.0f) return 255;
    return (uint32_t)x;
}
int Image::rgbToAscii(Vec3f col, char *out, size_t cap) const {
    if (max_px_value != 255) return PT_ERR_UNSUPPORTED;
    if (cap < px_size + 1) return NoSpaceLeft;
    char r[8], g[8], b[8];
    int rc;
    if ((rc = colNumToString(max_px_value, floatToU8(col.x), r, sizeof r)) != Ok) return rc;
    if ((rc = colNumToString(max_px_value, floatToU8(col.y), g, sizeof g)) != Ok) return rc;
    if ((rc = colNumToString(max_px_value, floatToU8(col.z), b, sizeof b)) != Ok) return rc;
    snprintf(out, cap, "%s %s %s\n", r, g, b);
    return Ok;
}

// ---- renderers/ppm_image.zig -----------------------------------------------------------------------
namespace ppm_renderer {

static int to_pt_config(const Config &cfg, pt_config *c) {
    if (!cfg.camera || !cfg.img || !cfg.ray_tracer_config || !cfg.slice || !cfg.scene) return PT_ERR_BAD_ARG;
    const Image &img = *cfg.img;
    const RayTracerConfig &rt = *cfg.ray_tracer_config;
    memset(c, 0, sizeof *c);
    c->width = (uint32_t)img.width; c->height = (uint32_t)img.height;
    c->subpixels = rt.subpixels; c->rays_per_subsample = rt.rays_per_subsample;
    c->rebounds = rt.rebounds; c->mode = cfg.deterministic ? 1u : 0u;
    c->seed = cfg.seed; c->t_min = rt.t_min; c->t_max = rt.t_max;
    c->blend_start[0] = img.blend_start.x; c->blend_start[1] = img.blend_start.y; c->blend_start[2] = img.blend_start.z;
    c->blend_stop[0] = img.blend_stop.x; c->blend_stop[1] = img.blend_stop.y; c->blend_stop[2] = img.blend_stop.z;
    c->idx_start = (uint32_t)cfg.istart; c->idx_stop = (uint32_t)cfg.istop;
    c->chunk = cfg.chunk;
    c->illumination = cfg.illumination;
    c->sampler = cfg.sampler;
    c->max_px_value = (uint32_t)img.max_px_value;
    c->tile_count = 1;
    return PT_OK;
}

int render(const Config &cfg) {
    if (!cfg.gpu) return PT_ERR_BAD_ARG;
    pt_config c;
    int rc = to_pt_config(cfg, &c);
    if (rc != PT_OK) return rc;
    const std::vector<pt_sphere> sp = cfg.scene->flatten();
    rc = ptgpu_set_scene(cfg.gpu, sp.data(), (uint32_t)sp.size());
    if (rc != PT_OK) return rc;
    const pt_camera cam = cfg.camera->flatten();
    rc = ptgpu_set_camera(cfg.gpu, &cam);
    if (rc != PT_OK) return rc;
    return ptgpu_render_ppm(cfg.gpu, &c, cfg.slice, cfg.slice_len, (uint32_t)cfg.img->header_size, (uint32_t)cfg.img->px_size);
}

int renderMultiGpu(const Config &cfg, ptgpu_multi *multi, int split) {
    if (!multi) return PT_ERR_BAD_ARG;
    pt_config c;
    int rc = to_pt_config(cfg, &c);
    if (rc != PT_OK) return rc;
    const std::vector<pt_sphere> sp = cfg.scene->flatten();
    rc = ptgpu_multi_set_scene(multi, sp.data(), (uint32_t)sp.size());
    if (rc != PT_OK) return rc;
    const pt_camera cam = cfg.camera->flatten();
    rc = ptgpu_multi_set_camera(multi, &cam);
    if (rc != PT_OK) return rc;
    return ptgpu_multi_render_ppm(multi, &c, split, cfg.slice, cfg.slice_len, (uint32_t)cfg.img->header_size, (uint32_t)cfg.img->px_size);
}

} // namespace ppm_renderer

std::string filepath(size_t num_scene, const RayTracerConfig &cfg, const Image &img, size_t seed) {
    char buf[256];
    snprintf(buf, sizeof buf, "images/scene%zu-w%zu-subpx%u-spp%u-depth%u-seed%zu.ppm", num_scene, img.width, cfg.subpixels,
             cfg.rays_per_subsample, cfg.rebounds, seed);
    return std::string(buf);
}

} // namespace pt

// ---- C view ---------------------------------------------------------------------------------------
using namespace pt;

extern "C" int pthost_setup_scene(int num_scene, uint64_t seed, pt_sphere *out, uint32_t cap) {
    return pthost_setup_scene_prng(num_scene, seed, (uint32_t)Prng::Default, out, cap);
}
extern "C" int pthost_setup_scene_prng(int num_scene, uint64_t seed, uint32_t prng, pt_sphere *out, uint32_t cap) {
    if (prng > (uint32_t)Prng::Xoshiro256) return PT_ERR_BAD_ARG;
    DefaultPrng prng_(seed, (Prng)prng); // main.zig:100
    Scene scene;
    const int rc = scene.setupScene(prng_, num_scene);
    if (rc != Ok) return rc;
    if (scene.spheres.size() > cap) return NoSpaceLeft;
    const std::vector<pt_sphere> f = scene.flatten();
    memcpy(out, f.data(), f.size() * sizeof(pt_sphere));
    return (int)f.size();
}
extern "C" int pthost_scene_synthetic(uint32_t n, uint64_t seed, pt_sphere *out, uint32_t cap) {
    if (n < 1 || n > cap) return NoSpaceLeft;
    Scene scene;
    scene.sceneSynthetic(n, seed);
    const std::vector<pt_sphere> f = scene.flatten();
    memcpy(out, f.data(), f.size() * sizeof(pt_sphere));
    return (int)f.size();
}
extern "C" int pthost_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov, float aspect_ratio,
                                 float aperture, float focus_dist, pt_camera *out) {
    if (!out) return PT_ERR_BAD_ARG;
    *out = Camera::make(Vec3f::make(lookfrom[0], lookfrom[1], lookfrom[2]), Vec3f::make(lookat[0], lookat[1], lookat[2]),
                        Vec3f::make(vup[0], vup[1], vup[2]), vfov, aspect_ratio, aperture, focus_dist).flatten();
    return Ok;
}
extern "C" int pthost_camera_preset(int num_scene, float aspect_ratio, pt_camera *out) {
    Camera c;
    const int rc = Camera::preset(num_scene, aspect_ratio, &c);
    if (rc != Ok) return rc;
    *out = c.flatten();
    return Ok;
}
static Image to_image(const pthost_image *p) { return Image::make(p->width, p->aspect_ratio, p->max_px_value); }
extern "C" int pthost_image_new(uint32_t width, float aspect_ratio, uint32_t max_px_value, pthost_image *out) {
    if (!out) return PT_ERR_BAD_ARG;
    const Image im = Image::make(width, aspect_ratio, max_px_value);
    out->aspect_ratio = im.aspect_ratio;
    out->width = (uint32_t)im.width; out->height = (uint32_t)im.height; out->max_px_value = (uint32_t)im.max_px_value;
    out->px_size = (uint32_t)im.px_size; out->header_size = (uint32_t)im.header_size;
    out->num_pixels = im.num_pixels; out->data_size = im.data_size; out->size = im.size;
    out->blend_start[0] = im.blend_start.x; out->blend_start[1] = im.blend_start.y; out->blend_start[2] = im.blend_start.z;
    out->blend_stop[0] = im.blend_stop.x; out->blend_stop[1] = im.blend_stop.y; out->blend_stop[2] = im.blend_stop.z;
    return Ok;
}
extern "C" int pthost_image_header(const pthost_image *img, char *buf, size_t cap) {
    const std::string h = to_image(img).header();
    if (h.size() + 1 > cap) return NoSpaceLeft;
    memcpy(buf, h.c_str(), h.size() + 1);
    return (int)h.size();
}
extern "C" int pthost_rgb_to_ascii(const pthost_image *img, const float col[3], char *out, size_t cap) {
    return to_image(img).rgbToAscii(Vec3f::make(col[0], col[1], col[2]), out, cap);
}
extern "C" uint32_t pthost_num_digits(uint64_t num) { return numDigits((size_t)num); }
extern "C" int pthost_col_num_to_string(uint64_t max_value, uint32_t value, char *out, size_t cap) {
    return colNumToString((size_t)max_value, value, out, cap);
}
extern "C" uint64_t pthost_chunk_size(uint64_t num_pixels, uint64_t num_threads) { return chunkSize(num_pixels, num_threads); }

extern "C" int pthost_render_scene_ppm(ptgpu_ctx *gpu, int num_scene, uint32_t width, uint32_t subpixels, uint32_t rays_per_subsample,
                                       uint32_t rebounds, uint64_t seed, int deterministic, uint8_t *slice, size_t slice_len) {
    // main.zig:78-182 without the prompts
    const Image img = Image::make(width, 16.0f / 9.0f, 255);
    const RayTracerConfig rt{subpixels, 0.1f, 1000.0f, rebounds, rays_per_subsample};
    DefaultPrng prng(seed);
    Scene scene;
    int rc = scene.setupScene(prng, num_scene);
    if (rc != Ok) return rc;
    Camera camera;
    rc = Camera::preset(num_scene, img.aspect_ratio, &camera);
    if (rc != Ok) return rc;
    if (slice_len < img.size) return NoSpaceLeft;
    ppm_renderer::Config cfg{gpu, &camera, &img, 0, img.num_pixels, seed, &rt, slice, slice_len, &scene};
    cfg.deterministic = deterministic != 0;
    rc = ppm_renderer::render(cfg);
    if (rc != Ok) return rc;
    const std::string h = img.header();
    memcpy(slice, h.data(), h.size()); // main.zig:180-182
    return Ok;
}
extern "C" int pthost_render_scene_ppm_multi(ptgpu_multi *multi, int split, int num_scene, uint32_t width, uint32_t subpixels,
                                             uint32_t rays_per_subsample, uint32_t rebounds, uint64_t seed, int deterministic, uint32_t chunk,
                                             uint8_t *slice, size_t slice_len) {
    // main.zig:78-182 without the prompts, the thread block replaced by the GPUs of `multi`
    const Image img = Image::make(width, 16.0f / 9.0f, 255);
    const RayTracerConfig rt{subpixels, 0.1f, 1000.0f, rebounds, rays_per_subsample};
    DefaultPrng prng(seed);
    Scene scene;
    int rc = scene.setupScene(prng, num_scene);
    if (rc != Ok) return rc;
    Camera camera;
    rc = Camera::preset(num_scene, img.aspect_ratio, &camera);
    if (rc != Ok) return rc;
    if (slice_len < img.size) return NoSpaceLeft;
    ppm_renderer::Config cfg{nullptr, &camera, &img, 0, img.num_pixels, seed, &rt, slice, slice_len, &scene};
    cfg.deterministic = deterministic != 0;
    cfg.chunk = chunk;
    rc = ppm_renderer::renderMultiGpu(cfg, multi, split);
    if (rc != Ok) return rc;
    const std::string h = img.header();
    memcpy(slice, h.data(), h.size()); // main.zig:180-182
    return Ok;
}
extern "C" int pthost_filepath(uint64_t num_scene, uint32_t subpixels, uint32_t rays_per_subsample, uint32_t rebounds, uint32_t width,
                               uint64_t seed, char *buf, size_t cap) {
    RayTracerConfig rt{subpixels, 0.1f, 1000.0f, rebounds, rays_per_subsample};
    const Image img = Image::make(width, 16.0f / 9.0f, 255);
    const std::string p = filepath((size_t)num_scene, rt, img, (size_t)seed);
    if (p.size() + 1 > cap) return NoSpaceLeft;
    memcpy(buf, p.c_str(), p.size() + 1);
    return (int)p.size();
}

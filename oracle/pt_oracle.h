/*
 * pt_oracle.h — CPU oracle for the path-tracing hot path of jackdbd/path-tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under path_tracer_amd/ or include/ may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity status: the reference is Zig 0.9.1 and cannot be built or
 * run in this image (no zig toolchain; deps fetched from the network).  This oracle
 * is a plain-C restatement, function by function, of the reference's arithmetic
 * (file:line cited at each function).  It is pinned by every known-answer test the
 * reference's own test blocks hold for this path (tests/test_oracle_kats.py).  The
 * reference's tests pin NOTHING about image-level output or RNG streams, so:
 *   - image-level parity is "pinned by restatement + reference KATs only";
 *   - the Zig std pieces the path uses (std.rand DefaultPrng/SplitMix64,
 *     Random.float(f32), math.pow) are restated from their published algorithms,
 *     recalled, NOT verifiable against a Zig build here.  The scene-construction stream IS
 *     pinned by the one image the reference holds (images/demo.png: Xoroshiro128+, low 32
 *     bits, the draw order of scene.zig:61-68; tests/test_demo_pin.py); the per-sample
 *     jitter / scatter streams of a real reference binary remain "parity unpinned".
 *
 * PODs are layout-identical to include/ptgpu.h's (declared again here so that the
 * oracle depends on nothing of the product).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_sphere {
    float cx, cy, cz, radius;
    uint32_t kind; /* 0 lambertian, 1 metal, 2 dielectric */
    float ar, ag, ab;
    float param;
} orc_sphere;

typedef struct orc_camera {
    float lookfrom[3], u[3], v[3], lower_left_corner[3], horizontal[3], vertical[3];
    float lens_radius;
} orc_camera;

typedef struct orc_config {
    uint32_t width, height;
    uint32_t subpixels, rays_per_subsample;
    uint32_t rebounds;
    uint32_t mode; /* 0 stochastic (Philox path streams), 1 deterministic (xi = 0.5), 2 Zig-RNG emulation */
    uint64_t seed;
    float t_min, t_max;
    float blend_start[3], blend_stop[3];
    uint32_t idx_start, idx_stop;
    uint32_t sample_start, sample_stop;
    uint32_t chunk;
    uint32_t max_px_value;
    uint32_t tile_pixels, tile_count, tile_index; /* interleaved tile ownership (multi-GPU split) */
    uint32_t illumination; /* 0 radiance (illumination.zig:76), 1 colorNormal (:42), 2 colorAlbedo (:60) */
    uint32_t sampler;      /* ORC_SAMPLER_*: how the unit-disk and unit-ball vectors are drawn (ignored in deterministic mode) */
    uint32_t reserved_;    /* 0 */
} orc_config;

/* ORC_SAMPLER_REJECTION: the reference's rejection loops (camera.zig:113-120, material.zig:27-34).
 * ORC_SAMPLER_DIRECT: NOT in the reference — the same two distributions drawn without a loop,
 * in operations gcc and the GPU evaluate identically; an opt-in whose statistical equivalence the tests show. */
enum { ORC_SAMPLER_REJECTION = 0, ORC_SAMPLER_DIRECT = 1 };

typedef struct orc_image {
    float aspect_ratio;
    uint32_t width, height, max_px_value;
    uint32_t px_size, header_size;
    uint64_t num_pixels, data_size, size;
    float blend_start[3], blend_stop[3];
} orc_image;

typedef struct orc_stats {
    uint64_t paths, rays, sphere_tests, not_unit;
} orc_stats;

enum { ORC_OK = 0, ORC_ERR_BAD_ARG = -1, ORC_ERR_DIRECTION_NOT_UNIT = -5, ORC_ERR_VALUE_EXCEEDS_MAX = -6,
       ORC_ERR_SCENE_NOT_AVAILABLE = -7, ORC_ERR_NO_SPACE = -8, ORC_ERR_UNSUPPORTED = -9 };

enum { ORC_MODE_PHILOX = 0, ORC_MODE_DETERMINISTIC = 1, ORC_MODE_ZIG = 2 };

/* ---- Zig std.rand emulation ------------------------------------------------
 * Two DefaultPrng generations (pt_oracle.c has the details): Xoroshiro128+ is pinned by std's own known-answer
 * sequence AND by the reference's images/demo.png (scene 21, seed 456); Xoshiro256++ (Zig 0.9.1 as recalled) by
 * its published vector only.  The default is the generator the reference's artefact was made with. */
enum { ORC_PRNG_XOROSHIRO128 = 0, ORC_PRNG_XOSHIRO256 = 1, ORC_PRNG_DEFAULT = ORC_PRNG_XOROSHIRO128 };
typedef struct orc_zigrng { uint64_t s[4]; uint32_t kind, pad_; } orc_zigrng;
void orc_zig_seed(orc_zigrng *r, uint64_t seed);     /* DefaultPrng.init(seed), ORC_PRNG_DEFAULT */
void orc_zig_seed_kind(orc_zigrng *r, uint64_t seed, uint32_t kind);
uint64_t orc_zig_next(orc_zigrng *r);                /* Xoroshiro128.next() / Xoshiro256.next() */
float orc_zig_float(orc_zigrng *r);                  /* Random.float(f32)      */
uint64_t orc_splitmix64_next(uint64_t *state);

/* ---- Philox4x32-10 ------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_u32_to_float(uint32_t s); /* Random.float(f32)'s bit trick on one u32 */

/* ---- vec3 / ray (vec3.zig, ray.zig) -------------------------------------- */
float orc_length(const float v[3]);
float orc_length_squared(const float v[3]);
void orc_unit_vector(const float v[3], float out[3]);
float orc_dot(const float a[3], const float b[3]);
void orc_cross(const float a[3], const float b[3], float out[3]);
void orc_point_at(const float o[3], const float d[3], float t, float out[3]);

/* ---- scene / camera / image construction (host side of the path) -------- */
int orc_scene_18_19_20(orc_sphere *out, uint32_t cap);                       /* scene.zig:32-50 */
int orc_scene_21(orc_zigrng *r, orc_sphere *out, uint32_t cap);              /* scene.zig:52-84 */
int orc_setup_scene(int n_scene, uint64_t seed, orc_sphere *out, uint32_t cap, orc_zigrng *rng_after); /* scene.zig:86-94 + main.zig:100-105 */
int orc_setup_scene_prng(int n_scene, uint64_t seed, uint32_t prng, orc_sphere *out, uint32_t cap, orc_zigrng *rng_after);
int orc_scene_synthetic(uint32_t n, uint64_t seed, orc_sphere *out, uint32_t cap); /* BASELINE config 5 */
void orc_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                    float aspect_ratio, float aperture, float focus_dist, orc_camera *out); /* camera.zig:48-97 */
int orc_camera_preset(int n_scene, float aspect_ratio, orc_camera *out);     /* main.zig:107-136 */
void orc_image_new(uint32_t width, float aspect_ratio, uint32_t max_px_value, orc_image *out); /* render_targets/ppm_image.zig:59-81 */
int orc_image_header(const orc_image *img, char *buf, size_t cap);           /* render_targets/ppm_image.zig:84-88 */
int orc_rgb_to_ascii(const orc_image *img, const float col[3], char *out, size_t cap); /* render_targets/ppm_image.zig:93-103 */
uint32_t orc_num_digits(uint64_t num);                                       /* utils.zig:12-20 */
int orc_col_num_to_string(uint64_t max_value, uint32_t value, char *out, size_t cap); /* utils.zig:62-84 */
float orc_degrees_to_radians(float deg);                                     /* utils.zig:113-115 */
uint64_t orc_chunk_size(uint64_t num_pixels, uint64_t num_threads);          /* multithreading.zig:40-44 */

/* ---- per-function hot-path pieces (for per-function parity vectors) ------ */
/* xi: caller-supplied uniform draws consumed in order; *n_used returns how many. */
void orc_cast_ray(const orc_camera *cam, float s, float t, const float *xi, uint32_t n_xi, uint32_t *n_used, float o[3], float d[3]); /* camera.zig:99-120 */
int orc_sphere_hit(const orc_sphere *sp, const float o[3], const float d[3], float t_min, float t_max,
                   float *t, float p[3], float n[3]);                        /* sphere.zig:63-103 */
int orc_scene_hit(const orc_sphere *sp, uint32_t n, const float o[3], const float d[3], float t_min, float t_max,
                  float *t, float p[3], float nrm[3]);                       /* scene.zig:96-110; returns index or -1 */
void orc_sample_disk(uint32_t sampler, const float *xi, uint32_t n_xi, uint32_t *n_used, float out[3]); /* camera.zig:113-120 / direct */
void orc_sample_ball(uint32_t sampler, const float *xi, uint32_t n_xi, uint32_t *n_used, float out[3]); /* material.zig:27-34 / direct */
void orc_sample_batch(int which /* 0 disk, 1 ball */, uint32_t sampler, const float *xi, uint32_t n, uint32_t stride, float *out4 /* p, n_used */);
void orc_scene_hit_batch(const orc_sphere *sp, uint32_t n_sp, const float *rays, uint32_t n, float t_min, float t_max, float *out8); /* scene.zig:96-110 on n rays */
void orc_reflect(const float d[3], const float n[3], float out[3]);          /* material.zig:59-65 */
int orc_refract(const float d[3], const float n[3], float n1_over_n2, float out[3]); /* material.zig:98-116; 0 = null */
float orc_schlick(float cosine, float refraction_index);                     /* material.zig:120-124 */
float orc_pow(float x, float y);                                             /* Zig std math.pow, integer-y path */
void orc_scatter(const orc_sphere *sp, const float d[3], const float p[3], const float n[3],
                 const float *xi, uint32_t n_xi, uint32_t *n_used, float atten[3], float so[3], float sd[3]); /* material.zig:43-51,79-88,129-167,191-198 */
int orc_lerp(const float u[3], const float p0[3], const float p1[3], float out[3]); /* illumination.zig:27-34 */

/* ---- the hot path -------------------------------------------------------- */
/* Per-pixel radiance SUMS over the selected samples (renderers/ppm_image.zig:49-69),
 * width*height*3 floats, pixels outside [idx_start,idx_stop) untouched.
 * nthreads <= 1: single thread.  nthreads > 1: the reference's split
 * (multithreading.zig:40-44, renderers/ppm_image.zig:100-108) over pthreads. */
int orc_render_sum(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg,
                   float *rgb_sum, uint32_t nthreads, orc_stats *stats);
/* mean = sum*(1/subpixels)*(1/rays_per_subsample) (renderers/ppm_image.zig:70). */
void orc_resolve_mean(const orc_config *cfg, const float *rgb_sum, float *rgb_mean);
/* The ASCII records render() writes (renderers/ppm_image.zig:70-82) for pixels
 * [idx_start, idx_stop) into slice[header_size + idx*px_size ..]. */
int orc_resolve_ppm(const orc_config *cfg, const orc_image *img, const float *rgb_sum, uint8_t *slice, size_t slice_len);
/* Whole render(): sum + resolve. */
int orc_render_ppm(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg, const orc_image *img,
                   uint8_t *slice, size_t slice_len, uint32_t nthreads, orc_stats *stats);
/* Radiance of ONE camera sample, recursive exactly as illumination.zig:76-90
 * (iterative = 0) or with the iterative bounce loop + innermost-first attenuation
 * unwinding the GPU kernel uses (iterative = 1); both must agree bit for bit. */
int orc_sample_radiance(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg,
                        uint32_t idx, uint32_t sample, int iterative, float rgb[3]);
/* Single-thread Zig-RNG render as main.zig:166-177 does it: the scene RNG stream
 * continues into the pixel jitter; scatter draws come from a separate
 * DefaultPrng.init(0) (material.zig:15).  Unpinned (see header comment). */
int orc_render_sum_zig(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg,
                       orc_zigrng *r, float *rgb_sum, orc_stats *stats);

#ifdef __cplusplus
}
#endif
#endif

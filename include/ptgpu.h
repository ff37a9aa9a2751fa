/*
 * ptgpu.h — C ABI of libptgpu, the MI355X (gfx950) path-tracing hot path.
 *
 * Drop-in boundary for jackdbd/path-tracer's renderer seam.  The reference has no
 * FFI; the seam is the Zig function pair
 *     render(cfg: Config) !void                  src/renderers/ppm_image.zig:30
 *     renderMultiThread(ctx: ThreadContext) !void src/renderers/ppm_image.zig:91
 * fed by Scene (src/scene.zig:19), Camera (src/camera.zig:18), Image
 * (src/render_targets/ppm_image.zig:28) and RayTracerConfig (src/raytracer.zig:17).
 * Zig structs / tagged unions have no defined layout, so what crosses here are the
 * `extern struct`-compatible PODs below (all little-endian, natural alignment).
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every host buffer it passes,
 *     the library owns every device buffer it allocates (freed by ptgpu_destroy);
 *   - every entry point returns an int status: PT_OK (0) or a negative PT_ERR_*;
 *   - one context drives one GPU; a context is not re-entrant (one host thread at a
 *     time per context); multi-GPU = ptgpu_multi_* below (one process, all GPUs of the
 *     node) or one process per GPU with the caller's own collective, see INTEGRATION.md;
 *   - there is NO CPU fallback: without a usable gfx950 device ptgpu_create fails.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
enum {
    PT_OK = 0,
    PT_ERR_BAD_ARG = -1,            /* null pointer, empty range, size mismatch      */
    PT_ERR_NO_DEVICE = -2,          /* no HIP device / wrong architecture            */
    PT_ERR_HIP = -3,                /* a HIP runtime call failed (ptgpu_last_hip)    */
    PT_ERR_OOM = -4,                /* device allocation failed (Zig: OutOfMemory)   */
    PT_ERR_DIRECTION_NOT_UNIT = -5, /* illumination.zig:29-31 DirectionIsNotUnitVector */
    PT_ERR_VALUE_EXCEEDS_MAX = -6,  /* utils.zig:63-65 ValueExceedsMaxValue          */
    PT_ERR_SCENE_NOT_AVAILABLE = -7,/* errors.zig:2  UserError.SceneNotAvailable     */
    PT_ERR_NO_SPACE = -8,           /* caller buffer too small (Zig: NoSpaceLeft)    */
    PT_ERR_UNSUPPORTED = -9,        /* e.g. max_px_value != 255                      */
    PT_ERR_SCRATCH_LIMIT = -10,     /* ptgpu_render_sum_device_noalloc: render does not fit its static scratch */
    PT_ERR_RCCL = -11               /* an RCCL call failed (ptgpu_multi_*)           */
};

/* ---- material kinds: the tags of material.zig:174-178 ------------------- */
enum { PT_LAMBERTIAN = 0, PT_METAL = 1, PT_DIELECTRIC = 2 };

/* One Sphere (sphere.zig:36-40) with its Material union (material.zig:174-189)
 * flattened: param = fuzziness for PT_METAL, refraction_index for PT_DIELECTRIC;
 * albedo is ignored for PT_DIELECTRIC (attenuation is (1,1,1), material.zig:135).
 * radius may be negative (scene.zig:48).  36 bytes. */
typedef struct pt_sphere {
    float cx, cy, cz;
    float radius;
    uint32_t kind;
    float ar, ag, ab;
    float param;
} pt_sphere;

/* The Camera fields castRay reads (camera.zig:99-106; built by Camera.new,
 * camera.zig:48-97).  76 bytes. */
typedef struct pt_camera {
    float lookfrom[3];
    float u[3];
    float v[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float lens_radius;
} pt_camera;

/* RayTracerConfig (raytracer.zig:17-23) + the Image fields render() reads
 * (renderers/ppm_image.zig:30-87) + the work range.  The reference's u8 counters
 * are widened to u32 (BASELINE configs need up to 4096 paths per pixel).
 *
 * Samples of a pixel are numbered s = sample*rays_per_subsample + rps in the
 * reference's loop order (renderers/ppm_image.zig:57-59), S = subpixels *
 * rays_per_subsample in total.  [sample_start, sample_stop) selects a sub-range
 * (sample-range split across GPUs); both 0 = all S samples.
 * `chunk` = samples per work item; each work item's samples are summed in order
 * and the partial sums of a pixel are then added in chunk order.  chunk = 0 (or
 * >= the range) reproduces the reference's single sequential sum exactly.
 * mode 1 = deterministic: every random draw returns 0.5 (pixel-centre rays, zero
 * lens offset, scatter vector (0.5,0.5,0.5), dielectric xi = 0.5).
 * Tile split (multi-GPU, large frames): with tile_pixels = T > 0 the range is cut into
 * runs of T consecutive pixel indices starting at idx_start (which must be a multiple
 * of T); run number t is rendered only by the owner with tile_index == t % tile_count,
 * so that cheap (sky) and expensive rows are dealt round-robin instead of in the
 * reference's contiguous bands (multithreading.zig:40-44).  Pixels of other owners are
 * left untouched. */
typedef struct pt_config {
    uint32_t width, height;                 /* Image.width/.height                  */
    uint32_t subpixels, rays_per_subsample; /* raytracer.zig:18,22                  */
    uint32_t rebounds;                      /* raytracer.zig:21                     */
    uint32_t mode;                          /* 0 stochastic (Philox), 1 deterministic */
    uint64_t seed;                          /* Philox key                           */
    float t_min, t_max;                     /* raytracer.zig:19-20; main.zig:87-88  */
    float blend_start[3], blend_stop[3];    /* render_targets/ppm_image.zig:69-70   */
    uint32_t idx_start, idx_stop;           /* renderers/ppm_image.zig:20-21 istart/istop */
    uint32_t sample_start, sample_stop;
    uint32_t chunk;
    uint32_t max_px_value;                  /* Image.max_px_value (255)             */
    uint32_t tile_pixels;                   /* tile split: pixels per tile, 0 = off */
    uint32_t tile_count, tile_index;        /* tile t of the range belongs to owner t % tile_count */
    uint32_t illumination;                  /* PT_ILLUM_*: which illumination.zig function colours a sample */
    uint32_t sampler;                       /* PT_SAMPLER_*: how unit-disk / unit-ball vectors are drawn (mode 0 only) */
    uint32_t reserved_;                     /* 0 */
} pt_config;

/* The reference draws its lens and scatter vectors with rejection loops (camera.zig:113-120: the unit disk;
 * material.zig:27-34: [0,1)^3 kept inside the unit ball, i.e. the POSITIVE OCTANT of the ball).
 *   PT_SAMPLER_REJECTION (default)  those loops, try by try, as the reference runs them;
 *   PT_SAMPLER_DIRECT               NOT in the reference: the same two distributions drawn without a loop from one
 *                                   try's worth of draws (sqrt(xi0) (cos, sin)(2 pi xi1); a uniform
 *                                   direction of the octant times xi2^(1/3)), in f32 operations the oracle repeats
 *                                   bit for bit.  Other numbers per path, the same image in distribution (the tests
 *                                   show the equivalence); a wave no longer waits for its slowest lane's retries.
 * Deterministic mode ignores the field (every draw is 0.5 and goes through the reference's expressions). */
enum { PT_SAMPLER_REJECTION = 0, PT_SAMPLER_DIRECT = 1 };

/* The reference has three per-sample colour functions; render() calls radiance and keeps the
 * other two as commented-out alternatives (renderers/ppm_image.zig:64-66).  The first-hit ones
 * trace one ray per sample and are cheap deterministic views for regression tests. */
enum {
    PT_ILLUM_RADIANCE = 0, /* illumination.zig:76-90  radiance (the path tracer)                      */
    PT_ILLUM_NORMAL = 1,   /* illumination.zig:42-51  colorNormal: (unit(p - (0,0,-1)) + 1) * 0.5 (sic:
                              the centre is the constant (0,0,-1), not the hit sphere's) or sky lerp  */
    PT_ILLUM_ALBEDO = 2    /* illumination.zig:60-73  colorAlbedo: albedo of the first hit ((1,1,1) for
                              a dielectric) or sky lerp                                                */
};

/* Counters of the last render on a context. */
typedef struct pt_stats {
    uint64_t paths;        /* camera samples traced                                  */
    uint64_t rays;         /* ray segments = closest-hit queries (Scene.is_hit calls)*/
    uint64_t sphere_tests; /* rays * n_spheres (what the reference's scan performs)  */
    uint64_t candidates;   /* exact Sphere.is_hit evaluations (after the LDS filter, in tree leaves, over candidate lists) */
    uint64_t not_unit;     /* lerp() directions failing illumination.zig:29          */
    uint64_t wave_steps;   /* ray steps executed by waves (rays / (64*wave_steps) = lane utilisation) */
    uint64_t flush_iters;  /* wave iterations of the exact-test loop / of the tree walk */
    double kernel_ms;      /* HIP-event time of the path-tracing kernels: primary pass + megakernel (of all segments, incl. their chunk sums, when segments > 1) */
    double total_ms;       /* kernel + chunk sums                                    */
    uint64_t segments;     /* megakernel launches the render took (PT_LIMIT_PARTIAL_BYTES, PT_LIMIT_PRIMARY_BYTES) */
    uint64_t primary_rays; /* of `rays`: camera rays whose closest hit came from their pixel's candidate list (primary pass), not from the scan */
    uint64_t list2_rays;   /* of `rays`: rays scattered off a pixel's main sphere whose closest hit came from that pixel's second-bounce list (primary pass) */
    double primary_ms;     /* of kernel_ms: the primary pass (candidate lists + camera rays + first radiance step), summed over the segments (0 beyond 64 segments) */
    uint64_t box_tests;    /* ray-box tests of the sphere-tree walk (two per inner node a ray visits); 0 for scenes traced by the scan */
    double lists_ms;       /* of primary_ms: the kernels that build the per-pixel candidate lists (pt_list_prep, pt_primary_lists) */
    double megakernel_ms;  /* of kernel_ms: pt_megakernel alone, summed over the segments (0 beyond 64 segments) */
    uint64_t primary_candidates; /* of `candidates`: exact Sphere.is_hit evaluations made by the primary pass over its lists */
} pt_stats;

typedef struct ptgpu_ctx ptgpu_ctx;

/* Create / destroy a context on HIP device `device` (replaces the thread set-up of
 * main.zig:142-161).  Fails with PT_ERR_NO_DEVICE when there is no gfx950 GPU. */
int ptgpu_create(int device, ptgpu_ctx **out);
int ptgpu_destroy(ptgpu_ctx *ctx);

/* Upload the scene (replaces passing *Scene, renderers/ppm_image.zig:25).  Builds
 * the LDS-tile filter table and the exact sphere/material tables on the device. */
int ptgpu_set_scene(ptgpu_ctx *ctx, const pt_sphere *spheres, uint32_t n);

/* Upload the camera (replaces passing *Camera, renderers/ppm_image.zig:18). */
int ptgpu_set_camera(ptgpu_ctx *ctx, const pt_camera *cam);

/* Radiance sums: for every pixel idx in [idx_start, idx_stop) the sum over the
 * selected samples of illumination.radiance (illumination.zig:76) of
 * camera.castRay (camera.zig:99) — the inner loops of render(),
 * renderers/ppm_image.zig:49-69.  Output: width*height*3 floats (row-major, RGB);
 * pixels outside the range are left untouched.
 *   _device: d_rgb_sum is device memory; work is enqueued on `stream`
 *            (a hipStream_t, NULL = default stream) and the call returns after the
 *            kernels are enqueued and the stream is synchronised.
 *   _host:   h_rgb_sum is host memory. */
int ptgpu_render_sum_device(ptgpu_ctx *ctx, const pt_config *cfg, float *d_rgb_sum, void *stream);
int ptgpu_render_sum_host(ptgpu_ctx *ctx, const pt_config *cfg, float *h_rgb_sum);

/* NO COUNTERPART IN THE REFERENCE.  ptgpu_render_sum_device for callers that cannot allocate device
 * memory (code under HIP graph capture set-up, tight serving loops, harnesses that forbid hipMalloc):
 * same arguments, same result bit for bit, but the kernels' working memory (counters, hit-id stack,
 * chunk partial sums) is a static device array of ptgpu_noalloc_scratch_bytes() = 128 MiB per GPU
 * inside the library instead of hipMalloc'ed buffers, and the call performs no hipMalloc / hipFree.
 * The hit-id stack (rebounds x resident threads x 4 B) must fit, else PT_ERR_SCRATCH_LIMIT: scenes traced by the
 * scan keep 327 680 threads resident (5 workgroups per CU: 1.25 MiB per bounce, depth 50 = 62.5 MiB, up to depth 102
 * fits), scenes traced through the sphere tree (256 spheres and more) 393 216 (6 per CU: 1.5 MiB per bounce, depth
 * 50 = 75 MiB, up to depth 85); what remains holds the chunk partials, in as many segments as it takes (chunk = 0
 * needs none).  The scratch is shared by every context on the same GPU and
 * keeps the last call's contents: calls that overlap on one GPU race — serialise them. */
int ptgpu_render_sum_device_noalloc(ptgpu_ctx *ctx, const pt_config *cfg, float *d_rgb_sum, void *stream);
uint64_t ptgpu_noalloc_scratch_bytes(void);

/* The resolve of renderers/ppm_image.zig:70-82 on the device: mean =
 * sum*(1/subpixels)*(1/rays_per_subsample), *max_px_value, truncating u8, the
 * 12-byte "%3d %3d %3d\n" record of render_targets/ppm_image.zig:93-103.
 * d_records receives (idx_stop-idx_start)*12 bytes (record of pixel idx at
 * (idx-idx_start)*12) and must be 4-byte aligned (records are written as three dwords);
 * either output may be NULL.  d_rgb8 receives 3 bytes/pixel.
 * A colour above max_px_value (col > 255: the reference asserts col <= 255,
 * renderers/ppm_image.zig:76, and its @floatToInt(u8) is undefined from 256 on) is
 * written clamped to 255 and the call returns PT_ERR_VALUE_EXCEEDS_MAX after writing
 * every record. */
int ptgpu_resolve_device(ptgpu_ctx *ctx, const pt_config *cfg, const float *d_rgb_sum,
                         uint8_t *d_records, uint8_t *d_rgb8, void *stream);

/* Whole render() replacement on host buffers: traces [idx_start, idx_stop) and
 * writes the pixel records into slice[header_size + idx*px_size ..], exactly the
 * bytes render() writes (renderers/ppm_image.zig:81-82).  The caller writes the
 * header afterwards (main.zig:180-182).  px_size must be 12 (max_px_value 255).
 * PT_ERR_VALUE_EXCEEDS_MAX as for ptgpu_resolve_device (records are written). */
int ptgpu_render_ppm(ptgpu_ctx *ctx, const pt_config *cfg, uint8_t *slice, size_t slice_len,
                     uint32_t header_size, uint32_t px_size);

/* Mean radiance (after the two reciprocal multiplies of renderers/ppm_image.zig:70)
 * as width*height*3 floats on the host; pixels outside the range untouched. */
int ptgpu_render_mean_host(ptgpu_ctx *ctx, const pt_config *cfg, float *h_rgb_mean);

/* Per-function device evaluation, used by the parity tests to pin each device
 * function against the oracle on arrays of inputs.
 * `fn` selects the function (PT_FN_*); `in`/`out` are host float arrays with
 * `n` records of the per-function strides given beside each id (xi = the uniform
 * draws the function consumes, in order; n_used = how many it consumed). */
enum {
    PT_FN_CAST_RAY = 0,    /* camera.zig:99     in: s,t,xi[16] (18)               out: o,d,n_used (7)      */
    PT_FN_SPHERE_HIT = 1,  /* sphere.zig:63     in: o,d,c,r,tmin,tmax (12)        out: hit,t,p,n (8)       */
    PT_FN_SCENE_HIT = 2,   /* scene.zig:96      in: o,d (6) (uses the ctx scene)  out: id,t,p,n (8)        */
    PT_FN_SCATTER = 3,     /* material.zig:191  in: d,p,n,kind,param,xi[16] (27)  out: scattered d,n_used (4) */
    PT_FN_REFRACT = 4,     /* material.zig:98   in: d,n,eta (7)                   out: ok,r (4)            */
    PT_FN_SCHLICK = 5,     /* material.zig:120  in: cos,ri (2)                    out: prob (1)            */
    PT_FN_REFLECT = 6,     /* material.zig:59   in: d,n (6)                       out: r (3)               */
    PT_FN_LERP = 7,        /* illumination.zig:27 in: d,p0,p1 (9)                 out: err,rgb (4)         */
    PT_FN_PHILOX = 8,      /* RNG: in: seed_lo,seed_hi,pixel,sample,block (5 x u32 bits) out: 4 floats     */
    PT_FN_SCENE_HIT_FILTERED = 9, /* scene.zig:96 as the megakernel evaluates it (LDS filter + exact pass): in/out as PT_FN_SCENE_HIT */
    PT_FN_FILTER_BIT = 10, /* the LDS filter's verdict for one (ray, sphere) pair: in: o,d,c,r (10)  out: candidate bit,
                              reference discriminant of sphere.zig:74, candidate bit of the shared-ordinate form of the
                              test (3).  Invariant: discriminant >= 0 => both bits == 1 */
    PT_FN_SCENE_HIT_TREE = 11, /* scene.zig:96 through the sphere tree of large scenes (PT_LIMIT_TREE_MIN_SPHERES): in/out as
                              PT_FN_SCENE_HIT; PT_ERR_UNSUPPORTED when the uploaded scene has no tree */
    PT_FN_SAMPLE_DISK = 12, /* camera.zig:113-120 or its direct equivalent: in: sampler (PT_SAMPLER_*), xi[16] (17)  out: p, n_used (4) */
    PT_FN_SAMPLE_BALL = 13, /* material.zig:27-34 or its direct equivalent: in: sampler, xi[16] (17)                  out: p, n_used (4) */
    PT_FN_TREE_BOX_SAFETY = 14 /* what the sphere tree rests on (scene.zig:96-110 must see every sphere the exact test accepts): the ray
                              visits every node; for each sphere Sphere.is_hit accepts at t, every ancestor's box must pass the walk's box
                              test on (t_min, t).  in: o,d (6)  out: spheres accepted, (sphere, box) pairs checked, pairs FAILED (must be 0), 0 */
};
int ptgpu_debug_eval(ptgpu_ctx *ctx, int fn, const float *in, float *out, uint32_t n,
                     float t_min, float t_max);

/* Proof by exhaustion of the library's lean correctly rounded f32 sequences (csrc/pt_device.h: v_sqrt_f32 / v_rcp_f32 plus
 * one FMA correction instead of the compiler's longer expansions): all 2^32 bit patterns through the lean sequence and
 * through the compiler's expression; *mismatches must come back 0.  which: 0 sqrtf(x), 1 1.0f / x, 2 1.0f / sqrtf(x)
 * (vec3.zig:61-64 unitVector), 3 the unit-length test of illumination.zig:28 taken on |d|^2.  About 0.1 s each. */
int ptgpu_debug_exhaustive(ptgpu_ctx *ctx, int which, uint64_t *mismatches);

/* The property the primary pass's candidate lists rest on, checked sphere by sphere: runs the lists and the primary
 * pass of `cfg` (stochastic mode, the reference's samplers) with every camera ray and every second-bounce ray that is about to use a
 * list ALSO given the reference's exact test on every sphere of the scene (scene.zig:96-110 as written); each sphere that test accepts
 * must be in the list.  No frame is produced.  out: [0] spheres accepted for camera rays, [1] of them NOT in list A (must be 0),
 * [2] accepted for second rays, [3] NOT in list B (must be 0), [4] rays whose closest hit (index or t) differs from the full scan's
 * (must be 0), [5] camera rays checked, [6] second rays checked.  PT_ERR_UNSUPPORTED when the configuration has no primary pass.
 * O(paths x spheres): a test tool.  No counterpart in the reference. */
int ptgpu_debug_list_safety(ptgpu_ctx *ctx, const pt_config *cfg, uint64_t out[7]);

/* ---- multi-GPU: the GPUs of one node behind one handle ---------------------------------
 * Replaces the reference's thread block, main.zig:142-165 (T threads over contiguous pixel chunks of
 * multithreading.zig:40-44, then join): one host thread per GPU drives that GPU's own context and
 * stream, then ONE exchange step — an RCCL reduce(SUM) of the fp32 frame sums to GPU 0 over xGMI.
 *   PT_SPLIT_TILES    runs of cfg->tile_pixels pixels (0 = 256) dealt round-robin to the GPUs (large
 *                     frames / scenes); other owners' pixels are 0, so the reduce assembles the frame
 *                     exactly: bit-identical to one GPU;
 *   PT_SPLIT_SAMPLES  GPU g traces samples [a + g*n/G, a + (g+1)*n/G) of every pixel, [a, a+n) being
 *                     cfg's own sample range (high-spp frames); equal to one GPU up to the fp32 add
 *                     order of the G partial frames.
 * cfg must not itself be one owner's share (tile_count <= 1).  librccl is dlopen'ed at the first
 * create with num_gpus > 1 (PT_ERR_RCCL if that or a collective fails); with one GPU no collective runs
 * (PTGPU_MULTI_FORCE_RCCL=1 in the environment runs it anyway, for tests).  Not re-entrant. */
typedef struct ptgpu_multi ptgpu_multi;
enum { PT_SPLIT_TILES = 0, PT_SPLIT_SAMPLES = 1 };
int ptgpu_multi_create(const int *devices /* NULL = 0..num_gpus-1 */, int num_gpus, ptgpu_multi **out);
int ptgpu_multi_destroy(ptgpu_multi *m);
int ptgpu_multi_num_gpus(const ptgpu_multi *m);
int ptgpu_multi_uses_rccl(const ptgpu_multi *m);
ptgpu_ctx *ptgpu_multi_ctx(ptgpu_multi *m, int gpu);            /* the single-GPU context of GPU `gpu` (borrowed) */
int ptgpu_multi_set_scene(ptgpu_multi *m, const pt_sphere *spheres, uint32_t n); /* replicated on every GPU */
int ptgpu_multi_set_camera(ptgpu_multi *m, const pt_camera *cam);
/* ptgpu_render_sum_host / ptgpu_render_ppm over all GPUs of the handle. */
int ptgpu_multi_render_sum_host(ptgpu_multi *m, const pt_config *cfg, int split, float *h_rgb_sum);
int ptgpu_multi_render_ppm(ptgpu_multi *m, const pt_config *cfg, int split, uint8_t *slice, size_t slice_len,
                           uint32_t header_size, uint32_t px_size);
/* The share GPU `gpu` of `num_gpus` renders (pure host arithmetic, no device needed): *out is the
 * pt_config for ptgpu_render_sum_device on that GPU; *has_work = 0 when the share is empty (more GPUs
 * than samples / tiles). */
int ptgpu_multi_plan(const pt_config *cfg, int split, int num_gpus, int gpu, pt_config *out, int *has_work);
/* Counters of the last render: of GPU `gpu`, or with gpu = -1 the totals (times: the slowest GPU's
 * kernel_ms; total_ms includes the reduce). */
int ptgpu_multi_get_stats(const ptgpu_multi *m, int gpu, pt_stats *out);
double ptgpu_multi_reduce_ms(const ptgpu_multi *m);             /* HIP-event time of the last RCCL reduce on GPU 0 */
int ptgpu_multi_last_rccl_error(const ptgpu_multi *m);          /* raw ncclResult_t of the last PT_ERR_RCCL */

/* Limits.  PT_LIMIT_PARTIAL_BYTES (default 2 GiB): upper bound of the device buffer that holds the
 * chunk partial sums (chunks x pixels x 12 B).  A render that needs more is traced in segments of
 * pixels, one kernel launch each, through the same buffer; the result is bit-identical (a pixel's
 * partial sums are always added in chunk order), pt_stats.segments tells how many launches it took.
 *
 * PT_LIMIT_PRIMARY_BYTES (default ~0 = automatic; 0 switches the primary pass off): upper bound of the device
 * buffer of the primary pass: one 32-byte record per path of a launch plus 176 B of candidate lists per
 * pixel.  Automatic = half of the device memory that is free when the render starts (hipMemGetInfo; what the context
 * already holds for the pass counts as free), at most 96 GiB — a host that shares the GPU with other allocators keeps
 * room.  The primary pass traces every camera ray against the few spheres its pixel's rays can reach
 * (a conservative per-pixel list built per launch) and shades the first hit; the megakernel continues the
 * paths from the records, so its scan over all spheres only serves scattered rays.  Same closest hits,
 * same draws, same sums: results do not depend on this limit, only segments and speed do.  The buffer is
 * allocated at the size a render needs, not at the limit, and kept for the next render; lowering the limit below
 * the held size frees it at once, ptgpu_trim frees every work buffer.  ptgpu_render_sum_device_noalloc never runs the
 * primary pass (its records would not fit the static scratch).
 *
 * PT_LIMIT_PRIMARY_BOUNCES (1 or 2, default 2): 2 lets the primary pass also trace the ray the sphere a pixel mainly
 * shows sends on (scattered, mirrored or refracted from near the pixel's central hit point) against a second per-pixel list;
 * 1 stops it after the camera ray.  Results do not depend on it.
 *
 * PT_LIMIT_TREE_MIN_SPHERES (default 256; 0 = never): scenes of at least this many spheres are traced through a tree
 * of boxes over the spheres instead of the scan over all of them (scene.zig:96-110 tests every sphere for every ray;
 * the tree only skips spheres the reference's f32 test provably rejects, and the closest hit is chosen by (t, index)
 * as the ascending scan chooses it: same hits, bit for bit).  Scenes of up to 512 spheres walk the tree in LDS, larger
 * ones in global memory.  The tree is built by ptgpu_set_scene (host, O(n log n)): set the limit before uploading
 * the scene.  Scenes with a non-finite centre or radius, or deeper than the traversal stack (possible beyond 3.3e7
 * spheres), keep the scan. */
enum { PT_LIMIT_PARTIAL_BYTES = 0, PT_LIMIT_PRIMARY_BYTES = 1, PT_LIMIT_PRIMARY_BOUNCES = 2, PT_LIMIT_TREE_MIN_SPHERES = 3 };
int ptgpu_set_limit(ptgpu_ctx *ctx, int which, uint64_t value);
/* Device memory the context holds besides the scene tables (records and lists of the primary pass, chunk partials,
 * hit-id stack, staging images): grown on demand by the renders, kept between them.  ptgpu_trim gives all of it back
 * (the next render allocates again what it needs); ptgpu_work_bytes reports the bytes held now. */
int ptgpu_trim(ptgpu_ctx *ctx);
uint64_t ptgpu_work_bytes(const ptgpu_ctx *ctx);

/* Diagnostics. */
/* The uploaded scene: its sphere count, and how many of them sit in mask words (32 consecutive spheres) that share
 * one centre ordinate and therefore take the filter's 5-FMA test instead of the 7-FMA one. */
int ptgpu_scene_info(const ptgpu_ctx *ctx, uint32_t *nspheres, uint32_t *shared_ordinate_tests);
/* The sphere tree of the uploaded scene (PT_LIMIT_TREE_MIN_SPHERES): its inner nodes (0: the scene has none and is
 * traced by the scan) and its depth in levels. */
int ptgpu_tree_info(const ptgpu_ctx *ctx, uint32_t *nodes, uint32_t *depth);
/* Host-only self-check of the tree builder (no device, no context): builds the tree ptgpu_set_scene would build for these
 * spheres (median_only != 0: without the surface-area heuristic) and counts structural violations — a sphere outside the box
 * of an ancestor, missing from the leaves or listed twice, an oversized leaf.  *violations must come back 0.
 * PT_ERR_UNSUPPORTED: no tree for this scene (non-finite spheres / too deep). */
int ptgpu_debug_tree_check(const pt_sphere *spheres, uint32_t n, int median_only, uint32_t *nodes, uint32_t *depth, uint32_t *leaves,
                           uint64_t *violations);
int ptgpu_get_stats(const ptgpu_ctx *ctx, pt_stats *out);
const char *ptgpu_strerror(int status);
int ptgpu_last_hip_error(const ptgpu_ctx *ctx); /* raw hipError_t of the last PT_ERR_HIP */
int ptgpu_device_info(const ptgpu_ctx *ctx, int *compute_units, int *clock_khz, char *arch, size_t arch_len);
uint32_t ptgpu_abi_version(void); /* 3 (round 3: pt_config is 112 bytes (sampler), pt_stats 136; a host built against 2 must be rebuilt) */

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */

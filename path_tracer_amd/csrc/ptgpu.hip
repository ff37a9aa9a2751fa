// ptgpu.hip — libptgpu: the MI355X (gfx950) path-tracing hot path behind the C ABI
// of include/ptgpu.h.  Hand-written HIP for CDNA4 only (wave64, LDS tiles,
// persistent workgroups); no CPU fallback.
//
// Kernel design (DESIGN.md has the long form):
//   * one lane = one path; a persistent wave keeps all 64 lanes busy by refilling a
//     lane's path as soon as it terminates (next sample of its work item, or a new
//     work item from an atomic queue; a wave reserves 64 items per atomic and deals them
//     out by ballot/popcount ranks) — replaces the thread split of multithreading.zig;
//   * Scene.is_hit (scene.zig:96-110) = a linear scan over a filter table
//     (cx,cz,threshold,cy) staged into LDS in tiles: 7 FMAs + 1 compare per sphere (5 + 1 when the 32 spheres
//     of a mask word share their centre ordinate)
//     decide conservatively whether the reference's discriminant can be >= 0 and one
//     v_addc shifts the verdict into a per-lane bit word (no branch in the scan); the
//     few survivors are evaluated afterwards with the reference's exact arithmetic
//     (sphere.zig:63-103), in sphere order, so the closest hit is the reference's bit
//     for bit;
//   * radiance (illumination.zig:76-90) = iterative bounce loop; the hit sphere of
//     every bounce is pushed on a per-lane stack in HBM and the attenuations are
//     multiplied innermost-first when the path ends, as the recursion does.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/ptgpu.h"
#include "pt_device.h"

using namespace ptdev;

#define PT_ABI_VERSION 3u // 3: pt_config grew to 112 bytes (sampler), pt_stats to 136 (lists_ms, megakernel_ms, primary_candidates); 2: pt_stats 112 bytes, limits, tree, no-allocation entry point

// ---------------------------------------------------------------------------
// kernel parameters
// ---------------------------------------------------------------------------
namespace {

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr int kTile = 512;           // spheres per LDS staging tile (8 KiB of float4 filter records)
#ifndef PT_MASK_TILE
#define PT_MASK_TILE 256
#endif
// Spheres scanned between two exact passes (= 32 x the candidate words a lane keeps in LDS).  Single-tile
// scenes use 256 (27 KiB of LDS per workgroup with the material tables: 5 workgroups per CU); multi-tile
// scenes have no tables in LDS and take the whole staging tile (fewer exact passes per ray step).
constexpr int kMaskTile = PT_MASK_TILE;
constexpr int mask_tile(bool multi_tile) { return multi_tile ? kTile : kMaskTile; }
constexpr int kPad = 8;              // the filter table is padded to a multiple of kPad records
#ifndef PT_WAVES_PER_SIMD
#define PT_WAVES_PER_SIMD 5
#endif
constexpr uint32_t kItemBatch = 64;  // work items a wave reserves per queue atomic
#ifndef PT_SUB
#define PT_SUB 4
#endif
#ifndef PT_REFILL_ROUNDS
#define PT_REFILL_ROUNDS 1 // with primary-pass records: refill rounds per ray step while some lane used its item up (rows of sky are all such
                           // items).  Round 2, headline frame, 2 / 4 / 16 rounds: lane utilisation 0.966 / 0.983 / 0.993 but 175.1 / 175.2 / 179.3 ms.
                           // Round 3 (tree walk, shorter steps), per 256 samples, 1 / 2 / 3 rounds: 33.9 / 34.35 / 34.6 ms — the second round's code is
                           // run by every wave step whether or not a lane needs it; one round
#endif
#ifndef PT_WORDS_PER_ITER
#define PT_WORDS_PER_ITER 4 // mask words (x32 tests) per unrolled scan-loop iteration; measured 1/2/4/8: 432.3/429.8/428.6/433.0 ms
#endif

constexpr int kSub = PT_SUB;         // filter records held in registers ahead of their tests (software pipeline depth)
static_assert(kPad % kSub == 0 && kMaskTile % 32 == 0 && kTile % kMaskTile == 0, "tile constants");
constexpr float kKeps = 1.52587890625e-05f; // 2^-16: relative safety margin of the filter
// primary pass: candidate-list capacity per pixel and the record tags (bits 31:30 of the first half's .w)
constexpr uint32_t kListCap = 16;
constexpr uint32_t kListEnd = 0xffffffffu, kListOverflow = 0xfffffffeu;
// per pixel slot: [0,16) list A (camera rays); 16: sphere S of the second-bounce list (kListEnd: none), 17..19: p0, 20: patch radius^2,
// 21..23: cone axis, 24: cosine of the cone's half-angle (f32 bits); [28,44) list B: spheres a ray can hit that starts in the patch
// on S and runs inside the cone
#ifndef PT_LIST_CAP_B
#define PT_LIST_CAP_B 16
#endif
constexpr uint32_t kListCapB = PT_LIST_CAP_B;
constexpr uint32_t kListTile = 256; // spheres per LDS tile of pt_primary_lists (72 B each)
constexpr uint32_t kListS = 16, kListP0 = 17, kListRho2 = 20, kListAxis = 21, kListCos = 24, kListB = 28, kListStride = kListB + kListCapB;
// record tags, bits 31:29 of the first half's .w (states have bit 31 set); bits 28:0 = first hit id
constexpr uint32_t kTagShift = 29, kTagFinal = 0u, kTagSkip = 1u, kTagState0 = 4u, kTagState1 = 5u, kTagState2 = 6u;
constexpr uint32_t kHitMask = (1u << kTagShift) - 1u;
constexpr uint32_t kDrawBits = 13;  // STATE2 packs (draws taken, second hit id) into the second half's .w: 13 + 19 bits
#define kNaN __builtin_nanf("")

enum { CNT_QUEUE = 0, CNT_PATHS = 1, CNT_RAYS = 2, CNT_CANDS = 3, CNT_NOT_UNIT = 4, CNT_WAVE_STEPS = 5, CNT_FLUSH_ITERS = 6, CNT_PRIMARY_RAYS = 7, CNT_LIST2_RAYS = 14, CNT_BOX_TESTS = 15, CNT_PRIMARY_CANDS = 20, CNT_CLAMPED = 23, CNT_PP0 = 24, CNT_N = 32 }; // 8..13, 16..19, CNT_PP0..+7: PT_PHASE_TIMING builds; CNT_CLAMPED: resolve's clamp count

// x / d for any u32 x without the ~35-instruction expansion of a run-time integer division: mulhi by a
// host-computed magic number (the "branch-free" scheme: t = mulhi(x, M); q = (t + ((x - t) >> 1)) >> s).
struct FastDiv {
    uint32_t d, M, s;
    __device__ __forceinline__ uint32_t div(uint32_t x) const {
        const uint32_t t = __umulhi(x, M);
        const uint32_t q = (t + ((x - t) >> 1)) >> s;
        return d == 1u ? x : q;
    }
};
static FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{d, 0u, 0u};
    if (d <= 1u) { f.d = 1u; return f; }
    uint32_t fl = 31u;
    while (!(d >> fl)) --fl;                       // floor(log2 d)
    if ((d & (d - 1u)) == 0u) { f.s = fl - 1u; return f; } // power of two: M = 0, q = (x >> 1) >> (fl - 1)
    const uint64_t num = (uint64_t)(1u << fl) << 32;
    uint32_t pm = (uint32_t)(num / d);
    const uint32_t rem = (uint32_t)(num - (uint64_t)pm * d);
    pm += pm;
    const uint32_t twice = rem + rem;
    if (twice >= d || twice < rem) pm += 1u;
    f.M = 1u + pm;
    f.s = fl;
    return f;
}

struct TreeNode;
struct KParams {
    Cam cam;
    V3 blend_start, blend_stop;
    float t_min, t_max;
    uint32_t width, height;
    uint32_t rebounds;
    uint32_t illum;                 // PT_ILLUM_*
    uint32_t seed_lo, seed_hi;
    uint32_t idx_start, idx_stop;   // pixel range
    uint32_t slot0;                 // first pixel slot of this launch (segment of the range / of this owner's tiles)
    uint32_t npix;                  // pixel slots this launch covers: [slot0, slot0 + npix)
    uint32_t tile_pixels, tile_count, tile_index; // interleaved tile ownership (0 = off)
    uint32_t direct;                // 1: one chunk, sums go straight to the frame at pixel*3
    uint32_t s0, s1, chunk, nchunks; // sample range and chunking
    uint32_t nspheres;
    uint32_t total_threads;
    FastDiv div_width, div_npix, div_chunk, div_tile, div_nchunks; // divisions by width, npix, chunk, tile_pixels, nchunks
    uint32_t blocks_per_row; // pt_primary: 64-record blocks per chunk index (npix * chunk / 64) when that is whole, else 0
    const float4 *filt; // (cx, cz, threshold, cy) padded to a multiple of kPad
    uint32_t has_shared; // any shared-ordinate word in the scene
    const float *ycy;   // per 32-sphere mask word (padded to whole tiles): the cy all its spheres share, or NaN
    const float4 *geom; // (cx, cy, cz, radius)
    const float4 *mat;  // (albedo r, g, b, param); dielectrics: (1/ri, schlick r0, 0, ri)
    const float *inv_radius; // 1/radius, correctly rounded (the value sphere.zig:88 computes per hit)
    const uint32_t *kind;
    float *part;        // direct ? frame[W*H][3] : [nchunks][npix][3] partial sums
    // primary pass (pt_primary): one record per path of the launch, index item * chunk + (sample - item's first sample)
    uint32_t primary;       // 1: the records exist; the megakernel starts its paths from them instead of casting camera rays
    float4 *rec;            // record r = rec[2r], rec[2r+1]: (colour | origin, tag<<29 | first hit id), (direction, draws taken) — 32 B, so that
                            // the two halves share a cache line and a line holds a lane's next record too.  (Round 3 tried the halves apart — first
                            // halves dense, 16 B per path; second halves compacted per item: pt_primary wrote 47.5 instead of 68.9 GB per headline
                            // frame, but the megakernel FETCHED 130 instead of 87 GB: a lane comes back to its item once per continued path, its
                            // lines long evicted, and then pays one line per array — and ran exactly as fast, 78.1 vs 78.5 ms: the fetches are not
                            // what it waits for.  Kept side by side.)
    uint32_t *item_mask;    // per item: bit j set = its sample j needs a look at its record (a state, or a dropped sample); clear = a finished
                            // colour.  use_mask = 0 (chunk does not divide 64, or > 32): not written, every record is looked at
    uint32_t use_mask;
    const uint32_t *lists;  // [npix][kListStride]: candidate spheres of each pixel slot's camera rays and of its second-bounce rays (pt_primary_lists)
    uint32_t *stack;    // [rebounds][total_threads] hit ids
    unsigned long long *counters;
    // sphere tree (pt_megakernel<.., TREE = true>): nodes, the spheres in leaf order (geometry, original index), root ref
    const TreeNode *tnodes;
    const float4 *tgeom;
    const uint32_t *tprim;
    uint32_t troot, tnode_count;
    uint32_t tree_exit_at; // a ray step's walk ends once at most this many lanes are still walking (0: when none is)
    uint32_t tree_flush_at; // the walk tests its postponed leaves once this many lanes hold one
};

// Pixel index of local slot `po`: the range itself, or the owner's interleaved tiles
// (tile t of the range belongs to owner t % tile_count; include/ptgpu.h).
__device__ __forceinline__ uint32_t slot_to_pixel(uint32_t po, uint32_t idx_start, uint32_t tile_pixels, uint32_t tile_count,
                                                  uint32_t tile_index) {
    if (tile_pixels == 0u) return idx_start + po;
    const uint32_t k = po / tile_pixels, r = po - k * tile_pixels;
    return idx_start + (k * tile_count + tile_index) * tile_pixels + r;
}
__device__ __forceinline__ uint32_t slot_to_pixel(uint32_t po, uint32_t idx_start, uint32_t tile_pixels, uint32_t tile_count,
                                                  uint32_t tile_index, const FastDiv &div_tile) {
    if (tile_pixels == 0u) return idx_start + po;
    const uint32_t k = div_tile.div(po), r = po - k * tile_pixels;
    return idx_start + (k * tile_count + tile_index) * tile_pixels + r;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Diagnostic build only (-DPT_PHASE_TIMING, tools/build_prof.sh): per-wave s_memtime stamps
// around the phases, summed into counters[8..15]; the shipped library has no stamps.
#ifdef PT_PHASE_TIMING
#define PT_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PT_ACC(idx, a, b) t_phase[idx] += (b) - (a)
#define PT_SUB_ACC(idx, a, b) t_sub[idx] += (b) - (a)
#else
#define PT_STAMP(var)
#define PT_ACC(idx, a, b)
#define PT_SUB_ACC(idx, a, b)
#endif

// Scene.is_hit (scene.zig:96-110) for the 64 rays of a wave: conservative filter scan over the LDS
// tile(s), then the reference's exact Sphere.is_hit (sphere.zig:63-103) on the survivors.  Shared by
// the megakernel and by the PT_FN_SCENE_HIT_FILTERED debug kernel (tests compare it with the plain
// exact scan).  All threads of the workgroup must call it together when MULTI_TILE (barriers).
#ifdef PT_PHASE_TIMING
#define PT_FLUSH_ACC(a, b) t_flush += (b) - (a)
#else
#define PT_FLUSH_ACC(a, b)
#endif
// Per-ray constants of the conservative filter: with dh = d/|d|,
//   F(c) = (c.dh)^2 + c.w + E = (oc.dh)^2 - |oc|^2 + |c|^2 + Keps*|o|^2,   candidate iff F >= T(sphere).
struct FilterRay { float hx, hy, hz, wx, wy, wz, E; };
__device__ __forceinline__ FilterRay filter_ray_setup(V3 o, V3 d, float a) {
    FilterRay r;
    const float inv = rsqrtf(a);
    r.hx = d.x * inv; r.hy = d.y * inv; r.hz = d.z * inv;
    const float od = fmaf(o.z, r.hz, fmaf(o.y, r.hy, o.x * r.hx));
    const float oo = fmaf(o.z, o.z, fmaf(o.y, o.y, o.x * o.x));
    r.wx = 2.0f * fmaf(-od, r.hx, o.x);
    r.wy = 2.0f * fmaf(-od, r.hy, o.y);
    r.wz = 2.0f * fmaf(-od, r.hz, o.z);
    r.E = fmaf(od, od, (kKeps - 1.0f) * oo);
    return r;
}
__device__ __forceinline__ float filter_value(const FilterRay &r, float cx, float cy, float cz) {
    float q = cx * r.hx;
    q = fmaf(cy, r.hy, q);
    q = fmaf(cz, r.hz, q);
    float s = fmaf(cx, r.wx, r.E);
    s = fmaf(cy, r.wy, s);
    s = fmaf(cz, r.wz, s);
    return fmaf(q, q, s);
}

// The same value for a sphere whose centre ordinate cy is shared by its whole mask word (32 consecutive
// spheres): the cy terms are loop invariants of that word — Qy = cy*hy and Ey = cy*wy + E are formed once per
// ray and word, and a test is 5 FMA-class instructions instead of 7.  (Scene 21: 484 of the 488 spheres rest on
// the ground with cy = 0.2, scene.zig:62.)  Same terms, another association: the rounding budget of the
// threshold (filter_threshold_of) bounds sums of |terms| and covers both.
__device__ __forceinline__ float filter_value_y(const FilterRay &r, float Qy, float Ey, float cx, float cz) {
    float q = fmaf(cx, r.hx, Qy);
    q = fmaf(cz, r.hz, q);
    float s = fmaf(cx, r.wx, Ey);
    s = fmaf(cz, r.wz, s);
    return fmaf(q, q, s);
}
#ifndef PT_YCLASS
#define PT_YCLASS 1 // 0: every word takes the general 7-FMA test (A/B builds)
#endif

template <bool MULTI_TILE, class GeomOf>
__device__ __forceinline__ void scene_is_hit_filtered(const float4 *g_filt, const float *g_ycy, bool has_shared, float *s_ycy_all, uint32_t n_pad, float4 *s_filt_all, uint32_t *s_mask,
                                                      uint32_t tid, GeomOf geom_of, V3 o, V3 d, float a, float t_min, float t_max,
                                                      float &closest, int &hit, uint32_t &n_cands,
                                                      uint32_t &n_flush_iters, unsigned long long &t_flush) {
    constexpr uint32_t TILE = (uint32_t)kTile;
    (void)t_flush;
    {
        const FilterRay fray = filter_ray_setup(o, d, a);
        // One bit per tested sphere, 32 tests per word: after the compare, VCC is shifted
        // into the lane's word with a single v_addc (m = m + m + vcc) — no branch, no
        // scalar work per test.  A word's first test ends up in its highest used bit.
        uint32_t m = 0;    // word under construction
        uint32_t S = 0;    // one bit per stored word of this tile: "word is non-zero"

        for (uint32_t tile0 = 0; tile0 < n_pad; tile0 += TILE) {
            const uint32_t tile_n = min(TILE, n_pad - tile0);
            if (MULTI_TILE) {
                __syncthreads();
                for (uint32_t i = tid; i < tile_n; i += kBlock) s_filt_all[i] = g_filt[tile0 + i];
                if (has_shared && tid < (uint32_t)(kTile / 32)) s_ycy_all[tid] = g_ycy[(tile0 >> 5) + tid];
                __syncthreads();
            }
          constexpr uint32_t MASK_TILE = (uint32_t)mask_tile(MULTI_TILE);
          for (uint32_t grp = 0; grp < tile_n; grp += MASK_TILE) { // scan MASK_TILE spheres, then the exact pass
            const uint32_t tile = tile0 + grp;                // first sphere of this group
            const float4 *s_filt = s_filt_all + grp;
            const float *s_ycy = s_ycy_all + (grp >> 5);      // per mask word: the shared cy, or NaN = no shared ordinate
            const uint32_t tn = min(MASK_TILE, tile_n - grp); // multiple of kPad
            const uint32_t nw = (tn + 31u) >> 5;              // words of this group
            const uint32_t g_last = tn - ((nw - 1u) << 5);    // tests in the last word
            m = 0; S = 0;
            auto store_word = [&](uint32_t w, uint32_t tests_in_word) {
                const uint32_t mv = (tests_in_word >= 32u) ? m : (m & ((1u << tests_in_word) - 1u));
                s_mask[w * kBlock + tid] = mv;
                S = S + S + (mv != 0u ? 1u : 0u);
            };
            // Software-pipelined scan: kSub records in registers, refilled one by one — as soon as
            // a record has been tested its registers take the record kSub places ahead (the LDS
            // arrays carry kSub records of padding, so the last refills read unused slots).
            float4 fr[kSub];
#pragma unroll
            for (int k = 0; k < kSub; ++k) fr[k] = s_filt[k];
            auto test_and_refill = [&](float4 &f, uint32_t next) {
                const float F = filter_value(fray, f.x, f.w, f.y); // records are (cx, cz, T, cy)
                // candidate iff F >= T
                asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(f.z), "v"(F) : "vcc");
                f = s_filt[next];
            };
            float Qy = 0.f, Ey = 0.f;
            auto test_and_refill_y = [&](float4 &f, uint32_t next) {
                const float F = filter_value_y(fray, Qy, Ey, f.x, f.y); // cy (f.w) is not read, but stays an operand so that
                // the record is still fetched with ONE ds_read_b128: measured on MI355X (64 samples of the headline frame),
                // general test 104.5 ms; this test with ds_read_b32 + ds_read_b64 96.0, ds_read_b96 113.0, ds_read_b128 92.2
                asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(f.z), "v"(F), "v"(f.w) : "vcc");
                f = s_filt[next];
            };
            // One full mask word (32 tests) starting at test j of the group.  The word's class is wave-uniform (one LDS broadcast read, one scalar branch).
            auto word32 = [&](uint32_t j) {
#if PT_YCLASS
                // has_shared (kernel argument): scenes without any shared-ordinate word skip the class look-up altogether
                const uint32_t cyb = has_shared ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(s_ycy[j >> 5])) : 0x7fc00000u;
                if (cyb != 0x7fc00000u) {
                    const float cyu = __uint_as_float(cyb);
                    Qy = cyu * fray.hy;
                    Ey = fmaf(cyu, fray.wy, fray.E);
#pragma unroll
                    for (int k = 0; k < 32; ++k) test_and_refill_y(fr[k % kSub], j + (uint32_t)k + (uint32_t)kSub);
                } else
#endif
                {
#pragma unroll
                    for (int k = 0; k < 32; ++k) test_and_refill(fr[k % kSub], j + (uint32_t)k + (uint32_t)kSub);
                }
                store_word(j >> 5, 32u);
            };
            uint32_t j0 = 0;
            for (; j0 + 32u * PT_WORDS_PER_ITER <= tn; j0 += 32u * PT_WORDS_PER_ITER) { // whole mask words per iteration: LDS offsets are immediates
#pragma unroll
                for (int w = 0; w < PT_WORDS_PER_ITER; ++w) word32(j0 + (uint32_t)(32 * w));
            }
            for (; j0 + 32u <= tn; j0 += 32u) word32(j0);
            if (j0 < tn) {                      // the last, partial word (a multiple of kPad tests)
                for (; j0 < tn; j0 += kPad) {
#pragma unroll
                    for (int k = 0; k < kPad; ++k) test_and_refill(fr[k % kSub], j0 + (uint32_t)k + (uint32_t)kSub);
                }
                store_word(nw - 1u, g_last);
            }
            // ---- exact Sphere.is_hit (sphere.zig:63-103) on the survivors, ascending sphere
            // index, strict '<' (scene.zig:102): each lane walks its own words/bits; the
            // wave iterates max-over-lanes(candidates) times.
            PT_STAMP(tF0);
            {
                uint32_t mw = 0, wcur = 0, gcur = 32;
#ifdef PT_EXPERIMENT_FLUSH_MAX
                uint32_t n_flush_iters_here = 0;
#endif
                for (;;) {
                    if (mw == 0u && S != 0u) {
                        const uint32_t b = 31u - (uint32_t)__clz((int)S);
                        S &= ~(1u << b);
                        wcur = nw - 1u - b;
                        gcur = (wcur == nw - 1u) ? g_last : 32u;
                        mw = s_mask[wcur * kBlock + tid];
                    }
                    const bool has = mw != 0u;
                    if (__ballot(has) == 0ull) break;
#ifdef PT_EXPERIMENT_FLUSH_MAX // timing experiment (breaks parity): at most this many exact-pass iterations per group
                    if (n_flush_iters_here++ >= PT_EXPERIMENT_FLUSH_MAX) break;
#endif
                    n_flush_iters++;
                    if (has) {
                        const uint32_t b = 31u - (uint32_t)__clz((int)mw);
                        mw &= ~(1u << b);
                        const uint32_t j = tile + (wcur << 5) + (gcur - 1u - b);
                        const float4 g = geom_of(j);
                        n_cands++;
                        float t;
                        if (sphere_near_root(o, d, a, g.x, g.y, g.z, g.w, t_min, t_max, t)) {
                            if (t < closest) { closest = t; hit = (int)j; }
                        }
                    }
                }
            }
            PT_STAMP(tF1);
            PT_FLUSH_ACC(tF0, tF1);
          }
        }
    }
}

// ---------------------------------------------------------------------------
// Scene.is_hit over a sphere tree (scenes too large for one LDS tile)
// ---------------------------------------------------------------------------
// The scan above costs every ray one filter test per sphere; beyond one LDS tile the table is also restaged
// per ray step.  For such scenes ptgpu_set_scene builds a binary tree of axis-aligned boxes over the spheres
// (leaves of <= kTreeLeaf spheres) and each lane walks it with its own stack (LDS, word-major).  The tree only
// selects which spheres get the exact Sphere.is_hit; the closest hit is the minimum of (t, sphere index) over
// the spheres that pass it, which is what the reference's ascending scan with strict '<' returns
// (scene.zig:96-110), so the outcome is the same bit for bit PROVIDED no box is skipped that holds a sphere
// the f32 test would accept with t below the current closest.  That test can accept a ray whose line misses
// the sphere in real arithmetic (rounding of the discriminant, sphere.zig:66-72): the accepted point
// o + t d then lies within r' of the centre, r'^2 = r^2 + g (2 |o - c|^2 + r^2), g = kListGamma (the bound the
// primary pass's lists use, with the same 9x reserve).  A box [lo, hi] that contains its
// spheres (centre +- radius, rounded outwards) is therefore tested after growing it by
//   mu = kTreeKappa * m,   m = max_axis max(|lo - o|, |hi - o|)  >=  r  and  >= |o - c| / sqrt 3  for each of its spheres,
// kTreeKappa = 8e-3 >= sqrt(g) (sqrt 6 + 1) = 6.9e-3 >= (r' - r) / m; the rest covers the slab arithmetic
// (every term is a difference to o, so its rounding is relative to m as well).  A ray on the boundary of
// a grown box (0 x inf = NaN in a slab) may go either way: it is >= mu away from every sphere in it.
// Boxes are stored as centre bc and half extents bh (bc - bh <= lo, hi <= bc + bh in exact arithmetic: the builder rounds bh up), and the
// test uses m = max_axis |bc - o| + max_axis bh, which is >= the m above (|lo - o|, |hi - o| <= |bc - o| + bh per axis), so the same
// argument holds with a slightly larger margin.  Per axis the ray is inside the grown slab for |t - tc| <= (bh + mu) |1/d|, tc = (bc - o) / d:
// 24 VALU instructions per box against 37 for the corner form (three subtractions instead of six, no separate growth of each corner).
struct TreeNode { float4 a, b, c; uint4 ch; }; // boxes of the two children: a = (bc0, bh0.x), b = (bh0.yz, bc1.xy), c = (bc1.z, bh1); ch.x, ch.y = child refs
#ifndef PT_TREE_LEAF
#define PT_TREE_LEAF 4 // measured 1 / 2 / 4: 280.3 / 259.3 / 253.7 ms per frame of the 100 000-sphere scene, 36.8 / 37.0 / 38.6 ms per 256 samples of 3 000 spheres
#endif
constexpr uint32_t kTreeLeaf = PT_TREE_LEAF; // spheres per leaf (at most 8: three bits of the leaf ref)
static_assert(kTreeLeaf >= 1 && kTreeLeaf <= 8, "leaf count field");
constexpr uint32_t kTreeStack = 24;        // stack entries per lane = deepest tree walked (set_scene falls back to the scan beyond it)
constexpr uint32_t kTreeLeafBit = 0x80000000u, kTreeNone = 0xffffffffu; // leaf ref: bit 31 | (count - 1) << 28 | first slot of the leaf arrays
#ifndef PT_TREE_KAPPA
#define PT_TREE_KAPPA 8e-3f // (tools/build_variant.sh with 0: the negative control of PT_FN_TREE_BOX_SAFETY — boxes that are not grown do fail it)
#endif
constexpr float kTreeKappa = PT_TREE_KAPPA;

__device__ __forceinline__ bool tree_box(V3 o, V3 inv, float bcx, float bcy, float bcz, float bhx, float bhy, float bhz, float t_lo, float t_hi,
                                         float &t_near) {
    const float dx = bcx - o.x, dy = bcy - o.y, dz = bcz - o.z;
    const float m = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) + fmaxf(fmaxf(bhx, bhy), bhz);
    const float mu = kTreeKappa * m;
    const float tcx = dx * inv.x, tcy = dy * inv.y, tcz = dz * inv.z;
    const float gx = bhx + mu, gy = bhy + mu, gz = bhz + mu;
    const float ax = fabsf(inv.x), ay = fabsf(inv.y), az = fabsf(inv.z);
    // (a zero direction component: tc = +-inf or NaN, the products inf — each bound is then +-inf on the open side or NaN, and min / max
    // drop NaN operands: the axis constrains nothing, which errs on the side of entering the box)
    t_near = fmaxf(fmaxf(fmaf(-gx, ax, tcx), fmaf(-gy, ay, tcy)), fmaxf(fmaf(-gz, az, tcz), t_lo));
    const float t_far = fminf(fminf(fmaf(gx, ax, tcx), fmaf(gy, ay, tcy)), fminf(fmaf(gz, az, tcz), t_hi));
    return t_near <= t_far;
}

// Where a walk reads the tree from.  In global memory (large scenes): 32-bit refs, a stack of kTreeStack words per lane in
// LDS, word-major (entry e of thread tid at [e * kBlock + tid]).
struct TreeInGlobal {
    const TreeNode *nodes; const float4 *tgeom; const uint32_t *tprim; uint32_t *stack;
    static constexpr uint32_t kLeafBit = kTreeLeafBit, kNone = kTreeNone;
    __device__ __forceinline__ static uint32_t first(uint32_t r) { return r & 0x0fffffffu; }
    __device__ __forceinline__ static uint32_t count(uint32_t r) { return ((r >> 28) & 7u) + 1u; }
    __device__ __forceinline__ void node(uint32_t i, float4 &A, float4 &B, float4 &C, uint32_t &c0, uint32_t &c1) const {
        const TreeNode *nd = nodes + i;
        A = nd->a; B = nd->b; C = nd->c;
        const uint4 ch = nd->ch;
        c0 = ch.x; c1 = ch.y;
    }
    __device__ __forceinline__ void sphere(uint32_t slot, float4 &g, uint32_t &id) const { g = tgeom[slot]; id = tprim[slot]; }
    __device__ __forceinline__ void push(uint32_t sp, uint32_t tid, uint32_t v) const { stack[sp * kBlock + tid] = v; }
    __device__ __forceinline__ uint32_t pop(uint32_t sp, uint32_t tid) const { return stack[sp * kBlock + tid]; }
};
// In LDS (scenes of one tile whose tree has at most kTreeLdsNodes inner nodes and kTreeLdsStack + 1 levels): the whole tree
// is staged once per workgroup — a walk step is then an LDS round trip instead of a trip to L1/L2 (the walk from global
// memory waited on memory 64 % of its wave cycles on scene 21).  16-bit refs: bit 15 leaf | (count - 1) << 9 | first slot (< 512).
// Capacity of the LDS copy and the occupancy it allows: 192 nodes + 12 stack entries = 24.75 KiB per workgroup, 6 per CU at 80 VGPRs
// (measured against 256 nodes + 16 entries = 30 KiB, 5 per CU: 38.4 vs 39.8 ms per 256 samples of the headline frame; the walk waits
// on LDS and on the record/table loads around it, a sixth wave hides some of that).  Median splits alone give at most 127 inner nodes
// and 8 levels for 512 spheres with leaves of 4, so every one-tile scene has a tree that fits.
#ifndef PT_TREE_LDS_NODES
#define PT_TREE_LDS_NODES 192
#endif
#ifndef PT_TREE_LDS_STACK
#define PT_TREE_LDS_STACK 12
#endif
#ifndef PT_TREE_LDS_WAVES
#define PT_TREE_LDS_WAVES 6
#endif
constexpr uint32_t kTreeLdsNodes = PT_TREE_LDS_NODES, kTreeLdsStack = PT_TREE_LDS_STACK;
struct TreeInLds {
    const float4 *boxes; const uint32_t *child; const float4 *tgeom; const uint16_t *tprim; uint16_t *stack;
    static constexpr uint32_t kLeafBit = 0x8000u, kNone = 0xffffu;
    __device__ __forceinline__ static uint32_t first(uint32_t r) { return r & 0x1ffu; }
    __device__ __forceinline__ static uint32_t count(uint32_t r) { return ((r >> 9) & 7u) + 1u; }
    __device__ __forceinline__ static uint32_t ref16(uint32_t r32) { // a ref of the global encoding
        return (r32 & kTreeLeafBit) ? (0x8000u | (((r32 >> 28) & 7u) << 9) | (r32 & 0x1ffu)) : r32;
    }
    __device__ __forceinline__ void node(uint32_t i, float4 &A, float4 &B, float4 &C, uint32_t &c0, uint32_t &c1) const {
        A = boxes[3u * i]; B = boxes[3u * i + 1u]; C = boxes[3u * i + 2u];
        const uint32_t ch = child[i];
        c0 = ch & 0xffffu; c1 = ch >> 16;
    }
    __device__ __forceinline__ void sphere(uint32_t slot, float4 &g, uint32_t &id) const { g = tgeom[slot]; id = tprim[slot]; }
    __device__ __forceinline__ void push(uint32_t sp, uint32_t tid, uint32_t v) const { stack[sp * kBlock + tid] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t pop(uint32_t sp, uint32_t tid) const { return stack[sp * kBlock + tid]; }
};

// A walk can be left unfinished and resumed: `cur` and `sp` are the lane's position (cur == kNone && sp == 0: finished; the
// stack lives in LDS), closest / hit its result so far.  exit_at > 0: the wave leaves the loop as soon as at most that many
// lanes are still walking, provided some lane has finished (the megakernel then shades the finished lanes and gives
// them their next rays while the stragglers keep their place; they go on in the next ray step).  exit_at = 0: to the end.
#ifdef PT_WALK_PROFILE // diagnostic build (tools/build_variant.sh walkprof -DPT_WALK_PROFILE): lanes per walk-loop iteration by what they do
static __device__ unsigned long long g_walk_prof[8];
#endif
template <class Tree>
__device__ __forceinline__ void scene_is_hit_tree(const Tree T, uint32_t &cur, uint32_t &sp, uint32_t &pend, uint32_t exit_at, uint32_t flush_at, uint32_t tid, V3 o, V3 d, float a,
                                                  float t_min, float t_max, float &closest, int &hit, unsigned long long &w_cands, unsigned long long &w_iters, unsigned long long &w_boxes) {
    // w_cands, w_iters, w_boxes: counts of the WAVE (sums of popcounts: scalar instructions; per-lane counters cost three VALU
    // additions and their loop-carried copies in every iteration of a loop that is bound by VALU issue)
    // 1/d to within an ulp or two is enough (kTreeKappa's reserve); a zero component gives inf: the slab is then all of t or none of it
    const V3 inv = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    // (a lane without a ray stays out — the caller gives it cur = kNone: its parked NaN direction would pass every box, min/max drop NaN operands)
    const unsigned long long walking0 = __ballot(cur != Tree::kNone || sp > 0u || pend != Tree::kNone);
    // One loop serves inner nodes and leaves (measured against "while-while" — inner nodes until every lane holds a leaf, then the
    // leaves together: 63 vs 75 ms on the 100 000-sphere scene, 43.5 vs 43.1 on 3 000 spheres, 30.2 vs 29.0 on scene 21).
    // LEAVES ARE POSTPONED: with every leaf tested where it is met, 59 % of the iterations ran the leaf body for 12 of 64 lanes
    // (scene 21; PT_WALK_PROFILE build).  A lane that reaches a leaf parks it in `pend` and goes on with its stack; the parked
    // leaves are tested together once `flush_at` lanes hold one (or no lane has an inner node left).  The order in which spheres
    // are tested does not matter for the result — the closest hit is the minimum of (t, index) over every sphere that is tested,
    // and a box is only ever skipped against the closest hit found SO FAR, which a later test can only lower.
    for (;;) {
        if (cur == Tree::kNone && sp > 0u) cur = T.pop(--sp, tid);
        if (cur != Tree::kNone && (cur & Tree::kLeafBit) && pend == Tree::kNone) {
            pend = cur;
            cur = Tree::kNone;
            if (sp > 0u) cur = T.pop(--sp, tid);
        }
        const bool at_inner = cur != Tree::kNone && !(cur & Tree::kLeafBit);
        const unsigned long long m_inner = __ballot(at_inner), m_pend = __ballot(pend != Tree::kNone);
        const unsigned long long walking = m_pend | __ballot(cur != Tree::kNone);
        if (walking == 0ull) break;
        if (exit_at != 0u && walking != walking0 && (uint32_t)__popcll(walking) <= exit_at) break;
        w_iters += (unsigned long long)__popcll(walking & (0ull - walking)); // + 1 (the lowest walking lane), in a form that stays a scalar popcount: as `n++` the compiler keeps this wave-uniform count in a VGPR, two VALU instructions per iteration
#ifdef PT_WALK_PROFILE
        const unsigned long long m_stalled = __ballot(cur != Tree::kNone && (cur & Tree::kLeafBit)); // at a second leaf while `pend` is taken: waits for the flush
        if ((tid & 63u) == 0u) {
            atomicAdd(&g_walk_prof[5], (unsigned long long)__popcll(walking)); atomicAdd(&g_walk_prof[6], (unsigned long long)__popcll(m_stalled));
            const bool fl = (uint32_t)__popcll(m_pend) >= flush_at || m_inner == 0ull;
            atomicAdd(&g_walk_prof[0], fl ? (unsigned long long)__popcll(m_pend) : 0ull); atomicAdd(&g_walk_prof[1], (unsigned long long)__popcll(m_inner));
            atomicAdd(&g_walk_prof[2], fl ? 1ull : 0ull); atomicAdd(&g_walk_prof[3], m_inner ? 1ull : 0ull);
            atomicAdd(&g_walk_prof[4], 1ull);
        }
#endif
        if ((uint32_t)__popcll(m_pend) >= flush_at || m_inner == 0ull) { // wave-uniform
            {   // (counted here, where control flow is still uniform — inside the lanes' branch the sum would be a per-lane value again —
                // as one popcount per bit of the leaves' count field: sum of counts = lanes + sum_b 2^b lanes with bit b of (count - 1))
                const uint32_t cm1 = pend != Tree::kNone ? Tree::count(pend) - 1u : 0u;
                w_cands += (unsigned long long)__popcll(m_pend);
#pragma unroll
                for (uint32_t b = 0; (1u << b) < kTreeLeaf; ++b) w_cands += (unsigned long long)__popcll(__ballot((cm1 >> b) & 1u)) << b;
            }
            if (pend != Tree::kNone) {
                const uint32_t first = Tree::first(pend), cnt = Tree::count(pend);
#pragma unroll
                for (uint32_t k = 0; k < kTreeLeaf; ++k) {
                    if (k < cnt) {
                        float4 g; uint32_t id;
                        T.sphere(first + k, g, id);
                        float t;
                        if (sphere_near_root(o, d, a, g.x, g.y, g.z, g.w, t_min, t_max, t)) {
                            // the scan's outcome: smallest t, and among equal t the lowest sphere index
                            if (t < closest || (t == closest && (int)id < hit)) { closest = t; hit = (int)id; }
                        }
                    }
                }
                pend = Tree::kNone;
            }
        }
        w_boxes += 2ull * (unsigned long long)__popcll(m_inner);
        if (at_inner) {
            float4 A, B, C; uint32_t c0, c1;
            T.node(cur, A, B, C, c0, c1);
            float n0, n1;
            const bool h0 = tree_box(o, inv, A.x, A.y, A.z, A.w, B.x, B.y, t_min, closest, n0);
            const bool h1 = tree_box(o, inv, B.z, B.w, C.x, C.y, C.z, C.w, t_min, closest, n1);
            if (h0 && h1) {
                const bool first0 = n0 <= n1; // nearer box first: a hit found there shortens (t_min, closest) for the boxes below the other
                T.push(sp++, tid, first0 ? c1 : c0);
                cur = first0 ? c0 : c1;
            } else {
                cur = h0 ? c0 : (h1 ? c1 : Tree::kNone);
            }
        }
    }
}

// The tree of a one-tile scene into LDS (every thread of the workgroup; followed by a barrier).
__device__ __forceinline__ void tree_stage_lds(const TreeNode *__restrict__ nodes, uint32_t n_nodes, const float4 *__restrict__ tgeom,
                                               const uint32_t *__restrict__ tprim, uint32_t nspheres, float4 *s_boxes, uint32_t *s_child, float4 *s_tgeom,
                                               uint16_t *s_tprim, uint32_t tid) {
    for (uint32_t i = tid; i < n_nodes; i += kBlock) {
        const TreeNode nd = nodes[i];
        s_boxes[3u * i] = nd.a; s_boxes[3u * i + 1u] = nd.b; s_boxes[3u * i + 2u] = nd.c;
        s_child[i] = TreeInLds::ref16(nd.ch.x) | (TreeInLds::ref16(nd.ch.y) << 16);
    }
    for (uint32_t i = tid; i < nspheres; i += kBlock) { s_tgeom[i] = tgeom[i]; s_tprim[i] = (uint16_t)tprim[i]; }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// the megakernel
// ---------------------------------------------------------------------------
// MULTI_TILE: the filter table is larger than one LDS tile and is restaged tile by tile in
// every ray step (block-wide barriers); otherwise it is staged once per workgroup.
//
// The wave-uniform filter records are read from LDS with broadcast ds_read_b128.  Feeding them
// through the scalar cache instead (s_load_dwordx16 into SGPR operands: no VGPRs, no LDS reads)
// was measured 15 % slower on MI355X — 16 B per wave per test saturates the scalar cache, which
// several CUs share, long before LDS.
// TREE: Scene.is_hit walks the sphere tree instead of scanning; no filter table, no barriers in the ray loop; the sphere
// tables are read from global memory.  With MULTI_TILE the tree stays in global memory and s_mask is the lanes' stacks;
// without (one-tile scenes) the whole tree is staged in LDS.
// MODE: 0 stochastic with the reference's rejection loops, 1 deterministic (every draw 0.5), 2 stochastic with the direct
// samplers (pt_config.sampler = PT_SAMPLER_DIRECT)
enum { kModeRejection = 0, kModeDet = 1, kModeDirect = 2 };
template <int MODE, bool MULTI_TILE, bool PRIMARY, bool TREE = false>
#ifndef PT_TREE_GLOBAL_WAVES
#define PT_TREE_GLOBAL_WAVES 6 // 5 / 6 waves per SIMD: 37.4 / 35.6 ms per 256 samples of 3 000 spheres, 230.9 / 215.4 ms per frame of 100 000 (the walk waits on L2)
#endif
__global__ __launch_bounds__(kBlock, TREE ? (MULTI_TILE ? PT_TREE_GLOBAL_WAVES : PT_TREE_LDS_WAVES) : PT_WAVES_PER_SIMD) void pt_megakernel(const KParams P) {
    constexpr bool DET = MODE == kModeDet, DIRECT = MODE == kModeDirect;
    constexpr bool TREE_LDS = TREE && !MULTI_TILE;
    constexpr bool TABLES_LDS = !MULTI_TILE && !TREE;      // filter table, radii and materials of a one-tile scene
    __shared__ float4 s_filt[TREE ? 1 : kTile + kSub];      // (cx, cz, T, cy) filter records of the current tile
    __shared__ uint32_t s_mask[TREE ? (TREE_LDS ? 1 : kTreeStack * kBlock) : mask_tile(MULTI_TILE) / 32 * kBlock]; // candidate bit words, word-major (conflict-free)
    __shared__ float s_ycy[kTile / 32];                     // per mask word of the tile: shared centre ordinate, or NaN
    __shared__ float4 s_tboxes[TREE_LDS ? 3 * kTreeLdsNodes : 1];
    __shared__ uint32_t s_tchild[TREE_LDS ? kTreeLdsNodes : 1];
    __shared__ float4 s_tgeom[TREE_LDS ? kTile : 1];
    __shared__ uint16_t s_tprim[TREE_LDS ? kTile : 2];
    __shared__ uint16_t s_tstack[TREE_LDS ? kTreeLdsStack * kBlock : 2];
    // single-tile scenes keep the exact-test radius and the materials in LDS too (no global
    // latency in the exact pass / shading); larger scenes read them from HBM/L2
    __shared__ float s_radius[TABLES_LDS ? kTile : 1];
    __shared__ float s_inv_radius[TABLES_LDS ? kTile : 1];
    __shared__ float4 s_mat[TABLES_LDS ? kTile : 1];   // (albedo r, g, b, param) | (1/ri, r0, 0, ri); kind in s_kind
    __shared__ uint8_t s_kind[TABLES_LDS ? kTile : 4];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t gtid = blockIdx.x * kBlock + tid;

    const uint32_t n_pad = (P.nspheres + kPad - 1) / kPad * kPad;
    if (TREE_LDS) tree_stage_lds(P.tnodes, P.tnode_count, P.tgeom, P.tprim, P.nspheres, s_tboxes, s_tchild, s_tgeom, s_tprim, tid);
    if (TABLES_LDS) {
        for (uint32_t i = tid; i < n_pad; i += kBlock) s_filt[i] = P.filt[i];
        if (tid < (uint32_t)(kTile / 32)) s_ycy[tid] = P.ycy[tid];
        for (uint32_t i = tid; i < P.nspheres; i += kBlock) {
            s_radius[i] = P.geom[i].w;
            s_inv_radius[i] = P.inv_radius[i];
            s_mat[i] = P.mat[i];
            s_kind[i] = (uint8_t)P.kind[i];
        }
        __syncthreads();
    }
    // exact sphere / material lookups (LDS for single-tile scenes)
    auto geom_of = [&](uint32_t j) -> float4 {
        if (!TABLES_LDS) return P.geom[j];
        const float4 f = s_filt[j]; // (cx, cz, T, cy)
        return make_float4(f.x, f.w, f.y, s_radius[j]);
    };
    auto inv_radius_of = [&](uint32_t j) -> float { return TABLES_LDS ? s_inv_radius[j] : P.inv_radius[j]; };
    auto mat_of = [&](uint32_t j) -> float4 { return TABLES_LDS ? s_mat[j] : P.mat[j]; };
    auto kind_of = [&](uint32_t j) -> uint32_t { return TABLES_LDS ? (uint32_t)s_kind[j] : P.kind[j]; };

    // per-lane path state
    bool alive = true;      // lane still has (or can get) work
    bool need_path = true;  // lane needs a new camera sample
    uint32_t pixel = 0, sample = 0, s_end = 0, out_slot = 0;
    uint32_t rec_base = 0;  // primary pass: record of (item, sample) = rec_base + sample
    uint32_t hmask = 0;     // primary pass: the item's mask, shifted so that bit 0 is the next sample to start
    bool has_item = false;
    V3 accum = v3(0.f, 0.f, 0.f);
    V3 o = v3(0.f, 0.f, 0.f), d = v3(kNaN, kNaN, kNaN); // parked (see phase A) until the first path starts
    uint32_t bounce = 0;
    // tree variants: the lane's walk (it may span ray steps, see scene_is_hit_tree) and its result so far
    uint32_t t_cur = TREE_LDS ? TreeInLds::kNone : TreeInGlobal::kNone, t_sp = 0, t_pend = t_cur;
    float t_closest = 0.f;
    int t_hit = -1;
    typename std::conditional<DET, ConstRng, PhiloxRng>::type rng;
    rng.init(P.seed_lo, P.seed_hi);
    uint32_t n_cands = 0, n_paths = 0, n_not_unit = 0; // per-lane counts (widened when the wave sums them)
    unsigned long long w_rays_traced = 0;             // wave-uniform
    uint32_t n_wave_steps = 0, n_flush_iters = 0; // wave-uniform (SALU) profile counters
    unsigned long long w_tree_cands = 0, w_tree_boxes = 0, w_tree_iters = 0;  // wave-uniform: exact tests, box tests and loop iterations of the tree walk
    unsigned long long t_flush = 0;               // PT_PHASE_TIMING builds only
#ifdef PT_PHASE_TIMING
    unsigned long long t_phase[5] = {0, 0, 0, 0, 0}; // A refill, B filter scan, B flush, C shade, whole
    unsigned long long t_sub[4] = {0, 0, 0, 0};      // within C: hit point, scatter, miss (lerp + unwinding); within A: path start
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif
    const uint32_t total_items = P.npix * P.nchunks;
    uint32_t pool_next = 0, pool_end = 0; // wave-uniform: items [pool_next, pool_end) are reserved for this wave

    for (;;) {
        // ---- phase A: refill lanes whose path has ended -------------------------
        PT_STAMP(tA0);
        // With primary-pass records a lane can use up its item while starting a path (every remaining sample
        // ended in the primary pass): such lanes get more rounds — a new item and its records — before the
        // wave goes into the scan (PT_REFILL_ROUNDS).
        for (int round = 0;; ++round) {
        if (alive && need_path && sample == s_end) {
            if (has_item) {
                float *dst = P.part + 3ull * out_slot;
                dst[0] = accum.x; dst[1] = accum.y; dst[2] = accum.z;
                has_item = false;
            }
        }
        {
            const bool want = alive && need_path && sample == s_end;
            const unsigned long long m = __ballot(want);
            if (m != 0ull) {
                // The wave draws items from a pool it reserved earlier and reserves kItemBatch more
                // with ONE atomic only when the pool cannot serve every requesting lane: the
                // atomic's round trip is paid once per ~64 items, not once per ray step.
                const uint32_t cnt = (uint32_t)__popcll(m);
                const uint32_t avail = pool_end - pool_next;
                uint32_t fresh = 0;
                if (cnt > avail) {
                    const int leader = __ffsll((long long)m) - 1;
                    if ((int)lane == leader) fresh = (uint32_t)atomicAdd(&P.counters[CNT_QUEUE], (unsigned long long)kItemBatch);
                    fresh = __shfl(fresh, leader, 64);
                }
                const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                const uint32_t item_ = (rank < avail) ? pool_next + rank : fresh + (rank - avail);
                if (cnt > avail) { pool_next = fresh + (cnt - avail); pool_end = fresh + kItemBatch; }
                else pool_next += cnt;
                if (want) {
                    const uint32_t item = item_;
                    // A lane without a path parks on a NaN direction: no filter compare passes, so it
                    // adds no candidates and no exact-pass iterations to its wave.
                    o = v3(0.f, 0.f, 0.f); d = v3(kNaN, kNaN, kNaN);
                    if (item < total_items) {
                        const uint32_t k = P.div_npix.div(item);
                        const uint32_t po = item - k * P.npix;
                        pixel = slot_to_pixel(P.slot0 + po, P.idx_start, P.tile_pixels, P.tile_count, P.tile_index, P.div_tile);
                        sample = P.s0 + k * P.chunk;
                        if (PRIMARY) {
                            rec_base = item * P.chunk - sample;
                            hmask = P.use_mask ? P.item_mask[item] : 0xffffffffu;
                        }
                        s_end = min(sample + P.chunk, P.s1);
                        out_slot = P.direct ? pixel : item; // frame[pixel] or part[k][po]
                        accum = v3(0.f, 0.f, 0.f);
                        has_item = pixel < P.idx_stop;      // the owner's last tile may be cut by the range:
                        if (!has_item) s_end = sample;      // an empty item — the lane idles this step and asks again
                        n_paths += s_end - sample;
                    } else {
                        alive = false;
                    }
                }
            }
        }
        // sample == s_end here only for a lane whose new item was empty (cut tile): it must not start a path
        PT_STAMP(tP0);
        if (PRIMARY) {
            // The primary pass has traced every camera ray of this launch against its pixel's candidate list
            // and shaded the first hit (pt_primary).  A record is the path's colour (it ended there: added
            // here, in sample order), or the ray after the first bounce, or — pixels without a list — the
            // camera ray itself.
            // The item's mask tells which of its samples ended in the primary pass: their colours (up to four at a
            // time) and the record after them are fetched together — independent loads, one round trip — instead of
            // walking the records one dependent load after the other.
            if (alive && need_path && sample != s_end) {
                for (;;) {
                    if (!P.use_mask) hmask = 0xffffffffu;
                    const uint32_t left = s_end - sample;
                    const uint32_t f = min(hmask ? (uint32_t)__builtin_ctz(hmask) : 32u, left); // finished samples in front
                    const uint32_t fb = min(f, 4u);
                    const bool look = fb == f && f < left;       // the record right behind this batch needs a look
                    const size_t r = 2u * (size_t)(rec_base + sample);
                    float4 c0, c1, c2, c3, ra, rb;
                    c0 = c1 = c2 = c3 = ra = rb = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (fb > 0u) c0 = P.rec[r];
                    if (fb > 1u) c1 = P.rec[r + 2u];
                    if (fb > 2u) c2 = P.rec[r + 4u];
                    if (fb > 3u) c3 = P.rec[r + 6u];
                    if (look) { ra = P.rec[r + 2u * f]; rb = P.rec[r + 2u * f + 1u]; }
                    if (fb > 0u) accum = vadd(accum, v3(c0.x, c0.y, c0.z)); // renderers/ppm_image.zig:67, in sample order
                    if (fb > 1u) accum = vadd(accum, v3(c1.x, c1.y, c1.z));
                    if (fb > 2u) accum = vadd(accum, v3(c2.x, c2.y, c2.z));
                    if (fb > 3u) accum = vadd(accum, v3(c3.x, c3.y, c3.z));
                    sample += fb;
                    hmask >>= fb;
                    if (look) {
                        const uint32_t meta = __float_as_uint(ra.w);
                        const uint32_t tag = meta >> kTagShift;
                        hmask >>= 1;
                        if (tag >= kTagState0) {
                            o = v3(ra.x, ra.y, ra.z); d = v3(rb.x, rb.y, rb.z);
                            bounce = tag - kTagState0; // 0: camera ray, 1 / 2: after the first / second bounce
                            const uint32_t w = __float_as_uint(rb.w);
                            rng.resume(pixel, sample, bounce == 2u ? (w & ((1u << kDrawBits) - 1u)) : w);
                            if (bounce) P.stack[gtid] = meta & kHitMask;                                  // stack[0][gtid]: the first hit's id
                            if (bounce == 2u) P.stack[(size_t)P.total_threads + gtid] = w >> kDrawBits;   // stack[1][gtid]
                            need_path = false;
                            break;
                        }
                        if (tag == kTagFinal) accum = vadd(accum, v3(ra.x, ra.y, ra.z)); // (no mask: every record comes this way)
                        sample++;                    // kTagSkip: DirectionIsNotUnitVector, counted by the primary pass
                    }
                    if (sample == s_end) {       // item finished without a ray to trace: ask for the next one, or park
                        o = v3(0.f, 0.f, 0.f); d = v3(kNaN, kNaN, kNaN);
                        break;
                    }
                }
            }
        } else if (alive && need_path && sample != s_end) {
            // renderers/ppm_image.zig:50-51,61-63
            rng.start_path(pixel, sample);
            const uint32_t i_row = P.div_width.div(pixel);
            const uint32_t i_col = pixel - i_row * P.width;
            float xi_u, xi_v;
            rng.next2(xi_u, xi_v);
            const float u = ((float)i_col + xi_u) / (float)P.width;
            const float v = ((float)(P.height - i_row + 1u) + xi_v) / (float)P.height;
            cast_ray<DIRECT>(P.cam, u, v, rng, o, d);
            bounce = 0;
            need_path = false;
        }
        PT_STAMP(tP1);
        PT_SUB_ACC(3, tP0, tP1);
        if (!PRIMARY || round == PT_REFILL_ROUNDS - 1) break;
        if (__ballot(alive && need_path) == 0ull) break;
        } // rounds
        if (MULTI_TILE && !TREE) {
            if (!__syncthreads_or(alive ? 1 : 0)) break;
        } else {
            if (__ballot(alive) == 0ull) break;
        }

        // ---- phase B: Scene.is_hit (scene.zig:96-110) ---------------------------
        PT_STAMP(tB0);
        PT_ACC(0, tA0, tB0);
        // conservative filter over the LDS tile(s), exact test on the survivors
        const float a = vlen2(d); // sphere.zig:68
        float closest = P.t_max;
        int hit = -1;
        bool tracing = alive && !need_path; // the lane has a ray in flight (not dead, not idling on an empty item)
        if (TREE) {
            constexpr uint32_t kNone = TREE_LDS ? TreeInLds::kNone : TreeInGlobal::kNone;
            const bool walking = t_cur != kNone || t_sp != 0u || t_pend != kNone;
            if (tracing && !walking) { // a new ray: start its walk
                t_cur = TREE_LDS ? TreeInLds::ref16(P.troot) : P.troot;
                t_closest = P.t_max; t_hit = -1;
            }
            if (TREE_LDS)
                scene_is_hit_tree(TreeInLds{s_tboxes, s_tchild, s_tgeom, s_tprim, s_tstack}, t_cur, t_sp, t_pend, P.tree_exit_at, P.tree_flush_at, tid, o, d, a, P.t_min, P.t_max,
                                  t_closest, t_hit, w_tree_cands, w_tree_iters, w_tree_boxes);
            else
                scene_is_hit_tree(TreeInGlobal{P.tnodes, P.tgeom, P.tprim, s_mask}, t_cur, t_sp, t_pend, P.tree_exit_at, P.tree_flush_at, tid, o, d, a, P.t_min, P.t_max, t_closest,
                                  t_hit, w_tree_cands, w_tree_iters, w_tree_boxes);
            tracing = tracing && t_cur == kNone && t_sp == 0u && t_pend == kNone; // lanes still walking skip this step's shading and keep their ray
            closest = t_closest; hit = t_hit;
        } else
            scene_is_hit_filtered<MULTI_TILE>(P.filt, P.ycy, P.has_shared != 0u, s_ycy, n_pad, s_filt, s_mask, tid, geom_of, o, d, a, P.t_min, P.t_max, closest, hit,
                                              n_cands, n_flush_iters, t_flush);
        w_rays_traced += (unsigned long long)__popcll(__ballot(tracing));
        n_wave_steps++;
        PT_STAMP(tC0);
        PT_ACC(1, tB0, tC0);

        // ---- phase C: radiance step (illumination.zig:76-90) ---------------------
        if (tracing) {
            bool done = false;
            V3 color = v3(0.f, 0.f, 0.f);
            bool ok = true;
            if (hit >= 0) {
                if (P.illum != PT_ILLUM_RADIANCE) {
                    // first-hit colour functions (renderers/ppm_image.zig:64-65)
                    if (P.illum == PT_ILLUM_NORMAL) { // illumination.zig:45-47; sic: about the constant centre (0,0,-1)
                        const V3 p = vadd(o, vmul(d, closest));
                        const V3 n = vunit(vsub(p, v3(0.0f, 0.0f, -1.0f)));
                        color = vmul(vadd(n, v3(1.0f, 1.0f, 1.0f)), 0.5f);
                    } else {                          // illumination.zig:65-69
                        const float4 m = mat_of((uint32_t)hit);
                        color = (kind_of((uint32_t)hit) == 2u) ? v3(1.0f, 1.0f, 1.0f) : v3(m.x, m.y, m.z);
                    }
                    done = true;
                } else if (bounce == P.rebounds) {
                    done = true; // illumination.zig:80-81: black
                } else {
                    const float4 g = geom_of((uint32_t)hit);
                    const float4 m = mat_of((uint32_t)hit);
                    const uint32_t kind = kind_of((uint32_t)hit);
                    V3 p, n;
                    PT_STAMP(tH0);
                    sphere_hit_point_inv(o, d, closest, g.x, g.y, g.z, inv_radius_of((uint32_t)hit), p, n);
                    PT_STAMP(tH1);
                    const V3 nd = scatter_pre<DIRECT>(kind, m.w, m.x, m.y, a, d, p, n, rng); // (a = |d|^2: lerp's unit check and the dielectric's cosine take |d| from it)
                    PT_STAMP(tH2);
                    PT_SUB_ACC(0, tH0, tH1); PT_SUB_ACC(1, tH1, tH2);
#ifndef PT_EXPERIMENT_NO_STACK // tools/build_variant.sh timing experiment (breaks parity)
                    P.stack[(size_t)bounce * P.total_threads + gtid] = (uint32_t)hit;
#endif
                    bounce++;
                    o = p; d = nd;
                }
            } else {
                PT_STAMP(tM0);
                ok = lerp_sky_len2(d, a, P.blend_start, P.blend_stop, color); // illumination.zig:88
                // v.elementwiseMul(s.attenuation), innermost first (illumination.zig:84-85)
                for (uint32_t b = bounce; b > 0; --b) {
#ifndef PT_EXPERIMENT_NO_STACK
                    const uint32_t id = P.stack[(size_t)(b - 1) * P.total_threads + gtid];
#else
                    const uint32_t id = b;
#endif
                    if (kind_of(id) != 2u) {
                        const float4 m = mat_of(id);
                        color = vemul(color, v3(m.x, m.y, m.z));
                    } else {
                        color = vemul(color, v3(1.0f, 1.0f, 1.0f)); // material.zig:135
                    }
                }
                PT_STAMP(tM1);
                PT_SUB_ACC(2, tM0, tM1);
                done = true;
            }
            if (done) {
                if (ok) accum = vadd(accum, color); // renderers/ppm_image.zig:67
                else n_not_unit++;
                sample++;
                need_path = true;
            }
        }
        PT_STAMP(tC1);
        PT_ACC(3, tC0, tC1);
    }

#ifdef PT_PHASE_TIMING
    if (lane == 0) {
        t_phase[4] = __builtin_amdgcn_s_memtime() - t_begin;
        atomicAdd(&P.counters[13], __builtin_amdgcn_s_memrealtime() - r_begin);
        t_phase[2] = t_flush;
        for (int i = 0; i < 5; ++i) atomicAdd(&P.counters[8 + i], t_phase[i]);
        for (int i = 0; i < 4; ++i) atomicAdd(&P.counters[16 + i], t_sub[i]);
    }
#endif
    // ---- statistics ---------------------------------------------------------------
    const unsigned long long w_rays = w_rays_traced, w_cands = TREE ? w_tree_cands : wave_sum_u64(n_cands),
                             w_paths = wave_sum_u64(n_paths), w_not_unit = wave_sum_u64(n_not_unit), w_boxes = w_tree_boxes;
    if (lane == 0) {
        atomicAdd(&P.counters[CNT_RAYS], w_rays);
        atomicAdd(&P.counters[CNT_CANDS], w_cands);
        atomicAdd(&P.counters[CNT_PATHS], w_paths);
        if (w_not_unit) atomicAdd(&P.counters[CNT_NOT_UNIT], w_not_unit);
        if (TREE) atomicAdd(&P.counters[CNT_BOX_TESTS], w_boxes);
        atomicAdd(&P.counters[CNT_WAVE_STEPS], (unsigned long long)n_wave_steps);
        atomicAdd(&P.counters[CNT_FLUSH_ITERS], TREE ? w_tree_iters : (unsigned long long)n_flush_iters);
    }
}

// ---------------------------------------------------------------------------
// the primary pass
// ---------------------------------------------------------------------------
// Camera rays are the one coherent ray population of the path: every ray of a pixel leaves the lens disk
// and crosses that pixel's footprint on the focal plane.  pt_primary_lists finds, per pixel, the spheres ANY
// such ray can hit (a conservative set, in ascending index); pt_primary then gives each camera ray the
// reference's exact Sphere.is_hit on that short list instead of the scan over all spheres — same ascending
// order, same strict '<', so the same closest hit bit for bit — and shades the first hit.  The megakernel
// takes the paths over from the records written here (its scan then only ever sees scattered rays).
//
// The set: a camera ray is the line through O + off and F (camera.zig:99-106), off in the lens disk
// (|off| <= Rl), F = llc + s*horizontal + t*vertical within the pixel's (s, t) box, i.e. |F - F0| <= Rp
// around the box centre F0.  Its point at parameter tau, (1 - tau)(O + off) + tau F, lies within
//   rho(tau) = |1 - tau| Rl + tau (Rp + e_d) + e_o        (tau >= 0)
// of the central line's point O + tau (F0 - O); e_o, e_d bound the f32 rounding of origin and direction.
// A sphere (c, r) the ray hits has a point of the ray at distance r from c, so
//   |c - O - tau Dv| <= r' + rho(tau)   for some tau >= 0,   Dv = F0 - O,
// r'^2 = r^2 + g (2 (|c - O| + Rl)^2 + r^2) covering the rounding of the f32 discriminant (sphere.zig:66-72).
// rho is linear on [0,1] and on [1,inf): on each piece the condition is a quadratic in tau, minimised in
// closed form.  Evaluated in double.  Rays behind tau = 0 do not exist (the list is used for t_min >= 0 only).
struct ListCam {
    double O[3], llc[3], H[3], V[3];
    double Rl, Rp, e_o, e_d;
};
#ifndef PT_METAL_CONE
#define PT_METAL_CONE 1.0 // second-bounce lists off a metal: cone width = spread of the incoming directions + this x the turn of the normal (4 = the bound; pt_primary checks each path; 1 / 2 / 4: 46.9 / 47.6 / 48.0 ms per 256 samples of the headline frame)
#endif
#ifndef PT_GLASS_CONE
#define PT_GLASS_CONE 1.0 // (0.8 / 1.5 / 3: 47.5 / 48.1 / 48.4 ms per 256 samples of the headline frame) second-bounce lists behind a dielectric: cone width in units of (spread of the incoming directions + turn of the normal)
#endif
#ifndef PT_PATCH_FACTOR
#define PT_PATCH_FACTOR 1.05 // measured on the headline frame, 0.5 / 0.75 / 1.05 / 1.3 / 2 / 3: 60.1 / 57.1 / 56.8 / 57.5 / 58.2 / 59.5 ms per 256 samples; second-bounce lists: patch radius in units of the camera rays' footprint on the sphere (a choice: pt_primary checks each hit point)
#endif
constexpr double kListGamma = 4e-6; // >= 9x the rounding bound of the f32 discriminant: 2^-24 (14 |oc|^2 + 6 r^2) a

__device__ __forceinline__ bool list_piece(double DD, double xx, double xd, double R, double beta, double ta, double tb) {
    const double A = DD - beta * beta;
    if (!(A > 0.0)) return true; // degenerate camera (footprint wider than the focal distance): keep the sphere
    const double Bh = xd + R * beta;
    double tm = Bh / A;
    tm = fmin(fmax(tm, ta), tb);
    const double g = (A * tm - 2.0 * Bh) * tm + xx - R * R;
    return g <= 1e-9 * (xx + R * R + DD * tm * tm);
}

// Per sphere, what pt_primary_lists needs and no pixel changes: x = c - O, |x|^2, r', and f32 upper bounds of
// |x| and r' for the quick reject.
struct ListSphere {
    double x[3], xx, reff;
    float xl, rub, r, xf[3]; // xf = (float)x
};
__global__ __launch_bounds__(256) void pt_list_prep(const ListCam C, const float4 *__restrict__ geom, uint32_t nspheres, ListSphere *__restrict__ out) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= nspheres) return;
    const float4 g = geom[j]; // (cx, cy, cz, radius)
    ListSphere S;
    S.x[0] = (double)g.x - C.O[0]; S.x[1] = (double)g.y - C.O[1]; S.x[2] = (double)g.z - C.O[2];
    S.xx = S.x[0] * S.x[0] + S.x[1] * S.x[1] + S.x[2] * S.x[2];
    const double r2 = (double)g.w * (double)g.w;
    const double xl = sqrt(S.xx) + C.Rl;
    S.reff = sqrt(r2 + kListGamma * (2.0 * xl * xl + r2));
    S.xl = sqrtf((float)S.xx) * (1.0f + 1e-6f);                  // >= |x|
    const float xr = S.xl + (float)C.Rl * (1.0f + 1e-6f);
    S.rub = (fabsf(g.w) + sqrtf((float)kListGamma * (2.0f * xr * xr + g.w * g.w))) * (1.0f + 1e-5f); // >= r'
    S.r = g.w;
    for (int i = 0; i < 3; ++i) S.xf[i] = (float)S.x[i];
    out[j] = S;
}

// One lane's walk of the sphere tree for the list kernel: box_ok(bc, bh) says whether a box (centre, half extents) can hold a sphere the
// list wants, leaf(slot) looks at the sphere in leaf slot `slot` and returns false to end the walk (the list has
// overflowed).  (Lanes walk independently: no wave-level operation inside; neighbouring pixels take nearly the same way.)
template <class BoxOk, class Leaf>
__device__ __forceinline__ void list_tree_walk(const TreeNode *__restrict__ nodes, uint32_t root, uint32_t *s_stack, uint32_t tid, bool active, BoxOk box_ok,
                                               Leaf leaf) {
    uint32_t cur = active ? root : kTreeNone, sp = 0;
    while (cur != kTreeNone) {
        if (cur & kTreeLeafBit) {
            const uint32_t first = TreeInGlobal::first(cur), cnt = TreeInGlobal::count(cur);
            bool go_on = true;
            for (uint32_t k = 0; k < cnt; ++k) go_on = leaf(first + k) && go_on;
            cur = (go_on && sp > 0u) ? s_stack[--sp * 256u + tid] : kTreeNone;
        } else {
            const TreeNode *nd = nodes + cur;
            const float4 A = nd->a, B = nd->b, Cc = nd->c;
            const uint4 ch = nd->ch;
            const bool h0 = box_ok(A.x, A.y, A.z, A.w, B.x, B.y);
            const bool h1 = box_ok(B.z, B.w, Cc.x, Cc.y, Cc.z, Cc.w);
            if (h0 && h1) { s_stack[sp++ * 256u + tid] = ch.y; cur = ch.x; }
            else if (h0 || h1) cur = h0 ? ch.x : ch.y;
            else cur = sp > 0u ? s_stack[--sp * 256u + tid] : kTreeNone;
        }
    }
}

// TREE: the two passes over the spheres walk the scene's sphere tree instead (large scenes): a box is entered when the
// ball around it — centre bc, radius hd + the largest r' - r of a sphere inside — passes the very test its spheres take
// (a sphere inside the box lies inside that ball, and both tests only get easier for a larger ball around the same
// points), so the lists hold the same spheres, in walk order instead of ascending (pt_primary's tie rule covers that).
template <bool TREE>
__global__ __launch_bounds__(256) void pt_primary_lists(const ListCam C, uint32_t width, uint32_t height, uint32_t slot0, uint32_t npix,
                                                        uint32_t idx_start, uint32_t idx_stop, uint32_t tile_pixels, uint32_t tile_count,
                                                        uint32_t tile_index, const ListSphere *__restrict__ sph, uint32_t nspheres,
                                                        const uint32_t *__restrict__ kind, const float4 *__restrict__ mat, float t_min,
                                                        uint32_t second_bounce, uint32_t *__restrict__ lists, const TreeNode *__restrict__ tnodes,
                                                        const uint32_t *__restrict__ tprim, uint32_t troot) {
    // The per-sphere table goes through LDS in tiles (broadcast reads): read straight from memory the loops below were
    // bound by the latency of one scalar load per sphere at three waves per SIMD.
    __shared__ ListSphere s_sph[TREE ? 1 : kListTile];
    __shared__ uint32_t s_stack[TREE ? kTreeStack * 256 : 1];
    const uint32_t po = blockIdx.x * 256u + threadIdx.x;
    bool act = po < npix;                      // this thread still builds list A
    uint32_t *L = lists + (size_t)(act ? po : 0u) * kListStride;
    uint32_t pixel = 0;
    if (act) {
        L[kListS] = kListEnd;
        pixel = slot_to_pixel(slot0 + po, idx_start, tile_pixels, tile_count, tile_index);
        if (pixel >= idx_stop) { L[0] = kListEnd; act = false; }
    }
    const uint32_t col = pixel % width, row = pixel / width;
    // the (s, t) box of renderers/ppm_image.zig:61-62 (sic: +1 in v), centre
    const double sm = ((double)col + 0.5) / (double)width;
    const double tm = ((double)(height - row + 1u) + 0.5) / (double)height;
    double Dv[3], DD = 0.0;
    for (int i = 0; i < 3; ++i) {
        Dv[i] = C.llc[i] + sm * C.H[i] + tm * C.V[i] - C.O[i];
        DD += Dv[i] * Dv[i];
    }
    const double a1 = C.Rl + C.e_o, b1 = (C.Rp + C.e_d) - C.Rl;   // tau in [0, 1]
    const double a2 = C.e_o - C.Rl, b2 = (C.Rp + C.e_d) + C.Rl;   // tau in [1, inf)
    // Quick reject, cheap and conservative, before the two quadratic tests: rho(tau) <= a1 + b2 tau; a sphere
    // touched at tau has tau |Dv| <= |x| + r' + rho(tau), i.e. tau <= tau_hi = (|x| + r' + a1) / (|Dv| - b2); and
    // its centre is no further from the central LINE than r' + a1 + b2 tau_hi.  Upper bounds in f32 (inflated),
    // the distance to the line in double.
    const double invDD = 1.0 / DD;
    const float lenD = (float)sqrt(DD) * (1.0f - 1e-6f) - (float)b2 * (1.0f + 1e-6f);
    const float inv_lenD = lenD > 0.0f ? (1.0f + 1e-5f) / lenD : 0.0f;
    const float a1f = (float)a1 * (1.0f + 1e-6f), b2f = (float)b2 * (1.0f + 1e-6f);
    uint32_t count = 0;
    // the ball around a box (centre bc, half extents bh), relative to the camera: x = bc - O, |x|^2, radius hd (rounded up)
    auto box_ball = [&](float cx, float cy, float cz, float hx, float hy, float hz, double x[3], double &xx, double &hd) {
        x[0] = (double)cx - C.O[0]; x[1] = (double)cy - C.O[1]; x[2] = (double)cz - C.O[2];
        xx = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
        hd = sqrt((double)hx * hx + (double)hy * hy + (double)hz * hz) * (1.0 + 1e-12);
    };
    if (TREE) {
        list_tree_walk(tnodes, troot, s_stack, threadIdx.x, act,
            [&](float cx, float cy, float cz, float hx, float hy, float hz) -> bool {
                double x[3], xx, hd;
                box_ball(cx, cy, cz, hx, hy, hz, x, xx, hd);
                const double xl = sqrt(xx) + hd + C.Rl;                                // >= |c - O| + Rl of every sphere inside
                double R = hd + sqrt(kListGamma * (2.0 * xl * xl + hd * hd)) * (1.0 + 1e-9); // >= hd - r + r' of each of them (pt_list_prep's r')
                // list_piece lets a sphere in that is outside by up to its tolerance (1e-9 relative in squared distance, < 1e-4 (|x| + R)
                // in distance): the ball takes that much more, so its own tolerance never matters
                R += 1e-4 * (xl + R + fabs(a1) + fabs(a2));
                const double xd = x[0] * Dv[0] + x[1] * Dv[1] + x[2] * Dv[2];
                if (lenD > 0.0f) { // the spheres' quick reject (below), for the ball: its centre is too far from the central line
                    const float base = (float)R * (1.0f + 1e-6f) + a1f;
                    const float tau_hi = ((float)xl * (1.0f + 1e-6f) + base) * inv_lenD;
                    const float Rmax = (base + b2f * tau_hi) * (1.0f + 1e-5f);
                    if (xx - xd * xd * invDD > (double)Rmax * (double)Rmax) return false;
                }
                return list_piece(DD, xx, xd, R + a1, b1, 0.0, 1.0) || list_piece(DD, xx, xd, R + a2, b2, 1.0, 1e300);
            },
            [&](uint32_t slot) {
                const uint32_t j = tprim[slot];
                const ListSphere &S = sph[j];
                const double xx = S.xx;
                const double xd = S.x[0] * Dv[0] + S.x[1] * Dv[1] + S.x[2] * Dv[2];
                if (list_piece(DD, xx, xd, S.reff + a1, b1, 0.0, 1.0) || list_piece(DD, xx, xd, S.reff + a2, b2, 1.0, 1e300)) {
                    if (count < kListCap) L[count] = j;
                    count++;
                }
                return count <= kListCap; // one too many: the pixel's camera rays take the megakernel's walk, no need to count on
            });
    }
    for (uint32_t t0 = 0; !TREE && t0 < nspheres; t0 += kListTile) {
        const uint32_t tn = min(kListTile, nspheres - t0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tn; i += 256u) s_sph[i] = sph[t0 + i];
        __syncthreads();
        if (!act) continue;
#pragma unroll 4
        for (uint32_t jj = 0; jj < tn; ++jj) {
            const ListSphere &S = s_sph[jj]; // wave-uniform address
            const double xx = S.xx;
            const double xd = S.x[0] * Dv[0] + S.x[1] * Dv[1] + S.x[2] * Dv[2];
            if (lenD > 0.0f) {
                const float base = S.rub + a1f;
                const float tau_hi = (S.xl + base) * inv_lenD;
                const float Rmax = (base + b2f * tau_hi) * (1.0f + 1e-5f);
                if (xx - xd * xd * invDD > (double)Rmax * (double)Rmax) continue;
            }
            if (list_piece(DD, xx, xd, S.reff + a1, b1, 0.0, 1.0) || list_piece(DD, xx, xd, S.reff + a2, b2, 1.0, 1e300)) {
                if (count < kListCap) L[count] = t0 + jj;
                count++;
            }
        }
    }
    if (act) {
        if (count > kListCap) { L[0] = kListOverflow; act = false; }  // too many: this pixel's camera rays take the scan
        else if (count < kListCap) L[count] = kListEnd;
    }
    bool doB = act && second_bounce && count != 0u; // this thread builds list B

    // ---- list B: what a ray scattered off the pixel's main sphere can hit ---------------------------------
    // S = a sphere the pixel's CENTRAL ray hits (Lambertian, material.zig:43-51: direction unit(n + rnd), rnd in [0,1)^3;
    // Metal, material.zig:79-88: unit(reflect(d, n) + fuzz rnd); Dielectric: the refracted direction).  The list is built for rays that start within rho_p of the central hit point p0 on S and run
    // inside a cone (axis a, half-angle alpha) — pt_primary checks both of each path's own hit point and direction before it
    // uses the list, so patch and cone are choices, not bounds; they are chosen to hold nearly all of the pixel's rays: with
    // n + rnd inside the box [n0 - eps, n0 + 1 + eps]^3 (eps: the turn of the normal over the patch + f32 rounding),
    // i.e. inside the circular cone (axis a, half-angle alpha) around that box.  A sphere (c, r) such a ray hits has
    // its centre within r' + rho_p of the cone with apex p0:  angle(c - p0, a) <= alpha + asin((r' + rho_p) / |c - p0|).
    // Candidates for S: the spheres of list A the central ray enters (near root beyond t_min), nearest first;
    // the first one whose patch is flat enough for a cone wins (a small sphere in front of the ground is too curved under
    // a wide lens: the ground behind it then gets the list, for the pixel's rays that pass the small one).
    const double lenDv = sqrt(DD), tau_min = (double)t_min / lenDv;
    constexpr uint32_t kNear = 4;   // the nearest few are enough: S is almost always the first or the second of them
    double taus[kNear];
    uint32_t order[kNear], ncand = 0;
    for (uint32_t i = 0; doB && i < count; ++i) {
        const uint32_t j = L[i];
        if (kind[j] > 2u) continue;   // every kind: their scattered rays fill a cone, or mostly do (below)
        const ListSphere &S = sph[j];
        const double xd = S.x[0] * Dv[0] + S.x[1] * Dv[1] + S.x[2] * Dv[2];
        const double disc = xd * xd - DD * (S.xx - (double)S.r * (double)S.r);
        if (disc < 0.0) continue;
        const double tau = (xd - sqrt(disc)) * invDD; // near root only, as sphere.zig:75-86
        if (!(tau > tau_min)) continue;
        // insert into the kNear nearest, ascending (fixed-size, fully unrolled: the arrays stay in registers)
        double tk = tau; uint32_t jk = j;
#pragma unroll
        for (uint32_t k = 0; k < kNear; ++k) {
            const bool empty = k >= ncand;
            if (empty || tk < taus[k]) {
                const double tt = empty ? tk : taus[k]; const uint32_t jt = empty ? jk : order[k];
                taus[k] = tk; order[k] = jk; tk = tt; jk = jt;
                if (empty) break;
            }
        }
        if (ncand < kNear) ncand++;
    }
    uint32_t Sj = kListEnd;
    double tau0 = 0.0, p0[3], ax[3], rho_p = 0.0, cosA = 1.0, sinA = 0.0;
    for (uint32_t ic = 0; ic < ncand && Sj == kListEnd; ++ic) {
        const ListSphere &SS = sph[order[ic]];
        tau0 = taus[ic];
        double y0[3], n0[3], yl0 = 0.0, pl0 = 0.0, ndv = 0.0;
        for (int i = 0; i < 3; ++i) {
            y0[i] = tau0 * Dv[i] - SS.x[i];          // p0 - c_S
            p0[i] = C.O[i] + tau0 * Dv[i];
            yl0 += y0[i] * y0[i]; pl0 += p0[i] * p0[i];
        }
        yl0 = sqrt(yl0); pl0 = sqrt(pl0);
        if (!(yl0 > 0.0)) continue;
        const double sgn = SS.r < 0.0f ? -1.0 : 1.0; // sphere.zig:88: (p - c) / radius, a negative radius turns the normal inwards
        for (int i = 0; i < 3; ++i) { n0[i] = sgn * y0[i] / yl0; ndv += n0[i] * Dv[i]; }
        const double cos_in = fabs(ndv) / lenDv;
        const double rho0 = tau0 <= 1.0 ? a1 + b1 * tau0 : a2 + b2 * tau0;
        rho_p = PT_PATCH_FACTOR * rho0 / fmax(cos_in, 0.05) + 2e-4 * (pl0 + 1.0);
        const double eps_n = rho_p / fabs((double)SS.r) * 1.001;
        if (!(eps_n <= 0.5)) continue;
        const double eps = eps_n + 1e-5 * (pl0 + rho_p + 2.0);
        if (kind[order[ic]] == 0u) {
            // Lambertian (material.zig:43-51): unit(n + rnd), rnd in [0,1)^3 — the box [n0 - eps, n0 + 1 + eps]^3
            double al = 0.0;
            for (int i = 0; i < 3; ++i) { ax[i] = n0[i] + 0.5; al += ax[i] * ax[i]; } // box centre
            al = sqrt(al);
            if (!(al > 1e-3)) continue;
            for (int i = 0; i < 3; ++i) ax[i] /= al;
            cosA = 1.0;
            bool ok = true;
            for (int c = 0; c < 8; ++c) {
                double u[3], ul = 0.0, au = 0.0;
                for (int i = 0; i < 3; ++i) {
                    u[i] = ((c >> i) & 1) ? n0[i] + 1.0 + eps : n0[i] - eps;
                    ul += u[i] * u[i]; au += ax[i] * u[i];
                }
                ul = sqrt(ul);
                if (!(ul > 1e-6)) { ok = false; break; }
                cosA = fmin(cosA, au / ul);
            }
            if (!ok || !(cosA >= 0.1)) continue;      // the box must lie well inside a half-space (it must not reach the origin)
            sinA = sqrt(fmax(0.0, 1.0 - cosA * cosA));
        } else if (kind[order[ic]] == 2u) {
            // Dielectric (material.zig:129-167) seen from outside: the refracted ray, which most paths take (the reflected ones
            // fail pt_primary's direction check and go on in the megakernel).  The cone is a guess around refract(d0, n0),
            // as wide as the incoming spread and the turn of the normal suggest.
            double dn = 0.0;
            for (int i = 0; i < 3; ++i) dn += Dv[i] * n0[i];
            dn /= lenDv;
            if (!(dn < 0.0)) continue;                        // from inside: not handled
            const double eta = (double)mat[order[ic]].x;      // 1 / refraction index (the table holds it)
            const double cos_t = fmin(-dn, 1.0);
            double rp[3], rp2 = 0.0;
            for (int i = 0; i < 3; ++i) { rp[i] = eta * (Dv[i] / lenDv + cos_t * n0[i]); rp2 += rp[i] * rp[i]; }
            if (!(rp2 < 1.0)) continue;
            const double par = -sqrt(1.0 - rp2);
            double al = 0.0;
            for (int i = 0; i < 3; ++i) { ax[i] = rp[i] + par * n0[i]; al += ax[i] * ax[i]; }
            al = sqrt(al);
            if (!(al > 0.5)) continue;
            for (int i = 0; i < 3; ++i) ax[i] /= al;
            const double w = PT_GLASS_CONE * ((C.Rl + C.Rp + C.e_d + C.e_o) / lenDv + eps) + 0.02;
            if (!(w <= 0.95)) continue;
            sinA = w;
            cosA = sqrt(1.0 - w * w);
        } else {
            // Metal (material.zig:79-88): unit(reflect(d, n) + fuzz * rnd).  reflect is an isometry in d and
            // |reflect(d0, n) - reflect(d0, n0)| <= 4 |n - n0| for unit vectors, so the vector before normalisation lies
            // within w = |d - d0| + 4 eps + sqrt(3) fuzz of the unit vector reflect(d0, n0): a cone of half-angle asin(w).
            // |d - d0| <= (Rl + Rp + e_d + e_o) / |Dv| (1 + small): directions from lens points to footprint points.
            const double fuzz = fabs((double)mat[order[ic]].w);
            double dn = 0.0;
            for (int i = 0; i < 3; ++i) dn += Dv[i] * n0[i];
            dn /= lenDv;
            double al = 0.0;
            for (int i = 0; i < 3; ++i) { ax[i] = Dv[i] / lenDv - 2.0 * dn * n0[i]; al += ax[i] * ax[i]; }
            al = sqrt(al);
            if (!(al > 0.5)) continue;
            for (int i = 0; i < 3; ++i) ax[i] /= al;
            const double dd_raw = (C.Rl + C.Rp + C.e_d + C.e_o) / lenDv;    // sin of the largest angle between d and d0
            if (!(dd_raw <= 0.2)) continue;
            const double dd = 1.02 * dd_raw;                                 // |d - d0| = 2 sin(angle / 2) <= 1.005 sin(angle) there
            const double w = fmax(dd + PT_METAL_CONE * eps + 1.7320508075688773 * fuzz * (1.0 + 1e-6) + 1e-5, 0.02); // (not narrower than the run-time check can tell)
            if (!(w <= 0.95)) continue;
            sinA = w;
            cosA = sqrt(1.0 - w * w);
        }
        Sj = order[ic];
    }
    doB = doB && Sj != kListEnd;
    if (!TREE && !__syncthreads_or(doB ? 1 : 0)) return;   // nobody in this workgroup: skip the second pass over the spheres
    uint32_t *B = L + kListB;
    uint32_t cnt2 = 0;
    // The condition, (y.a) >= cosA sqrt(|y|^2 - R^2) - sinA R with y = c_j - p0 and R = r' + rho_p, is evaluated in f32
    // with every bound taken on the permissive side (relative margins of 1e-5 against rounding of 1e-7: the lists grow by
    // nothing measurable); v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf this file is compiled with.
    float tdf[3], axf[3], tdl = 0.f;
    for (int i = 0; i < 3; ++i) { tdf[i] = (float)(tau0 * Dv[i]); axf[i] = (float)ax[i]; tdl += tdf[i] * tdf[i]; }
    tdl = __builtin_amdgcn_sqrtf(tdl) * (1.0f + 1e-6f);
    const float cosAf = (float)cosA * (1.0f - 1e-5f) - 1e-5f, sinAf = (float)sinA * (1.0f + 1e-5f) + 1e-5f;
    const float rho_pf = (float)rho_p * (1.0f + 1e-6f), gammaf = (float)kListGamma * (1.0f + 1e-6f);
    // sphere j against the cone; true = on the list
    auto cone_sphere = [&](const ListSphere &S) -> bool {
        const float yf0 = S.xf[0] - tdf[0], yf1 = S.xf[1] - tdf[1], yf2 = S.xf[2] - tdf[2];
        const float err = 1e-6f * (S.xl + tdl);              // >= the error of yf (inputs rounded to f32, three subtractions)
        const float yl = __builtin_amdgcn_sqrtf(yf0 * yf0 + yf1 * yf1 + yf2 * yf2);
        const float ylu = yl * (1.0f + 1e-6f) + err + rho_pf;
        const float Rf = (fabsf(S.r) + __builtin_amdgcn_sqrtf(gammaf * (2.0f * ylu * ylu + S.r * S.r)) + rho_pf) * (1.0f + 1e-5f) + err;
        if (yl > Rf) {
            const float ya = yf0 * axf[0] + yf1 * axf[1] + yf2 * axf[2] + err + 1e-6f * yl;
            const float yll = fmaxf(yl * (1.0f - 1e-6f) - err, 0.0f);
            const float rhs = cosAf * __builtin_amdgcn_sqrtf(fmaxf(yll * yll - Rf * Rf, 0.0f)) * (1.0f - 1e-5f) - sinAf * Rf;
            if (ya < rhs - 1e-5f * (yl + Rf)) return false;
        }
        return true;
    };
    if (TREE) {
        // the ball of a box against the cone, in double with the same permissive bounds: radius hd + (r' - r) + rho_p + what the f32 test
        // above may add to a sphere's radius (1e-5 relative + err), so that a sphere that test lets in always finds its boxes open
        const double cosAd = (double)cosAf, sinAd = (double)sinAf;
        list_tree_walk(tnodes, troot, s_stack, threadIdx.x, doB,
            [&](float cx, float cy, float cz, float hx, float hy, float hz) -> bool {
                double x[3], xx, hd;
                box_ball(cx, cy, cz, hx, hy, hz, x, xx, hd);
                double y[3], yy = 0.0, ya = 0.0;
                for (int i = 0; i < 3; ++i) { y[i] = x[i] - tau0 * Dv[i]; yy += y[i] * y[i]; ya += y[i] * ax[i]; }
                const double yl = sqrt(yy), ylu = yl + hd + rho_p;
                const double err = 2e-6 * (sqrt(xx) + hd + (double)tdl);
                const double R = (hd + sqrt(kListGamma * 1.001 * (2.0 * ylu * ylu + hd * hd)) + rho_p) * (1.0 + 3e-5) + 2.0 * err;
                if (!(yl > R)) return true;
                const double rhs = cosAd * sqrt(fmax(yy - R * R, 0.0)) * (1.0 - 1e-5) - sinAd * R;
                return !(ya + 1e-4 * yl + 1e-5 * (yl + R) < rhs); // (1e-4: more than the permissive terms of the sphere test can differ between a sphere and its box)
            },
            [&](uint32_t slot) {
                const uint32_t j = tprim[slot];
                if (cone_sphere(sph[j])) {
                    if (cnt2 < kListCapB) B[cnt2] = j;
                    cnt2++;
                }
                return cnt2 <= kListCapB;
            });
    }
    for (uint32_t t0 = 0; !TREE && t0 < nspheres; t0 += kListTile) {
        const uint32_t tn = min(kListTile, nspheres - t0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tn; i += 256u) s_sph[i] = sph[t0 + i];
        __syncthreads();
        if (!doB) continue;
#pragma unroll 4
        for (uint32_t jj = 0; jj < tn; ++jj) {
            if (!cone_sphere(s_sph[jj])) continue;
            if (cnt2 < kListCapB) B[cnt2] = t0 + jj;
            cnt2++;
        }
    }
    if (!doB || cnt2 > kListCapB) return;
    if (cnt2 < kListCapB) B[cnt2] = kListEnd;
    L[kListP0] = __float_as_uint((float)p0[0]); L[kListP0 + 1] = __float_as_uint((float)p0[1]); L[kListP0 + 2] = __float_as_uint((float)p0[2]);
    const double rc = 0.98 * rho_p;
    L[kListRho2] = __float_as_uint((float)(rc * rc) * (1.0f - 1e-6f));
    L[kListAxis] = __float_as_uint(axf[0]); L[kListAxis + 1] = __float_as_uint(axf[1]); L[kListAxis + 2] = __float_as_uint(axf[2]);
    L[kListCos] = __float_as_uint(fminf((float)cosA * (1.0f + 1e-5f) + 1e-5f, 2.0f)); // what pt_primary demands of each path's direction
    L[kListS] = Sj;
}

// One thread per path of the launch, in record order (item-major, the item's samples consecutive): the
// camera ray (renderers/ppm_image.zig:50-51,61-63), Scene.is_hit over the pixel's list, one radiance step
// (illumination.zig:76-90 at depth 0).
// CHECK (ptgpu_debug_list_safety; never in a render): at both list loops the ray is ALSO given the exact test on every sphere of the
// scene, and each sphere that test accepts must be in the list the path is about to use — the property the lists rest on, checked
// sphere by sphere under exactly the conditions pt_primary uses a list, instead of through frames (where a missing sphere only shows
// when it was the closest).  Counted: accepted spheres and missing ones per list, closest hits that differ.
enum { CNT_CHECK_ACC_A = 24, CNT_CHECK_MISS_A = 25, CNT_CHECK_ACC_B = 26, CNT_CHECK_MISS_B = 27, CNT_CHECK_DIFF = 28, CNT_CHECK_RAYS_A = 29, CNT_CHECK_RAYS_B = 30 }; // (the phase-timing slots)
template <int MODE, bool MULTI_TILE, bool CHECK = false>
__global__ __launch_bounds__(256) void pt_primary(const KParams P, uint32_t nrec) {
    constexpr bool DET = MODE == kModeDet, DIRECT = MODE == kModeDirect;
    // scenes of one tile keep the sphere tables in LDS (19 KiB; the grid-stride loop reuses them): the candidate loop's
    // geometry reads were dependent global loads, and the kernel waited on memory 42 % of its wave cycles
    __shared__ float4 s_geom[MULTI_TILE ? 1 : kTile];
    __shared__ float4 s_mat[MULTI_TILE ? 1 : kTile];
    __shared__ float s_inv_radius[MULTI_TILE ? 1 : kTile];
    __shared__ uint8_t s_kind[MULTI_TILE ? 4 : kTile];
    if (!MULTI_TILE) {
        for (uint32_t i = threadIdx.x; i < P.nspheres; i += 256u) {
            s_geom[i] = P.geom[i]; s_mat[i] = P.mat[i]; s_inv_radius[i] = P.inv_radius[i]; s_kind[i] = (uint8_t)P.kind[i];
        }
        __syncthreads();
    }
    auto geom_of = [&](uint32_t j) -> float4 { return MULTI_TILE ? P.geom[j] : s_geom[j]; };
    auto mat_of = [&](uint32_t j) -> float4 { return MULTI_TILE ? P.mat[j] : s_mat[j]; };
    auto inv_radius_of = [&](uint32_t j) -> float { return MULTI_TILE ? P.inv_radius[j] : s_inv_radius[j]; };
    auto kind_of = [&](uint32_t j) -> uint32_t { return MULTI_TILE ? P.kind[j] : (uint32_t)s_kind[j]; };
    // grid-stride: the counters below cost three atomics per WAVE OF THE GRID, not per 64 paths (same-address
    // atomics serialise in L2: one per wave of paths made this kernel 15 times slower)
    uint32_t rays = 0, rays2 = 0, cands = 0, not_unit = 0;
    uint32_t chk[5] = {0, 0, 0, 0, 0}; // CHECK: accepted / missing of list A, of list B, closest hits that differ
#ifdef PT_PHASE_TIMING
    unsigned long long t_pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // path start, list trace, shade, record write, whole
    const unsigned long long tpp_begin = __builtin_amdgcn_s_memtime();
#define PP_ACC(i, a, b) t_pp[i] += (b) - (a)
#else
#define PP_ACC(i, a, b)
#endif
    // A wave takes 64 consecutive records (whole items) per turn.  The order of the turns walks the CHUNKS OF A PIXEL GROUP before it moves to
    // the next pixels (records are laid out chunk-major, item = chunk index * npix + pixel slot): the 176-byte candidate lists of those pixels
    // are then fetched from memory once and found in the caches by the other chunks' waves, instead of once per chunk (64 times per frame:
    // 24 GB of the headline frame's HBM reads, and the loads the candidate loop waits for).
    const uint32_t nblocks = (nrec + 63u) >> 6, wave0 = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
    for (uint32_t v = wave0; v < nblocks; v += nwaves) {
        PT_STAMP(tq0);
        uint32_t blk = v;
        if (P.blocks_per_row) { const uint32_t pb = P.div_nchunks.div(v); blk = (v - pb * P.nchunks) * P.blocks_per_row + pb; }
        const uint32_t r = (blk << 6) + (threadIdx.x & 63u);
        const uint32_t item = P.div_chunk.div(r), jj = r - item * P.chunk;
        const uint32_t k = P.div_npix.div(item), po = item - k * P.npix;
        const uint32_t pixel = slot_to_pixel(P.slot0 + po, P.idx_start, P.tile_pixels, P.tile_count, P.tile_index, P.div_tile);
        const uint32_t first = P.s0 + k * P.chunk;
        const uint32_t sample = first + jj;
        bool look = false; // the megakernel has to look at this path's record (a state to continue, or a dropped sample)
        if (r < nrec && pixel < P.idx_stop && sample < min(first + P.chunk, P.s1)) {
            PT_STAMP(tq00);
            PP_ACC(6, tq0, tq00);
            typename std::conditional<DET, ConstRng, PhiloxRng>::type rng;
            rng.init(P.seed_lo, P.seed_hi);
            rng.start_path(pixel, sample);
            const uint32_t i_row = P.div_width.div(pixel);
            const uint32_t i_col = pixel - i_row * P.width;
            float xi_u, xi_v;
            rng.next2(xi_u, xi_v);
            const float u = ((float)i_col + xi_u) / (float)P.width;
            const float v = ((float)(P.height - i_row + 1u) + xi_v) / (float)P.height;
            const V3 disk = random_vector_in_unit_disk<DIRECT>(rng);
            V3 o, d;
            cast_ray_from_disk(P.cam, u, v, disk, o, d);
            const uint32_t *L = P.lists + (size_t)po * kListStride;
            // a disk draw kept after PT_MAX_TRIES failed tries may lie outside the lens the list was built for
            // (asking for the list before the path's draws, so that the loads run under them, costs 16 VGPRs and the sixth wave: 14.3 vs 14.2 ms)
#ifdef PT_EXPERIMENT_TRIES1 // tools/build_variant.sh with -DPT_MAX_TRIES=1: what the rejection loops' retries cost this kernel (timing only, breaks parity)
            const bool listed = L[0] != kListOverflow;
#else
            const bool listed = L[0] != kListOverflow && vlen2(disk) < 1.0f;
#endif
            // the record if nothing below gets further: the camera ray itself
            float4 ra = make_float4(o.x, o.y, o.z, __uint_as_float(kTagState0 << kTagShift));
            float4 rb = make_float4(d.x, d.y, d.z, __uint_as_float(rng.drawn()));
            PT_STAMP(tq1);
            PP_ACC(0, tq0, tq1);
            if (listed) {
                uint32_t hit0 = 0;
                // the pixel's second-bounce header (sphere S, patch centre and radius^2, cone axis and cosine), fetched now with the
                // list itself: asked for only when the first scatter is done, these loads were a round trip in the open
                static_assert(kListS == 16 && kListP0 == 17 && kListRho2 == 20 && kListAxis == 21 && kListCos == 24, "header read as 2 x uint4 + 1");
                const uint4 hd0 = reinterpret_cast<const uint4 *>(L)[4], hd1 = reinterpret_cast<const uint4 *>(L)[5];
                const uint32_t hd_cos = L[kListCos];
                // depth 0: the camera ray over list A; depth 1: the ray scattered off the pixel's main sphere over list B
                for (uint32_t depth = 0;; ++depth) {
                    PT_STAMP(tl0);
                    const uint32_t *Ld = depth ? L + kListB : L;
                    const float a = vlen2(d); // sphere.zig:68
                    float closest = P.t_max;
                    int hit = -1;
                    // the whole list at once (four independent 16-byte loads), not entry by entry
                    static_assert(kListCap == 16 && kListCapB == 16 && kListB % 4 == 0 && kListStride % 4 == 0, "lists are read as 4 x uint4");
                    const uint4 *L4 = reinterpret_cast<const uint4 *>(Ld);
                    const uint4 e0 = L4[0], e1 = L4[1], e2 = L4[2], e3 = L4[3];
                    const uint32_t ent[16] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w, e3.x, e3.y, e3.z, e3.w};
                    bool more = true; // (a flag instead of a break: the steps unroll, ent[] stays in registers)
                    auto test_entry = [&](uint32_t j) { // ascending sphere index, strict '<' (scene.zig:102)
                        more = more && j != kListEnd;
                        if (more) {
                            const float4 g = geom_of(j);
                            cands++;
                            float t;
                            if (sphere_near_root(o, d, a, g.x, g.y, g.z, g.w, P.t_min, P.t_max, t)) {
                                // lists built by a tree walk (large scenes) are in no particular order: among equal t the lower index
                                // wins explicitly, which is what the ascending order gives the strict '<' for free
                                if (t < closest || (MULTI_TILE && t == closest && (int)j < hit)) { closest = t; hit = (int)j; }
                            }
                        }
                    };
                    // four entries at a time; a group is skipped by the whole wave once every lane's list has ended (two to four spheres as a rule)
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) {
                        if (q > 0u && __ballot(more) == 0ull) continue;
#pragma unroll
                        for (uint32_t i = 0; i < 4u; ++i) test_entry(ent[4u * q + i]);
                    }
                    if (depth) rays2++; else rays++;
                    if (CHECK) {
                        float c_closest = P.t_max; int c_hit = -1;
                        for (uint32_t j = 0; j < P.nspheres; ++j) { // scene.zig:96-110 as written
                            const float4 g = geom_of(j);
                            float t;
                            if (!sphere_near_root(o, d, a, g.x, g.y, g.z, g.w, P.t_min, P.t_max, t)) continue;
                            if (t < c_closest) { c_closest = t; c_hit = (int)j; }
                            bool in_list = false, open = true;
                            for (uint32_t i = 0; i < 16u; ++i) { open = open && ent[i] != kListEnd; in_list = in_list || (open && ent[i] == j); }
                            chk[depth ? 2 : 0]++;
                            if (!in_list) chk[depth ? 3 : 1]++;
                        }
                        if (c_hit != hit || (hit >= 0 && __float_as_uint(c_closest) != __float_as_uint(closest))) chk[4]++;
                    }
                    PT_STAMP(tl1);
                    PP_ACC(1 + (depth ? 2 : 0), tl0, tl1);
                    V3 color = v3(0.f, 0.f, 0.f);
                    uint32_t tag = kTagFinal;
                    if (hit >= 0) {
                        if (P.illum != PT_ILLUM_RADIANCE) {
                            if (P.illum == PT_ILLUM_NORMAL) { // illumination.zig:45-47; sic: about the constant centre (0,0,-1)
                                const V3 p = vadd(o, vmul(d, closest));
                                const V3 n = vunit(vsub(p, v3(0.0f, 0.0f, -1.0f)));
                                color = vmul(vadd(n, v3(1.0f, 1.0f, 1.0f)), 0.5f);
                            } else {                          // illumination.zig:65-69
                                const float4 m = mat_of((uint32_t)hit);
                                color = (kind_of((uint32_t)hit) == 2u) ? v3(1.0f, 1.0f, 1.0f) : v3(m.x, m.y, m.z);
                            }
                        } else if (depth != P.rebounds) {     // depth == rebounds: illumination.zig:80-81, black
                            const float4 g = geom_of((uint32_t)hit);
                            const float4 m = mat_of((uint32_t)hit);
                            const uint32_t kind = kind_of((uint32_t)hit);
                            V3 p, n;
                            sphere_hit_point_inv(o, d, closest, g.x, g.y, g.z, inv_radius_of((uint32_t)hit), p, n);
                            const V3 nd = scatter_pre<DIRECT>(kind, m.w, m.x, m.y, a, d, p, n, rng);
                            const uint32_t drawn = rng.drawn();
                            PT_STAMP(tl2);
                            PP_ACC(2 + (depth ? 2 : 0), tl1, tl2);
                            if (depth == 0u) {
                                hit0 = (uint32_t)hit;
                                ra = make_float4(p.x, p.y, p.z, __uint_as_float((kTagState1 << kTagShift) | hit0));
                                rb = make_float4(nd.x, nd.y, nd.z, __uint_as_float(drawn));
                                // one more bounce here if this ray is one list B was built for: it leaves the pixel's main sphere
                                // within the patch and runs inside the cone
                                const V3 dp = vsub(p, v3(__uint_as_float(hd0.y), __uint_as_float(hd0.z), __uint_as_float(hd0.w)));
                                const V3 axis = v3(__uint_as_float(hd1.y), __uint_as_float(hd1.z), __uint_as_float(hd1.w));
                                if (hit0 == hd0.x && vlen2(dp) <= __uint_as_float(hd1.x) && vdot(nd, axis) >= __uint_as_float(hd_cos)) {
                                    o = p; d = nd;
                                    continue;
                                }
                            } else if (drawn < (1u << kDrawBits) && (uint32_t)hit < (1u << (32u - kDrawBits))) {
                                ra = make_float4(p.x, p.y, p.z, __uint_as_float((kTagState2 << kTagShift) | hit0));
                                rb = make_float4(nd.x, nd.y, nd.z, __uint_as_float(drawn | ((uint32_t)hit << kDrawBits)));
                            } else rays2--; // the record stays the one after the first bounce; the megakernel repeats (and counts) this step
                            break;
                        }
                    } else if (!lerp_sky_len2(d, a, P.blend_start, P.blend_stop, color)) { // illumination.zig:88
                        tag = kTagSkip;
                        not_unit++;
                    } else if (depth) {
                        // v.elementwiseMul(s.attenuation) of the first hit (illumination.zig:84-85)
                        const float4 m = mat_of(hit0);
                        color = (kind_of(hit0) != 2u) ? vemul(color, v3(m.x, m.y, m.z)) : vemul(color, v3(1.0f, 1.0f, 1.0f));
                    }
                    ra = make_float4(color.x, color.y, color.z, __uint_as_float(tag << kTagShift));
                    break;
                }
            }
            PT_STAMP(tw0);
            P.rec[2u * (size_t)r] = ra;
            if (__float_as_uint(ra.w) >> 31) P.rec[2u * (size_t)r + 1u] = rb;
            look = (__float_as_uint(ra.w) >> kTagShift) != kTagFinal;
            PT_STAMP(tw1);
            PP_ACC(5, tw0, tw1);
        }
        if (P.use_mask) { // 64 % chunk == 0: an item's samples sit in one wave, lanes [lane - jj, lane - jj + chunk)
            const unsigned long long m = __ballot(look);
            if (jj == 0u && r < nrec) P.item_mask[item] = (uint32_t)(m >> (threadIdx.x & 63u)) & (uint32_t)((1ull << P.chunk) - 1ull);
        }
    }
#ifdef PT_PHASE_TIMING
    if ((threadIdx.x & 63u) == 0u) {
        t_pp[7] = __builtin_amdgcn_s_memtime() - tpp_begin;
        for (int i = 0; i < 8; ++i) atomicAdd(&P.counters[CNT_PP0 + i], t_pp[i]);
    }
#endif
    const unsigned long long w_rays = wave_sum_u64(rays), w_rays2 = wave_sum_u64(rays2), w_cands = wave_sum_u64(cands), w_nu = wave_sum_u64(not_unit);
    if (CHECK) {
        unsigned long long w_chk[5];
        for (int i = 0; i < 5; ++i) w_chk[i] = wave_sum_u64(chk[i]);
        if ((threadIdx.x & 63u) == 0u) {
            for (int i = 0; i < 5; ++i) if (w_chk[i]) atomicAdd(&P.counters[CNT_CHECK_ACC_A + i], w_chk[i]);
            if (w_rays) atomicAdd(&P.counters[CNT_CHECK_RAYS_A], w_rays);
            if (w_rays2) atomicAdd(&P.counters[CNT_CHECK_RAYS_B], w_rays2);
        }
    }
    if ((threadIdx.x & 63u) == 0u) {
        if (w_rays) { atomicAdd(&P.counters[CNT_RAYS], w_rays + w_rays2); atomicAdd(&P.counters[CNT_PRIMARY_RAYS], w_rays); }
        if (w_rays2) atomicAdd(&P.counters[CNT_LIST2_RAYS], w_rays2);
        if (w_cands) { atomicAdd(&P.counters[CNT_CANDS], w_cands); atomicAdd(&P.counters[CNT_PRIMARY_CANDS], w_cands); }
        if (w_nu) atomicAdd(&P.counters[CNT_NOT_UNIT], w_nu);
    }
}

// Sum the chunk partials of a pixel in chunk order: total = p0; total += p1; ...
__global__ void pt_sum_chunks(const float *part, uint32_t slot0, uint32_t npix, uint32_t nchunks, uint32_t idx_start, uint32_t idx_stop,
                              uint32_t tile_pixels, uint32_t tile_count, uint32_t tile_index, float *rgb_sum) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; // over npix*3
    if (i >= npix * 3u) return;
    const uint32_t po = i / 3u, c = i - po * 3u;
    const uint32_t pixel = slot_to_pixel(slot0 + po, idx_start, tile_pixels, tile_count, tile_index);
    if (pixel >= idx_stop) return;
    float total = part[i];
    for (uint32_t k = 1; k < nchunks; ++k) total = total + part[(size_t)k * npix * 3u + i];
    rgb_sum[(size_t)pixel * 3u + c] = total;
}

// renderers/ppm_image.zig:70 — the two reciprocal multiplies
__global__ void pt_mean(const float *rgb_sum, uint32_t idx_start, uint32_t npix, float k1, float k2, float *rgb_mean) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 3u) return;
    const size_t at = (size_t)idx_start * 3u + i;
    rgb_mean[at] = k2 * (k1 * rgb_sum[at]);
}

// "%3d" of utils.zig:62-84: left-padded to three characters
__device__ __forceinline__ void put3(uint8_t *dst, uint32_t v) {
    const uint32_t h = v / 100u, t = (v / 10u) % 10u, u = v % 10u;
    dst[0] = (v >= 100u) ? (uint8_t)('0' + h) : (uint8_t)' ';
    dst[1] = (v >= 10u) ? (uint8_t)('0' + t) : (uint8_t)' ';
    dst[2] = (uint8_t)('0' + u);
}

// renderers/ppm_image.zig:70-82 + render_targets/ppm_image.zig:93-103
__global__ void pt_resolve(const float *rgb_sum, uint32_t idx_start, uint32_t npix, float k1, float k2, float max_val_f,
                           uint8_t *records, uint8_t *rgb8, unsigned long long *counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t at = ((size_t)idx_start + i) * 3u;
    uint32_t q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float mean = k2 * (k1 * rgb_sum[at + c]);
        const float col = max_val_f * mean;
        // @floatToInt(u8, col): truncation; out of range is undefined in the
        // reference's ReleaseFast build — clamped here and counted.
        uint32_t v;
        if (!(col > 0.0f)) v = 0u;
        else if (col > max_val_f) { v = 255u; atomicAdd(&counters[CNT_CLAMPED], 1ull); } // renderers/ppm_image.zig:76 asserts col <= 255
        else v = (uint32_t)col;
        q[c] = v;
    }
    if (rgb8) { rgb8[(size_t)i * 3u] = (uint8_t)q[0]; rgb8[(size_t)i * 3u + 1] = (uint8_t)q[1]; rgb8[(size_t)i * 3u + 2] = (uint8_t)q[2]; }
    if (records) {
        uint8_t rec[12];
        put3(rec, q[0]); rec[3] = ' ';
        put3(rec + 4, q[1]); rec[7] = ' ';
        put3(rec + 8, q[2]); rec[11] = '\n';
        uint32_t *dst = reinterpret_cast<uint32_t *>(records + (size_t)i * 12u);
        dst[0] = (uint32_t)rec[0] | ((uint32_t)rec[1] << 8) | ((uint32_t)rec[2] << 16) | ((uint32_t)rec[3] << 24);
        dst[1] = (uint32_t)rec[4] | ((uint32_t)rec[5] << 8) | ((uint32_t)rec[6] << 16) | ((uint32_t)rec[7] << 24);
        dst[2] = (uint32_t)rec[8] | ((uint32_t)rec[9] << 8) | ((uint32_t)rec[10] << 16) | ((uint32_t)rec[11] << 24);
    }
}

// ---------------------------------------------------------------------------
// per-function evaluation kernel (parity tests pin each device function)
// ---------------------------------------------------------------------------
__global__ void pt_debug_eval(int fn, const float *in, float *out, uint32_t n, Cam cam, const float4 *geom,
                              const float4 *mat, const uint32_t *kind, uint32_t nspheres, float t_min, float t_max) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
    case PT_FN_SPHERE_HIT: {
        const float *r = in + (size_t)i * 12; float *w = out + (size_t)i * 8;
        const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
        float t = 0.f; V3 p = v3(0, 0, 0), nn = v3(0, 0, 0);
        const bool h = sphere_near_root(o, d, vlen2(d), r[6], r[7], r[8], r[9], r[10], r[11], t);
        if (h) sphere_hit_point(o, d, t, r[6], r[7], r[8], r[9], p, nn);
        w[0] = h ? 1.f : 0.f; w[1] = t; w[2] = p.x; w[3] = p.y; w[4] = p.z; w[5] = nn.x; w[6] = nn.y; w[7] = nn.z;
        break;
    }
    case PT_FN_SCENE_HIT: { // plain exact scan (the reference's loop), no filter
        const float *r = in + (size_t)i * 6; float *w = out + (size_t)i * 8;
        const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
        const float a = vlen2(d);
        float closest = t_max; int hit = -1;
        for (uint32_t j = 0; j < nspheres; ++j) {
            const float4 g = geom[j]; float t;
            if (sphere_near_root(o, d, a, g.x, g.y, g.z, g.w, t_min, t_max, t) && t < closest) { closest = t; hit = (int)j; }
        }
        V3 p = v3(0, 0, 0), nn = v3(0, 0, 0);
        if (hit >= 0) { const float4 g = geom[hit]; sphere_hit_point(o, d, closest, g.x, g.y, g.z, g.w, p, nn); }
        w[0] = (float)hit; w[1] = (hit >= 0) ? closest : 0.f; w[2] = p.x; w[3] = p.y; w[4] = p.z; w[5] = nn.x; w[6] = nn.y; w[7] = nn.z;
        break;
    }
    case PT_FN_REFRACT: {
        const float *r = in + (size_t)i * 7; float *w = out + (size_t)i * 4;
        V3 res = v3(0, 0, 0);
        const bool ok = refract(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), r[6], res);
        w[0] = ok ? 1.f : 0.f; w[1] = res.x; w[2] = res.y; w[3] = res.z;
        break;
    }
    case PT_FN_SCHLICK: out[i] = schlick(in[(size_t)i * 2], in[(size_t)i * 2 + 1]); break;
    case PT_FN_REFLECT: {
        const float *r = in + (size_t)i * 6;
        const V3 res = reflect(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]));
        out[(size_t)i * 3] = res.x; out[(size_t)i * 3 + 1] = res.y; out[(size_t)i * 3 + 2] = res.z;
        break;
    }
    case PT_FN_LERP: {
        const float *r = in + (size_t)i * 9; float *w = out + (size_t)i * 4;
        V3 res = v3(0, 0, 0);
        const bool ok = lerp_sky(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), v3(r[6], r[7], r[8]), res);
        w[0] = ok ? 0.f : 1.f; w[1] = res.x; w[2] = res.y; w[3] = res.z;
        break;
    }
    case PT_FN_PHILOX: {
        const uint32_t *r = reinterpret_cast<const uint32_t *>(in) + (size_t)i * 5;
        uint32_t a, b, c, dd;
        philox4x32_10(r[2], r[3], r[4], 0u, r[0], r[1], a, b, c, dd);
        out[(size_t)i * 4] = u32_to_float(a); out[(size_t)i * 4 + 1] = u32_to_float(b);
        out[(size_t)i * 4 + 2] = u32_to_float(c); out[(size_t)i * 4 + 3] = u32_to_float(dd);
        break;
    }
    default: break;
    }
}

// PT_FN_FILTER_BIT: the conservative filter's verdict for one (ray, sphere) pair — exactly the scan's arithmetic
// (filter_ray_setup / filter_value, threshold from the host's filter_threshold) — beside the reference's
// discriminant (sphere.zig:64-74).  The invariant the exact pass relies on: discriminant >= 0  =>  bit set.
__global__ void pt_debug_filter_bit(const float *in, const float *thr, float *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = in + (size_t)i * 10;
    const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
    const float a = vlen2(d);
    const FilterRay fr = filter_ray_setup(o, d, a);
    const float F = filter_value(fr, r[6], r[7], r[8]);
    const V3 oc = vsub(o, v3(r[6], r[7], r[8]));
    const float half_b = vdot(oc, d);
    const float c = vlen2(oc) - r[9] * r[9];
    const float Fy = filter_value_y(fr, r[7] * fr.hy, fmaf(r[7], fr.wy, fr.E), r[6], r[8]); // as in a word that shares cy
    out[(size_t)i * 3] = (thr[i] <= F) ? 1.0f : 0.0f; // v_cmp_le_f32 T, F
    out[(size_t)i * 3 + 1] = half_b * half_b - a * c;
    out[(size_t)i * 3 + 2] = (thr[i] <= Fy) ? 1.0f : 0.0f;
}

// ptgpu_debug_exhaustive: the lean unary sequences of pt_device.h (sqrt_rn, rcp_rn, inv_sqrt_rn, not_unit_len2) against the
// compiler's correctly rounded expressions on ALL 2^32 bit patterns; counts the patterns that differ (NaN == NaN).
__global__ void pt_debug_exhaustive(int which, unsigned long long *bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; // the low 24 bits
    uint32_t n = 0;
    for (uint32_t hi = 0; hi < 256u; ++hi) {
        const uint32_t bits = (hi << 24) | tid;
        const float x = __uint_as_float(bits);
        float lean = 0.f, ref = 0.f;
        if (which == 0) { lean = sqrt_rn(x); ref = sqrtf(x); }
        else if (which == 1) { lean = rcp_rn(x); ref = 1.0f / x; }
        else if (which == 2) { lean = inv_sqrt_rn(x); ref = 1.0f / sqrtf(x); }
        else if (!(bits >> 31)) { lean = not_unit_len2(x) ? 1.f : 0.f; ref = (fabsf(sqrtf(x) - 1.0f) > 0.00001f) ? 1.f : 0.f; } // a = |d|^2: never negative
        if (!(__float_as_uint(lean) == __float_as_uint(ref) || (lean != lean && ref != ref))) n++;
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}

// cast_ray and scatter consume random draws: the debug variants take the draws from
// the input record (ArrayRng) so that the oracle can be fed the same numbers.
__global__ void pt_debug_eval_rng(int fn, const float *in, float *out, uint32_t n, Cam cam) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn == PT_FN_CAST_RAY) {
        // in: s, t, xi[16] (18)   out: o, d, n_used (7)
        const float *r = in + (size_t)i * 18; float *w = out + (size_t)i * 7;
        ArrayRng ar{r + 2, 0u, 16u};
        V3 origin, dir;
        cast_ray(cam, r[0], r[1], ar, origin, dir);
        w[0] = origin.x; w[1] = origin.y; w[2] = origin.z; w[3] = dir.x; w[4] = dir.y; w[5] = dir.z; w[6] = (float)ar.n;
    } else if (fn == PT_FN_SCATTER) {
        // in: d(3), p(3), n(3), kind, param, xi[16] (27)   out: scattered d (3), n_used (4)
        const float *r = in + (size_t)i * 27; float *w = out + (size_t)i * 4;
        ArrayRng ar{r + 11, 0u, 16u};
        const V3 nd = scatter((uint32_t)r[9], r[10], v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), v3(r[6], r[7], r[8]), ar);
        w[0] = nd.x; w[1] = nd.y; w[2] = nd.z; w[3] = (float)ar.n;
    } else if (fn == PT_FN_SAMPLE_DISK || fn == PT_FN_SAMPLE_BALL) {
        // in: sampler, xi[16] (17)   out: p (3), n_used (4)
        const float *r = in + (size_t)i * 17; float *w = out + (size_t)i * 4;
        ArrayRng ar{r + 1, 0u, 16u};
        const bool direct = r[0] != 0.0f;
        V3 p;
        if (fn == PT_FN_SAMPLE_DISK) p = direct ? random_vector_in_unit_disk<true>(ar) : random_vector_in_unit_disk<false>(ar);
        else p = direct ? random_vector_in_unit_sphere<true>(ar) : random_vector_in_unit_sphere<false>(ar);
        w[0] = p.x; w[1] = p.y; w[2] = p.z; w[3] = (float)ar.n;
    }
}

// PT_FN_SCENE_HIT_FILTERED: the megakernel's own Scene.is_hit (LDS filter + exact pass) on caller
// rays, one ray per thread, same output record as PT_FN_SCENE_HIT (the plain exact scan).
template <bool MULTI_TILE>
__global__ __launch_bounds__(kBlock) void pt_debug_scene_hit_filtered(const float *in, float *out, uint32_t n, const float4 *g_filt, const float *g_ycy,
                                                                      const float4 *g_geom, uint32_t nspheres, float t_min, float t_max) {
    __shared__ float4 s_filt[kTile + kSub];
    __shared__ uint32_t s_mask[mask_tile(MULTI_TILE) / 32 * kBlock];
    __shared__ float s_ycy[kTile / 32];
    const uint32_t tid = threadIdx.x;
    const uint32_t i = blockIdx.x * kBlock + tid;
    const uint32_t n_pad = (nspheres + kPad - 1) / kPad * kPad;
    if (!MULTI_TILE) {
        for (uint32_t k = tid; k < n_pad; k += kBlock) s_filt[k] = g_filt[k];
        if (tid < (uint32_t)(kTile / 32)) s_ycy[tid] = g_ycy[tid];
        __syncthreads();
    }
    const bool live = i < n;
    const float *r = in + (size_t)(live ? i : 0) * 6;
    const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
    const float a = vlen2(d);
    float closest = t_max;
    int hit = -1;
    unsigned long long t_flush = 0;
    uint32_t n_cands = 0, n_flush_iters = 0;
    auto geom_of = [&](uint32_t j) -> float4 { return g_geom[j]; };
    scene_is_hit_filtered<MULTI_TILE>(g_filt, g_ycy, true, s_ycy, n_pad, s_filt, s_mask, tid, geom_of, o, d, a, t_min, t_max, closest, hit, n_cands,
                                      n_flush_iters, t_flush);
    if (!live) return;
    float *w = out + (size_t)i * 8;
    V3 p = v3(0, 0, 0), nn = v3(0, 0, 0);
    if (hit >= 0) { const float4 g = g_geom[hit]; sphere_hit_point(o, d, closest, g.x, g.y, g.z, g.w, p, nn); }
    w[0] = (float)hit; w[1] = (hit >= 0) ? closest : 0.f; w[2] = p.x; w[3] = p.y; w[4] = p.z; w[5] = nn.x; w[6] = nn.y; w[7] = nn.z;
}

// PT_FN_SCENE_HIT_TREE: the same through the sphere tree (same record).
template <bool LDS>
__global__ __launch_bounds__(kBlock) void pt_debug_scene_hit_tree(const float *in, float *out, uint32_t n, const TreeNode *nodes, uint32_t n_nodes, const float4 *tgeom,
                                                                  const uint32_t *tprim, uint32_t nspheres, uint32_t root, const float4 *g_geom, float t_min, float t_max, uint32_t flush_at) {
    __shared__ uint32_t s_stack[LDS ? 1 : kTreeStack * kBlock];
    __shared__ float4 s_tboxes[LDS ? 3 * kTreeLdsNodes : 1];
    __shared__ uint32_t s_tchild[LDS ? kTreeLdsNodes : 1];
    __shared__ float4 s_tgeom[LDS ? kTile : 1];
    __shared__ uint16_t s_tprim[LDS ? kTile : 2];
    __shared__ uint16_t s_tstack[LDS ? kTreeLdsStack * kBlock : 2];
    const uint32_t tid = threadIdx.x;
    if (LDS) tree_stage_lds(nodes, n_nodes, tgeom, tprim, nspheres, s_tboxes, s_tchild, s_tgeom, s_tprim, tid);
    const uint32_t i = blockIdx.x * kBlock + tid;
    const bool live = i < n;
    const float *r = in + (size_t)(live ? i : 0) * 6;
    const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
    const float a = vlen2(d);
    float closest = t_max;
    int hit = -1;
    unsigned long long n_cands = 0, n_boxes = 0, n_iters = 0;
    uint32_t sp = 0;
    if (LDS) {
        uint32_t cur = live ? TreeInLds::ref16(root) : TreeInLds::kNone, pend = TreeInLds::kNone;
        scene_is_hit_tree(TreeInLds{s_tboxes, s_tchild, s_tgeom, s_tprim, s_tstack}, cur, sp, pend, 0u, flush_at, tid, o, d, a, t_min, t_max, closest, hit, n_cands, n_iters, n_boxes);
    } else {
        uint32_t cur = live ? root : kTreeNone, pend = kTreeNone;
        scene_is_hit_tree(TreeInGlobal{nodes, tgeom, tprim, s_stack}, cur, sp, pend, 0u, flush_at, tid, o, d, a, t_min, t_max, closest, hit, n_cands, n_iters, n_boxes);
    }
    if (!live) return;
    float *w = out + (size_t)i * 8;
    V3 p = v3(0, 0, 0), nn = v3(0, 0, 0);
    if (hit >= 0) { const float4 g = g_geom[hit]; sphere_hit_point(o, d, closest, g.x, g.y, g.z, g.w, p, nn); }
    w[0] = (float)hit; w[1] = (hit >= 0) ? closest : 0.f; w[2] = p.x; w[3] = p.y; w[4] = p.z; w[5] = nn.x; w[6] = nn.y; w[7] = nn.z;
}

// PT_FN_TREE_BOX_SAFETY: the property the tree walk rests on, checked pair by pair instead of through closest hits.  The ray visits
// EVERY node (no pruning); for every sphere the exact test accepts at t, the box of every ancestor must pass tree_box — the
// function the walk uses — with the interval clipped to (t_min, t): the tightest interval under which the walk could still need
// that sphere (it skips a box against the closest hit so far, which is >= t until the sphere is tested).  out: spheres accepted,
// (sphere, ancestor box) pairs checked, pairs whose box test FAILED (must be 0), 0.
__global__ __launch_bounds__(kBlock) void pt_debug_tree_box_safety(const float *in, float *out, uint32_t n, const TreeNode *nodes, const float4 *tgeom,
                                                                   uint32_t root, float t_min, float t_max) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *r = in + (size_t)i * 6;
    const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
    const float a = vlen2(d);
    const V3 inv = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); // as the walk takes it
    uint32_t path_node[kTreeStack + 1], path_next[kTreeStack + 1]; // the inner nodes from the root down and which child comes next
    uint32_t sp = 0;
    float accepted = 0.f, pairs = 0.f, failed = 0.f;
    auto child_box = [&](uint32_t node, uint32_t c, float t_hi) -> bool {
        const TreeNode nd = nodes[node];
        float t_near;
        return c == 0u ? tree_box(o, inv, nd.a.x, nd.a.y, nd.a.z, nd.a.w, nd.b.x, nd.b.y, t_min, t_hi, t_near)
                       : tree_box(o, inv, nd.b.z, nd.b.w, nd.c.x, nd.c.y, nd.c.z, nd.c.w, t_min, t_hi, t_near);
    };
    auto leaf = [&](uint32_t ref) {
        const uint32_t first = TreeInGlobal::first(ref), cnt = TreeInGlobal::count(ref);
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4 g = tgeom[first + k];
            float t;
            if (!sphere_near_root(o, d, a, g.x, g.y, g.z, g.w, t_min, t_max, t)) continue;
            accepted += 1.f;
            for (uint32_t f = 0; f < sp; ++f) { // path_next[f] - 1 is the child of path_node[f] this leaf hangs under
                pairs += 1.f;
                if (!child_box(path_node[f], path_next[f] - 1u, t)) failed += 1.f;
            }
        }
    };
    if (root & kTreeLeafBit) leaf(root);
    else { path_node[0] = root; path_next[0] = 0u; sp = 1; }
    while (sp > 0u) {
        if (path_next[sp - 1u] == 2u) { --sp; continue; }
        const uint32_t c = path_next[sp - 1u]++;
        const uint4 ch = nodes[path_node[sp - 1u]].ch;
        const uint32_t ref = c == 0u ? ch.x : ch.y;
        if (ref & kTreeLeafBit) leaf(ref);
        else if (sp <= kTreeStack) { path_node[sp] = ref; path_next[sp] = 0u; ++sp; }
    }
    float *w = out + (size_t)i * 4;
    w[0] = accepted; w[1] = pairs; w[2] = failed; w[3] = 0.f;
}

} // namespace

// ---------------------------------------------------------------------------
// host side: context, uploads, launches
// ---------------------------------------------------------------------------
// Scratch of the no-allocation entry point (ptgpu_render_sum_device_noalloc): counters, hit-id stack and
// chunk partials live in this file-scope device array instead of hipMalloc'ed buffers.  Every GPU the
// library runs on holds its own copy at its own address (resolved once per context with
// hipGetSymbolAddress, with that device current); host code only ever uses that resolved pointer.
#define PT_SCRATCH_BYTES (128ull << 20)
static __device__ __align__(256) unsigned char scratch[PT_SCRATCH_BYTES];
constexpr uint32_t kMaxTimedSegments = 64; // renders of more segments report primary_ms = lists_ms = megakernel_ms = 0
constexpr uint32_t kSegEvents = 4;         // per timed segment: before the lists, after them, after pt_primary, after the megakernel
constexpr uint32_t kPrecreatedSegments = 4; // event sets made by ptgpu_create (the no-allocation entry point creates none: further segments go untimed)
constexpr size_t kScratchCounters = 256; // bytes reserved for the CNT_N counters at the start of the scratch

constexpr uint64_t kPrimaryAuto = ~0ull;          // PT_LIMIT_PRIMARY_BYTES default: half of the device memory that is free at the render, at most kPrimaryAutoMax
constexpr uint64_t kPrimaryAutoMax = 96ull << 30;

struct ptgpu_ctx {
    int device = -1;
    int compute_units = 0;
    int clock_khz = 0;
    char arch[64] = {0};
    int last_hip = 0;
    uint32_t nspheres = 0, n_pad = 0;
    float4 *d_filt = nullptr, *d_geom = nullptr, *d_mat = nullptr;
    uint32_t *d_kind = nullptr;
    float *d_inv_radius = nullptr;
    float *d_ycy = nullptr;
    uint32_t shared_tests = 0; // spheres in shared-ordinate mask words (5-FMA filter test)
    // sphere tree of the uploaded scene (tree_nodes = 0: none, the scan serves it)
    TreeNode *d_tnodes = nullptr; float4 *d_tgeom = nullptr; uint32_t *d_tprim = nullptr;
    uint32_t tree_root = 0, tree_nodes = 0, tree_depth = 0;
    uint32_t tree_min_spheres = 256u;                 // PT_LIMIT_TREE_MIN_SPHERES (measured: the walk in LDS overtakes the scan between 150 and 250 spheres)
    int resident_tree[2][2] = {{0, 0}, {0, 0}};      // [det][multi]: workgroups per CU of the tree variants (multi = 0: tree staged in LDS)
    bool have_camera = false;
    bool debug_list_check = false; // ptgpu_debug_list_safety: render_sum_impl runs the lists and pt_primary<.., CHECK> only
    Cam cam;
    float *d_part = nullptr; size_t part_bytes = 0;
    uint32_t *d_stack = nullptr; size_t stack_bytes = 0;
    float *d_tmp = nullptr; size_t tmp_bytes = 0;    // scratch image for host-pointer variants
    uint8_t *d_rec = nullptr; size_t rec_bytes = 0;  // scratch records for ptgpu_render_ppm
    unsigned long long *d_counters = nullptr;
    unsigned long long *h_counters = nullptr;        // pinned: counters come back with an async copy, no staging
    unsigned char *d_scratch = nullptr;              // this device's copy of `scratch` (hipGetSymbolAddress)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    int resident[2][2] = {{0, 0}, {0, 0}};           // [det][multi]: workgroups per CU the occupancy query reports
    int blocks_per_cu = 0; // 0: what the occupancy query reports; PTGPU_BLOCKS_PER_CU overrides (experiments)
    uint64_t partial_cap = 2ull << 30;               // PT_LIMIT_PARTIAL_BYTES
    uint64_t primary_cap = kPrimaryAuto;             // PT_LIMIT_PRIMARY_BYTES (0: no primary pass; kPrimaryAuto: sized per render from hipMemGetInfo)
    uint32_t primary_bounces = 2;                    // PT_LIMIT_PRIMARY_BOUNCES: 1 = camera rays only, 2 = also the rays scattered off each pixel's main sphere
    unsigned char *d_prim = nullptr; size_t prim_bytes = 0; // primary-pass records + candidate lists
    void *d_lsph = nullptr; size_t lsph_bytes = 0;          // ListSphere per sphere (pt_list_prep)
    std::vector<hipEvent_t> evp;                     // kSegEvents per timed segment (kPrecreatedSegments sets from ptgpu_create, grown on demand)
    pt_stats stats;
};

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) {                              \
            (ctx)->last_hip = (int)e_;                       \
            return (e_ == hipErrorOutOfMemory) ? PT_ERR_OOM : PT_ERR_HIP; \
        }                                                    \
    } while (0)

static int ensure(ptgpu_ctx *ctx, void **ptr, size_t *have, size_t want) {
    if (*have >= want && *ptr) return PT_OK;
    if (*ptr) { HIP_TRY(ctx, hipFree(*ptr)); *ptr = nullptr; *have = 0; }
    HIP_TRY(ctx, hipMalloc(ptr, want));
    *have = want;
    return PT_OK;
}

extern "C" uint32_t ptgpu_abi_version(void) { return PT_ABI_VERSION; }

extern "C" const char *ptgpu_strerror(int status) {
    switch (status) {
    case PT_OK: return "ok";
    case PT_ERR_BAD_ARG: return "bad argument";
    case PT_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
    case PT_ERR_HIP: return "HIP runtime error";
    case PT_ERR_OOM: return "out of device memory";
    case PT_ERR_DIRECTION_NOT_UNIT: return "DirectionIsNotUnitVector";
    case PT_ERR_VALUE_EXCEEDS_MAX: return "ValueExceedsMaxValue";
    case PT_ERR_SCENE_NOT_AVAILABLE: return "SceneNotAvailable";
    case PT_ERR_NO_SPACE: return "NoSpaceLeft";
    case PT_ERR_UNSUPPORTED: return "unsupported";
    case PT_ERR_SCRATCH_LIMIT: return "render does not fit the static scratch of the no-allocation entry point";
    case PT_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
    }
}

extern "C" int ptgpu_create(int device, ptgpu_ctx **out) {
    if (!out) return PT_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return PT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PT_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PT_ERR_NO_DEVICE; // built for gfx950 only
    ptgpu_ctx *ctx = new (std::nothrow) ptgpu_ctx();
    if (!ctx) return PT_ERR_OOM;
    ctx->device = device;
    ctx->compute_units = prop.multiProcessorCount;
    ctx->clock_khz = prop.clockRate;
    strncpy(ctx->arch, prop.gcnArchName, sizeof(ctx->arch) - 1);
    memset(&ctx->stats, 0, sizeof ctx->stats);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_counters, CNT_N * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_counters, CNT_N * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev2);
    for (uint32_t i = 0; i < kSegEvents * kPrecreatedSegments && e == hipSuccess; ++i) {
        hipEvent_t ev = nullptr;
        e = hipEventCreate(&ev);
        if (e == hipSuccess) ctx->evp.push_back(ev);
    }
    // this device's address of the static scratch array (device `device` is current here)
    if (e == hipSuccess) e = hipGetSymbolAddress((void **)&ctx->d_scratch, HIP_SYMBOL(scratch));
    if (e == hipSuccess) {
        // workgroups per CU each kernel variant can keep resident (the persistent grid's size)
        int r = 0;
        // (the variants that start paths from primary-pass records have the same launch bounds and LDS)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeRejection, false, false>, kBlock, 0) == hipSuccess && r > 0) ctx->resident[0][0] = r;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeRejection, true, false>, kBlock, 0) == hipSuccess && r > 0) ctx->resident[0][1] = r;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeDet, false, false>, kBlock, 0) == hipSuccess && r > 0) ctx->resident[1][0] = r;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeDet, true, false>, kBlock, 0) == hipSuccess && r > 0) ctx->resident[1][1] = r;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) if (ctx->resident[i][j] <= 0) ctx->resident[i][j] = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeRejection, true, false, true>, kBlock, 0) == hipSuccess && r > 0) ctx->resident_tree[0][1] = r;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeDet, true, false, true>, kBlock, 0) == hipSuccess && r > 0) ctx->resident_tree[1][1] = r;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeRejection, false, false, true>, kBlock, 0) == hipSuccess && r > 0) ctx->resident_tree[0][0] = r;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, pt_megakernel<kModeDet, false, false, true>, kBlock, 0) == hipSuccess && r > 0) ctx->resident_tree[1][0] = r;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) if (ctx->resident_tree[i][j] <= 0) ctx->resident_tree[i][j] = 1;
    }
    if (e != hipSuccess || !ctx->d_scratch) {
        if (ctx->d_counters) (void)hipFree(ctx->d_counters);
        if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
        if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
        if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
        if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
        for (hipEvent_t ev : ctx->evp) (void)hipEventDestroy(ev);
        delete ctx;
        return PT_ERR_HIP;
    }
    const char *bpc = getenv("PTGPU_BLOCKS_PER_CU");
    if (bpc && atoi(bpc) > 0 && atoi(bpc) <= 8) ctx->blocks_per_cu = atoi(bpc);
    *out = ctx;
    return PT_OK;
}

extern "C" int ptgpu_destroy(ptgpu_ctx *ctx) {
    if (!ctx) return PT_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    void *ptrs[] = {ctx->d_tnodes, ctx->d_tgeom, ctx->d_tprim, ctx->d_filt, ctx->d_geom, ctx->d_mat, ctx->d_kind, ctx->d_inv_radius, ctx->d_ycy, ctx->d_part, ctx->d_stack, ctx->d_tmp, ctx->d_rec, ctx->d_counters, ctx->d_prim, ctx->d_lsph};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    for (hipEvent_t e : ctx->evp) (void)hipEventDestroy(e);
    delete ctx;
    return PT_OK;
}

extern "C" int ptgpu_device_info(const ptgpu_ctx *ctx, int *compute_units, int *clock_khz, char *arch, size_t arch_len) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (compute_units) *compute_units = ctx->compute_units;
    if (clock_khz) *clock_khz = ctx->clock_khz;
    if (arch && arch_len) { strncpy(arch, ctx->arch, arch_len - 1); arch[arch_len - 1] = 0; }
    return PT_OK;
}

extern "C" int ptgpu_last_hip_error(const ptgpu_ctx *ctx) { return ctx ? ctx->last_hip : 0; }

extern "C" int ptgpu_get_stats(const ptgpu_ctx *ctx, pt_stats *out) {
    if (!ctx || !out) return PT_ERR_BAD_ARG;
    *out = ctx->stats;
    return PT_OK;
}

extern "C" int ptgpu_scene_info(const ptgpu_ctx *ctx, uint32_t *nspheres, uint32_t *shared_ordinate_tests) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (nspheres) *nspheres = ctx->nspheres;
    if (shared_ordinate_tests) *shared_ordinate_tests = ctx->shared_tests;
    return PT_OK;
}

extern "C" int ptgpu_tree_info(const ptgpu_ctx *ctx, uint32_t *nodes, uint32_t *depth) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (nodes) *nodes = ctx->tree_nodes;
    if (depth) *depth = ctx->tree_depth;
    return PT_OK;
}

extern "C" int ptgpu_set_limit(ptgpu_ctx *ctx, int which, uint64_t value) {
    if (!ctx) return PT_ERR_BAD_ARG;
    switch (which) {
    case PT_LIMIT_PARTIAL_BYTES:
        if (value < 4096) return PT_ERR_BAD_ARG;
        ctx->partial_cap = value;
        return PT_OK;
    case PT_LIMIT_PRIMARY_BYTES:
        ctx->primary_cap = value; // (~0: back to the automatic size)
        if (ctx->d_prim && (uint64_t)ctx->prim_bytes > value) { // a lowered limit gives the memory back now, not at ptgpu_destroy
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            HIP_TRY(ctx, hipFree(ctx->d_prim));
            ctx->d_prim = nullptr; ctx->prim_bytes = 0;
        }
        return PT_OK;
    case PT_LIMIT_PRIMARY_BOUNCES:
        if (value < 1 || value > 2) return PT_ERR_BAD_ARG;
        ctx->primary_bounces = (uint32_t)value;
        return PT_OK;
    case PT_LIMIT_TREE_MIN_SPHERES:
        if (value > 0xffffffffull) return PT_ERR_BAD_ARG;
        ctx->tree_min_spheres = (uint32_t)value; // takes effect at the next ptgpu_set_scene
        return PT_OK;
    default: return PT_ERR_BAD_ARG;
    }
}

// Gives back every work buffer the context has grown (records and lists of the primary pass, chunk partials, hit-id stack,
// staging images); scene tables stay.  The next render allocates again what it needs.
extern "C" int ptgpu_trim(ptgpu_ctx *ctx) {
    if (!ctx) return PT_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void **ptrs[] = {(void **)&ctx->d_prim, (void **)&ctx->d_part, (void **)&ctx->d_stack, (void **)&ctx->d_tmp, (void **)&ctx->d_rec, &ctx->d_lsph};
    size_t *sizes[] = {&ctx->prim_bytes, &ctx->part_bytes, &ctx->stack_bytes, &ctx->tmp_bytes, &ctx->rec_bytes, &ctx->lsph_bytes};
    for (int i = 0; i < 6; ++i) {
        if (*ptrs[i]) HIP_TRY(ctx, hipFree(*ptrs[i]));
        *ptrs[i] = nullptr; *sizes[i] = 0;
    }
    return PT_OK;
}
extern "C" uint64_t ptgpu_work_bytes(const ptgpu_ctx *ctx) {
    return ctx ? (uint64_t)ctx->prim_bytes + ctx->part_bytes + ctx->stack_bytes + ctx->tmp_bytes + ctx->rec_bytes + ctx->lsph_bytes : 0u;
}

extern "C" uint64_t ptgpu_noalloc_scratch_bytes(void) { return PT_SCRATCH_BYTES; }

// The filter threshold of sphere j (see the kernel comment): with c the centre,
// cc = |c|^2, r2 = radius^2 and M = |o|^2 + cc, the kernel evaluates
//   F = (oc.dh)^2 - |oc|^2 + cc + Keps*|o|^2     (dh = d/|d|, oc = o - c)
// and queues the sphere iff F >= cc - r2 - Keps*cc, i.e. iff
//   (oc.dh)^2 - |oc|^2 + r2 + Keps*M >= 0.
// The reference's float discriminant can only be >= 0 when the real-arithmetic one is
// >= -~90*2^-24*M (rounding of both sides), Keps = 2^-16 = 256*2^-24.  Evaluated in
// double and rounded toward -inf.
static float filter_threshold_of(float cx_, float cy_, float cz_, float radius) {
    const double cx = cx_, cy = cy_, cz = cz_, r = radius;
    const double cc = cx * cx + cy * cy + cz * cz;
    const double thr = cc - r * r - (double)kKeps * cc;
    float f = (float)thr;
    if ((double)f > thr) f = nextafterf(f, -INFINITY);
    return f;
}
static float filter_threshold(const pt_sphere &s) { return filter_threshold_of(s.cx, s.cy, s.cz, s.radius); }

// The sphere tree (scene_is_hit_tree), built top-down.  A range of spheres is split where the surface-area
// heuristic is cheapest — area(left box) x count(left) + area(right box) x count(right), evaluated at 15 bin
// boundaries per axis over the centres (a ground sphere of radius 1000 among small ones ends up alone near the
// root instead of inflating every box on its side) — as long as the larger part can still be finished by median
// splits within the traversal stack; otherwise, and when the heuristic finds nothing to separate (coincident
// centres), at the median centre along the widest axis.  Leaves of <= kTreeLeaf spheres.  Boxes hold
// centre +- |radius| rounded outwards.  Returns false (no tree: the scan serves the scene) for non-finite spheres
// or a depth beyond the traversal stack.
struct TreeBuild {
    std::vector<TreeNode> nodes;
    std::vector<float4> tgeom;
    std::vector<uint32_t> tprim;
    uint32_t root = 0, depth = 0;
    bool finite = true; // every stored box has finite centre and half extents
    bool sah = true; // false: median splits only (at most n / 2 leaves, ceil(log2(n / kTreeLeaf)) + 1 levels)
};
struct TreeBox {
    float lo[3], hi[3];
    void clear() { for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
    void add(const TreeBox &o) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
    double area() const {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return (x < 0.0) ? 0.0 : x * y + y * z + z * x;
    }
    // the box as the walk stores it: centre bc and half extents bh with bc - bh <= lo and hi <= bc + bh in exact arithmetic (bh rounded up);
    // false when a coordinate is not finite (centre +- radius overflowed: such a scene keeps the scan)
    bool centre_form(float bc[3], float bh[3]) const {
        for (int k = 0; k < 3; ++k) {
            bc[k] = (float)(0.5 * ((double)lo[k] + (double)hi[k]));
            const double h = std::max((double)hi[k] - (double)bc[k], (double)bc[k] - (double)lo[k]); // exact in double
            float hf = (float)h;
            if ((double)hf < h) hf = nextafterf(hf, INFINITY);
            bh[k] = hf;
            if (!std::isfinite(bc[k]) || !std::isfinite(bh[k])) return false;
        }
        return true;
    }
};
static TreeBox tree_sphere_box(const pt_sphere &s) {
    TreeBox b;
    const float c[3] = {s.cx, s.cy, s.cz}, r = fabsf(s.radius);
    for (int k = 0; k < 3; ++k) { b.lo[k] = nextafterf(c[k] - r, -INFINITY); b.hi[k] = nextafterf(c[k] + r, INFINITY); }
    return b;
}
static uint32_t tree_levels_of_median_split(uint32_t count) { // levels (leaf level included) a balanced subtree of `count` spheres takes
    uint32_t levels = 1;
    while (count > kTreeLeaf) { count = (count + 1u) / 2u; ++levels; }
    return levels;
}
static uint32_t tree_build_range(const pt_sphere *sp, std::vector<uint32_t> &idx, uint32_t b, uint32_t e, uint32_t level, TreeBuild &T, TreeBox &box) {
    T.depth = std::max(T.depth, level);
    box.clear();
    if (e - b <= kTreeLeaf) {
        const uint32_t first = (uint32_t)T.tprim.size();
        for (uint32_t i = b; i < e; ++i) {
            const pt_sphere &s = sp[idx[i]];
            box.add(tree_sphere_box(s));
            T.tgeom.push_back(make_float4(s.cx, s.cy, s.cz, s.radius));
            T.tprim.push_back(idx[i]);
        }
        return kTreeLeafBit | ((e - b - 1u) << 28) | first;
    }
    auto centre = [&](uint32_t i, int k) -> float { const pt_sphere &s = sp[i]; return k == 0 ? s.cx : k == 1 ? s.cy : s.cz; };
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = b; i < e; ++i)
        for (int k = 0; k < 3; ++k) { clo[k] = std::min(clo[k], centre(idx[i], k)); chi[k] = std::max(chi[k], centre(idx[i], k)); }
    // binned surface-area heuristic
    constexpr int kBins = 16;
    double best_cost = INFINITY;
    int best_axis = -1, best_bin = 0;
    uint32_t best_left = 0;
    auto bin_of = [&](uint32_t i, int k) -> int {
        const double f = ((double)centre(i, k) - clo[k]) / ((double)chi[k] - clo[k]) * kBins;
        return std::min(kBins - 1, std::max(0, (int)f));
    };
    for (int k = 0; k < 3; ++k) {
        if (!(chi[k] > clo[k])) continue;
        TreeBox bb[kBins]; uint32_t cnt[kBins] = {0};
        for (int j = 0; j < kBins; ++j) bb[j].clear();
        for (uint32_t i = b; i < e; ++i) { const int j = bin_of(idx[i], k); bb[j].add(tree_sphere_box(sp[idx[i]])); cnt[j]++; }
        double right_area[kBins]; uint32_t right_cnt[kBins];
        TreeBox acc; acc.clear(); uint32_t n = 0;
        for (int j = kBins - 1; j > 0; --j) { acc.add(bb[j]); n += cnt[j]; right_area[j] = acc.area(); right_cnt[j] = n; }
        acc.clear(); n = 0;
        for (int j = 0; j + 1 < kBins; ++j) { // split between bins j and j + 1
            acc.add(bb[j]); n += cnt[j];
            if (n == 0u || right_cnt[j + 1] == 0u) continue;
            const double cost = acc.area() * n + right_area[j + 1] * right_cnt[j + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = j; best_left = n; }
        }
    }
    uint32_t mid = b + (e - b) / 2u;
    const uint32_t larger = std::max(best_left, (e - b) - best_left);
    if (T.sah && best_axis >= 0 && level + 1u + tree_levels_of_median_split(larger) <= kTreeStack) {
        mid = (uint32_t)(std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t i) { return bin_of(i, best_axis) <= best_bin; }) - idx.begin());
    } else {
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
        std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e, [&](uint32_t x, uint32_t y) {
            const float cx = centre(x, axis), cy = centre(y, axis);
            return cx < cy || (cx == cy && x < y);
        });
    }
    const uint32_t me = (uint32_t)T.nodes.size();
    T.nodes.emplace_back();
    TreeBox b0, b1;
    const uint32_t c0 = tree_build_range(sp, idx, b, mid, level + 1u, T, b0);
    const uint32_t c1 = tree_build_range(sp, idx, mid, e, level + 1u, T, b1);
    TreeNode &N = T.nodes[me];
    float bc0[3], bh0[3], bc1[3], bh1[3];
    if (!b0.centre_form(bc0, bh0) || !b1.centre_form(bc1, bh1)) T.finite = false;
    N.a = make_float4(bc0[0], bc0[1], bc0[2], bh0[0]);
    N.b = make_float4(bh0[1], bh0[2], bc1[0], bc1[1]);
    N.c = make_float4(bc1[2], bh1[0], bh1[1], bh1[2]);
    N.ch = make_uint4(c0, c1, 0u, 0u);
    box.add(b0); box.add(b1);
    return me;
}
static bool tree_build(const pt_sphere *sp, uint32_t n, TreeBuild &T) {
    if (n > 0x0fffffffu) return false;
    for (uint32_t i = 0; i < n; ++i)
        if (!std::isfinite(sp[i].cx) || !std::isfinite(sp[i].cy) || !std::isfinite(sp[i].cz) || !std::isfinite(sp[i].radius)) return false;
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    T.nodes.reserve(n / 2u + 1u); T.tgeom.reserve(n); T.tprim.reserve(n);
    TreeBox box;
    T.root = tree_build_range(sp, idx, 0u, n, 1u, T, box);
    if (T.nodes.empty()) T.nodes.emplace_back(); // a one-leaf scene: the root ref is the leaf, the node array is never read
    return T.depth <= kTreeStack && T.finite;
}

// Host-only check of the builder (no device needed; tests/test_host.py): builds the tree ptgpu_set_scene would build and
// walks it, counting what must never happen — a sphere (centre +- |radius|) sticking out of the box of any of its ancestors, a
// sphere missing from the leaves or listed twice, a leaf beyond kTreeLeaf spheres, a leaf geometry that is not its sphere's.
extern "C" int ptgpu_debug_tree_check(const pt_sphere *spheres, uint32_t n, int median_only, uint32_t *nodes, uint32_t *depth, uint32_t *leaves,
                                      uint64_t *violations) {
    if (!spheres || n == 0 || !violations) return PT_ERR_BAD_ARG;
    TreeBuild T;
    T.sah = !median_only;
    if (!tree_build(spheres, n, T)) return PT_ERR_UNSUPPORTED;
    uint64_t bad = 0;
    uint32_t n_leaves = 0;
    std::vector<uint8_t> seen(n, 0);
    // every stored box (bc -+ bh, taken exactly, in double) with a link to its parent's: stored boxes need not nest (each is rounded on
    // its own), so a sphere is checked against the box of EVERY ancestor — a walk skips a subtree on any one of them
    struct CheckBox { double lo[3], hi[3]; int parent; };
    std::vector<CheckBox> boxes;
    struct Item { uint32_t ref; int box; };
    std::vector<Item> todo;
    todo.push_back(Item{T.root, -1});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.ref & kTreeLeafBit) {
            const uint32_t first = it.ref & 0x0fffffffu, cnt = ((it.ref >> 28) & 7u) + 1u;
            n_leaves++;
            if (cnt > kTreeLeaf || (size_t)first + cnt > T.tprim.size()) { bad++; continue; }
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t id = T.tprim[first + k];
                if (id >= n || seen[id]) { bad++; continue; }
                seen[id] = 1;
                const pt_sphere &s = spheres[id];
                const float4 g = T.tgeom[first + k];
                if (memcmp(&g.x, &s.cx, 4) || memcmp(&g.y, &s.cy, 4) || memcmp(&g.z, &s.cz, 4) || memcmp(&g.w, &s.radius, 4)) bad++;
                const float c[3] = {s.cx, s.cy, s.cz}, r = fabsf(s.radius);
                for (int bx = it.box; bx >= 0; bx = boxes[bx].parent) {
                    bool out = false;
                    for (int a = 0; a < 3; ++a)
                        if (!(boxes[bx].lo[a] <= (double)c[a] - r && (double)c[a] + r <= boxes[bx].hi[a])) out = true;
                    if (out) { bad++; break; }
                }
            }
        } else {
            if (it.ref >= T.nodes.size()) { bad++; continue; }
            const TreeNode &N = T.nodes[it.ref];
            CheckBox b0, b1;
            const float bc0[3] = {N.a.x, N.a.y, N.a.z}, bh0[3] = {N.a.w, N.b.x, N.b.y}, bc1[3] = {N.b.z, N.b.w, N.c.x}, bh1[3] = {N.c.y, N.c.z, N.c.w};
            for (int a = 0; a < 3; ++a) {
                b0.lo[a] = (double)bc0[a] - bh0[a]; b0.hi[a] = (double)bc0[a] + bh0[a];
                b1.lo[a] = (double)bc1[a] - bh1[a]; b1.hi[a] = (double)bc1[a] + bh1[a];
                if (!(bh0[a] >= 0.0f) || !(bh1[a] >= 0.0f) || !std::isfinite(bc0[a]) || !std::isfinite(bc1[a]) || !std::isfinite(bh0[a]) || !std::isfinite(bh1[a])) bad++;
            }
            b0.parent = b1.parent = it.box;
            boxes.push_back(b0); todo.push_back(Item{N.ch.x, (int)boxes.size() - 1});
            boxes.push_back(b1); todo.push_back(Item{N.ch.y, (int)boxes.size() - 1});
        }
    }
    for (uint32_t i = 0; i < n; ++i) if (!seen[i]) bad++;
    if (nodes) *nodes = (uint32_t)T.nodes.size();
    if (depth) *depth = T.depth;
    if (leaves) *leaves = n_leaves;
    *violations = bad;
    return PT_OK;
}

extern "C" int ptgpu_set_scene(ptgpu_ctx *ctx, const pt_sphere *spheres, uint32_t n) {
    if (!ctx || !spheres || n == 0) return PT_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n; ++i)
        if (spheres[i].kind > PT_DIELECTRIC) return PT_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n_pad = (n + kPad - 1) / kPad * kPad;
    std::vector<float4> filt(n_pad), geom(n), mat(n);
    std::vector<float> inv_radius(n);
    std::vector<uint32_t> kind(n);
    for (uint32_t i = 0; i < n; ++i) {
        const pt_sphere &s = spheres[i];
        filt[i] = make_float4(s.cx, s.cz, filter_threshold(s), s.cy);
        geom[i] = make_float4(s.cx, s.cy, s.cz, s.radius);
        // per-sphere values the reference recomputes at every hit, in the same f32 arithmetic (this file is
        // built with -ffp-contract=off; host float division is correctly rounded like the device's)
        inv_radius[i] = 1.0f / s.radius;                                   // sphere.zig:88
        mat[i] = (s.kind == PT_DIELECTRIC) ? make_float4(1.0f / s.param, schlick_r0(s.param), 0.0f, s.param) // material.zig:143,121
                                           : make_float4(s.ar, s.ag, s.ab, s.param);
        kind[i] = s.kind;
    }
    for (uint32_t i = n; i < n_pad; ++i) filt[i] = make_float4(0.f, 0.f, INFINITY, 0.f); // never a candidate
    // Full mask words (32 consecutive spheres) whose spheres all share one finite centre ordinate take the 5-FMA test
    // (filter_value_y); the table is padded to whole LDS tiles, NaN = general test.  Pad records match any class.
    const uint32_t n_words = (n_pad + (uint32_t)kTile - 1) / (uint32_t)kTile * (uint32_t)(kTile / 32);
    std::vector<float> ycy(n_words, __builtin_nanf(""));
    for (uint32_t w = 0; (w + 1) * 32u <= n; ++w) { // full words only: the last, partial word always takes the general test
        const uint32_t lo = w * 32u, hi = lo + 32u;
        bool same = std::isfinite(spheres[lo].cy);
        for (uint32_t i = lo + 1; i < hi && same; ++i) same = memcmp(&spheres[i].cy, &spheres[lo].cy, sizeof(float)) == 0;
        if (same) ycy[w] = spheres[lo].cy;
    }
    for (float &v : ycy) if (v != v) { const uint32_t q = 0x7fc00000u; memcpy(&v, &q, 4); } // the kernel compares these bits
    // Build the new tables aside and swap them in only when every step succeeded: a failure leaves the
    // context with its previous scene intact (never half-updated).
    TreeBuild tree;
    bool with_tree = ctx->tree_min_spheres != 0u && n >= ctx->tree_min_spheres && tree_build(spheres, n, tree);
    // a one-tile scene walks its tree in LDS: it has to fit there.  Median splits alone always do (a range of more than 4 spheres
    // is halved: seven halvings bring 512 spheres down to 4, so at most 128 leaves, 127 inner nodes and 8 levels).
    if (with_tree && n_pad <= (uint32_t)kTile && (tree.nodes.size() > kTreeLdsNodes || tree.depth > kTreeLdsStack + 1u)) {
        tree = TreeBuild();
        tree.sah = false;
        with_tree = tree_build(spheres, n, tree) && tree.nodes.size() <= kTreeLdsNodes && tree.depth <= kTreeLdsStack + 1u;
    }
    constexpr int kTables = 9;
    void *fresh[kTables] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const void *src[kTables] = {filt.data(), geom.data(), mat.data(), kind.data(), inv_radius.data(), ycy.data(), tree.nodes.data(), tree.tgeom.data(),
                                tree.tprim.data()};
    const size_t bytes[kTables] = {n_pad * sizeof(float4), n * sizeof(float4), n * sizeof(float4), n * sizeof(uint32_t), n * sizeof(float),
                                   n_words * sizeof(float), tree.nodes.size() * sizeof(TreeNode), tree.tgeom.size() * sizeof(float4),
                                   tree.tprim.size() * sizeof(uint32_t)};
    hipError_t e = hipSuccess;
    for (int i = 0; i < (with_tree ? kTables : 6) && e == hipSuccess; ++i) {
        e = hipMalloc(&fresh[i], bytes[i]);
        if (e == hipSuccess) e = hipMemcpy(fresh[i], src[i], bytes[i], hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        for (void *p : fresh) if (p) (void)hipFree(p);
        ctx->last_hip = (int)e;
        return (e == hipErrorOutOfMemory) ? PT_ERR_OOM : PT_ERR_HIP;
    }
    void *olds[] = {ctx->d_filt, ctx->d_geom, ctx->d_mat, ctx->d_kind, ctx->d_inv_radius, ctx->d_ycy, ctx->d_tnodes, ctx->d_tgeom, ctx->d_tprim};
    ctx->d_tnodes = (TreeNode *)fresh[6]; ctx->d_tgeom = (float4 *)fresh[7]; ctx->d_tprim = (uint32_t *)fresh[8];
    ctx->tree_root = with_tree ? tree.root : 0u;
    ctx->tree_nodes = with_tree ? (uint32_t)tree.nodes.size() : 0u;
    ctx->tree_depth = with_tree ? tree.depth : 0u;
    ctx->d_filt = (float4 *)fresh[0]; ctx->d_geom = (float4 *)fresh[1]; ctx->d_mat = (float4 *)fresh[2];
    ctx->d_kind = (uint32_t *)fresh[3]; ctx->d_inv_radius = (float *)fresh[4]; ctx->d_ycy = (float *)fresh[5];
    ctx->nspheres = n;
    ctx->n_pad = n_pad;
    ctx->shared_tests = 0;
    for (uint32_t w = 0; w * 32u < n; ++w) if (ycy[w] == ycy[w]) ctx->shared_tests += 32u;
    for (void *p : olds) if (p) (void)hipFree(p);
    return PT_OK;
}

static Cam to_cam(const pt_camera *c) {
    Cam k;
    k.lookfrom = v3(c->lookfrom[0], c->lookfrom[1], c->lookfrom[2]);
    k.u = v3(c->u[0], c->u[1], c->u[2]);
    k.v = v3(c->v[0], c->v[1], c->v[2]);
    k.llc = v3(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    k.horizontal = v3(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    k.vertical = v3(c->vertical[0], c->vertical[1], c->vertical[2]);
    k.lens_radius = c->lens_radius;
    return k;
}

extern "C" int ptgpu_set_camera(ptgpu_ctx *ctx, const pt_camera *cam) {
    if (!ctx || !cam) return PT_ERR_BAD_ARG;
    ctx->cam = to_cam(cam);
    ctx->have_camera = true;
    return PT_OK;
}

static int check_cfg(const ptgpu_ctx *ctx, const pt_config *cfg, uint32_t *s0, uint32_t *s1, uint32_t *chunk) {
    if (!ctx || !cfg) return PT_ERR_BAD_ARG;
    if (!ctx->d_filt || !ctx->have_camera) return PT_ERR_BAD_ARG;
    if (cfg->width == 0 || cfg->height == 0 || cfg->subpixels == 0 || cfg->rays_per_subsample == 0) return PT_ERR_BAD_ARG;
    const uint64_t np = (uint64_t)cfg->width * cfg->height;
    if (np > 0x7fffffffull / 3) return PT_ERR_BAD_ARG;
    if (cfg->idx_start > cfg->idx_stop || cfg->idx_stop > np) return PT_ERR_BAD_ARG;
    if (cfg->mode > 1 || cfg->illumination > PT_ILLUM_ALBEDO || cfg->sampler > PT_SAMPLER_DIRECT) return PT_ERR_BAD_ARG;
    if (cfg->tile_pixels != 0 &&
        (cfg->tile_count == 0 || cfg->tile_index >= cfg->tile_count || cfg->idx_start % cfg->tile_pixels != 0))
        return PT_ERR_BAD_ARG;
    const uint64_t S = (uint64_t)cfg->subpixels * cfg->rays_per_subsample;
    if (S > 0xffffffffull) return PT_ERR_BAD_ARG;
    uint32_t a = cfg->sample_start, b = cfg->sample_stop;
    if (a == 0 && b == 0) b = (uint32_t)S;
    if (a > b || b > S) return PT_ERR_BAD_ARG;
    uint32_t c = cfg->chunk;
    if (c == 0 || c > b - a) c = (b > a) ? b - a : 1;
    *s0 = a; *s1 = b; *chunk = c;
    return PT_OK;
}

// Host side of pt_primary_lists (see its comment): the camera in double, the lens and footprint radii, the
// rounding allowances.
static ListCam make_list_cam(const Cam &c, uint32_t width, uint32_t height) {
    ListCam L;
    const V3 *src[4] = {&c.lookfrom, &c.llc, &c.horizontal, &c.vertical};
    double *dst[4] = {L.O, L.llc, L.H, L.V};
    double len[4];
    for (int i = 0; i < 4; ++i) {
        dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z;
        len[i] = sqrt(dst[i][0] * dst[i][0] + dst[i][1] * dst[i][1] + dst[i][2] * dst[i][2]);
    }
    // |u*x + v*y| <= sigma_max([u v]) |(x, y)|: largest eigenvalue of the 2x2 Gram matrix
    const double uu = (double)c.u.x * c.u.x + (double)c.u.y * c.u.y + (double)c.u.z * c.u.z;
    const double vv = (double)c.v.x * c.v.x + (double)c.v.y * c.v.y + (double)c.v.z * c.v.z;
    const double uv = (double)c.u.x * c.v.x + (double)c.u.y * c.v.y + (double)c.u.z * c.v.z;
    const double lam = 0.5 * ((uu + vv) + sqrt((uu - vv) * (uu - vv) + 4.0 * uv * uv));
    const double ulp = 5.9604644775390625e-08; // 2^-24
    L.Rl = fabs((double)c.lens_radius) * sqrt(lam) * (1.0 + 1e-6);
    // (col + xi) and (height - row + 1 + xi) are rounded to f32 before the division: up to an ulp of the index outside the box
    const double wh = (double)(width > height ? width : height);
    L.Rp = 0.5 * (len[2] / (double)width + len[3] / (double)height) * (1.0 + 1e-3 + 8.0 * wh * ulp);
#ifdef PT_LIST_NEGATIVE_CONTROL // tools/build_variant.sh: the pixel footprint taken at half its size — ptgpu_debug_list_safety must then report missing spheres
    L.Rp *= 0.5;
#endif
    const double mag = len[0] + len[1] + len[2] + len[3] + L.Rl;
    L.e_o = 16.0 * ulp * (len[0] + L.Rl);
    L.e_d = 32.0 * ulp * mag;
    return L;
}

// Device memory a render works in besides its inputs and output.
struct Workspace {
    unsigned long long *counters; // CNT_N
    uint32_t *stack; size_t stack_bytes;
    float *part; size_t part_bytes;
};

template <int MODE, bool PRIMARY>
static void launch_megakernel_mp(bool multi, uint32_t grid, hipStream_t stream, const KParams &P) {
    if (P.tnodes) {
        if (multi) hipLaunchKernelGGL((pt_megakernel<MODE, true, PRIMARY, true>), dim3(grid), dim3(kBlock), 0, stream, P);
        else hipLaunchKernelGGL((pt_megakernel<MODE, false, PRIMARY, true>), dim3(grid), dim3(kBlock), 0, stream, P);
        return;
    }
    if (multi) hipLaunchKernelGGL((pt_megakernel<MODE, true, PRIMARY>), dim3(grid), dim3(kBlock), 0, stream, P);
    else hipLaunchKernelGGL((pt_megakernel<MODE, false, PRIMARY>), dim3(grid), dim3(kBlock), 0, stream, P);
}
template <int MODE>
static void launch_megakernel_m(bool multi, uint32_t grid, hipStream_t stream, const KParams &P) {
    if (P.primary) launch_megakernel_mp<MODE, true>(multi, grid, stream, P);
    else launch_megakernel_mp<MODE, false>(multi, grid, stream, P);
}
static void launch_megakernel(int mode, bool multi, uint32_t grid, hipStream_t stream, const KParams &P) {
    if (mode == kModeDet) launch_megakernel_m<kModeDet>(multi, grid, stream, P);
    else if (mode == kModeDirect) launch_megakernel_m<kModeDirect>(multi, grid, stream, P);
    else launch_megakernel_m<kModeRejection>(multi, grid, stream, P);
}
template <int MODE>
static void launch_primary_m(bool multi, uint32_t pgrid, hipStream_t stream, const KParams &P, uint32_t nrec) {
    if (multi) hipLaunchKernelGGL((pt_primary<MODE, true>), dim3(pgrid), dim3(256), 0, stream, P, nrec);
    else hipLaunchKernelGGL((pt_primary<MODE, false>), dim3(pgrid), dim3(256), 0, stream, P, nrec);
}
static void launch_primary_check(bool multi, uint32_t pgrid, hipStream_t stream, const KParams &P, uint32_t nrec) {
    if (multi) hipLaunchKernelGGL((pt_primary<kModeRejection, true, true>), dim3(pgrid), dim3(256), 0, stream, P, nrec);
    else hipLaunchKernelGGL((pt_primary<kModeRejection, false, true>), dim3(pgrid), dim3(256), 0, stream, P, nrec);
}
static void launch_primary(int mode, bool multi, uint32_t pgrid, hipStream_t stream, const KParams &P, uint32_t nrec) {
    if (mode == kModeDet) launch_primary_m<kModeDet>(multi, pgrid, stream, P, nrec);
    else if (mode == kModeDirect) launch_primary_m<kModeDirect>(multi, pgrid, stream, P, nrec);
    else launch_primary_m<kModeRejection>(multi, pgrid, stream, P, nrec);
}

// The render behind both ptgpu_render_sum_device (workspace from hipMalloc, grown on demand) and
// ptgpu_render_sum_device_noalloc (workspace carved out of the static scratch; allocates nothing).
//
// Chunk partials are bounded: when (chunks x pixel slots x 12 B) exceeds the partial buffer the slots are
// traced in SEGMENTS, one megakernel launch + one pt_sum_chunks each, in stream order through the same
// buffer.  A pixel's partials are still added in chunk order, so the result does not depend on the
// segmentation (bit-identical to one launch); only the launch tails are paid once per segment.
static int render_sum_impl(ptgpu_ctx *ctx, const pt_config *cfg, float *d_rgb_sum, hipStream_t stream, bool noalloc) {
    uint32_t s0, s1, chunk;
    int rc = check_cfg(ctx, cfg, &s0, &s1, &chunk);
    if (rc != PT_OK) return rc;
    if (!d_rgb_sum) return PT_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t npix = cfg->idx_stop - cfg->idx_start;
    if (cfg->tile_pixels != 0) { // slots of this owner's tiles (the last one may be cut by idx_stop)
        const uint32_t ntiles = (npix + cfg->tile_pixels - 1) / cfg->tile_pixels;
        const uint32_t mine = (ntiles > cfg->tile_index) ? (ntiles - cfg->tile_index + cfg->tile_count - 1) / cfg->tile_count : 0;
        npix = mine * cfg->tile_pixels;
    }
    memset(&ctx->stats, 0, sizeof ctx->stats);
    if (npix == 0 || s1 == s0) return PT_OK;
    const uint32_t nchunks = (s1 - s0 + chunk - 1) / chunk;

    const bool tree = ctx->tree_nodes != 0u;
    const bool multi = ctx->n_pad > (uint32_t)kTile; // with a tree: walked in global memory (else staged in LDS)
    const bool det = cfg->mode == 1;
    const int mode = det ? kModeDet : (cfg->sampler == PT_SAMPLER_DIRECT ? kModeDirect : kModeRejection); // (the deterministic mode has no sampler: every draw is 0.5)
    // persistent grid: as many workgroups per CU as the kernel variant can keep resident
    const int bpc = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : (tree ? ctx->resident_tree : ctx->resident)[det ? 1 : 0][multi ? 1 : 0];
    const uint32_t grid = (uint32_t)(ctx->compute_units * bpc);
    const uint32_t total_threads = grid * kBlock;
    const size_t stack_need = (size_t)(cfg->rebounds ? cfg->rebounds : 1u) * total_threads * sizeof(uint32_t);

    Workspace ws{};
    size_t part_avail; // bytes the partial buffer may take
    if (noalloc) {
        const size_t stack_off = kScratchCounters;
        const size_t part_off = (stack_off + stack_need + 255u) & ~(size_t)255u;
        if (part_off > PT_SCRATCH_BYTES) return PT_ERR_SCRATCH_LIMIT;
        ws.counters = reinterpret_cast<unsigned long long *>(ctx->d_scratch);
        ws.stack = reinterpret_cast<uint32_t *>(ctx->d_scratch + stack_off); ws.stack_bytes = stack_need;
        ws.part = reinterpret_cast<float *>(ctx->d_scratch + part_off);
        part_avail = PT_SCRATCH_BYTES - part_off;
    } else {
        rc = ensure(ctx, (void **)&ctx->d_stack, &ctx->stack_bytes, stack_need);
        if (rc != PT_OK) return rc;
        ws.counters = ctx->d_counters;
        ws.stack = ctx->d_stack; ws.stack_bytes = ctx->stack_bytes;
        ws.part = nullptr;
        part_avail = (size_t)ctx->partial_cap;
    }
    // segment = pixel slots traced per launch: bounded by the partial buffer and, with the primary pass, by
    // the buffer of its records (32 B per path) and candidate lists
    auto fit_slots = [](uint64_t avail, uint64_t per_slot) -> uint64_t { return per_slot ? avail / per_slot : ~0ull; };
    const uint64_t part_per_slot = (nchunks > 1) ? (uint64_t)nchunks * 3u * sizeof(float) : 0u;
    const uint64_t prim_per_slot = (uint64_t)nchunks * chunk * 32u + (uint64_t)kListStride * sizeof(uint32_t) + (uint64_t)nchunks * sizeof(uint32_t);
    // the lists hold rays that leave the lens forwards; the no-allocation entry point keeps the camera rays in the megakernel
    uint64_t prim_cap = (noalloc || cfg->t_min < 0.0f || ctx->nspheres > kHitMask) ? 0u : ctx->primary_cap;
    if (prim_cap == kPrimaryAuto) {
        // Sized from what the device has free NOW (plus what this context already holds for the pass): half of it, so that a
        // host sharing the GPU with other allocators keeps room; the buffer is then allocated at the size the render needs.
        // A context that already holds enough for the whole frame (a render loop's second and later calls) asks nothing.
        const uint64_t whole = (uint64_t)npix * prim_per_slot + 2048u;
        if (ctx->d_prim && (uint64_t)ctx->prim_bytes >= whole) prim_cap = std::min<uint64_t>(kPrimaryAutoMax, (uint64_t)ctx->prim_bytes);
        else {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
            prim_cap = std::min<uint64_t>(kPrimaryAutoMax, ((uint64_t)free_b + (uint64_t)ctx->prim_bytes) / 2u);
        }
    }
    const uint64_t part_fit = fit_slots(part_avail, part_per_slot);
    if (part_fit < 64) return noalloc ? PT_ERR_SCRATCH_LIMIT : PT_ERR_BAD_ARG; // not even one wave of slots: raise the limit / chunk
    bool primary;
    uint32_t seg;
    size_t prim_rec_bytes = 0;
    for (;;) { // sized once; should the allocation still fail (another process took the memory since the query): with half of it, ..., then without the pass
        primary = prim_cap > 1024u && fit_slots(prim_cap - 1024u, prim_per_slot) >= 64;
        uint64_t fit = part_fit;
        if (primary) fit = std::min(fit, fit_slots(prim_cap - 1024u, prim_per_slot));
        uint64_t seg64 = npix;
        if (fit < seg64) seg64 = fit / 64u * 64u;
        // u32 item ids incl. the waves' last (overshooting) reservations; u32 record ids
        const uint64_t max_items = 0xf0000000ull;
        if (seg64 * nchunks > max_items) seg64 = max_items / nchunks / 64u * 64u;
        if (primary && seg64 * nchunks * chunk > max_items) seg64 = max_items / ((uint64_t)nchunks * chunk) / 64u * 64u;
        if (seg64 == 0) return PT_ERR_BAD_ARG;
        seg = (uint32_t)seg64;
        if (!primary) break;
        prim_rec_bytes = ((size_t)seg * nchunks * chunk * 32u + 255u) & ~(size_t)255u;
        rc = ensure(ctx, (void **)&ctx->d_prim, &ctx->prim_bytes, prim_rec_bytes + (size_t)seg * kListStride * sizeof(uint32_t) + (size_t)seg * nchunks * sizeof(uint32_t));
        if (rc == PT_OK) rc = ensure(ctx, (void **)&ctx->d_lsph, &ctx->lsph_bytes, (size_t)ctx->nspheres * sizeof(ListSphere));
        if (rc == PT_OK) break;
        if (rc != PT_ERR_OOM) return rc;
        (void)hipGetLastError();  // the failed allocation is not this render's error
        prim_cap = std::min<uint64_t>(prim_cap, (uint64_t)seg * prim_per_slot) / 2u;
    }
    if (nchunks > 1) {
        const size_t part_need = (size_t)seg * part_per_slot;
        if (noalloc) ws.part_bytes = part_need;
        else {
            rc = ensure(ctx, (void **)&ctx->d_part, &ctx->part_bytes, part_need);
            if (rc != PT_OK) return rc;
            ws.part = ctx->d_part; ws.part_bytes = ctx->part_bytes;
        }
    }

    KParams P;
    P.cam = ctx->cam;
    P.blend_start = v3(cfg->blend_start[0], cfg->blend_start[1], cfg->blend_start[2]);
    P.blend_stop = v3(cfg->blend_stop[0], cfg->blend_stop[1], cfg->blend_stop[2]);
    P.t_min = cfg->t_min; P.t_max = cfg->t_max;
    P.width = cfg->width; P.height = cfg->height;
    P.rebounds = cfg->rebounds;
    P.illum = cfg->illumination;
    P.seed_lo = (uint32_t)cfg->seed; P.seed_hi = (uint32_t)(cfg->seed >> 32);
    P.idx_start = cfg->idx_start; P.idx_stop = cfg->idx_stop;
    P.tile_pixels = cfg->tile_pixels; P.tile_count = cfg->tile_count; P.tile_index = cfg->tile_index;
    P.direct = (nchunks == 1) ? 1u : 0u;
    P.s0 = s0; P.s1 = s1; P.chunk = chunk; P.nchunks = nchunks;
    P.div_width = make_fastdiv(cfg->width); P.div_nchunks = make_fastdiv(nchunks); P.div_chunk = make_fastdiv(chunk); P.div_tile = make_fastdiv(cfg->tile_pixels ? cfg->tile_pixels : 1u);
    P.nspheres = ctx->nspheres;
    P.total_threads = total_threads;
    P.filt = ctx->d_filt; P.ycy = ctx->d_ycy; P.has_shared = ctx->shared_tests ? 1u : 0u; P.geom = ctx->d_geom; P.mat = ctx->d_mat; P.kind = ctx->d_kind; P.inv_radius = ctx->d_inv_radius;
    P.part = (nchunks == 1) ? d_rgb_sum : ws.part; // one chunk: sums go straight to the frame
    P.stack = ws.stack; P.counters = ws.counters;
    P.tnodes = tree ? ctx->d_tnodes : nullptr; P.tgeom = ctx->d_tgeom; P.tprim = ctx->d_tprim; P.troot = ctx->tree_root; P.tnode_count = ctx->tree_nodes;
    // When a ray step's walk is left to shade the finished lanes (scene_is_hit_tree): the longer the walks, the earlier it pays.
    // Measured per 256 samples of 1080p, threshold 0 / 4 / 8 / 12 / 16 / 24 / 32 lanes: scene 21 in LDS 40.7 / 39.8 / 39.7 / - / 40.7 / 41.8 / 44.1 ms,
    // 3 000 spheres 41.1 / - / 37.2 / 37.2 / 37.6 / 38.5 / 40.3, 100 000 spheres 276.5 / - / 261.0 / 253.6 / 245.5 / 226.4 / 224.6.
    P.tree_exit_at = !multi ? 8u : (ctx->tree_nodes > 8192u ? 24u : 12u);
    { const char *e = getenv("PTGPU_TREE_EXIT_AT"); if (e) P.tree_exit_at = (uint32_t)atoi(e); } // A/B runs
    // Postponed leaves are tested once this many lanes hold one.  Measured per 256 samples of 1080p, threshold 1 (every leaf where it is met) / 4 / 8 / 12 / 16 / 24 / 32 / 48:
    // scene 21 in LDS 36.6 / 35.5 / 35.3 / 35.5 / 36.0 / 36.3 / 36.6 / 37.3 ms; 3 000 spheres 34.0 / - / 31.7 / - / 32.0 / 32.2; 100 000 spheres (64 samples) 68.8 / - / 65.4 / - / 63.9 / 63.9 / 73.6
    P.tree_flush_at = (multi && ctx->tree_nodes > 8192u) ? 16u : 8u;
    { const char *e = getenv("PTGPU_TREE_FLUSH_AT"); if (e) P.tree_flush_at = (uint32_t)atoi(e); } // A/B runs (1: every leaf where it is met, as before)
    P.primary = primary ? 1u : 0u;
    P.rec = primary ? reinterpret_cast<float4 *>(ctx->d_prim) : nullptr;
    uint32_t *d_lists = primary ? reinterpret_cast<uint32_t *>(ctx->d_prim + prim_rec_bytes) : nullptr;
    P.lists = d_lists;
    P.item_mask = primary ? d_lists + (size_t)seg * kListStride : nullptr;
    P.use_mask = (primary && chunk <= 32u && 64u % chunk == 0u) ? 1u : 0u;
    const ListCam LC = primary ? make_list_cam(ctx->cam, cfg->width, cfg->height) : ListCam{};
    const bool one_seg = seg >= npix;

    HIP_TRY(ctx, hipMemsetAsync(ws.counters, 0, CNT_N * sizeof(unsigned long long), stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    uint32_t nseg = 0, n_timed = 0;
    for (uint32_t slot0 = 0; slot0 < npix; slot0 += seg, ++nseg) {
        P.slot0 = slot0;
        P.npix = std::min(seg, npix - slot0);
        P.div_npix = make_fastdiv(P.npix);
        P.blocks_per_row = ((uint64_t)P.npix * chunk % 64u == 0u) ? (uint32_t)((uint64_t)P.npix * chunk / 64u) : 0u;
        if (slot0 != 0) HIP_TRY(ctx, hipMemsetAsync(&ws.counters[CNT_QUEUE], 0, sizeof(unsigned long long), stream)); // the item queue restarts
        bool timed = nseg < kMaxTimedSegments;
        if (timed && ctx->evp.size() < (size_t)kSegEvents * (nseg + 1u)) {
            if (noalloc) timed = false; // no event is created on this path
            else while (ctx->evp.size() < (size_t)kSegEvents * (nseg + 1u)) {
                hipEvent_t e = nullptr;
                HIP_TRY(ctx, hipEventCreate(&e));
                ctx->evp.push_back(e);
            }
        }
        if (timed) n_timed++;
        hipEvent_t *evs = timed ? &ctx->evp[(size_t)kSegEvents * nseg] : nullptr;
        if (timed) HIP_TRY(ctx, hipEventRecord(evs[0], stream));
        if (!primary && timed) { HIP_TRY(ctx, hipEventRecord(evs[1], stream)); HIP_TRY(ctx, hipEventRecord(evs[2], stream)); }
        if (primary) {
            if (nseg == 0) { // what the lists need of each sphere, once per render
                hipLaunchKernelGGL(pt_list_prep, dim3((ctx->nspheres + 255u) / 256u), dim3(256), 0, stream, LC, (const float4 *)ctx->d_geom, ctx->nspheres,
                                   (ListSphere *)ctx->d_lsph);
                HIP_TRY(ctx, hipGetLastError());
            }
            const uint32_t second = (ctx->primary_bounces >= 2 && cfg->illumination == PT_ILLUM_RADIANCE && cfg->rebounds >= 2u) ? 1u : 0u;
            if (tree && multi) // (a one-tile scene's lists come from the loop over its spheres either way: 1.5 ms per launch, ascending order)
                hipLaunchKernelGGL(pt_primary_lists<true>, dim3((P.npix + 255u) / 256u), dim3(256), 0, stream, LC, cfg->width, cfg->height, slot0, P.npix,
                                   cfg->idx_start, cfg->idx_stop, cfg->tile_pixels, cfg->tile_count, cfg->tile_index, (const ListSphere *)ctx->d_lsph,
                                   ctx->nspheres, (const uint32_t *)ctx->d_kind, (const float4 *)ctx->d_mat, cfg->t_min, second, d_lists,
                                   (const TreeNode *)ctx->d_tnodes, (const uint32_t *)ctx->d_tprim, ctx->tree_root);
            else
                hipLaunchKernelGGL(pt_primary_lists<false>, dim3((P.npix + 255u) / 256u), dim3(256), 0, stream, LC, cfg->width, cfg->height, slot0, P.npix,
                                   cfg->idx_start, cfg->idx_stop, cfg->tile_pixels, cfg->tile_count, cfg->tile_index, (const ListSphere *)ctx->d_lsph,
                                   ctx->nspheres, (const uint32_t *)ctx->d_kind, (const float4 *)ctx->d_mat, cfg->t_min, second, d_lists,
                                   (const TreeNode *)nullptr, (const uint32_t *)nullptr, 0u);
            HIP_TRY(ctx, hipGetLastError());
            if (timed) HIP_TRY(ctx, hipEventRecord(evs[1], stream));
            const uint32_t nrec = P.npix * nchunks * chunk; // <= 0xf0000000 by the segment size
            const uint32_t pgrid = std::min<uint32_t>((nrec + 255u) / 256u, (uint32_t)ctx->compute_units * 32u);
            if (ctx->debug_list_check) launch_primary_check(multi, pgrid, stream, P, nrec);
            else launch_primary(mode, multi, pgrid, stream, P, nrec);
            HIP_TRY(ctx, hipGetLastError());
            if (timed) HIP_TRY(ctx, hipEventRecord(evs[2], stream));
        }
        if (!ctx->debug_list_check) launch_megakernel(mode, multi, grid, stream, P); // (the check stops after the primary pass)
        HIP_TRY(ctx, hipGetLastError());
        if (timed) HIP_TRY(ctx, hipEventRecord(evs[3], stream));
        if (one_seg) HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream)); // one segment: ev0..ev1 is the tracing kernels alone
        if (nchunks > 1 && !ctx->debug_list_check) {
            const uint32_t n3 = P.npix * 3u;
            hipLaunchKernelGGL(pt_sum_chunks, dim3((n3 + 255u) / 256u), dim3(256), 0, stream, (const float *)ws.part, slot0, P.npix, nchunks,
                               cfg->idx_start, cfg->idx_stop, cfg->tile_pixels, cfg->tile_count, cfg->tile_index, d_rgb_sum);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    if (nseg > 1) HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream)); // several segments: the span includes their (tiny) chunk sums
    HIP_TRY(ctx, hipEventRecord(ctx->ev2, stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ws.counters, CNT_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    const unsigned long long *h = ctx->h_counters;
    float ms01 = 0.f, ms02 = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms01, ctx->ev0, ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(&ms02, ctx->ev0, ctx->ev2));
    ctx->stats.paths = h[CNT_PATHS];
    ctx->stats.rays = h[CNT_RAYS];
    ctx->stats.sphere_tests = h[CNT_RAYS] * ctx->nspheres;
    ctx->stats.candidates = h[CNT_CANDS];
    ctx->stats.primary_rays = h[CNT_PRIMARY_RAYS];
    ctx->stats.list2_rays = h[CNT_LIST2_RAYS];
    ctx->stats.box_tests = h[CNT_BOX_TESTS];
    ctx->stats.primary_candidates = h[CNT_PRIMARY_CANDS];
    ctx->stats.primary_ms = ctx->stats.lists_ms = ctx->stats.megakernel_ms = 0.0;
    if (n_timed == nseg) { // every segment carried its events
        for (uint32_t i = 0; i < nseg; ++i) {
            const hipEvent_t *evs = &ctx->evp[(size_t)kSegEvents * i];
            float ms_l = 0.f, ms_p = 0.f, ms_m = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms_l, evs[0], evs[1]));
            HIP_TRY(ctx, hipEventElapsedTime(&ms_p, evs[0], evs[2]));
            HIP_TRY(ctx, hipEventElapsedTime(&ms_m, evs[2], evs[3]));
            ctx->stats.lists_ms += ms_l; ctx->stats.primary_ms += ms_p; ctx->stats.megakernel_ms += ms_m;
        }
    }
    ctx->stats.not_unit = h[CNT_NOT_UNIT];
#ifdef PT_PHASE_TIMING
    {
        const double tot = (double)h[12];
        fprintf(stderr, "[phase timing] shader clock %.0f MHz (s_memtime / s_memrealtime)\n", tot / (double)h[13] * 100.0);
        fprintf(stderr, "[phase timing] share of wave cycles: refill %.3f  scan %.3f  (of which flush %.3f)  shade %.3f\n",
                h[8] / tot, h[9] / tot, h[10] / tot, h[11] / tot);
        if (h[CNT_PP0 + 7]) {
            const double tp = (double)h[CNT_PP0 + 7];
            fprintf(stderr, "[phase timing] pt_primary, share of its wave cycles: path start (Philox, jitter, lens, castRay) %.3f  list A %.3f  shade 0 (hit point, scatter) %.3f  "
                            "list B %.3f  shade 1 %.3f  record write %.3f  (of path start: index arithmetic %.3f)\n", h[CNT_PP0] / tp, h[CNT_PP0 + 1] / tp, h[CNT_PP0 + 2] / tp, h[CNT_PP0 + 3] / tp, h[CNT_PP0 + 4] / tp, h[CNT_PP0 + 5] / tp, h[CNT_PP0 + 6] / tp);
        }
        fprintf(stderr, "[phase timing]   within shade: hit point %.3f  scatter %.3f  miss (lerp + unwinding) %.3f;  within refill: path start %.3f\n",
                h[16] / tot, h[17] / tot, h[18] / tot, h[19] / tot);
    }
#endif
#ifdef PT_WALK_PROFILE
    {
        unsigned long long wp[8];
        if (hipMemcpyFromSymbol(wp, HIP_SYMBOL(g_walk_prof), sizeof wp) == hipSuccess && wp[4]) {
            fprintf(stderr, "[walk profile] loop iterations %llu: with a lane at a leaf %.3f (%.1f lanes each), with a lane at an inner node %.3f (%.1f lanes each)\n", wp[4],
                    (double)wp[2] / wp[4], (double)wp[0] / (wp[2] ? wp[2] : 1), (double)wp[3] / wp[4], (double)wp[1] / (wp[3] ? wp[3] : 1));
            fprintf(stderr, "[walk profile] lanes still walking per iteration %.1f, of them stalled at a second leaf %.1f\n", (double)wp[5] / wp[4], (double)wp[6] / wp[4]);
            memset(wp, 0, sizeof wp);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_prof), wp, sizeof wp);
        }
    }
#endif
    ctx->stats.wave_steps = h[CNT_WAVE_STEPS];
    ctx->stats.flush_iters = h[CNT_FLUSH_ITERS];
    ctx->stats.kernel_ms = ms01;
    ctx->stats.total_ms = ms02;
    ctx->stats.segments = nseg;
    if (h[CNT_NOT_UNIT]) return PT_ERR_DIRECTION_NOT_UNIT;
    return PT_OK;
}

extern "C" int ptgpu_render_sum_device(ptgpu_ctx *ctx, const pt_config *cfg, float *d_rgb_sum, void *stream) {
    return render_sum_impl(ctx, cfg, d_rgb_sum, (hipStream_t)stream, false);
}

extern "C" int ptgpu_render_sum_device_noalloc(ptgpu_ctx *ctx, const pt_config *cfg, float *d_rgb_sum, void *stream) {
    return render_sum_impl(ctx, cfg, d_rgb_sum, (hipStream_t)stream, true);
}

extern "C" int ptgpu_render_sum_host(ptgpu_ctx *ctx, const pt_config *cfg, float *h_rgb_sum) {
    if (!ctx || !cfg || !h_rgb_sum) return PT_ERR_BAD_ARG;
    uint32_t s0_, s1_, chunk_;
    int rc = check_cfg(ctx, cfg, &s0_, &s1_, &chunk_); // before anything is allocated from cfg's sizes
    if (rc != PT_OK) return rc;
    const size_t bytes = (size_t)cfg->width * cfg->height * 3u * sizeof(float);
    rc = ensure(ctx, (void **)&ctx->d_tmp, &ctx->tmp_bytes, bytes);
    if (rc != PT_OK) return rc;
    const size_t off = (size_t)cfg->idx_start * 3u, cnt = (size_t)(cfg->idx_stop - cfg->idx_start) * 3u;
    // tile split: other owners' pixels inside the range must come back untouched
    if (cfg->tile_pixels != 0 && cnt) HIP_TRY(ctx, hipMemcpy(ctx->d_tmp + off, h_rgb_sum + off, cnt * sizeof(float), hipMemcpyHostToDevice));
    rc = ptgpu_render_sum_device(ctx, cfg, ctx->d_tmp, nullptr);
    if (rc != PT_OK && rc != PT_ERR_DIRECTION_NOT_UNIT) return rc;
    if (cnt) HIP_TRY(ctx, hipMemcpy(h_rgb_sum + off, ctx->d_tmp + off, cnt * sizeof(float), hipMemcpyDeviceToHost));
    return rc;
}

extern "C" int ptgpu_render_mean_host(ptgpu_ctx *ctx, const pt_config *cfg, float *h_rgb_mean) {
    if (!ctx || !cfg || !h_rgb_mean) return PT_ERR_BAD_ARG;
    if (cfg->tile_pixels != 0) return PT_ERR_UNSUPPORTED; // resolve whole ranges only: combine the owners' sums first
    uint32_t s0_, s1_, chunk_;
    int rc = check_cfg(ctx, cfg, &s0_, &s1_, &chunk_); // before anything is allocated from cfg's sizes
    if (rc != PT_OK) return rc;
    const size_t bytes = (size_t)cfg->width * cfg->height * 3u * sizeof(float);
    rc = ensure(ctx, (void **)&ctx->d_tmp, &ctx->tmp_bytes, bytes);
    if (rc != PT_OK) return rc;
    rc = ptgpu_render_sum_device(ctx, cfg, ctx->d_tmp, nullptr);
    if (rc != PT_OK && rc != PT_ERR_DIRECTION_NOT_UNIT) return rc;
    const uint32_t npix = cfg->idx_stop - cfg->idx_start;
    if (npix) {
        const float k1 = 1.0f / (float)cfg->subpixels, k2 = 1.0f / (float)cfg->rays_per_subsample;
        const uint32_t n3 = npix * 3u;
        hipLaunchKernelGGL(pt_mean, dim3((n3 + 255u) / 256u), dim3(256), 0, nullptr, (const float *)ctx->d_tmp, cfg->idx_start, npix, k1, k2, ctx->d_tmp);
        HIP_TRY(ctx, hipGetLastError());
        const size_t off = (size_t)cfg->idx_start * 3u;
        HIP_TRY(ctx, hipMemcpy(h_rgb_mean + off, ctx->d_tmp + off, (size_t)n3 * sizeof(float), hipMemcpyDeviceToHost));
    }
    return rc;
}

extern "C" int ptgpu_resolve_device(ptgpu_ctx *ctx, const pt_config *cfg, const float *d_rgb_sum, uint8_t *d_records,
                                    uint8_t *d_rgb8, void *stream_) {
    if (!ctx || !cfg || !d_rgb_sum) return PT_ERR_BAD_ARG;
    if (cfg->max_px_value != 255u) return PT_ERR_UNSUPPORTED;
    if (cfg->idx_start > cfg->idx_stop || (uint64_t)cfg->idx_stop > (uint64_t)cfg->width * cfg->height) return PT_ERR_BAD_ARG;
    if (cfg->subpixels == 0 || cfg->rays_per_subsample == 0) return PT_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t npix = cfg->idx_stop - cfg->idx_start;
    if (npix == 0) return PT_OK;
    const float k1 = 1.0f / (float)cfg->subpixels, k2 = 1.0f / (float)cfg->rays_per_subsample;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters + CNT_CLAMPED, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(pt_resolve, dim3((npix + 255u) / 256u), dim3(256), 0, stream, d_rgb_sum, cfg->idx_start, npix, k1, k2,
                       (float)cfg->max_px_value, d_records, d_rgb8, ctx->d_counters);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    // renderers/ppm_image.zig:75-77 asserts col <= 255 (and @floatToInt(u8) is undefined beyond it in
    // ReleaseFast): such values are written clamped to 255 and reported
    unsigned long long clamped = 0;
    HIP_TRY(ctx, hipMemcpy(&clamped, ctx->d_counters + CNT_CLAMPED, sizeof clamped, hipMemcpyDeviceToHost));
    return clamped ? PT_ERR_VALUE_EXCEEDS_MAX : PT_OK;
}

extern "C" int ptgpu_render_ppm(ptgpu_ctx *ctx, const pt_config *cfg, uint8_t *slice, size_t slice_len, uint32_t header_size,
                                uint32_t px_size) {
    if (!ctx || !cfg || !slice) return PT_ERR_BAD_ARG;
    if (cfg->max_px_value != 255u || px_size != 12u || cfg->tile_pixels != 0) return PT_ERR_UNSUPPORTED;
    uint32_t s0_, s1_, chunk_;
    int rc = check_cfg(ctx, cfg, &s0_, &s1_, &chunk_); // before anything is allocated from cfg's sizes
    if (rc != PT_OK) return rc;
    const uint32_t npix = cfg->idx_stop - cfg->idx_start;
    if ((size_t)header_size + (size_t)cfg->idx_stop * px_size > slice_len) return PT_ERR_NO_SPACE;
    const size_t bytes = (size_t)cfg->width * cfg->height * 3u * sizeof(float);
    rc = ensure(ctx, (void **)&ctx->d_tmp, &ctx->tmp_bytes, bytes);
    if (rc != PT_OK) return rc;
    rc = ptgpu_render_sum_device(ctx, cfg, ctx->d_tmp, nullptr);
    if (rc != PT_OK) return rc; // DirectionIsNotUnitVector aborts render() (renderers/ppm_image.zig:66 `try`)
    if (npix == 0) return PT_OK;
    const pt_stats keep = ctx->stats;
    rc = ensure(ctx, (void **)&ctx->d_rec, &ctx->rec_bytes, (size_t)npix * 12u);
    if (rc != PT_OK) return rc;
    rc = ptgpu_resolve_device(ctx, cfg, ctx->d_tmp, ctx->d_rec, nullptr, nullptr);
    if (rc != PT_OK && rc != PT_ERR_VALUE_EXCEEDS_MAX) return rc;
    HIP_TRY(ctx, hipMemcpy(slice + header_size + (size_t)cfg->idx_start * px_size, ctx->d_rec, (size_t)npix * 12u, hipMemcpyDeviceToHost));
    ctx->stats = keep;
    return rc; // PT_ERR_VALUE_EXCEEDS_MAX: records written with the offending values clamped to 255
}

extern "C" int ptgpu_debug_eval(ptgpu_ctx *ctx, int fn, const float *in, float *out, uint32_t n, float t_min, float t_max) {
    if (!ctx || !in || !out) return PT_ERR_BAD_ARG;
    if (n == 0) return PT_OK;
    size_t in_stride = 0, out_stride = 0;
    switch (fn) {
    case PT_FN_CAST_RAY: in_stride = 18; out_stride = 7; break;
    case PT_FN_SCATTER: in_stride = 27; out_stride = 4; break;
    case PT_FN_SPHERE_HIT: in_stride = 12; out_stride = 8; break;
    case PT_FN_SCENE_HIT: in_stride = 6; out_stride = 8; break;
    case PT_FN_SCENE_HIT_FILTERED: in_stride = 6; out_stride = 8; break;
    case PT_FN_REFRACT: in_stride = 7; out_stride = 4; break;
    case PT_FN_SCHLICK: in_stride = 2; out_stride = 1; break;
    case PT_FN_REFLECT: in_stride = 6; out_stride = 3; break;
    case PT_FN_LERP: in_stride = 9; out_stride = 4; break;
    case PT_FN_PHILOX: in_stride = 5; out_stride = 4; break;
    case PT_FN_FILTER_BIT: in_stride = 10; out_stride = 3; break;
    case PT_FN_SCENE_HIT_TREE: in_stride = 6; out_stride = 8; break;
    case PT_FN_TREE_BOX_SAFETY: in_stride = 6; out_stride = 4; break;
    case PT_FN_SAMPLE_DISK: case PT_FN_SAMPLE_BALL: in_stride = 17; out_stride = 4; break;
    default: return PT_ERR_UNSUPPORTED;
    }
    if ((fn == PT_FN_SCENE_HIT || fn == PT_FN_SCENE_HIT_FILTERED) && !ctx->d_geom) return PT_ERR_BAD_ARG;
    if (fn == PT_FN_CAST_RAY && !ctx->have_camera) return PT_ERR_BAD_ARG;
    if ((fn == PT_FN_SCENE_HIT_TREE || fn == PT_FN_TREE_BOX_SAFETY) && ctx->tree_nodes == 0u) return PT_ERR_UNSUPPORTED;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d_in = nullptr, *d_out = nullptr, *d_thr = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_in, in_stride * n * sizeof(float)));
    hipError_t e = hipMalloc((void **)&d_out, out_stride * n * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(d_in); ctx->last_hip = (int)e; return PT_ERR_OOM; }
    int rc = PT_OK;
    e = hipMemcpy(d_in, in, in_stride * n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && fn == PT_FN_FILTER_BIT) { // the per-sphere threshold ptgpu_set_scene would tabulate
        std::vector<float> thr(n);
        for (uint32_t i = 0; i < n; ++i) thr[i] = filter_threshold_of(in[(size_t)i * 10 + 6], in[(size_t)i * 10 + 7], in[(size_t)i * 10 + 8], in[(size_t)i * 10 + 9]);
        e = hipMalloc((void **)&d_thr, n * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d_thr, thr.data(), n * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        Cam cam = ctx->have_camera ? ctx->cam : Cam();
        if (fn == PT_FN_FILTER_BIT) {
            hipLaunchKernelGGL(pt_debug_filter_bit, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, (const float *)d_in, (const float *)d_thr, d_out, n);
        } else if (fn == PT_FN_SCENE_HIT_TREE) {
            if (ctx->n_pad <= (uint32_t)kTile) // the walk the megakernel would use for this scene: tree staged in LDS
                hipLaunchKernelGGL(pt_debug_scene_hit_tree<true>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, (const float *)d_in, d_out, n,
                                   (const TreeNode *)ctx->d_tnodes, ctx->tree_nodes, (const float4 *)ctx->d_tgeom, (const uint32_t *)ctx->d_tprim, ctx->nspheres,
                                   ctx->tree_root, (const float4 *)ctx->d_geom, t_min, t_max, 8u);
            else
                hipLaunchKernelGGL(pt_debug_scene_hit_tree<false>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, (const float *)d_in, d_out, n,
                                   (const TreeNode *)ctx->d_tnodes, ctx->tree_nodes, (const float4 *)ctx->d_tgeom, (const uint32_t *)ctx->d_tprim, ctx->nspheres,
                                   ctx->tree_root, (const float4 *)ctx->d_geom, t_min, t_max, 8u);
        } else if (fn == PT_FN_TREE_BOX_SAFETY) {
            hipLaunchKernelGGL(pt_debug_tree_box_safety, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, (const float *)d_in, d_out, n,
                               (const TreeNode *)ctx->d_tnodes, (const float4 *)ctx->d_tgeom, ctx->tree_root, t_min, t_max);
        } else if (fn == PT_FN_SCENE_HIT_FILTERED) {
            const uint32_t blocks = (n + kBlock - 1) / kBlock;
            if (ctx->n_pad > (uint32_t)kTile)
                hipLaunchKernelGGL((pt_debug_scene_hit_filtered<true>), dim3(blocks), dim3(kBlock), 0, nullptr, (const float *)d_in, d_out, n,
                                   (const float4 *)ctx->d_filt, (const float *)ctx->d_ycy, (const float4 *)ctx->d_geom, ctx->nspheres, t_min, t_max);
            else
                hipLaunchKernelGGL((pt_debug_scene_hit_filtered<false>), dim3(blocks), dim3(kBlock), 0, nullptr, (const float *)d_in, d_out, n,
                                   (const float4 *)ctx->d_filt, (const float *)ctx->d_ycy, (const float4 *)ctx->d_geom, ctx->nspheres, t_min, t_max);
        } else if (fn == PT_FN_CAST_RAY || fn == PT_FN_SCATTER || fn == PT_FN_SAMPLE_DISK || fn == PT_FN_SAMPLE_BALL)
            hipLaunchKernelGGL(pt_debug_eval_rng, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, fn, (const float *)d_in, d_out, n, cam);
        else
            hipLaunchKernelGGL(pt_debug_eval, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, fn, (const float *)d_in, d_out, n, cam,
                               (const float4 *)ctx->d_geom, (const float4 *)ctx->d_mat, (const uint32_t *)ctx->d_kind, ctx->nspheres, t_min, t_max);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(out, d_out, out_stride * n * sizeof(float), hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) { ctx->last_hip = (int)e; rc = PT_ERR_HIP; }
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (d_thr) (void)hipFree(d_thr);
    return rc;
}

extern "C" int ptgpu_debug_list_safety(ptgpu_ctx *ctx, const pt_config *cfg, uint64_t out[7]) {
    if (!ctx || !cfg || !out) return PT_ERR_BAD_ARG;
    if (cfg->mode != 0u || cfg->sampler != PT_SAMPLER_REJECTION) return PT_ERR_UNSUPPORTED; // the check kernel is built for the default mode
    ctx->debug_list_check = true;
    // (the frame argument is not written in this mode: no megakernel, no chunk sums; any non-null device pointer will do)
    const int rc = render_sum_impl(ctx, cfg, reinterpret_cast<float *>(ctx->d_counters), nullptr, false);
    ctx->debug_list_check = false;
    if (rc != PT_OK) return rc;
    if (ctx->stats.primary_rays == 0u) return PT_ERR_UNSUPPORTED; // no primary pass for this scene / configuration / limit: nothing was checked
    for (int i = 0; i < 7; ++i) out[i] = ctx->h_counters[CNT_CHECK_ACC_A + i];
    return PT_OK;
}

extern "C" int ptgpu_debug_exhaustive(ptgpu_ctx *ctx, int which, uint64_t *mismatches) {
    if (!ctx || !mismatches || which < 0 || which > 3) return PT_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(pt_debug_exhaustive, dim3((1u << 24) / 256u), dim3(256), 0, nullptr, which, ctx->d_counters);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    unsigned long long bad = 0;
    HIP_TRY(ctx, hipMemcpy(&bad, ctx->d_counters, sizeof bad, hipMemcpyDeviceToHost));
    *mismatches = bad;
    return PT_OK;
}

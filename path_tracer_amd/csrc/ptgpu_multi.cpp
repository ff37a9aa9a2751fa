// ptgpu_multi.cpp — the GPUs of one node behind the C ABI (include/ptgpu.h, ptgpu_multi_*).
//
// Replaces the reference's only work split — main.zig:142-165: T OS threads, each handed a contiguous
// pixel chunk (multithreading.zig:40-44), joined before the header is written — with one host thread per
// GPU, each driving its own single-GPU context on its own stream, and ONE exchange step: an RCCL
// reduce(SUM) of the fp32 frame sums to GPU 0 over xGMI.
//   PT_SPLIT_SAMPLES  GPU g traces samples [s0 + g*S/G, s0 + (g+1)*S/G) of every pixel (high-spp frames);
//   PT_SPLIT_TILES    runs of tile_pixels pixels dealt round-robin; other owners' pixels stay 0, so the
//                     same reduce assembles the frame exactly (x + 0 = x), bit-identical to one GPU.
// Built on the single-GPU entry points only (no access to their internals).  RCCL is loaded with dlopen
// at the first multi-GPU create: single-GPU users never pay for the 570 MB library, and a process that
// has already loaded an RCCL (PyTorch ships one under the same soname) shares it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/ptgpu.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::once_flag once;
    bool load() { // (two threads may create their first multi-GPU handle at the same time)
        std::call_once(once, [this] { load_once(); });
        return handle != nullptr;
    }
    bool load_once() {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(h, "ncclReduce");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        if (!CommInitAll || !CommDestroy || !Reduce || !GroupStart || !GroupEnd) { dlclose(h); return false; }
        handle = h;
        return true;
    }
};
Rccl g_rccl;

} // namespace

struct ptgpu_multi {
    int n = 0;
    std::vector<int> devices;
    std::vector<ptgpu_ctx *> ctx;
    std::vector<hipStream_t> stream;
    std::vector<float *> frame;      // per-device W*H*3 sums (device 0's receives the reduce)
    std::vector<size_t> frame_bytes;
    std::vector<ncclComm_t> comm;    // empty: no communicator (one GPU, RCCL not forced)
    bool host_sum = false;           // tests only (PTGPU_MULTI_ALLOW_DUPLICATE_DEVICES=1 and a device listed twice): frames added on the host
    uint8_t *d_rec = nullptr; size_t rec_bytes = 0; // device 0: records of ptgpu_multi_render_ppm
    std::vector<pt_stats> stats;
    double reduce_ms = 0.0;
    int last_rccl = 0;
};

// The share of GPU `gpu` of `num_gpus`.  Returns PT_OK and sets *has_work (0: this GPU has nothing to trace).
extern "C" int ptgpu_multi_plan(const pt_config *cfg, int split, int num_gpus, int gpu, pt_config *out, int *has_work) {
    if (!cfg || !out || !has_work || num_gpus <= 0 || gpu < 0 || gpu >= num_gpus) return PT_ERR_BAD_ARG;
    if (cfg->tile_pixels != 0 && cfg->tile_count > 1) return PT_ERR_UNSUPPORTED; // already one owner's share: splits do not nest
    *out = *cfg;
    *has_work = 1;
    if (split == PT_SPLIT_SAMPLES) {
        const uint64_t S = (uint64_t)cfg->subpixels * cfg->rays_per_subsample;
        if (S == 0 || S > 0xffffffffull) return PT_ERR_BAD_ARG;
        uint64_t a = cfg->sample_start, b = cfg->sample_stop;
        if (a == 0 && b == 0) b = S; // both 0 = all samples (include/ptgpu.h)
        if (a > b || b > S) return PT_ERR_BAD_ARG;
        const uint64_t n = b - a;
        const uint64_t s0 = a + (uint64_t)gpu * n / (uint64_t)num_gpus, s1 = a + (uint64_t)(gpu + 1) * n / (uint64_t)num_gpus;
        out->sample_start = (uint32_t)s0; out->sample_stop = (uint32_t)s1;
        out->tile_pixels = 0; out->tile_count = 1; out->tile_index = 0;
        if (s1 == s0) *has_work = 0; // more GPUs than samples: (0,0) must not come to mean "all samples"
        return PT_OK;
    }
    if (split == PT_SPLIT_TILES) {
        out->tile_pixels = cfg->tile_pixels ? cfg->tile_pixels : 256u;
        out->tile_count = (uint32_t)num_gpus; out->tile_index = (uint32_t)gpu;
        if (cfg->idx_start % out->tile_pixels != 0) return PT_ERR_BAD_ARG;
        const uint32_t npix = cfg->idx_stop > cfg->idx_start ? cfg->idx_stop - cfg->idx_start : 0u;
        const uint32_t ntiles = (npix + out->tile_pixels - 1) / out->tile_pixels;
        if (ntiles <= (uint32_t)gpu) *has_work = 0;
        return PT_OK;
    }
    return PT_ERR_BAD_ARG;
}

extern "C" int ptgpu_multi_destroy(ptgpu_multi *m) {
    if (!m) return PT_ERR_BAD_ARG;
    for (size_t g = 0; g < m->comm.size(); ++g)
        if (m->comm[g] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm[g]);
    for (int g = 0; g < (int)m->ctx.size(); ++g) {
        (void)hipSetDevice(m->devices[g]);
        if (g < (int)m->frame.size() && m->frame[g]) (void)hipFree(m->frame[g]);
        if (g == 0 && m->d_rec) (void)hipFree(m->d_rec);
        if (g < (int)m->stream.size() && m->stream[g]) (void)hipStreamDestroy(m->stream[g]);
        if (m->ctx[g]) (void)ptgpu_destroy(m->ctx[g]);
    }
    delete m;
    return PT_OK;
}

extern "C" int ptgpu_multi_create(const int *devices, int num_gpus, ptgpu_multi **out) {
    if (!out || num_gpus <= 0 || num_gpus > 64) return PT_ERR_BAD_ARG;
    *out = nullptr;
    ptgpu_multi *m = new (std::nothrow) ptgpu_multi();
    if (!m) return PT_ERR_OOM;
    m->n = num_gpus;
    for (int g = 0; g < num_gpus; ++g) m->devices.push_back(devices ? devices[g] : g);
    // One context per GPU.  Tests only: PTGPU_MULTI_ALLOW_DUPLICATE_DEVICES=1 lets a device appear more than once, so that the
    // whole N > 1 plan — threads, zeroed frames, shares, the sum on GPU 0 — runs on a one-GPU box; RCCL cannot put two ranks
    // on one device, so the frames are then added on the host, in GPU order (x + 0 = x under the tile split).
    const char *dup_ok = getenv("PTGPU_MULTI_ALLOW_DUPLICATE_DEVICES");
    for (int g = 0; g < num_gpus; ++g)
        for (int h = 0; h < g; ++h)
            if (m->devices[g] == m->devices[h]) {
                if (!(dup_ok && atoi(dup_ok) > 0)) { delete m; return PT_ERR_BAD_ARG; }
                m->host_sum = true;
            }
    m->stats.assign(num_gpus, pt_stats());
    m->frame.assign(num_gpus, nullptr);
    m->frame_bytes.assign(num_gpus, 0);
    for (int g = 0; g < num_gpus; ++g) {
        ptgpu_ctx *c = nullptr;
        int rc = ptgpu_create(m->devices[g], &c);
        if (rc != PT_OK) { ptgpu_multi_destroy(m); return rc; }
        m->ctx.push_back(c);
        hipStream_t s = nullptr;
        if (hipSetDevice(m->devices[g]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            ptgpu_multi_destroy(m);
            return PT_ERR_HIP;
        }
        m->stream.push_back(s);
    }
    const char *force = getenv("PTGPU_MULTI_FORCE_RCCL"); // tests: run the collective on a single GPU too
    if (!m->host_sum && (num_gpus > 1 || (force && atoi(force) > 0))) {
        if (!g_rccl.load()) { ptgpu_multi_destroy(m); return PT_ERR_RCCL; }
        m->comm.assign(num_gpus, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), num_gpus, m->devices.data());
        if (r != ncclSuccess) { m->comm.clear(); ptgpu_multi_destroy(m); return PT_ERR_RCCL; }
    }
    *out = m;
    return PT_OK;
}

extern "C" int ptgpu_multi_num_gpus(const ptgpu_multi *m) { return m ? m->n : 0; }
extern "C" int ptgpu_multi_uses_rccl(const ptgpu_multi *m) { return (m && !m->comm.empty()) ? 1 : 0; }
extern "C" ptgpu_ctx *ptgpu_multi_ctx(ptgpu_multi *m, int gpu) { return (m && gpu >= 0 && gpu < m->n) ? m->ctx[gpu] : nullptr; }

extern "C" int ptgpu_multi_set_scene(ptgpu_multi *m, const pt_sphere *spheres, uint32_t n) {
    if (!m) return PT_ERR_BAD_ARG;
    for (int g = 0; g < m->n; ++g) { // the scene is KBs..MBs: replicated, not partitioned
        const int rc = ptgpu_set_scene(m->ctx[g], spheres, n);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

extern "C" int ptgpu_multi_set_camera(ptgpu_multi *m, const pt_camera *cam) {
    if (!m) return PT_ERR_BAD_ARG;
    for (int g = 0; g < m->n; ++g) {
        const int rc = ptgpu_set_camera(m->ctx[g], cam);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

// Every GPU traces its share into its own zeroed frame (one host thread per GPU, as the reference spawns one
// per pixel chunk), then the frames are summed on GPU 0.  Returns with the complete sums of
// [idx_start, idx_stop) in m->frame[0] and every stream idle.
static int render_sum_on_devices(ptgpu_multi *m, const pt_config *cfg, int split) {
    if (!m || !cfg) return PT_ERR_BAD_ARG;
    if (cfg->width == 0 || cfg->height == 0) return PT_ERR_BAD_ARG;
    const uint64_t np = (uint64_t)cfg->width * cfg->height;
    if (np > 0x7fffffffull / 3 || cfg->idx_start > cfg->idx_stop || cfg->idx_stop > np) return PT_ERR_BAD_ARG;
    const size_t bytes = (size_t)np * 3u * sizeof(float);
    std::vector<pt_config> share(m->n);
    std::vector<int> has(m->n, 0);
    for (int g = 0; g < m->n; ++g) {
        const int rc = ptgpu_multi_plan(cfg, split, m->n, g, &share[g], &has[g]);
        if (rc != PT_OK) return rc;
    }
    const size_t off = (size_t)cfg->idx_start * 3u, cnt = (size_t)(cfg->idx_stop - cfg->idx_start) * 3u;
    std::vector<int> rcs(m->n, PT_OK);
    auto work = [&](int g) {
        int rc = PT_OK;
        hipError_t e = hipSetDevice(m->devices[g]);
        if (e == hipSuccess && (m->frame_bytes[g] < bytes || !m->frame[g])) {
            if (m->frame[g]) { (void)hipFree(m->frame[g]); m->frame[g] = nullptr; m->frame_bytes[g] = 0; }
            e = hipMalloc((void **)&m->frame[g], bytes);
            if (e == hipSuccess) m->frame_bytes[g] = bytes;
        }
        if (e == hipSuccess && cnt) e = hipMemsetAsync(m->frame[g] + off, 0, cnt * sizeof(float), m->stream[g]);
        if (e != hipSuccess) { rcs[g] = (e == hipErrorOutOfMemory) ? PT_ERR_OOM : PT_ERR_HIP; return; }
        memset(&m->stats[g], 0, sizeof(pt_stats));
        if (has[g]) {
            rc = ptgpu_render_sum_device(m->ctx[g], &share[g], m->frame[g], (void *)m->stream[g]);
            (void)ptgpu_get_stats(m->ctx[g], &m->stats[g]);
        } else if (hipStreamSynchronize(m->stream[g]) != hipSuccess) rc = PT_ERR_HIP;
        rcs[g] = rc;
    };
    if (m->n == 1) work(0);
    else {
        // (nothing thrown here may cross the extern "C" boundary: a thread that cannot be started leaves its share to this one)
        std::vector<std::thread> threads;
        std::vector<int> started(m->n, 0);
        try { threads.reserve(m->n); } catch (...) { return PT_ERR_OOM; }
        for (int g = 0; g < m->n; ++g) {
            try { threads.emplace_back(work, g); started[g] = 1; } // main.zig:155-161
            catch (...) { started[g] = 0; }
        }
        for (int g = 0; g < m->n; ++g) if (!started[g]) work(g);
        for (auto &t : threads) t.join();                            // main.zig:163-165
    }
    int not_unit = PT_OK;
    for (int g = 0; g < m->n; ++g) {
        if (rcs[g] == PT_ERR_DIRECTION_NOT_UNIT) not_unit = rcs[g]; // the frame is still complete (include/ptgpu.h)
        else if (rcs[g] != PT_OK) return rcs[g];
    }
    m->reduce_ms = 0.0;
    if (m->host_sum && m->n > 1 && cnt) { // tests only: see ptgpu_multi_create
        std::vector<float> acc, part;
        try { acc.resize(cnt); part.resize(cnt); } catch (...) { return PT_ERR_OOM; }
        for (int g = 0; g < m->n; ++g) {
            if (hipSetDevice(m->devices[g]) != hipSuccess) return PT_ERR_HIP;
            if (hipMemcpy(g == 0 ? acc.data() : part.data(), m->frame[g] + off, cnt * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return PT_ERR_HIP;
            if (g > 0) for (size_t i = 0; i < cnt; ++i) acc[i] = acc[i] + part[i];
        }
        if (hipSetDevice(m->devices[0]) != hipSuccess) return PT_ERR_HIP;
        if (hipMemcpy(m->frame[0] + off, acc.data(), cnt * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return PT_ERR_HIP;
    }
    if (!m->comm.empty() && cnt) {
        // the path's one exchange step: fp32 sums -> GPU 0 (in place on the root)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipSetDevice(m->devices[0]);
        const bool timed = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        if (timed) (void)hipEventRecord(e0, m->stream[0]);
        ncclResult_t r = g_rccl.GroupStart();
        for (int g = 0; g < m->n && r == ncclSuccess; ++g)
            r = g_rccl.Reduce(m->frame[g] + off, m->frame[0] + off, cnt, ncclFloat, ncclSum, 0, m->comm[g], m->stream[g]);
        const ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        (void)hipSetDevice(m->devices[0]);
        if (timed) (void)hipEventRecord(e1, m->stream[0]);
        hipError_t e = hipSuccess;
        for (int g = 0; g < m->n; ++g) {
            (void)hipSetDevice(m->devices[g]);
            const hipError_t eg = hipStreamSynchronize(m->stream[g]);
            if (eg != hipSuccess) e = eg;
        }
        if (timed && e == hipSuccess && r == ncclSuccess) { float ms = 0.f; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) m->reduce_ms = ms; }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (r != ncclSuccess) { m->last_rccl = (int)r; return PT_ERR_RCCL; }
        if (e != hipSuccess) return PT_ERR_HIP;
    }
    return not_unit;
}

extern "C" int ptgpu_multi_render_sum_host(ptgpu_multi *m, const pt_config *cfg, int split, float *h_rgb_sum) {
    if (!h_rgb_sum) return PT_ERR_BAD_ARG;
    const int rc = render_sum_on_devices(m, cfg, split);
    if (rc != PT_OK && rc != PT_ERR_DIRECTION_NOT_UNIT) return rc;
    const size_t off = (size_t)cfg->idx_start * 3u, cnt = (size_t)(cfg->idx_stop - cfg->idx_start) * 3u;
    if (cnt) {
        if (hipSetDevice(m->devices[0]) != hipSuccess) return PT_ERR_HIP;
        if (hipMemcpy(h_rgb_sum + off, m->frame[0] + off, cnt * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return PT_ERR_HIP;
    }
    return rc;
}

extern "C" int ptgpu_multi_render_ppm(ptgpu_multi *m, const pt_config *cfg, int split, uint8_t *slice, size_t slice_len,
                                      uint32_t header_size, uint32_t px_size) {
    if (!m || !cfg || !slice) return PT_ERR_BAD_ARG;
    if (cfg->max_px_value != 255u || px_size != 12u) return PT_ERR_UNSUPPORTED;
    if (cfg->idx_start > cfg->idx_stop) return PT_ERR_BAD_ARG;
    if ((size_t)header_size + (size_t)cfg->idx_stop * px_size > slice_len) return PT_ERR_NO_SPACE;
    int rc = render_sum_on_devices(m, cfg, split);
    if (rc != PT_OK) return rc; // DirectionIsNotUnitVector aborts render() (renderers/ppm_image.zig:66 `try`)
    const uint32_t npix = cfg->idx_stop - cfg->idx_start;
    if (npix == 0) return PT_OK;
    if (hipSetDevice(m->devices[0]) != hipSuccess) return PT_ERR_HIP;
    if (m->rec_bytes < (size_t)npix * 12u || !m->d_rec) {
        if (m->d_rec) { (void)hipFree(m->d_rec); m->d_rec = nullptr; m->rec_bytes = 0; }
        if (hipMalloc((void **)&m->d_rec, (size_t)npix * 12u) != hipSuccess) return PT_ERR_OOM;
        m->rec_bytes = (size_t)npix * 12u;
    }
    pt_config whole = *cfg; // the resolve runs once, on the combined sums (renderers/ppm_image.zig:70-82)
    whole.tile_pixels = 0; whole.tile_count = 1; whole.tile_index = 0;
    rc = ptgpu_resolve_device(m->ctx[0], &whole, m->frame[0], m->d_rec, nullptr, (void *)m->stream[0]);
    if (rc != PT_OK && rc != PT_ERR_VALUE_EXCEEDS_MAX) return rc;
    if (hipMemcpy(slice + header_size + (size_t)cfg->idx_start * px_size, m->d_rec, (size_t)npix * 12u, hipMemcpyDeviceToHost) != hipSuccess)
        return PT_ERR_HIP;
    return rc;
}

extern "C" int ptgpu_multi_get_stats(const ptgpu_multi *m, int gpu, pt_stats *out) {
    if (!m || !out || gpu < -1 || gpu >= m->n) return PT_ERR_BAD_ARG;
    if (gpu >= 0) { *out = m->stats[gpu]; return PT_OK; }
    pt_stats t;
    memset(&t, 0, sizeof t);
    for (int g = 0; g < m->n; ++g) { // totals; times = the slowest GPU (they run side by side)
        const pt_stats &s = m->stats[g];
        t.paths += s.paths; t.rays += s.rays; t.sphere_tests += s.sphere_tests; t.candidates += s.candidates;
        t.not_unit += s.not_unit; t.wave_steps += s.wave_steps; t.flush_iters += s.flush_iters; t.segments += s.segments; t.primary_rays += s.primary_rays; t.list2_rays += s.list2_rays; t.box_tests += s.box_tests; t.primary_candidates += s.primary_candidates;
        if (s.primary_ms > t.primary_ms) t.primary_ms = s.primary_ms;
        if (s.lists_ms > t.lists_ms) t.lists_ms = s.lists_ms;
        if (s.megakernel_ms > t.megakernel_ms) t.megakernel_ms = s.megakernel_ms;
        if (s.kernel_ms > t.kernel_ms) t.kernel_ms = s.kernel_ms;
        if (s.total_ms > t.total_ms) t.total_ms = s.total_ms;
    }
    t.total_ms += m->reduce_ms;
    *out = t;
    return PT_OK;
}

extern "C" double ptgpu_multi_reduce_ms(const ptgpu_multi *m) { return m ? m->reduce_ms : 0.0; }
extern "C" int ptgpu_multi_last_rccl_error(const ptgpu_multi *m) { return m ? m->last_rccl : 0; }

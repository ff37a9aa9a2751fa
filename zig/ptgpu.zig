//! ptgpu.zig — Zig 0.9.1 binding of libptgpu (include/ptgpu.h) for jackdbd/path-tracer.
//!
//! Drop this file into the reference's `src/` and link libptgpu from build.zig (see
//! zig/build_snippet.zig).  It keeps the reference's renderer call shape: `render(cfg)` takes the
//! same `Config` as renderers/ppm_image.zig:16-26 and fills the same pixel records of `cfg.slice`,
//! but traces the whole `[istart, istop)` range on the GPU in one call, so main.zig:142-178 (the
//! thread-spawn block) becomes a single `try ptgpu.render(cfg, seed)`.
//!
//! UNTESTED: no Zig toolchain exists in the image this was written in.  The same C ABI is
//! exercised by the C++ host (path_tracer_amd/host/) and the Python ctypes binding.

const std = @import("std");

const Camera = @import("camera.zig").Camera;
const Image = @import("render_targets/ppm_image.zig").Image;
const Material = @import("material.zig").Material;
const RendererConfig = @import("renderers/ppm_image.zig").Config;
const Scene = @import("scene.zig").Scene;
const Vec3f = @import("vec3.zig").Vec3f;

// ---- extern-struct mirrors of include/ptgpu.h (plain Zig structs and tagged unions have no
// ---- defined layout and cannot cross the ABI) ------------------------------------------------
pub const PtSphere = extern struct {
    cx: f32,
    cy: f32,
    cz: f32,
    radius: f32,
    kind: u32, // 0 lambertian, 1 metal, 2 dielectric
    ar: f32,
    ag: f32,
    ab: f32,
    param: f32, // metal: fuzziness; dielectric: refraction_index
};

pub const PtCamera = extern struct {
    lookfrom: [3]f32,
    u: [3]f32,
    v: [3]f32,
    lower_left_corner: [3]f32,
    horizontal: [3]f32,
    vertical: [3]f32,
    lens_radius: f32,
};

pub const PtConfig = extern struct {
    width: u32,
    height: u32,
    subpixels: u32,
    rays_per_subsample: u32,
    rebounds: u32,
    mode: u32, // 0 stochastic, 1 deterministic
    seed: u64,
    t_min: f32,
    t_max: f32,
    blend_start: [3]f32,
    blend_stop: [3]f32,
    idx_start: u32,
    idx_stop: u32,
    sample_start: u32,
    sample_stop: u32,
    chunk: u32,
    max_px_value: u32,
    tile_pixels: u32,
    tile_count: u32,
    tile_index: u32,
    illumination: u32, // 0 radiance, 1 colorNormal, 2 colorAlbedo (illumination.zig)
    sampler: u32, // 0 the reference's rejection loops (camera.zig:113-120, material.zig:27-34), 1 the direct equivalents (not in the reference)
    reserved_: u32, // 0
};

pub const PtCtx = opaque {};

pub extern fn ptgpu_create(device: c_int, out: *?*PtCtx) c_int;
pub extern fn ptgpu_destroy(ctx: *PtCtx) c_int;
pub extern fn ptgpu_set_scene(ctx: *PtCtx, spheres: [*]const PtSphere, n: u32) c_int;
pub extern fn ptgpu_set_camera(ctx: *PtCtx, cam: *const PtCamera) c_int;
pub extern fn ptgpu_render_ppm(ctx: *PtCtx, cfg: *const PtConfig, slice: [*]u8, slice_len: usize, header_size: u32, px_size: u32) c_int;
pub extern fn ptgpu_render_sum_host(ctx: *PtCtx, cfg: *const PtConfig, rgb_sum: [*]f32) c_int;
// ---- all GPUs of the node (replaces the thread block of main.zig:142-165) -----------------------
pub const PtMulti = opaque {};
pub const PT_SPLIT_TILES: c_int = 0;
pub const PT_SPLIT_SAMPLES: c_int = 1;
pub extern fn ptgpu_multi_create(devices: ?[*]const c_int, num_gpus: c_int, out: *?*PtMulti) c_int;
pub extern fn ptgpu_multi_destroy(m: *PtMulti) c_int;
pub extern fn ptgpu_multi_set_scene(m: *PtMulti, spheres: [*]const PtSphere, n: u32) c_int;
pub extern fn ptgpu_multi_set_camera(m: *PtMulti, cam: *const PtCamera) c_int;
pub extern fn ptgpu_multi_render_ppm(m: *PtMulti, cfg: *const PtConfig, split: c_int, slice: [*]u8, slice_len: usize, header_size: u32, px_size: u32) c_int;
pub extern fn ptgpu_multi_render_sum_host(m: *PtMulti, cfg: *const PtConfig, split: c_int, rgb_sum: [*]f32) c_int;

// no counterpart in the reference: the same render without any device allocation (static 128 MiB scratch per GPU)
pub extern fn ptgpu_render_sum_device_noalloc(ctx: *PtCtx, cfg: *const PtConfig, d_rgb_sum: [*]f32, stream: ?*anyopaque) c_int;
pub extern fn ptgpu_noalloc_scratch_bytes() u64;
// device-memory limits of a context (INTEGRATION.md 4c); PT_LIMIT_PRIMARY_BYTES = 0 switches the primary pass off
pub const PT_LIMIT_PARTIAL_BYTES: c_int = 0;
pub const PT_LIMIT_PRIMARY_BYTES: c_int = 1;
pub const PT_LIMIT_PRIMARY_BOUNCES: c_int = 2; // 1: camera rays only, 2 (default): also the rays scattered off each pixel's main diffuse sphere
pub const PT_LIMIT_TREE_MIN_SPHERES: c_int = 3; // scenes of at least this many spheres (default 256; 0: never) are traced through a sphere tree; set before ptgpu_set_scene
pub extern fn ptgpu_set_limit(ctx: *PtCtx, which: c_int, value: u64) c_int;
pub extern fn ptgpu_trim(ctx: *PtCtx) c_int; // frees every work buffer the renders have grown (records, partials, stack); the scene stays
pub extern fn ptgpu_work_bytes(ctx: *const PtCtx) u64;
pub const PT_SAMPLER_REJECTION: u32 = 0; // the reference's rejection loops (camera.zig:113-120, material.zig:27-34)
pub const PT_SAMPLER_DIRECT: u32 = 1; // the same distributions without a loop (not in the reference)
pub extern fn ptgpu_tree_info(ctx: *const PtCtx, nodes: ?*u32, depth: ?*u32) c_int;
pub extern fn ptgpu_debug_tree_check(spheres: [*]const PtSphere, n: u32, median_only: c_int, nodes: ?*u32, depth: ?*u32, leaves: ?*u32, violations: *u64) c_int; // host-only builder self-check
pub extern fn ptgpu_strerror(status: c_int) [*:0]const u8;
pub extern fn ptgpu_abi_version() u32;

pub const Error = error{ NoGpu, BadArgument, HipFailure, OutOfMemory, DirectionIsNotUnitVector, ValueExceedsMaxValue, NoSpaceLeft, Unsupported, ScratchLimit, RcclFailure };

fn check(rc: c_int) Error!void {
    return switch (rc) {
        0 => {},
        -2 => error.NoGpu,
        -3 => error.HipFailure,
        -4 => error.OutOfMemory,
        -5 => error.DirectionIsNotUnitVector, // illumination.zig:29-31
        -6 => error.ValueExceedsMaxValue, // utils.zig:63-65
        -8 => error.NoSpaceLeft,
        -9 => error.Unsupported,
        -10 => error.ScratchLimit,
        -11 => error.RcclFailure,
        else => error.BadArgument,
    };
}

fn arr(v: Vec3f) [3]f32 {
    return [3]f32{ v.x, v.y, v.z };
}

/// Flattens Sphere + Material (sphere.zig:36-39, material.zig:174-177) to the POD table.
pub fn flattenScene(allocator: std.mem.Allocator, scene: *const Scene) ![]PtSphere {
    var flat = try allocator.alloc(PtSphere, scene.spheres.items.len);
    for (scene.spheres.items) |s, i| {
        var p = PtSphere{ .cx = s.center.x, .cy = s.center.y, .cz = s.center.z, .radius = s.radius, .kind = 0, .ar = 1.0, .ag = 1.0, .ab = 1.0, .param = 0.0 };
        switch (s.material) {
            Material._lambertian => |m| {
                p.kind = 0;
                p.ar = m.albedo.x;
                p.ag = m.albedo.y;
                p.ab = m.albedo.z;
            },
            Material._metal => |m| {
                p.kind = 1;
                p.ar = m.albedo.x;
                p.ag = m.albedo.y;
                p.ab = m.albedo.z;
                p.param = m.fuzziness;
            },
            Material._dielectric => |m| {
                p.kind = 2;
                p.param = m.refraction_index;
            },
        }
        flat[i] = p;
    }
    return flat;
}

/// The 19 floats Camera.castRay reads (camera.zig:99-106).
pub fn flattenCamera(c: *const Camera) PtCamera {
    return PtCamera{
        .lookfrom = arr(c.lookfrom),
        .u = arr(c.u),
        .v = arr(c.v),
        .lower_left_corner = arr(c.lower_left_corner),
        .horizontal = arr(c.horizontal),
        .vertical = arr(c.vertical),
        .lens_radius = c.lens_radius,
    };
}

fn toPtConfig(cfg: RendererConfig, seed: u64) PtConfig {
    const img = cfg.img;
    const rt = cfg.ray_tracer_config;
    return PtConfig{
        .width = @intCast(u32, img.width),
        .height = @intCast(u32, img.height),
        .subpixels = rt.subpixels,
        .rays_per_subsample = rt.rays_per_subsample,
        .rebounds = rt.rebounds,
        .mode = 0,
        .seed = seed,
        .t_min = rt.t_min,
        .t_max = rt.t_max,
        .blend_start = arr(img.blend_start),
        .blend_stop = arr(img.blend_stop),
        .idx_start = @intCast(u32, cfg.istart),
        .idx_stop = @intCast(u32, cfg.istop),
        .sample_start = 0,
        .sample_stop = 0, // 0,0 = all subpixels * rays_per_subsample samples
        .chunk = 16,
        .max_px_value = @intCast(u32, img.max_px_value),
        .tile_pixels = 0,
        .tile_count = 1,
        .tile_index = 0,
        .illumination = 0,
        .sampler = 0,
        .reserved_ = 0,
    };
}

/// GPU replacement of renderers/ppm_image.zig:30 `render(cfg)`: same Config, same bytes
/// written into `cfg.slice` at `header_size + idx * px_size` for idx in [istart, istop).
/// `seed` keys the Philox streams (cfg.r, the sequential std.rand stream, is not used).
pub fn render(cfg: RendererConfig, seed: u64) !void {
    var gpu: ?*PtCtx = null;
    try check(ptgpu_create(0, &gpu));
    defer _ = ptgpu_destroy(gpu.?);

    const flat = try flattenScene(cfg.allocator.*, cfg.scene);
    defer cfg.allocator.free(flat);
    try check(ptgpu_set_scene(gpu.?, flat.ptr, @intCast(u32, flat.len)));
    const cam = flattenCamera(cfg.camera);
    try check(ptgpu_set_camera(gpu.?, &cam));

    const pc = toPtConfig(cfg, seed);
    try check(ptgpu_render_ppm(gpu.?, &pc, cfg.slice.ptr, cfg.slice.len, @intCast(u32, cfg.img.header_size), @intCast(u32, cfg.img.px_size)));
}

/// The multi-GPU counterpart of the `use_multithreaded_version` branch (main.zig:142-165): `num_gpus` GPUs
/// share [istart, istop) by interleaved tiles (PT_SPLIT_TILES) or by sample ranges (PT_SPLIT_SAMPLES); the
/// library spawns one host thread per GPU, reduces the fp32 sums over RCCL and resolves once.
pub fn renderMultiGpu(cfg: RendererConfig, seed: u64, num_gpus: c_int, split: c_int) !void {
    var multi: ?*PtMulti = null;
    try check(ptgpu_multi_create(null, num_gpus, &multi));
    defer _ = ptgpu_multi_destroy(multi.?);

    const flat = try flattenScene(cfg.allocator.*, cfg.scene);
    defer cfg.allocator.free(flat);
    try check(ptgpu_multi_set_scene(multi.?, flat.ptr, @intCast(u32, flat.len)));
    const cam = flattenCamera(cfg.camera);
    try check(ptgpu_multi_set_camera(multi.?, &cam));

    const pc = toPtConfig(cfg, seed);
    try check(ptgpu_multi_render_ppm(multi.?, &pc, split, cfg.slice.ptr, cfg.slice.len, @intCast(u32, cfg.img.header_size), @intCast(u32, cfg.img.px_size)));
}

test "extern struct sizes match include/ptgpu.h" {
    try std.testing.expectEqual(@as(usize, 36), @sizeOf(PtSphere));
    try std.testing.expectEqual(@as(usize, 76), @sizeOf(PtCamera));
    try std.testing.expectEqual(@as(usize, 112), @sizeOf(PtConfig)); // ABI 3: sampler, reserved_
    try std.testing.expectEqual(@as(usize, 24), @offsetOf(PtConfig, "seed"));
    try std.testing.expectEqual(@as(usize, 88), @offsetOf(PtConfig, "tile_pixels"));
}

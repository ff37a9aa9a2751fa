// Lines to add to the reference's build.zig (Zig 0.9.1 std.build API) after `deps.addAllTo(exe);`
// so that src/ptgpu.zig's extern fns resolve.  PTGPU_ROOT = a checkout of this repository with
// path_tracer_amd/csrc/build.sh already run (libptgpu.so needs libamdhip64 from ROCm at run time).
//
//     exe.linkLibC();
//     exe.addIncludeDir(PTGPU_ROOT ++ "/include");
//     exe.addLibPath(PTGPU_ROOT ++ "/path_tracer_amd");
//     exe.linkSystemLibrary("ptgpu");
//     exe.addRPath(PTGPU_ROOT ++ "/path_tracer_amd");
//
// and for the binding's own layout test:
//
//     var ptgpu_tests = b.addTest("src/ptgpu.zig");
//     ptgpu_tests.setBuildMode(mode);
//     test_step.dependOn(&ptgpu_tests.step);
//
// main.zig:142-178 (spawn T threads over ceil(P/T) pixel chunks, join) then becomes:
//
//     try ptgpu.render(.{ .allocator = &allocator, .camera = &camera, .img = &img, .istart = 0,
//         .istop = img.num_pixels, .r = &r, .ray_tracer_config = &ray_tracer_cfg, .slice = slice,
//         .scene = &scene }, seed);

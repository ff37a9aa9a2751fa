"""ptgpu_render_sum_device_noalloc — the entry point for callers that cannot allocate device memory: same result
as ptgpu_render_sum_device, bit for bit, with the kernels' working memory in the library's static 128 MiB scratch.
Two calls in a row, the second with a smaller input (the scratch keeps the first call's contents)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_noalloc_matches_allocating_entry_point_twice(gpu):
    from path_tracer_amd import _lib, host
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    assert _lib.load().ptgpu_noalloc_scratch_bytes() == 128 << 20
    side = torch.cuda.Stream()
    # first call: 640x360, 64 samples in 32 chunks of 2: 84 MiB of partials beside the hit-id stack (scene 21 walks its sphere
    # tree: 6 workgroups per CU, 393 216 threads x depth 50 x 4 B = 75 MiB) -> the partials are carved from the remaining
    # 53 MiB segment by segment, the item queue restarting in between; second call: a smaller frame, fewer samples, other chunking
    segments = []
    for width, rps, chunk, depth in ((640, 16, 2, 50), (200, 3, 5, 50), (200, 2, 0, 6)):
        img = host.Image(width)
        rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=rps, rebounds=depth)
        cfg = host.make_pt_config(img, rt, seed=456, chunk=chunk)
        n = img.num_pixels * 3
        want = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
        got = torch.full((n,), -2.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        gpu.render_sum_device(cfg, want.data_ptr(), side.cuda_stream)
        st_a = gpu.stats()
        gpu.render_sum_device_noalloc(cfg, got.data_ptr(), side.cuda_stream)
        st_b = gpu.stats()
        side.synchronize()
        # (the allocating entry point runs the primary pass, this one does not: the same paths and rays, other exact-test counts)
        assert (st_a.paths, st_a.rays) == (st_b.paths, st_b.rays) and st_a.primary_rays > 0 and st_b.primary_rays == 0
        assert st_a.paths == img.num_pixels * 4 * rps
        np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())
        segments.append(st_b.segments)
    assert segments[0] >= 2 and segments[-1] == 1, segments   # the segmented no-allocation path did run


def test_noalloc_rejects_what_does_not_fit(gpu):
    from path_tracer_amd import _lib, host
    from path_tracer_amd._lib import PtGpuError
    scene = host.Scene().setupScene(456, 18)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(18).pod)
    img = host.Image(64)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=1, rebounds=255)   # hit-id stack: 255 x 327 680 x 4 B = 319 MiB
    frame = torch.zeros(img.num_pixels * 3, dtype=torch.float32, device="cuda")
    with pytest.raises(PtGpuError) as e:
        gpu.render_sum_device_noalloc(host.make_pt_config(img, rt), frame.data_ptr(), 0)
    assert e.value.status == _lib.PT_ERR_SCRATCH_LIMIT
    gpu.render_sum_device(host.make_pt_config(img, rt), frame.data_ptr(), 0)     # the allocating entry point takes it

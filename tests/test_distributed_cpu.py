"""The multi-GPU host logic (path_tracer_amd/distributed.py) on CPU: world_size-2 `gloo` processes.

No GPU here, so the single-device render is a TEST DOUBLE — the oracle — injected as `render_into`;
what is under test is the product's split planning (sample ranges / interleaved tiles) and its one
collective (reduce(SUM) of the fp32 sum buffers to rank 0).  The product's own render_into
(GpuContext) is exercised by the -m gpu tests.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_render_into(spheres, cam):
    from oracle import oracle

    def fn(cfg, frame):
        ocfg = oracle.Config()
        C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
        ocam = oracle.Camera()
        C.memmove(C.byref(ocam), C.byref(cam), C.sizeof(ocam))
        rc, out, _ = oracle.render_sum(spheres, ocam, ocfg, nthreads=2)
        assert rc == 0
        frame.copy_(torch.from_numpy(out.reshape(-1)))

    return fn


def _worker(rank, world, port, split, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from path_tracer_amd import host
    from path_tracer_amd.distributed import render_sum_distributed
    img = host.Image(40)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=3, rebounds=50)
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21, img.aspect_ratio)
    frame = render_sum_distributed(_oracle_render_into(scene.spheres, cam.pod), img, rt, seed=9, split=split, chunk=4,
                                   tile_pixels=16, device="cpu")
    if rank == 0:
        q.put(frame.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def _single(split):
    sys.path.insert(0, ROOT)
    from path_tracer_amd import host
    from path_tracer_amd.distributed import render_sum_distributed
    img = host.Image(40)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=3, rebounds=50)
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21, img.aspect_ratio)
    return render_sum_distributed(_oracle_render_into(scene.spheres, cam.pod), img, rt, seed=9, split=split, chunk=4,
                                  tile_pixels=16, device="cpu").numpy()


@pytest.mark.parametrize("split,port", [("samples", 29611), ("tiles", 29612)])
def test_world2_matches_single(split, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, split, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _single(split)
    if split == "tiles":
        np.testing.assert_array_equal(got, ref)       # disjoint supports: x + 0 = x
    else:
        np.testing.assert_allclose(got, ref, rtol=2e-6)  # fp32 add order of the two halves
        assert np.abs(got).sum() > 0


def test_sample_ranges_partition():
    sys.path.insert(0, ROOT)
    from path_tracer_amd.distributed import sample_range
    for total in (1, 7, 600, 1024, 4096):
        for world in (1, 2, 3, 4, 8):
            spans = [sample_range(total, world, g) for g in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_plan_empty_sample_range_is_not_all_samples():
    sys.path.insert(0, ROOT)
    from path_tracer_amd import host
    from path_tracer_amd.distributed import plan
    img = host.Image(32)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=1)
    assert plan(img, rt, 0, world=4, rank=0) is None      # 1 sample over 4 ranks: ranks 0..2 get nothing
    cfg = plan(img, rt, 0, world=4, rank=3)
    assert (cfg.sample_start, cfg.sample_stop) == (0, 1)

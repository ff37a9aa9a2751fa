"""ptgpu_multi_plan (include/ptgpu.h) — the split of one frame over the GPUs of a node, as the one-process
ptgpu_multi_* path and the one-process-per-GPU path (path_tracer_amd/distributed.py) both use it.  Pure host
arithmetic inside libptgpu, so it runs here without a GPU; the oracle stands in for the single-GPU render to show that
the shares partition the work exactly (tiles: disjoint union, bit-identical; samples: every sample once)."""
import ctypes as C

import numpy as np
import pytest


def _cfg(host, img, rt, **kw):
    return host.make_pt_config(img, rt, seed=9, **kw)


def _oracle_sum(orc, spheres, cam, cfg):
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    rc, out, st = orc.render_sum(spheres, cam, ocfg, nthreads=4)
    assert rc == 0
    return out, st


@pytest.mark.parametrize("gpus", [1, 2, 3, 8])
def test_sample_split_covers_every_sample_once(gpus):
    from path_tracer_amd import host
    from path_tracer_amd.gpu import multi_plan
    img = host.Image(40)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=5)          # S = 20
    for a, b in ((0, 0), (3, 17), (6, 7)):
        lo = a if (a, b) != (0, 0) else 0
        hi = b if (a, b) != (0, 0) else 20
        at = lo
        for g in range(gpus):
            share = multi_plan(_cfg(host, img, rt, sample_start=a, sample_stop=b, chunk=4), "samples", gpus, g)
            if share is None:
                assert (lo + g * (hi - lo) // gpus) == (lo + (g + 1) * (hi - lo) // gpus)   # really empty
                continue
            assert share.sample_start == at and share.sample_stop > at and share.tile_pixels == 0 and share.chunk == 4
            at = share.sample_stop
        assert at == hi


@pytest.mark.parametrize("gpus,tile", [(2, 16), (3, 64), (8, 0), (5, 4096)])
def test_tile_split_is_an_exact_partition(orc, gpus, tile):
    from path_tracer_amd import host
    from path_tracer_amd.gpu import multi_plan
    img = host.Image(40)                                                   # 880 pixels
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=2, rebounds=8)
    sp, _ = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    whole, st = _oracle_sum(orc, sp, cam, _cfg(host, img, rt))
    acc = np.zeros_like(whole)
    paths = 0
    for g in range(gpus):
        share = multi_plan(_cfg(host, img, rt, tile_pixels=tile), "tiles", gpus, g)
        ntiles = -(-880 // (tile or 256))
        if share is None:
            assert g >= ntiles
            continue
        assert (share.tile_pixels, share.tile_count, share.tile_index) == (tile or 256, gpus, g)
        part, pst = _oracle_sum(orc, sp, cam, share)
        assert not ((acc != 0) & (part != 0)).any()
        acc += part
        paths += pst.paths
    assert paths == st.paths
    np.testing.assert_array_equal(acc, whole)


def test_sample_split_sums_to_the_whole(orc):
    from path_tracer_amd import host
    from path_tracer_amd.gpu import multi_plan
    img = host.Image(32)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=4, rebounds=8)
    sp, _ = orc.setup_scene(18)
    cam = orc.camera_preset(18)
    whole, st = _oracle_sum(orc, sp, cam, _cfg(host, img, rt, chunk=4))
    acc = np.zeros_like(whole)
    rays = 0
    for g in range(4):
        part, pst = _oracle_sum(orc, sp, cam, multi_plan(_cfg(host, img, rt, chunk=4), "samples", 4, g))
        acc += part
        rays += pst.rays
    assert rays == st.rays
    np.testing.assert_array_equal(acc, whole)   # 4 ranges of one chunk each, added in range order = the chunked sum


def test_plan_rejects_nested_and_bad_arguments():
    from path_tracer_amd import _lib, host
    from path_tracer_amd._lib import PtGpuError
    from path_tracer_amd.gpu import multi_plan
    img = host.Image(40)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=2)
    with pytest.raises(PtGpuError) as e:     # already one owner's share
        multi_plan(_cfg(host, img, rt, tile_pixels=16, tile_count=2, tile_index=1), "tiles", 2, 0)
    assert e.value.status == _lib.PT_ERR_UNSUPPORTED
    for kw, split, n, g in ((dict(), "samples", 0, 0), (dict(), "samples", 2, 2), (dict(), 7, 2, 0),
                            (dict(sample_start=3, sample_stop=2), "samples", 2, 0), (dict(sample_stop=5), "samples", 2, 0),
                            (dict(istart=8, tile_pixels=16), "tiles", 2, 0)):
        with pytest.raises(PtGpuError) as e:
            multi_plan(_cfg(host, img, rt, **kw), split, n, g)
        assert e.value.status == _lib.PT_ERR_BAD_ARG

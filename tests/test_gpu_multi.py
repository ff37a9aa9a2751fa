"""ptgpu_multi_* on the one GPU of the test box: the whole multi-GPU code path of the C ABI — per-GPU thread,
context, stream and frame, the plan, and (with PTGPU_MULTI_FORCE_RCCL=1) a real RCCL communicator and
ncclReduce — must reproduce the single-GPU entry points bit for bit.  N > 1 is the same code with more threads;
its partition arithmetic is covered on CPU (tests/test_multi_plan_cpu.py, tests/test_distributed_cpu.py)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("force_rccl", [False, True])
def test_multi_one_gpu_matches_single_gpu(gpu, force_rccl, monkeypatch):
    from path_tracer_amd import host
    from path_tracer_amd.gpu import MultiGpu
    monkeypatch.setenv("PTGPU_MULTI_FORCE_RCCL", "1" if force_rccl else "0")
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(160)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=4, rebounds=50)
    with MultiGpu(1) as m:
        assert m.num_gpus == 1 and m.uses_rccl == force_rccl
        m.set_scene(scene.spheres)
        m.set_camera(cam.pod)
        for kw in (dict(chunk=4), dict(chunk=4, istart=160 * 10, istop=160 * 50 + 7), dict(chunk=0, sample_start=3, sample_stop=11)):
            cfg = host.make_pt_config(img, rt, seed=5, **kw)
            want = gpu.render_sum(cfg)
            for split in ("tiles", "samples"):
                if split == "tiles" and kw.get("istart", 0) % 256:
                    continue
                got = m.render_sum(cfg, split)
                np.testing.assert_array_equal(got, want)
                st, one = m.stats(), gpu.stats()
                assert (st.paths, st.rays) == (one.paths, one.rays) and m.stats(0).paths == st.paths
        cfg = host.make_pt_config(img, rt, seed=5, chunk=4)
        a, b = np.zeros(img.size, dtype=np.uint8), np.zeros(img.size, dtype=np.uint8)
        gpu.render_ppm(cfg, a, img.header_size, img.px_size)
        m.render_ppm(cfg, "samples", b, img.header_size, img.px_size)
        np.testing.assert_array_equal(a, b)
        if force_rccl:
            assert m.reduce_ms() > 0.0


@pytest.mark.parametrize("n", [2, 3])
def test_multi_threaded_plan_with_n_ranks_on_one_gpu(gpu, n, monkeypatch):
    """N > 1 on the one-GPU box: PTGPU_MULTI_ALLOW_DUPLICATE_DEVICES=1 lets device 0 stand in for every rank, so the
    concurrent worker threads, one context / stream / zeroed frame per rank, the shares of ptgpu_multi_plan and the sum on
    rank 0 all run (the sum on the host instead of ncclReduce: RCCL cannot put two ranks on one device).  Tile split:
    the one-GPU frame bit for bit.  Sample split: the ranks' partial frames added in rank order."""
    from path_tracer_amd import host
    from path_tracer_amd.gpu import MultiGpu, multi_plan
    monkeypatch.setenv("PTGPU_MULTI_ALLOW_DUPLICATE_DEVICES", "1")
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(192)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=4, rebounds=50)
    with MultiGpu(n, devices=[0] * n) as m:
        assert m.num_gpus == n and not m.uses_rccl
        m.set_scene(scene.spheres)
        m.set_camera(cam.pod)
        for kw in (dict(chunk=4), dict(chunk=2, istart=256 * 3, istop=192 * 90 + 11)):
            cfg = host.make_pt_config(img, rt, seed=7, **kw)
            want = gpu.render_sum(cfg)
            one = gpu.stats()
            got = m.render_sum(cfg, "tiles")
            np.testing.assert_array_equal(got, want)
            st = m.stats()
            assert (st.paths, st.rays) == (one.paths, one.rays)
            assert sum(m.stats(g).paths for g in range(n)) == st.paths and all(m.stats(g).paths > 0 for g in range(n))
            got = m.render_sum(cfg, "samples")
            acc = None
            for g in range(n):
                share = multi_plan(cfg, "samples", n, g)
                part = gpu.render_sum(share)
                acc = part if acc is None else acc + part
            np.testing.assert_array_equal(got, acc)
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
        cfg = host.make_pt_config(img, rt, seed=7, chunk=4)
        a, b = np.zeros(img.size, dtype=np.uint8), np.zeros(img.size, dtype=np.uint8)
        gpu.render_ppm(cfg, a, img.header_size, img.px_size)
        m.render_ppm(cfg, "tiles", b, img.header_size, img.px_size)
        np.testing.assert_array_equal(a, b)


def test_multi_rejects_bad_devices(monkeypatch):
    from path_tracer_amd import _lib
    monkeypatch.delenv("PTGPU_MULTI_ALLOW_DUPLICATE_DEVICES", raising=False)
    from path_tracer_amd._lib import PtGpuError
    from path_tracer_amd.gpu import MultiGpu
    with pytest.raises(PtGpuError) as e:
        MultiGpu(2, devices=[0, 0])                      # one context per GPU
    assert e.value.status == _lib.PT_ERR_BAD_ARG
    with pytest.raises(PtGpuError) as e:
        MultiGpu(1, devices=[99])
    assert e.value.status == _lib.PT_ERR_NO_DEVICE


def test_pt_render_cli_gpus_flag(tmp_path):
    """pt_render --gpus 1 (renderMultiGpu, main.zig's "multi-threaded version") writes the same file as the single-GPU path."""
    exe = os.path.join(ROOT, "path_tracer_amd", "pt_render")
    args = ["--scene", "21", "--width", "96", "--spp", "3", "--depth", "12", "--seed", "456", "--chunk", "4"]
    one, multi_t, multi_s = (str(tmp_path / n) for n in ("one.ppm", "mt.ppm", "ms.ppm"))
    env = dict(os.environ, PTGPU_MULTI_FORCE_RCCL="1")
    subprocess.run([exe] + args + ["--out", one], check=True, timeout=120, capture_output=True)
    subprocess.run([exe] + args + ["--gpus", "1", "--split", "tiles", "--out", multi_t], check=True, timeout=300, capture_output=True, env=env)
    r = subprocess.run([exe] + args + ["--gpus", "1", "--split", "samples", "--out", multi_s], check=True, timeout=300, capture_output=True, env=env)
    assert b"RCCL reduce" in r.stderr
    ref = open(one, "rb").read()
    assert ref.startswith(b"P3\n96 54\n255\n") and open(multi_t, "rb").read() == ref and open(multi_s, "rb").read() == ref


@pytest.mark.parametrize("split", ["tiles", "samples"])
def test_bench_two_ranks_on_one_gpu_assemble_the_one_gpu_frame(gpu, split, tmp_path):
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one process per rank), rehearsed with both
    ranks on the one GPU over gloo: rank 0's frame after the collective is the one-GPU frame — bit for bit under the tile
    split, the two sample halves added under the sample split — and the JSON line says what ran."""
    import json
    import socket
    import sys
    from path_tracer_amd import host
    from path_tracer_amd.gpu import multi_plan
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--width", "480",
           "--spp", "4", "--split", split, "--rehearse-gloo", "--dump-frame", out]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, check=True, timeout=280, capture_output=True, env=env, cwd=ROOT)
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "rehearsal" in line
    assert ("tile split x2" if split == "tiles" else "sample-range split x2") in line["config"]["parallelism"]
    assert "gloo" in line["config"]["parallelism"] and "RCCL" not in line["config"]["parallelism"]
    assert "480x270" in line["metric"] and "16 spp" in line["metric"] and line["metric"] != "paths/sec (scene 21, 1920x1080, 1024 spp, depth 50)"
    assert line["config"]["global_paths_per_pixel"] == 16 and "cpu_baseline" not in line
    got = np.load(out)
    scene = host.Scene().setupScene(456, 21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(21).pod)
    img = host.Image(480)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=4, rebounds=50)
    chunk = line["config"]["chunk"]
    whole = host.make_pt_config(img, rt, seed=456, chunk=chunk)
    if split == "tiles":
        want = gpu.render_sum(whole)
        assert abs(line["value"] * line["ms_per_step"] * 1e-3 - img.num_pixels * 16) < 1.0   # all ranks' paths over the slowest rank's time
    else:
        want = gpu.render_sum(multi_plan(whole, "samples", 2, 0)) + gpu.render_sum(multi_plan(whole, "samples", 2, 1))
    np.testing.assert_array_equal(got, want)

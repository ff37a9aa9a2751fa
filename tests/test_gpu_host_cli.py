"""The C++ host (path_tracer_amd/host/, the compiled mirror of main.zig + renderers/ppm_image.zig) on the
GPU: the `pt_render` CLI writes the same .ppm file, byte for byte, as header + oracle records."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "path_tracer_amd", "pt_render")


def _oracle_file(orc, n_scene, width, spp, depth, seed, deterministic, sampler=0):
    spheres, _ = orc.setup_scene(n_scene, seed)
    cam = orc.camera_preset(n_scene)
    img = orc.image_new(width)
    cfg = orc.make_config(img, subpixels=4, rays_per_subsample=spp, rebounds=depth, seed=seed, mode=1 if deterministic else 0, sampler=sampler)
    rc, ref_sum, _ = orc.render_sum(spheres, cam, cfg, nthreads=8)
    rc2, ppm = orc.resolve_ppm(cfg, img, ref_sum)
    assert rc == 0 and rc2 == 0
    header = ("P3\n%d %d\n255\n" % (img.width, img.height)).encode()   # render_targets/ppm_image.zig:84-88
    assert len(header) == img.header_size
    return header + bytes(ppm[img.header_size:])


@pytest.mark.parametrize("n_scene,width,spp,depth,seed,det", [(21, 64, 3, 6, 456, False), (18, 48, 2, 50, 42, False),
                                                               (20, 40, 1, 6, 7, True)])
def test_pt_render_cli_writes_the_oracle_file(orc, tmp_path, n_scene, width, spp, depth, seed, det):
    assert os.path.exists(EXE), "build path_tracer_amd/host/build.sh first"
    out = tmp_path / "frame.ppm"
    cmd = [EXE, "--scene", str(n_scene), "--width", str(width), "--spp", str(spp), "--depth", str(depth), "--seed", str(seed),
           "--out", str(out)] + (["--deterministic"] if det else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "paths/s" in r.stderr and "Program took" in r.stderr          # main.zig:191's log line
    assert out.read_bytes() == _oracle_file(orc, n_scene, width, spp, depth, seed, det)


def test_pt_render_cli_direct_sampler(orc, tmp_path):
    """--sampler direct: pt_config.sampler = PT_SAMPLER_DIRECT through the C++ host; the file is the oracle's in that mode, and differs
    from the default mode's."""
    out, ref = tmp_path / "direct.ppm", tmp_path / "rejection.ppm"
    base = [EXE, "--scene", "21", "--width", "64", "--spp", "3", "--depth", "50", "--seed", "456"]
    r = subprocess.run(base + ["--sampler", "direct", "--out", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == _oracle_file(orc, 21, 64, 3, 50, 456, False, sampler=orc.SAMPLER_DIRECT)
    r = subprocess.run(base + ["--sampler", "rejection", "--out", str(ref)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and ref.read_bytes() == _oracle_file(orc, 21, 64, 3, 50, 456, False) and ref.read_bytes() != out.read_bytes()
    assert subprocess.run(base + ["--sampler", "sobol"], capture_output=True, timeout=60).returncode == 2


def test_pt_render_cli_default_file_name_and_bad_scene(tmp_path):
    r = subprocess.run([EXE, "--scene", "17"], capture_output=True, text=True, timeout=60, cwd=tmp_path)
    assert r.returncode == 1 and "SceneNotAvailable" in r.stderr          # scene.zig:92 error
    os.makedirs(tmp_path / "images", exist_ok=True)
    r = subprocess.run([EXE, "--scene", "19", "--width", "32", "--spp", "1", "--seed", "5"], capture_output=True, text=True,
                       timeout=60, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    from path_tracer_amd import host
    name = host.filepath(19, host.RayTracerConfig(rays_per_subsample=1), host.Image(32), 5)   # utils.zig:113-115
    assert (tmp_path / name).exists(), (name, r.stderr)


@pytest.mark.parametrize("aov,illumination", [("normal", 1), ("albedo", 2)])
def test_pt_render_cli_aov(orc, tmp_path, aov, illumination):
    """--aov: the reference's commented-out colour functions (renderers/ppm_image.zig:64-65) through the C++ host."""
    out = tmp_path / "aov.ppm"
    r = subprocess.run([EXE, "--scene", "21", "--width", "64", "--spp", "2", "--seed", "456", "--aov", aov, "--out", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    spheres, _ = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    img = orc.image_new(64)
    cfg = orc.make_config(img, subpixels=4, rays_per_subsample=2, rebounds=6, seed=456, illumination=illumination)
    rc, ref_sum, _ = orc.render_sum(spheres, cam, cfg, nthreads=4)
    rc2, ppm = orc.resolve_ppm(cfg, img, ref_sum)
    assert rc == 0 and rc2 == 0
    assert out.read_bytes()[img.header_size:] == bytes(ppm[img.header_size:])

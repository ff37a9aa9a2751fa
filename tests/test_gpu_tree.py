"""The sphere tree of large scenes (scene_is_hit_tree) against the scan over all spheres.

scene.zig:96-110 tests every sphere for every ray.  With PT_LIMIT_TREE_MIN_SPHERES = 0 the megakernel does exactly that
(conservative LDS filter + exact pass); with a tree it walks boxes and gives the exact Sphere.is_hit only to the spheres
of the leaves it reaches.  Both must return the same closest hit for every ray — index and t, bit for bit — and so the
same frames, path and ray counts.  PT_FN_SCENE_HIT is the plain ascending scan with strict '<' on the GPU (pinned
against the oracle's scene_is_hit and the committed vectors in tests/test_golden.py and tests/test_analytic_kats.py); PT_FN_SCENE_HIT_TREE the tree walk on the same rays.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _set_scene(gpu, spheres, min_spheres):
    from path_tracer_amd import _lib
    gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, min_spheres)
    try:
        gpu.set_scene(spheres)
    finally:
        gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, _lib.PT_TREE_MIN_SPHERES_DEFAULT)


def _rays(rng, spheres, n):
    """origins on sphere surfaces, near them, far away and at the centres; directions random, axis-aligned (zero components),
    aimed at sphere centres and grazing (aimed at a point one radius off a centre); some not normalised"""
    c = np.stack([spheres["cx"], spheres["cy"], spheres["cz"]], axis=1).astype(np.float64)
    r = np.abs(spheres["radius"].astype(np.float64))
    pick = rng.integers(0, len(spheres), n)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    kind = rng.integers(0, 5, n)
    o = np.where((kind == 0)[:, None], c[pick] + u * r[pick, None],                       # on a surface
        np.where((kind == 1)[:, None], c[pick] + u * r[pick, None] * rng.uniform(1.0, 3.0, (n, 1)),
        np.where((kind == 2)[:, None], rng.normal(0, 30, (n, 3)),
        np.where((kind == 3)[:, None], c[pick], rng.normal(0, 3, (n, 3))))))
    target = rng.integers(0, len(spheres), n)
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    dk = rng.integers(0, 5, n)
    to_c = c[target] - o
    graze = c[target] + v * r[target, None] * rng.choice([0.999, 1.0, 1.001], (n, 1)) - o
    axis = np.zeros((n, 3)); axis[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    d = np.where((dk == 0)[:, None], v, np.where((dk == 1)[:, None], to_c, np.where((dk == 2)[:, None], graze,
        np.where((dk == 3)[:, None], axis, v * rng.choice([1e-3, 0.5, 7.0], (n, 1))))))
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    unit = (rng.random((n, 1)) < 0.7) & (nrm > 0)
    d = np.where(unit, d / np.where(nrm > 0, nrm, 1.0), d)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def _compare_hits(gpu, rays, t_min, t_max):
    from path_tracer_amd import _lib
    scan = gpu.debug_eval(_lib.PT_FN_SCENE_HIT, rays, t_min, t_max)
    tree = gpu.debug_eval(_lib.PT_FN_SCENE_HIT_TREE, rays, t_min, t_max)
    np.testing.assert_array_equal(tree.view(np.uint32), scan.view(np.uint32))
    return scan


def test_tree_is_built_for_scenes_of_256_spheres_and_more(gpu):
    from path_tracer_amd import _lib, host
    gpu.set_scene(host.Scene().setupScene(456, 19).spheres)                 # 5 spheres
    assert gpu.tree_info() == {"nodes": 0, "depth": 0}
    with pytest.raises(Exception):
        gpu.debug_eval(_lib.PT_FN_SCENE_HIT_TREE, np.zeros((1, 6), np.float32))
    gpu.set_scene(host.Scene().sceneSynthetic(255, 7).spheres)
    assert gpu.tree_info()["nodes"] == 0
    gpu.set_scene(host.Scene().setupScene(456, 21).spheres)                 # 488 spheres: one LDS tile, the tree is walked in LDS
    info = gpu.tree_info()
    assert 488 // 4 - 1 <= info["nodes"] <= 192 and 8 <= info["depth"] <= 13
    big = host.Scene().sceneSynthetic(3000, 7).spheres
    gpu.set_scene(big)
    info = gpu.tree_info()
    assert 3000 // 4 - 1 <= info["nodes"] < 3000 and 11 <= info["depth"] <= 24   # at best ceil(log2(3000 / 4)) + 1 levels, at most the stack
    _set_scene(gpu, big, 0)
    assert gpu.tree_info()["nodes"] == 0
    _set_scene(gpu, big[:3], 1)                                             # a scene that is one leaf
    assert gpu.tree_info() == {"nodes": 1, "depth": 1}
    bad = big.copy(); bad["cx"][17] = np.nan                                # non-finite spheres keep the scan
    gpu.set_scene(bad)
    assert gpu.tree_info()["nodes"] == 0


@pytest.mark.parametrize("n_scene", [18, 19, 20, 21])
def test_closest_hits_of_the_reference_scenes(gpu, n_scene):
    from path_tracer_amd import host
    sp = host.Scene().setupScene(456, n_scene).spheres
    _set_scene(gpu, sp, 1)
    rng = np.random.default_rng(n_scene)
    for t_min, t_max in ((0.1, 1000.0), (0.001, 1.0e6), (0.0, 50.0), (-5.0, 1000.0)):
        scan = _compare_hits(gpu, _rays(rng, sp, 60000), t_min, t_max)
        assert 0.1 < np.mean(scan[:, 0] >= 0) < 1.0


def test_closest_hits_of_adversarial_scenes(gpu):
    """radii from 1e-3 to 5000 and negative, coincident spheres (the lower index must win), concentric shells, a scene far
    from the origin (coordinates ~1e4), everything on one point (the median split has nothing to split on)"""
    from path_tracer_amd import _lib
    rng = np.random.default_rng(5)
    hits = 0
    for trial in range(40):
        n = int(rng.choice([2, 5, 40, 300, 2500]))
        sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
        spread = float(rng.choice([0.0, 4.0, 15.0, 60.0]))
        off = float(rng.choice([0.0, 0.0, 1.0e4]))
        sp["cx"], sp["cy"], sp["cz"] = off + rng.normal(0, spread, n), rng.normal(0, spread, n), rng.normal(0, spread, n) - off
        sp["radius"] = rng.choice([1e-3, 0.02, 0.2, 0.5, 3.0, 40.0, -0.7], n)
        sp["kind"] = 0
        if trial % 3 == 0:
            sp[0] = (0, -5000, 0, 5000, 0, 0.5, 0.5, 0.5, 0)
        if n > 5:
            sp[2] = sp[n - 1] = sp[n // 2]                                  # coincident triples, far apart in index
            for f in ("cx", "cy", "cz"):
                sp[f][4] = sp[f][3]                                         # concentric shells
            sp["radius"][4] = 2.0 * abs(sp["radius"][3])
        _set_scene(gpu, sp, 1)
        assert gpu.tree_info()["nodes"] >= 1
        for t_min, t_max in ((0.001, 1.0e6), (-2.0, 100.0)):
            scan = _compare_hits(gpu, _rays(rng, sp, 30000), t_min, t_max)
            hits += int(np.sum(scan[:, 0] >= 0))
    assert hits > 100000


def test_no_box_rejects_a_sphere_the_exact_test_accepts(gpu):
    """The safety of the tree, pair by pair (PT_FN_TREE_BOX_SAFETY) instead of through closest hits, where a wrongly skipped box
    only shows when it held THE closest sphere: every ray visits every node, and for each sphere Sphere.is_hit accepts at t, the
    box of every ancestor must pass the walk's own box test on (t_min, t).  Reference scenes and the adversarial ones (radii 1e-3
    to 5000 and negative, coincident and concentric spheres, coordinates ~1e4, grazing / axis-aligned / non-unit rays from surfaces
    and centres)."""
    from path_tracer_amd import _lib, host
    rng = np.random.default_rng(77)
    scenes = [host.Scene().setupScene(456, 21).spheres, host.Scene().sceneSynthetic(3000, 2).spheres]
    for trial in range(24):
        n = int(rng.choice([5, 40, 300, 2500]))
        sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
        spread = float(rng.choice([0.0, 4.0, 15.0, 60.0]))
        off = float(rng.choice([0.0, 0.0, 1.0e4]))
        sp["cx"], sp["cy"], sp["cz"] = off + rng.normal(0, spread, n), rng.normal(0, spread, n), rng.normal(0, spread, n) - off
        sp["radius"] = rng.choice([1e-3, 0.02, 0.2, 0.5, 3.0, 40.0, -0.7], n)
        if trial % 3 == 0:
            sp[0] = (0, -5000, 0, 5000, 0, 0.5, 0.5, 0.5, 0)
        sp[2] = sp[n - 1] = sp[n // 2]
        scenes.append(sp)
    accepted = pairs = 0
    for sp in scenes:
        _set_scene(gpu, sp, 1)
        assert gpu.tree_info()["nodes"] >= 1
        for t_min, t_max in ((0.001, 1.0e6), (0.0, 100.0), (-2.0, 1000.0)):
            out = gpu.debug_eval(_lib.PT_FN_TREE_BOX_SAFETY, _rays(rng, sp, 20000), t_min, t_max)
            assert np.all(out[:, 2] == 0), (len(sp), t_min, int(out[:, 2].sum()), np.flatnonzero(out[:, 2])[:5])
            accepted += int(out[:, 0].sum()); pairs += int(out[:, 1].sum())
    assert accepted > 300_000 and pairs > 2_000_000, (accepted, pairs)


def test_non_finite_rays_hit_nothing_either_way(gpu):
    from path_tracer_amd import host
    sp = host.Scene().sceneSynthetic(1000, 3).spheres
    gpu.set_scene(sp)
    assert gpu.tree_info()["nodes"] > 0
    rays = np.zeros((8, 6), np.float32)
    rays[:, 3:] = [[np.nan, 1, 0], [0, 0, 0], [np.inf, 0, 0], [1, np.nan, np.nan], [0, -np.inf, 1], [np.nan] * 3, [0, 1, 0], [1e-40, 1, 0]]
    rays[5, :3] = np.nan
    _compare_hits(gpu, rays, 0.001, 1.0e6)


def _render_both(gpu, spheres, cfg, primary):
    from path_tracer_amd import _lib
    out = []
    try:
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT if primary else 0)
        for min_spheres in (1, 0):
            _set_scene(gpu, spheres, min_spheres)
            assert (gpu.tree_info()["nodes"] > 0) == (min_spheres == 1)
            frame = gpu.render_sum(cfg, allow_not_unit=True)
            st = gpu.stats()
            out.append((frame, (st.paths, st.rays, st.not_unit, st.primary_rays, st.list2_rays, st.box_tests)))
    finally:
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)
    (tree, st_tree), (scan, st_scan) = out
    assert st_tree[:5] == st_scan[:5]
    assert st_tree[5] > 0 and st_scan[5] == 0      # ray-box tests are counted by the walk only
    np.testing.assert_array_equal(tree, scan)
    return tree


@pytest.mark.parametrize("primary", [True, False])
@pytest.mark.parametrize("deterministic", [False, True])
def test_frames_tree_against_scan(gpu, primary, deterministic):
    """one-tile scenes forced through the tree, and many-tile scenes forced through the scan: same frames, same counts"""
    from path_tracer_amd import host
    img = host.Image(96)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=6, rebounds=50)
    for n_scene in (19, 21):
        gpu.set_camera(host.Camera.preset(n_scene).pod)
        _render_both(gpu, host.Scene().setupScene(456, n_scene).spheres, host.make_pt_config(img, rt, seed=3, chunk=4, deterministic=deterministic), primary)
    gpu.set_camera(host.Camera.preset(21).pod)
    for n, seed in ((600, 1), (3000, 7)):
        _render_both(gpu, host.Scene().sceneSynthetic(n, seed).spheres, host.make_pt_config(img, rt, seed=4, chunk=4, deterministic=deterministic), primary)


def test_frames_tree_against_scan_odd_cases(gpu):
    """negative t_min (hits behind the origin count), first-hit colour functions, rebounds 0, tiles and ranges"""
    from path_tracer_amd import host
    sp = host.Scene().sceneSynthetic(1500, 2).spheres
    gpu.set_camera(host.Camera.preset(21).pod)
    img = host.Image(80)
    _render_both(gpu, sp, host.make_pt_config(img, host.RayTracerConfig(subpixels=2, rays_per_subsample=3, t_min=-5.0, t_max=1000.0, rebounds=6), seed=2), True)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=4, rebounds=12)
    for kw in (dict(chunk=4, illumination=1), dict(chunk=4, illumination=2), dict(chunk=3, istart=333, istop=2999),
               dict(chunk=4, tile_pixels=256, tile_count=3, tile_index=1), dict(chunk=0)):
        _render_both(gpu, sp, host.make_pt_config(img, rt, seed=8, **kw), True)
    _render_both(gpu, sp, host.make_pt_config(img, host.RayTracerConfig(subpixels=2, rays_per_subsample=4, rebounds=0), seed=8, chunk=4), True)


def test_tree_frame_against_the_oracle(orc, gpu):
    """and directly against the CPU restatement of the reference's scan"""
    import ctypes as C
    from path_tracer_amd import host
    scene = host.Scene().sceneSynthetic(2000, 11)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    assert gpu.tree_info()["nodes"] > 0
    gpu.set_camera(cam.pod)
    img = host.Image(64)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=4, rebounds=50)
    cfg = host.make_pt_config(img, rt, seed=6, chunk=4)
    got = gpu.render_sum(cfg)
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam.pod), C.sizeof(ocam))
    rc, ref, _ = orc.render_sum(scene.spheres, ocam, ocfg, nthreads=4)
    assert rc == 0
    np.testing.assert_array_equal(got, ref)


def test_tree_walk_against_the_oracle_ray_by_ray(orc, gpu):
    """The tree walk tied to the ORACLE directly (not to the GPU's own scan): Scene.is_hit of scene.zig:96-110 as the C restatement runs
    it, ray by ray, on the adversarial rays of this file — scene 21 (tree in LDS), a 3 000-sphere scene (tree in global memory) and a
    scene of huge, tiny, negative, coincident and concentric spheres; index, t, hit point and normal, bit for bit; t_min 0.1, 0 and < 0."""
    from path_tracer_amd import _lib, host
    rng = np.random.default_rng(77)
    odd = np.zeros(700, dtype=_lib.SPHERE_DTYPE)
    c = np.round(rng.normal(0, 4, (700, 3)))                       # a lattice: many coincident and concentric centres
    odd["cx"], odd["cy"], odd["cz"] = c[:, 0], c[:, 1], c[:, 2]
    odd["radius"] = rng.choice([1e-3, 0.02, 0.3, 0.5, 2.0, 60.0, -0.7], 700)
    scenes = [host.Scene().setupScene(456, 21).spheres, host.Scene().sceneSynthetic(3000, 1).spheres, odd]
    total_hits = 0
    for sp in scenes:
        gpu.set_scene(sp)
        assert gpu.tree_info()["nodes"] > 0
        rays = _rays(rng, sp, 40_000)
        for t_min in (0.1, 0.0, -1.0):
            got = gpu.debug_eval(_lib.PT_FN_SCENE_HIT_TREE, rays, t_min=t_min, t_max=1000.0)
            want = orc.scene_hit_batch(sp, rays, t_min=t_min, t_max=1000.0)
            np.testing.assert_array_equal(got[:, 0], want[:, 0])
            hit = want[:, 0] >= 0
            np.testing.assert_array_equal(got[hit].view(np.uint32), want[hit].view(np.uint32))
            total_hits += int(hit.sum())
    assert total_hits > 100_000


def test_lists_built_by_the_tree_walk_random_cameras_and_adversarial_scenes(gpu):
    """The primary pass of a tree scene builds its per-pixel lists by walking the tree (pt_primary_lists<true>): frames with
    those lists + the tree megakernel against frames without the primary pass through the scan, over pinhole to very wide
    lenses, cameras inside spheres, radii from 1e-3 to 5000 and negative, coincident spheres, dense scenes whose lists overflow."""
    from path_tracer_amd import _lib, host
    rng = np.random.default_rng(23)
    img = host.Image(64)    # 64 x 36
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=8, t_min=0.001, t_max=1.0e6, rebounds=8)
    listed = second = total = 0
    for trial in range(48):
        n = int(rng.choice([3, 40, 300, 1200]))
        sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
        spread = float(rng.choice([4.0, 15.0, 60.0]))
        sp["cx"], sp["cy"], sp["cz"] = rng.normal(0, spread, n), rng.normal(0, spread, n), rng.normal(0, spread, n)
        sp["radius"] = rng.choice([1e-3, 0.02, 0.2, 0.5, 3.0, 40.0, -0.7], n)
        sp["kind"] = rng.choice([0, 1, 2], n, p=[0.6, 0.25, 0.15])
        sp["ar"], sp["ag"], sp["ab"] = rng.random(n), rng.random(n), rng.random(n)
        sp["param"] = np.where(sp["kind"] == 2, 1.5, 0.3 * rng.random(n))
        sp[0] = (0, -5000, 0, 5000, 0, 0.5, 0.5, 0.5, 0)
        if n > 3:
            sp[1] = (2000, 0, 0, 1.0, 0, 1, 1, 1, 0)
            sp[2] = sp[3] = (1, 1, 1, 0.25, 0, 1, 0.5, 0.2, 0)     # coincident: the lower index must win
        lookfrom = rng.normal(0, 6, 3)
        lookat = lookfrom + rng.normal(0, 1, 3) * rng.choice([0.5, 3.0, 20.0])
        vup = np.array([0.0, 1.0, 0.0]) + rng.normal(0, 0.3, 3)
        cam = host.Camera.new(lookfrom, lookat, vup, float(rng.choice([5.0, 20.0, 60.0, 120.0])), img.aspect_ratio,
                              float(rng.choice([0.0, 0.0, 0.05, 0.5, 4.0])), float(rng.choice([0.3, 3.0, 10.0, 40.0])))
        gpu.set_camera(cam.pod)
        cfg = host.make_pt_config(img, rt, seed=300 + trial, chunk=16)
        try:
            _set_scene(gpu, sp, 1)
            tree = gpu.render_sum(cfg, allow_not_unit=True)
            st = gpu.stats()
            gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 0)
            _set_scene(gpu, sp, 0)
            scan = gpu.render_sum(cfg, allow_not_unit=True)
            st0 = gpu.stats()
        finally:
            gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)
        assert (st.paths, st.rays, st.not_unit) == (st0.paths, st0.rays, st0.not_unit) and st0.primary_rays == 0
        np.testing.assert_array_equal(tree, scan)
        listed += st.primary_rays
        second += st.list2_rays
        total += st.paths
    assert 0.5 * total < listed < total     # listed pixels and overflowing ones
    assert second > 0.02 * total            # second-bounce lists too

"""Generates tests/golden/*.npz from the CPU oracle (goldens the build creates itself).

The reference cannot be run here, so these vectors pin the oracle's CURRENT behaviour (regression
fixtures) and give the GPU tests committed data that needs neither the oracle nor /root/reference at
run time.  Re-run only when the oracle deliberately changes:  python tools/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def images():
    out = {}
    for n_scene, width in ((18, 32), (21, 64)):
        sp, _ = o.setup_scene(n_scene, 456)
        cam = o.camera_preset(n_scene)
        img = o.image_new(width)
        for depth in (6, 50):
            cfg = o.make_config(img, subpixels=1, rays_per_subsample=1, rebounds=depth, mode=1)
            rc, s, st = o.render_sum(sp, cam, cfg, 4)
            assert rc == 0
            out["det_scene%d_w%d_d%d" % (n_scene, width, depth)] = s
            rc, ppm = o.resolve_ppm(cfg, img, s)
            assert rc == 0
            out["det_scene%d_w%d_d%d_ppm" % (n_scene, width, depth)] = ppm
        # a stochastic image pins the Philox stream definition (draw order, block alignment)
        cfg = o.make_config(img, subpixels=4, rays_per_subsample=2, rebounds=50, mode=0, seed=456)
        rc, s, st = o.render_sum(sp, cam, cfg, 4)
        assert rc == 0
        out["sto_scene%d_w%d_d50_seed456" % (n_scene, width)] = s
        out["sto_scene%d_w%d_d50_seed456_rays" % (n_scene, width)] = np.array([st.paths, st.rays], dtype=np.uint64)
    out["scene21_seed456_spheres"] = o.setup_scene(21, 456)[0].view(np.uint8)
    np.savez_compressed(os.path.join(OUT, "images.npz"), **out)


def unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def per_function(n=400, seed=2024):
    rng = np.random.default_rng(seed)
    L = o.lib()
    out = {}
    f32 = np.float32
    sp21, _ = o.setup_scene(21, 456)
    cam = o.camera_preset(21)

    # cast_ray: in s,t,xi[16] -> o,d,n_used
    inp = rng.random((n, 18), dtype=np.float32)
    res = np.zeros((n, 7), dtype=np.float32)
    for i in range(n):
        used = C.c_uint32()
        oo, dd = o.f3((0, 0, 0)), o.f3((0, 0, 0))
        xi = np.ascontiguousarray(inp[i, 2:])
        L.orc_cast_ray(C.byref(cam), float(inp[i, 0]), float(inp[i, 1]), o.fptr(xi), 16, C.byref(used), oo, dd)
        res[i] = list(oo) + list(dd) + [used.value]
    out["cast_ray_in"], out["cast_ray_out"] = inp, res

    # sphere_hit: o,d,c,r,tmin,tmax -> hit,t,p,n   (rays aimed near the sphere so that ~half hit)
    c = rng.uniform(-5, 5, (n, 3)).astype(f32)
    r = rng.uniform(0.1, 2.0, (n, 1)).astype(f32)
    r[::7] *= -1  # negative radii (scene.zig:48)
    org = rng.uniform(-8, 8, (n, 3)).astype(f32)
    aim = c + rng.normal(0, 1, (n, 3)).astype(f32) * np.abs(r)
    d = unit(aim - org)
    inp = np.concatenate([org, d, c, r, np.full((n, 1), 0.1, f32), np.full((n, 1), 1000.0, f32)], axis=1).astype(f32)
    res = np.zeros((n, 8), dtype=f32)
    sph = np.zeros(1, dtype=o.SPHERE_DTYPE)
    for i in range(n):
        sph[0] = (inp[i, 6], inp[i, 7], inp[i, 8], inp[i, 9], 0, 1, 1, 1, 0)
        t = C.c_float()
        p, nn = o.f3((0, 0, 0)), o.f3((0, 0, 0))
        h = L.orc_sphere_hit(sph.ctypes.data, o.f3(inp[i, 0:3]), o.f3(inp[i, 3:6]), float(inp[i, 10]), float(inp[i, 11]),
                             C.byref(t), p, nn)
        res[i] = [h, t.value if h else 0] + (list(p) + list(nn) if h else [0] * 6)
    out["sphere_hit_in"], out["sphere_hit_out"] = inp, res

    # scene_hit over scene 21: o,d -> id,t,p,n
    org = np.array([13, 2, 3], f32) + rng.normal(0, 0.05, (n, 3)).astype(f32)
    tgt = np.stack([rng.uniform(-11, 11, n), rng.uniform(-0.5, 2.5, n), rng.uniform(-11, 11, n)], axis=1).astype(f32)
    d = unit(tgt - org)
    inp = np.concatenate([org, d], axis=1).astype(f32)
    res = np.zeros((n, 8), dtype=f32)
    for i in range(n):
        t = C.c_float()
        p, nn = o.f3((0, 0, 0)), o.f3((0, 0, 0))
        idx = L.orc_scene_hit(sp21.ctypes.data, len(sp21), o.f3(inp[i, 0:3]), o.f3(inp[i, 3:6]), 0.1, 1000.0, C.byref(t), p, nn)
        res[i] = [idx, t.value if idx >= 0 else 0] + (list(p) + list(nn) if idx >= 0 else [0] * 6)
    out["scene_hit_in"], out["scene_hit_out"] = inp, res

    # scatter: d,p,n,kind,param,xi[16] -> scattered d, n_used
    d = unit(rng.normal(0, 1, (n, 3)))
    nrm = unit(rng.normal(0, 1, (n, 3)))
    p = rng.uniform(-3, 3, (n, 3)).astype(f32)
    kind = rng.integers(0, 3, (n, 1)).astype(f32)
    param = np.where(kind == 1, rng.uniform(0, 0.5, (n, 1)), np.where(kind == 2, rng.choice([1.5, 1.3, 2.4], (n, 1)), 0)).astype(f32)
    xi = rng.random((n, 16), dtype=np.float32)
    inp = np.concatenate([d, p, nrm, kind, param, xi], axis=1).astype(f32)
    res = np.zeros((n, 4), dtype=f32)
    for i in range(n):
        sph[0] = (0, 0, 0, 1, int(inp[i, 9]), 0.5, 0.6, 0.7, inp[i, 10])
        used = C.c_uint32()
        at, so, sd = o.f3((0, 0, 0)), o.f3((0, 0, 0)), o.f3((0, 0, 0))
        x = np.ascontiguousarray(inp[i, 11:])
        L.orc_scatter(sph.ctypes.data, o.f3(inp[i, 0:3]), o.f3(inp[i, 3:6]), o.f3(inp[i, 6:9]), o.fptr(x), 16, C.byref(used), at, so, sd)
        res[i] = list(sd) + [used.value]
    out["scatter_in"], out["scatter_out"] = inp, res

    # refract (incl. total internal reflection), schlick, reflect, lerp
    eta = rng.choice([1.5, 1 / 1.5, 0.3, 2.4], (n, 1)).astype(f32)
    flip = np.where((d * nrm).sum(1, keepdims=True) > 0, -1, 1).astype(f32)
    inp = np.concatenate([d, nrm * flip, eta], axis=1).astype(f32)
    res = np.zeros((n, 4), dtype=f32)
    for i in range(n):
        rr = o.f3((0, 0, 0))
        ok = L.orc_refract(o.f3(inp[i, 0:3]), o.f3(inp[i, 3:6]), float(inp[i, 6]), rr)
        res[i] = [ok] + (list(rr) if ok else [0, 0, 0])
    assert 0 < res[:, 0].sum() < n  # both branches present
    out["refract_in"], out["refract_out"] = inp, res

    inp = np.concatenate([rng.uniform(-0.2, 1.5, (n, 1)), rng.choice([1.5, 1.3, 2.4], (n, 1))], axis=1).astype(f32)
    out["schlick_in"] = inp
    out["schlick_out"] = np.array([[L.orc_schlick(float(a), float(b))] for a, b in inp], dtype=f32)

    inp = np.concatenate([d, nrm], axis=1).astype(f32)
    res = np.zeros((n, 3), dtype=f32)
    for i in range(n):
        rr = o.f3((0, 0, 0))
        L.orc_reflect(o.f3(inp[i, 0:3]), o.f3(inp[i, 3:6]), rr)
        res[i] = list(rr)
    out["reflect_in"], out["reflect_out"] = inp, res

    dd = d.copy()
    dd[::9] *= 1.001  # not unit -> DirectionIsNotUnitVector
    inp = np.concatenate([dd, np.tile(np.array([[1, 1, 1, 0.5, 0.7, 1.0]], f32), (n, 1))], axis=1).astype(f32)
    res = np.zeros((n, 4), dtype=f32)
    for i in range(n):
        rr = o.f3((0, 0, 0))
        rc = L.orc_lerp(o.f3(inp[i, 0:3]), o.f3(inp[i, 3:6]), o.f3(inp[i, 6:9]), rr)
        res[i] = [1 if rc != 0 else 0] + (list(rr) if rc == 0 else [0, 0, 0])
    out["lerp_in"], out["lerp_out"] = inp, res

    # philox: seed_lo, seed_hi, pixel, sample, block (u32) -> 4 floats
    inp = rng.integers(0, 2 ** 32, (n, 5), dtype=np.uint64).astype(np.uint32)
    res = np.zeros((n, 4), dtype=f32)
    for i in range(n):
        ctr = (C.c_uint32 * 4)(int(inp[i, 2]), int(inp[i, 3]), int(inp[i, 4]), 0)
        key = (C.c_uint32 * 2)(int(inp[i, 0]), int(inp[i, 1]))
        w = (C.c_uint32 * 4)()
        L.orc_philox4x32_10(ctr, key, w)
        res[i] = [L.orc_u32_to_float(x) for x in w]
    out["philox_in"], out["philox_out"] = inp, res
    np.savez_compressed(os.path.join(OUT, "per_function.npz"), **out)


def sampler_direct(n=600, seed=77):
    """ORC_SAMPLER_DIRECT (not in the reference: the opt-in samplers without rejection loops): per-function vectors of both
    samplers on the same draws, and stochastic frames in direct mode.  A file of its own: the reference-faithful fixtures
    above are not regenerated for it."""
    rng = np.random.default_rng(seed)
    out = {}
    xi = (rng.integers(0, 1 << 23, size=(n, 16)).astype(np.float32) / np.float32(1 << 23))   # the values Random.float(f32) can take
    xi[0, :3] = 0.0
    xi[1, :3] = np.float32(1.0) - np.float32(2.0 ** -23)
    xi[2, :3] = 0.5
    xi[3, :3] = (0.0, 0.25, 1.0 - 2.0 ** -23)
    for which in ("disk", "ball"):
        for name, smp in (("rejection", o.SAMPLER_REJECTION), ("direct", o.SAMPLER_DIRECT)):
            inp = np.concatenate([np.full((n, 1), smp, np.float32), xi], axis=1)
            out["sample_%s_%s_in" % (which, name)] = inp
            out["sample_%s_%s_out" % (which, name)] = o.sample_batch(which, smp, xi)
    for n_scene, width in ((18, 32), (21, 64)):
        sp, _ = o.setup_scene(n_scene, 456)
        cam = o.camera_preset(n_scene)
        img = o.image_new(width)
        cfg = o.make_config(img, subpixels=4, rays_per_subsample=2, rebounds=50, mode=0, seed=456, sampler=o.SAMPLER_DIRECT)
        rc, s, st = o.render_sum(sp, cam, cfg, 4)
        assert rc == 0
        out["direct_scene%d_w%d_d50_seed456" % (n_scene, width)] = s
        out["direct_scene%d_w%d_d50_seed456_rays" % (n_scene, width)] = np.array([st.paths, st.rays], dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "sampler_direct.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    o.build()
    if "--only-direct" in sys.argv:
        sampler_direct()
    else:
        images()
        per_function()
        sampler_direct()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))

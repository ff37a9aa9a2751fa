"""Stress run of the sphere tree against the plain ascending scan (development tool; the committed tests hold a bounded
version of the same comparison): random scenes of many shapes x random rays, closest hit (index and t) must agree bit for bit.
usage: python tools/stress_tree.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_tracer_amd import _lib
from path_tracer_amd.gpu import GpuContext

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_tree import _rays


def scene(rng):
    n = int(rng.choice([2, 7, 33, 200, 511, 512, 513, 2000, 20000]))
    sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
    shape = rng.integers(0, 5)
    spread = float(rng.choice([0.0, 0.5, 4.0, 15.0, 60.0, 1000.0]))
    off = rng.normal(0, 1, 3) * float(rng.choice([0.0, 0.0, 100.0, 1.0e4, 1.0e6]))
    c = rng.normal(0, 1, (n, 3)) * spread
    if shape == 1: c[:, 1] = 0.0                                  # on a plane
    if shape == 2: c[:, 1:] = 0.0                                 # on a line
    if shape == 3: c = np.round(c)                                # on a lattice: many coincident centres
    if shape == 4: c = c * rng.choice([1e-3, 1.0], (n, 1))        # a dense cluster inside a sparse cloud
    c += off
    sp["cx"], sp["cy"], sp["cz"] = c[:, 0], c[:, 1], c[:, 2]
    r = rng.choice([1e-4, 1e-3, 0.02, 0.2, 0.5, 3.0, 40.0, 5000.0, -0.7], n, p=[.05, .1, .2, .3, .15, .1, .05, .01, .04])
    sp["radius"] = r * float(rng.choice([1.0, 1.0, 1e-2, 30.0]))
    sp["kind"] = 0
    return sp


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    g = GpuContext(0)
    g.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, 1)
    t0, trials, rays_total, hits, pairs = time.time(), 0, 0, 0, 0
    while time.time() - t0 < budget:
        sp = scene(rng)
        g.set_scene(sp)
        assert g.tree_info()["nodes"] >= 1
        for t_min, t_max in ((0.001, 1.0e6), (0.0, 1.0e30), (-3.0, 50.0)):
            rays = _rays(rng, sp, 100000)
            a = g.debug_eval(_lib.PT_FN_SCENE_HIT, rays, t_min, t_max)
            b = g.debug_eval(_lib.PT_FN_SCENE_HIT_TREE, rays, t_min, t_max)
            bad = np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))
            if len(bad):
                i = bad[0]
                print("MISMATCH trial %d n=%d ray %s scan %s tree %s" % (trials, len(sp), rays[i], a[i], b[i]), flush=True)
                np.save("gpurun_out/stress_tree_scene.npy", sp); np.save("gpurun_out/stress_tree_rays.npy", rays)
                sys.exit(1)
            rays_total += len(rays); hits += int((a[:, 0] >= 0).sum())
            c = g.debug_eval(_lib.PT_FN_TREE_BOX_SAFETY, rays[:20000], t_min, t_max)   # every accepted sphere against every ancestor's box
            if c[:, 2].any():
                i = int(np.flatnonzero(c[:, 2])[0])
                print("BOX FAILED trial %d n=%d ray %s: %s" % (trials, len(sp), rays[i], c[i]), flush=True)
                np.save("gpurun_out/stress_tree_scene.npy", sp); np.save("gpurun_out/stress_tree_rays.npy", rays)
                sys.exit(1)
            pairs += int(c[:, 1].sum())
        trials += 1
        if trials % 20 == 0:
            print("%d scenes, %.1f M rays, %.1f M hits, %.0f s" % (trials, rays_total * 1e-6, hits * 1e-6, time.time() - t0), flush=True)
    print("OK: %d scenes, %.1f M rays, %.1f M hits, no mismatch; %.1f M (accepted sphere, ancestor box) pairs, no box failed" % (trials, rays_total * 1e-6, hits * 1e-6, pairs * 1e-6))


main()

"""Stress run of the primary pass's candidate lists (scan-built and tree-built) and of the tree megakernel against the
render that uses neither: random scenes x random cameras at small size, frames and counters must agree bit for bit.
(Development tool; tests/test_gpu_tree.py and tests/test_gpu_primary_pass.py hold bounded versions.)
usage: python tools/stress_lists.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_tracer_amd import _lib, host
from path_tracer_amd.gpu import GpuContext


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    g = GpuContext(0)
    img = host.Image(64)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=8, t_min=0.001, t_max=1.0e6, rebounds=8)
    t0, trials, listed, second, total, accepted = time.time(), 0, 0, 0, 0, 0
    while time.time() - t0 < budget:
        n = int(rng.choice([3, 40, 300, 512, 700, 4000]))
        sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
        spread = float(rng.choice([4.0, 15.0, 60.0]))
        sp["cx"], sp["cy"], sp["cz"] = rng.normal(0, spread, n), rng.normal(0, spread, n) * rng.choice([0.05, 1.0]), rng.normal(0, spread, n)
        sp["radius"] = rng.choice([1e-3, 0.02, 0.2, 0.5, 3.0, 40.0, -0.7], n)
        sp["kind"] = rng.choice([0, 1, 2], n, p=[0.6, 0.25, 0.15])
        sp["ar"], sp["ag"], sp["ab"] = rng.random(n), rng.random(n), rng.random(n)
        sp["param"] = np.where(sp["kind"] == 2, 1.5, 0.3 * rng.random(n))
        sp[0] = (0, -5000, 0, 5000, 0, 0.5, 0.5, 0.5, 0)
        lookfrom = rng.normal(0, 6, 3)
        lookat = lookfrom + rng.normal(0, 1, 3) * rng.choice([0.5, 3.0, 20.0])
        vup = np.array([0.0, 1.0, 0.0]) + rng.normal(0, 0.3, 3)
        cam = host.Camera.new(lookfrom, lookat, vup, float(rng.choice([5.0, 20.0, 60.0, 120.0])), img.aspect_ratio,
                              float(rng.choice([0.0, 0.0, 0.05, 0.5, 4.0])), float(rng.choice([0.3, 3.0, 10.0, 40.0])))
        g.set_camera(cam.pod)
        cfg = host.make_pt_config(img, rt, seed=1000 + trials, chunk=16)
        g.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)
        g.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, 1)
        g.set_scene(sp)
        a = g.render_sum(cfg, allow_not_unit=True); sa = g.stats()
        if sa.primary_rays:  # the lists' property sphere by sphere (ptgpu_debug_list_safety), on the same scene, camera and draws
            chk = g.debug_list_safety(cfg)
            if chk["missing_a"] or chk["missing_b"] or chk["closest_differs"]:
                print("LIST CHECK FAILED trial %d n=%d: %s" % (trials, n, chk), flush=True)
                np.save("gpurun_out/stress_lists_scene.npy", sp)
                sys.exit(1)
            accepted += chk["accepted_a"] + chk["accepted_b"]
        g.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 0)
        g.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, 0)
        g.set_scene(sp)
        b = g.render_sum(cfg, allow_not_unit=True); sb = g.stats()
        if (sa.paths, sa.rays, sa.not_unit) != (sb.paths, sb.rays, sb.not_unit) or not np.array_equal(a, b):
            print("MISMATCH trial %d n=%d" % (trials, n), flush=True)
            np.save("gpurun_out/stress_lists_scene.npy", sp)
            sys.exit(1)
        listed += sa.primary_rays; second += sa.list2_rays; total += sa.paths
        trials += 1
        if trials % 50 == 0:
            print("%d renders pairs, listed %.2f, second bounce %.2f of the paths, %.0f s" % (trials, listed / total, second / total, time.time() - t0), flush=True)
    print("OK: %d scenes x cameras, listed %.2f, second bounce %.2f of the paths, no mismatch; %.1f M accepted spheres all in their lists" % (trials, listed / total, second / total, accepted * 1e-6))


main()

"""Quick single-GPU throughput probe (development tool, not the benchmark)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # before libptgpu: one HIP runtime per process (torch bundles its own)
torch.cuda.init()
import numpy as np
from path_tracer_amd import host
from path_tracer_amd.gpu import GpuContext

def main():
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16      # rays_per_subsample (x4 subpixels)
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    scene = int(sys.argv[5]) if len(sys.argv) > 5 else 21
    if scene > 100:
        sp = host.Scene().sceneSynthetic(scene, 1).spheres; cam = host.Camera.preset(21).pod
    else:
        sp = host.Scene().setupScene(456, scene).spheres; cam = host.Camera.preset(scene).pod
    img = host.Image(width)
    tiles = os.environ.get("PROBE_TILES")  # "count,index": one owner's share of an interleaved 256-pixel tile split
    tkw = dict(tile_pixels=256, tile_count=int(tiles.split(",")[0]), tile_index=int(tiles.split(",")[1])) if tiles else {}
    cfg = host.make_pt_config(img, host.RayTracerConfig(subpixels=4, rays_per_subsample=spp, rebounds=depth), seed=456, chunk=chunk, **tkw)
    g = GpuContext(0)
    info = g.device_info()
    if "PROBE_TREE_MIN" in os.environ:  # 0: scan whatever the size; 1: sphere tree whatever the size
        from path_tracer_amd import _lib
        g.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, int(os.environ["PROBE_TREE_MIN"]))
    t = time.time(); g.set_scene(sp); g.set_camera(cam)
    print("set_scene %.1f ms, tree %s" % ((time.time() - t) * 1e3, g.tree_info()))
    if "PROBE_PRIMARY_BYTES" in os.environ:  # 0: camera rays inside the megakernel
        from path_tracer_amd import _lib
        g.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, int(float(os.environ["PROBE_PRIMARY_BYTES"])))
    if "PROBE_PRIMARY_BOUNCES" in os.environ:
        from path_tracer_amd import _lib
        g.set_limit(_lib.PT_LIMIT_PRIMARY_BOUNCES, int(os.environ["PROBE_PRIMARY_BOUNCES"]))
    buf = torch.zeros(img.height * img.width * 3, dtype=torch.float32, device="cuda")
    for it in range(3):
        t = time.time(); g.render_sum_device(cfg, buf.data_ptr()); dt = time.time() - t
        st = g.stats()
        tests = st.rays * len(sp)
        flops = tests * 18 + st.rays * 60 + st.paths * 40
        print("   lane util %.3f  flush iters/step %.2f  segments %d  primary: %.3e + %.3e rays (%.1f%% of rays) %.1f ms" % (
            (st.rays - st.primary_rays - st.list2_rays) / (64.0 * max(st.wave_steps, 1)), st.flush_iters / max(st.wave_steps, 1), st.segments, st.primary_rays,
            st.list2_rays, 100.0 * (st.primary_rays + st.list2_rays) / max(st.rays, 1), st.primary_ms))
        print("iter %d: wall %.3fs kernel %.1f ms paths %.3e rays %.3e (%.2f rays/path) cands/ray %.2f | %.3e paths/s %.1f Mrays/s %.2f Gtests/s %.1f TFLOP/s (%.1f%% of 157.3)" % (
            it, dt, st.kernel_ms, st.paths, st.rays, st.rays / st.paths, st.candidates / st.rays,
            st.paths / (st.kernel_ms * 1e-3), st.rays / (st.kernel_ms * 1e-3) * 1e-6, tests / (st.kernel_ms * 1e-3) * 1e-9,
            flops / (st.kernel_ms * 1e-3) * 1e-12, flops / (st.kernel_ms * 1e-3) / 157.3e12 * 100), flush=True)
    print(info)

main()

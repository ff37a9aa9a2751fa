import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch; torch.cuda.init()
from path_tracer_amd import host, _lib
from path_tracer_amd.gpu import GpuContext
g = GpuContext(0)
img = host.Image(1920)
buf = torch.zeros(img.height * img.width * 3, dtype=torch.float32, device="cuda")
cfg = host.make_pt_config(img, host.RayTracerConfig(subpixels=4, rays_per_subsample=16, rebounds=50), seed=456, chunk=16)
cases = [("scene%d" % k, host.Scene().setupScene(456, k).spheres, host.Camera.preset(k).pod) for k in (18, 19, 20, 21)]
cases += [("synth%d" % n, host.Scene().sceneSynthetic(n, 1).spheres, host.Camera.preset(21).pod) for n in (16, 48, 101, 150, 250, 400, 512)]
for name, sp, cam in cases:
    row = []
    for m in (0, 1):
        g.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, m)
        g.set_scene(sp); g.set_camera(cam)
        best = 1e9
        for it in range(3):
            g.render_sum_device(cfg, buf.data_ptr()); st = g.stats(); best = min(best, st.kernel_ms)
        row.append((best, st.primary_ms, g.tree_info()["nodes"]))
    print("%-9s n=%4d  scan %.2f ms (primary %.2f) | tree %.2f ms (primary %.2f, %d nodes)" % (name, len(sp), row[0][0], row[0][1], row[1][0], row[1][1], row[1][2]), flush=True)

#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

A "step" = one full render of BASELINE configs[2]: scene 21 (488 spheres), 1920x1080, 1024 paths per
pixel (subpixels 4 x rays_per_subsample 256), depth 50, seed 456, with the scene/camera already
resident in HBM and the fp32 radiance sums left in HBM.
N > 1 GPUs: STRONG scaling of that same frame (BASELINE's metric: "scene 21 1080p, 1/2/4/8 GPUs") by
interleaved 256-pixel tiles — rank g traces all 1024 samples of the tiles t with t % N == g, other pixels
stay 0 — then ONE RCCL reduce(SUM) of the fp32 sum buffers to rank 0 (inside the timed region), which
assembles the one-GPU frame bit for bit.  `--split samples` splits the sample range of every pixel
instead; `--weak` keeps the per-GPU work fixed (N x 1024 paths per pixel, sample split).  value = paths traced by all ranks / max-over-ranks
wall time.

Rank 0 prints one JSON line (contract in the task statement) with `roofline` (FP32 VALU) and,
at N = 1, `cpu_baseline` (the oracle's C restatement timed on the host cores).

`roofline.frac` is the EXECUTED fraction of the dominant kernel (pt_megakernel): the arithmetic its active lanes
performed — device-counted ray-box tests x 30 FLOP + its exact sphere tests x 18 + 60 per ray it traces (for scenes
traced by the filter scan: 15 / 11 FLOP per filter test) — over its own HIP-event time, against the FP32 VALU peak.
What the reference's scan WOULD have cost for the same rays (18 FLOP per ray-sphere test) is reported
beside it as `algorithmic_rate`, a work rate that can exceed the peak because the sphere tree and the per-pixel lists
skip most of those tests; it is not a utilisation.  `roofline.kernels` lists every tracing kernel with its time and
executed arithmetic; `roofline.traffic` and the VALU lane utilisations come from the committed rocprofv3 counter
summary of this same command (profiles/r03_counters.json; null when the run's configuration is not the one profiled).
`roofline.scan_reference` (N = 1, headline config) is one more frame timed in the same run with the sphere tree and
the primary pass switched off: every ray through the LDS-tiled filter scan the north star describes.
"""
import argparse
import json
import os
import sys
import time

import torch  # imported BEFORE libptgpu: torch bundles its own HIP runtime; one runtime per process
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_VALU = 157.3e12  # FLOP/s per MI355X: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md)


def algorithmic_flops(paths, rays, nspheres):
    """18 FLOP per ray-sphere test + 60 per ray + 40 per path."""
    return rays * nspheres * 18.0 + rays * 60.0 + paths * 40.0


def load_counters(steps_spp):
    """profiles/r03_counters.json: per-kernel rocprofv3 PMC sums of ONE launch of the headline frame (tools/counters_json.py;
    separate --pmc passes of `bench.py --steps 1 --warmup 0`).  None when absent or collected at another sample count."""
    path = os.path.join(ROOT, "profiles", "r03_counters.json")
    try:
        with open(path) as f:
            c = json.load(f)
    except (OSError, ValueError):
        return None
    return c if c.get("spp") == steps_spp else None


def cpu_baseline(width, depth, seed):
    """The oracle (C restatement of the reference algorithm, reference's thread split) on a bounded
    sample of the same workload: scene 21 at the same resolution and depth, 32 paths per pixel
    (subpixels 4 x rays_per_subsample 8) instead of 1024 — the rate does not depend on spp."""
    from oracle import oracle
    oracle.select_native()  # -O3 -march=native -ffp-contract=off, built on THIS box (the ReleaseFast stand-in of BASELINE.md)
    allowed = sorted(os.sched_getaffinity(0))
    cores = min(16, len(allowed))              # the GPU box's CPU share for one GPU is 16
    os.sched_setaffinity(0, set(allowed[:cores]))
    try:
        threads = 2 * cores                    # main.zig:143-145: two threads per logical CPU
        spheres, _ = oracle.setup_scene(21, seed)
        cam = oracle.camera_preset(21)
        img = oracle.image_new(width)
        cfg = oracle.make_config(img, subpixels=4, rays_per_subsample=8, rebounds=depth, seed=seed)
        t0 = time.perf_counter()
        rc, _, st = oracle.render_sum(spheres, cam, cfg, nthreads=threads)
        dt = time.perf_counter() - t0
    finally:
        os.sched_setaffinity(0, set(allowed))
    assert rc == 0
    return {"value": st.paths / dt, "unit": "paths/s", "cores": cores, "threads": threads, "kind": "port",
            "mrays_per_s": st.rays / dt * 1e-6, "seconds": dt,
            "sample": "scene 21 %dx%d depth %d, 32 paths/pixel (%d paths, %d rays), oracle C restatement (gcc -O3 -march=native "
                      "-ffp-contract=off, built on this box), contiguous ceil(P/T) pixel chunks over T=%d pthreads pinned to %d cores"
                      % (img.width, img.height, depth, st.paths, st.rays, threads, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=256, help="rays_per_subsample per GPU (x4 subpixels = paths/pixel)")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seed", type=int, default=456)
    ap.add_argument("--chunk", type=int, default=0,
                    help="samples per GPU work item; 0 = 16 on one GPU (and for --weak), 8 when N ranks share one frame by tiles, "
                         "max(4, 16/N) by samples (a rank's launch is N times shorter, so its tail must be: measured at one rank's "
                         "eighth, tiles, chunk 8 / 16 = 22.9 / 23.7 ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tree-min-spheres", type=int, default=None,
                    help="A/B only: PT_LIMIT_TREE_MIN_SPHERES (library default 256; 0 = scan for every scene, 1 = sphere tree for every scene)")
    ap.add_argument("--primary-bytes", type=int, default=None, help="A/B only: PT_LIMIT_PRIMARY_BYTES (0 = no primary pass)")
    ap.add_argument("--sampler", choices=["rejection", "direct"], default="rejection",
                    help="rejection (default): the reference's rejection loops, try by try; direct: PT_SAMPLER_DIRECT, the same "
                         "distributions without a loop (not in the reference; include/ptgpu.h).  The headline line is always the "
                         "reference-faithful default; at N = 1 it carries the direct mode's time beside it (`sampler_direct`)")
    ap.add_argument("--dump-frame", default=None, help="tests: rank 0 saves the last step's fp32 sums (H x W x 3) to this .npy file")
    ap.add_argument("--no-extras", action="store_true", help="skip the scan_reference and sampler_direct frames (profiling runs)")
    ap.add_argument("--spheres", type=int, default=0,
                    help="0 (default) = scene 21, the headline config; N > 0 = the synthetic random-sphere scene of N spheres "
                         "(BASELINE configs[4] uses 100000 with --spp 64 --split tiles)")
    ap.add_argument("--split", choices=["samples", "tiles"], default=None,
                    help="multi-GPU partition of the frame: interleaved 256-pixel tiles (default: a rank then builds the candidate "
                         "lists of its own pixels only, and the frame equals the one-GPU frame bit for bit) or sample ranges of "
                         "every pixel (default with --weak); measured on one GPU at one rank's eighth of the headline frame: "
                         "22.9 ms (tiles) / 25.4 ms (samples) against 22.8 = 182.5 / 8")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling (sample split only): every rank traces 4 x spp paths per pixel of a frame of N x 4 x spp; "
                         "default is strong scaling: ONE frame of 4 x spp paths per pixel shared by the N ranks")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="development only: run the N>1 plumbing on ONE GPU (all ranks on cuda:0, gloo all_reduce "
                         "instead of the RCCL reduce); the JSON line is marked and is not a benchmark result")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from path_tracer_amd import host
    from path_tracer_amd.distributed import gpu_render_into, render_sum_distributed
    from path_tracer_amd.gpu import GpuContext

    img = host.Image(args.width)
    weak = args.weak
    if args.split is None:
        args.split = "samples" if weak else "tiles"
    if args.chunk == 0:
        args.chunk = 16 if (world == 1 or weak) else (8 if args.split == "tiles" else max(4, 16 // world))
    if weak and args.split != "samples":
        raise SystemExit("bench.py: --weak needs --split samples")
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=args.spp * (world if weak else 1), rebounds=args.depth)
    scene = host.Scene().sceneSynthetic(args.spheres, 1) if args.spheres > 0 else host.Scene().setupScene(args.seed, 21)
    camera = host.Camera.preset(21, img.aspect_ratio)
    gpu = GpuContext(local_rank)
    if args.tree_min_spheres is not None:
        from path_tracer_amd import _lib
        gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, args.tree_min_spheres)
    if args.primary_bytes is not None:
        from path_tracer_amd import _lib
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, args.primary_bytes)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(camera.pod)
    sampler = 1 if args.sampler == "direct" else 0
    render_into = gpu_render_into(gpu, sampler=sampler)
    frame = torch.zeros(img.height * img.width * 3, dtype=torch.float32, device="cuda")

    def step():
        render_sum_distributed(render_into, img, rt, args.seed, split=args.split, chunk=args.chunk, frame=frame,
                               everywhere=args.rehearse_gloo)  # gloo has no CUDA reduce, only all_reduce
        return gpu.stats()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, primary_ms, lists_ms, mega_ms = [], [], [], []
    paths = rays = wave_steps = primary_rays = list2_rays = segments = cands = boxes = prim_cands = 0
    for _ in range(args.steps):
        st = step()
        kernel_ms.append(st.kernel_ms)   # HIP events around the tracing kernels (primary pass + megakernel), on the stream they ran on
        primary_ms.append(st.primary_ms) # ... around the primary pass alone (lists + pt_primary), summed over the segments
        lists_ms.append(st.lists_ms)     # ... around the list kernels alone
        mega_ms.append(st.megakernel_ms) # ... around pt_megakernel alone
        prim_cands += st.primary_candidates
        paths += st.paths
        rays += st.rays
        cands += st.candidates
        boxes += st.box_tests
        wave_steps += st.wave_steps
        primary_rays += st.primary_rays                   # camera rays the primary pass resolved against their pixel's list
        list2_rays += st.list2_rays
        segments = st.segments
    fence()
    dt = time.perf_counter() - t0

    tot = torch.tensor([dt, float(paths), float(rays)], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max = float(mx[0])
    else:
        dt_max = dt
    all_paths, all_rays = float(tot[1]), float(tot[2])

    if rank == 0 and args.dump_frame:
        import numpy as np
        np.save(args.dump_frame, frame.detach().cpu().numpy().reshape(img.height, img.width, 3))
    if rank == 0:
        nsph = len(scene.spheres)
        scene_name = ("synthetic random-sphere scene (%d spheres, generator seed 1)" % nsph if args.spheres > 0
                      else "scene 21 (488 spheres, seed %d)" % args.seed)
        headline = args.spheres == 0 and args.width == 1920 and args.spp == 256 and args.depth == 50 and sampler == 0
        K = float(args.steps)
        k_ms = sum(kernel_ms) / K
        p_ms, l_ms, m_ms = sum(primary_ms) / K, sum(lists_ms) / K, sum(mega_ms) / K
        if m_ms <= 0.0:                      # more than 64 segments: the per-kernel events were not kept
            m_ms = k_ms - p_ms
        shared = gpu.scene_info()["shared_ordinate_tests"]
        tree = gpu.tree_info()   # nodes > 0: Scene.is_hit walks the sphere tree (scenes of >= 256 spheres)
        paths_l, rays_l = paths / K, rays / K
        prim_rays_l, list2_l = primary_rays / K, list2_rays / K           # camera rays / second rays resolved against per-pixel lists
        list_rays_l = prim_rays_l + list2_l
        mk_rays = rays_l - list_rays_l                                    # rays the megakernel traces
        cands_l, pcands_l, boxes_l = cands / K, prim_cands / K, boxes / K
        shipped = args.tree_min_spheres is None and args.primary_bytes is None   # (A/B switches select other kernels than the ones profiled)
        counters = load_counters(args.spp) if (headline and shipped) else None
        ck = (counters or {}).get("kernels", {})

        def peak_frac(flops, ms):
            return flops / (ms * 1e-3) / PEAK_FP32_VALU if ms > 0 else None

        # ---- what each kernel EXECUTED (active lanes' arithmetic) --------------------------------------------------
        if tree["nodes"]:   # ~30 FLOP per ray-box test (24 instructions: 3 differences to the centre, the margin, 3 grown half extents, 3 centre products, 6 FMAs to the slab ends, min/max chains; kept at the round-2 figure of 30), 18 per exact sphere test, 60 per ray
            mk_exec = boxes_l * 30.0 + (cands_l - pcands_l) * 18.0 + mk_rays * 60.0
            mk_basis = ("executed: %.1f ray-box tests x 30 FLOP + %.1f exact sphere tests x 18 FLOP + 60 FLOP per ray the megakernel traces (device counts)"
                        % (boxes_l / max(mk_rays, 1.0), (cands_l - pcands_l) / max(mk_rays, 1.0)))
        else:               # filter scan: 7 FMA + compare (15 FLOP) per test on every lane, 5 FMA + compare (11) in shared-ordinate words; exact test on the survivors
            mk_exec = mk_rays * ((nsph - shared) * 15.0 + shared * 11.0) + (cands_l - pcands_l) * 18.0 + mk_rays * 60.0
            mk_basis = ("executed: %d filter tests per ray x 15 FLOP (%d of them x 11: shared-ordinate words) + %.1f exact sphere tests x 18 + 60 FLOP per ray"
                        % (nsph, shared, (cands_l - pcands_l) / max(mk_rays, 1.0)))
        if not list_rays_l:
            mk_exec += paths_l * 40.0        # camera rays are cast inside the megakernel
        pr_exec = pcands_l * 18.0 + list_rays_l * 60.0 + paths_l * 40.0 if list_rays_l else 0.0
        mk_algo = mk_rays * (nsph * 18.0 + 60.0) + (0.0 if list_rays_l else paths_l * 40.0)   # SURVEY 8d for the rays it traces
        render_algo = algorithmic_flops(paths_l, rays_l, nsph)                                 # ... for every ray of the render

        def kernel_entry(name, ms, exec_flops, what):
            e = {"name": name, "ms": ms, "share_of_kernel_ms": ms / k_ms if k_ms > 0 else None, "what": what,
                 "executed_flops": exec_flops,
                 "executed_tflops": exec_flops / (ms * 1e-3) * 1e-12 if (exec_flops is not None and ms > 0) else None,
                 "executed_frac": peak_frac(exec_flops, ms) if exec_flops is not None else None}
            c = ck.get(name)
            if c:   # rocprofv3 PMC sums of one launch of this same command (profiles/r03_counters.json)
                e.update({"valu_lane_utilisation": c.get("valu_lane_utilisation"), "valu_wave_instructions": c.get("SQ_INSTS_VALU"),
                          "hbm_read_bytes": c.get("hbm_read_bytes"), "hbm_write_bytes": c.get("hbm_write_bytes"),
                          "wait_share_of_wave_cycles": c.get("wait_any_share")})
            return e

        kernels = [kernel_entry("pt_megakernel", m_ms, mk_exec,
                                "%s for the %.3e rays per frame no per-pixel list resolves + radiance steps" % ("sphere-tree walk" if tree["nodes"] else "filter scan + exact pass", mk_rays))]
        if list_rays_l:
            kernels.append(kernel_entry("pt_primary", p_ms - l_ms, pr_exec,
                                        "camera rays (%.3e) and second rays (%.3e) against their pixel's candidate list (exact Sphere.is_hit, %.1f per ray) + radiance steps"
                                        % (prim_rays_l, list2_l, pcands_l / max(list_rays_l, 1.0))))
            kernels.append(kernel_entry("pt_primary_lists", l_ms, None,
                                        "per-pixel candidate lists: fp64 tube / fp32 cone tests of each pixel against the spheres (not FP32-roofline work)"))
        traffic = None
        if counters:
            traffic = sum((ck.get(k["name"], {}).get("hbm_read_bytes") or 0) + (ck.get(k["name"], {}).get("hbm_write_bytes") or 0) for k in kernels)
        frame_bytes = img.width * img.height * 12
        render = {"kernels": " + ".join(k["name"] for k in reversed(kernels)),
                  "kernel_ms": k_ms, "primary_pass_ms": p_ms, "segments": segments,
                  "paths": paths_l, "rays": rays_l, "primary_rays": list_rays_l, "of_which_second_bounce": list2_l,
                  "executed_flops": mk_exec + pr_exec, "executed_frac": peak_frac(mk_exec + pr_exec, k_ms),
                  "algorithmic_flops": render_algo, "algorithmic_tflops": render_algo / (k_ms * 1e-3) * 1e-12,
                  "algorithmic_rate_over_peak": render_algo / (k_ms * 1e-3) / PEAK_FP32_VALU}
        out = {
            "metric": "paths/sec (scene 21, 1920x1080, 1024 spp, depth 50)" if headline else
                      "paths/sec (%s, %dx%d, %d spp, depth %d%s)" % (scene_name, img.width, img.height, 4 * args.spp, args.depth, ", PT_SAMPLER_DIRECT" if sampler else ""),
            "value": all_paths / dt_max,
            "unit": "paths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, %d paths/pixel per %s (subpixels 4 x %d), depth %d"
                                   % (("BASELINE configs[2]: " if headline else "") + scene_name, img.width, img.height,
                                      4 * args.spp, "GPU" if weak else "frame", args.spp, args.depth),
                       "parallelism": "1 GPU" if world == 1 else
                                      ("sample-range split x%d" if args.split == "samples" else "interleaved 256-pixel tile split x%d") % world
                                      + (" + gloo all_reduce(SUM) of fp32 sums (rehearsal: all ranks on one GPU)" if args.rehearse_gloo else " + RCCL reduce(SUM) of fp32 sums"),
                       "chunk": args.chunk, "global_paths_per_pixel": 4 * args.spp * (world if weak else 1),
                       "sampler": "direct (PT_SAMPLER_DIRECT: not the reference's rejection loops)" if sampler else "rejection (the reference's loops)"},
            "mrays_per_s": all_rays / dt_max * 1e-6,
            "rays_per_path": all_rays / all_paths,
            "roofline": {"bound": "valu", "kernel": "pt_megakernel", "kernel_ms": m_ms, "launches": segments,
                         "achieved": mk_exec / (m_ms * 1e-3) * 1e-12, "peak": PEAK_FP32_VALU * 1e-12, "unit": "TFLOP/s",
                         "frac": peak_frac(mk_exec, m_ms),
                         "basis": mk_basis,
                         "flops_per_launch": mk_exec / max(segments, 1),
                         "rays_per_launch": mk_rays / max(segments, 1), "wave_steps_per_launch": wave_steps / K / max(segments, 1),
                         "traffic": traffic, "traffic_unit": "bytes per frame, all tracing kernels (rocprofv3 FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE)",
                         "traffic_source": (counters or {}).get("source"),
                         "algorithmic_bytes": frame_bytes + nsph * 36,
                         "algorithmic_bytes_note": "12 B of fp32 sums per pixel + the scene once (SURVEY 8d); the rest of `traffic` is the 32-byte-per-path records "
                                                   "pt_primary hands to the megakernel, the hit-id stack and the chunk partials",
                         "algorithmic_rate": {"tflops": mk_algo / (m_ms * 1e-3) * 1e-12, "over_peak": mk_algo / (m_ms * 1e-3) / PEAK_FP32_VALU,
                                              "note": "the reference's scan priced for the rays this kernel traces (SURVEY 8d: 18 FLOP x %d spheres + 60 per ray): a work rate, "
                                                      "NOT a utilisation — %s" % (nsph, "the sphere tree skips nearly all of those tests" if tree["nodes"] else "the filter executes 15/11 of the 18 FLOP per test")},
                         "spheres": nsph, "shared_ordinate_tests": shared,
                         "scene_is_hit": ("sphere tree (%s): %d inner nodes, %d levels, %.1f ray-box tests and %.1f exact sphere tests per megakernel ray instead of %d sphere tests"
                                          % ("walked in LDS" if nsph <= 512 else "walked in global memory", tree["nodes"], tree["depth"], boxes_l / max(mk_rays, 1.0),
                                             (cands_l - pcands_l) / max(mk_rays, 1.0), nsph))
                                         if tree["nodes"] else "filter scan over all spheres + exact pass",
                         "kernels": kernels,
                         "render": render,
                         "note": "FP32 VALU roofline (no MFMA, not HBM-bound: SURVEY.md 8d).  `frac` = what pt_megakernel, the dominant kernel, EXECUTED over its HIP-event time "
                                 "(summed over `launches` segment launches; rocprofv3's average duration agrees: profiles/) against 157.3 TFLOP/s.  The walk is divergence- and "
                                 "latency-bound, not FMA-bound: see `kernels[].valu_lane_utilisation`.  `scan_reference` is the same frame with every ray through the LDS-tiled "
                                 "filter scan (the north star's design), timed in this same run."},
        }
        if args.rehearse_gloo:
            out["rehearsal"] = "all ranks on ONE GPU over gloo: plumbing check only, not a benchmark result"
        if world == 1 and headline and not args.no_extras:
            from path_tracer_amd import _lib

            def one_frame(smp):
                """one warm-up + one timed frame of the same workload outside the timed region above: (wall ms, stats)"""
                fn = gpu_render_into(gpu, sampler=smp)
                res = None
                for _ in range(2):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    render_sum_distributed(fn, img, rt, args.seed, split=args.split, chunk=args.chunk, frame=frame)
                    torch.cuda.synchronize()
                    res = ((time.perf_counter() - t1) * 1e3, gpu.stats())
                return res

            # (1) the north star's own design for Scene.is_hit, timed beside the shipped path: no sphere tree, no per-pixel lists —
            # every ray of the frame through the LDS-tiled filter scan of pt_megakernel
            gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, 0)
            gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 0)
            gpu.set_scene(scene.spheres)
            ms, st = one_frame(0)
            sc_algo = algorithmic_flops(st.paths, st.rays, nsph)
            sc_exec = st.rays * ((nsph - shared) * 15.0 + shared * 11.0) + st.candidates * 18.0 + st.rays * 60.0 + st.paths * 40.0
            out["roofline"]["scan_reference"] = {
                "what": "the same frame with PT_LIMIT_TREE_MIN_SPHERES = 0 and PT_LIMIT_PRIMARY_BYTES = 0: camera rays cast in the megakernel, every ray tested "
                        "against all %d spheres by the LDS-tiled filter scan (7 / 5 FMA + compare per test), exact Sphere.is_hit on the survivors; same image bit for bit" % nsph,
                "ms_per_frame": ms, "kernel_ms": st.kernel_ms, "paths_per_s": st.paths / (ms * 1e-3), "rays": st.rays,
                "algorithmic_tflops": sc_algo / (st.kernel_ms * 1e-3) * 1e-12, "algorithmic_frac": sc_algo / (st.kernel_ms * 1e-3) / PEAK_FP32_VALU,
                "executed_tflops": sc_exec / (st.kernel_ms * 1e-3) * 1e-12, "executed_frac": sc_exec / (st.kernel_ms * 1e-3) / PEAK_FP32_VALU,
                "note": "here the algorithmic figure IS work the kernel does test by test (the reference's 18 FLOP per ray-sphere test, of which the filter "
                        "executes 15 or 11 on every lane): the north star's '>= 25 % of the FP32 VALU peak' refers to this path"}
            gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, _lib.PT_TREE_MIN_SPHERES_DEFAULT)
            gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)
            gpu.set_scene(scene.spheres)
            # (2) the opt-in samplers without rejection loops (same distributions: tests/test_sampler_direct.py), shipped path otherwise
            ms, st = one_frame(1)
            out["sampler_direct"] = {
                "what": "the same frame with pt_config.sampler = PT_SAMPLER_DIRECT (not the reference's rejection loops; the same distributions, "
                        "GPU == oracle bit for bit in this mode too): reported beside the reference-faithful headline, never as `value`",
                "ms_per_frame": ms, "kernel_ms": st.kernel_ms, "primary_pass_ms": st.primary_ms, "megakernel_ms": st.megakernel_ms,
                "paths_per_s": st.paths / (ms * 1e-3), "mrays_per_s": st.rays / (ms * 1e-3) * 1e-6, "rays_per_path": st.rays / max(st.paths, 1)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.width, args.depth, args.seed)
            except Exception as e:  # the GPU line must not be lost to a host-side problem (e.g. no compiler for the oracle)
                out["cpu_baseline"] = {"value": None, "unit": "paths/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    gpu.close()


if __name__ == "__main__":
    main()

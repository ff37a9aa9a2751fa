"""gpurun_out/profile_<tag>/ (tools/collect_r03.sh) -> profiles/<tag>_counters.json + profiles/<tag>_counters.md + the kernel stats.

The JSON is what bench.py reads back (roofline.traffic, kernels[].valu_lane_utilisation): per tracing kernel the rocprofv3 PMC
sums of ONE launch of the headline frame, each counter group collected in a run of its own.
usage: python tools/counters_json.py <tag>
"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/profile_%s" % tag
os.makedirs("profiles", exist_ok=True)
if os.path.exists(os.path.join(src, "kt", "kt_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), "profiles/%s_bench_kernel_stats.csv" % tag)
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), "profiles/%s_bench_under_rocprof.json" % tag)


def kernel_of(name):
    if "pt_primary_lists" in name or "pt_list_prep" in name:
        return "pt_primary_lists"
    if "pt_primary" in name:
        return "pt_primary"
    if "pt_megakernel" in name:
        return "pt_megakernel"
    return None


vals, bench = {}, None
for name in ("pmc_valu", "pmc_lds", "pmc_grbm", "pmc_fetch", "pmc_write"):
    f = os.path.join(src, name, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = kernel_of(row["Kernel_Name"])
            if k is None:
                continue
            d = vals.setdefault(k, {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    bench = json.load(open(os.path.join(src, name + ".json")))

spp = None
if bench:
    spp = bench["config"]["global_paths_per_pixel"] // 4
kern_ms = {k["name"]: k["ms"] for k in bench["roofline"]["kernels"]} if bench else {}
out = {"spp": spp, "tag": tag,
       "source": "profiles/%s_counters.json: rocprofv3 --pmc passes (one run per counter group) of `python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0`; "
                 "hbm_read_bytes = FETCH_SIZE [KB] x 1024 x 2 (MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B), hbm_write_bytes = WRITE_SIZE [KB] x 1024" % tag,
       "kernels": {}}
lines = ["# %s — rocprofv3 counters of the tracing kernels (ONE launch each: scene 21, 1920x1080, %s paths/pixel, depth 50)" % (tag, 4 * spp if spp else "?"), "",
         "Collected by tools/collect_r03.sh in separate --pmc runs (no tracing); kernel times are the HIP-event times of the counted run (`pmc_write` pass).", ""]
for k in ("pt_megakernel", "pt_primary", "pt_primary_lists"):
    v = vals.get(k)
    if not v:
        continue
    e = dict(v)
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        e["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0)
    if v.get("SQ_WAVE_CYCLES"):
        e["wait_any_share"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
        e["issue_stall_share"] = v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
        e["valu_active_share"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in v:
        e["hbm_read_bytes"] = v["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in v:
        e["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024.0
    ms = kern_ms.get(k)
    if ms:
        e["ms_in_counted_run"] = ms
        if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
            e["hbm_gb_per_s"] = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / (ms * 1e-3) * 1e-9
        if v.get("SQ_INSTS_VALU"):
            e["valu_issue_interval_ns_per_simd"] = ms * 1e6 * 1024 / v["SQ_INSTS_VALU"]
    out["kernels"][k] = e
    lines += ["## %s" % k, "", "| counter / derived | value |", "|---|---|"]
    for name in sorted(e):
        lines.append("| %s | %.6g |" % (name, e[name]))
    lines.append("")
if bench:
    rl = bench["roofline"]
    mk = out["kernels"].get("pt_megakernel", {})
    if mk.get("SQ_INSTS_VALU") and rl.get("rays_per_launch"):
        rays = rl["rays_per_launch"] * rl.get("launches", 1)
        mk["valu_wave_instructions_per_64_rays"] = mk["SQ_INSTS_VALU"] / (rays / 64.0)
        lines += ["pt_megakernel: %.0f VALU wave-instructions per 64 rays it traces (%.3e rays)" % (mk["valu_wave_instructions_per_64_rays"], rays), ""]
    pr = out["kernels"].get("pt_primary", {})
    if pr.get("SQ_INSTS_VALU"):
        pr["valu_wave_instructions_per_64_paths"] = pr["SQ_INSTS_VALU"] / (rl["render"]["paths"] / 64.0)
        lines += ["pt_primary: %.0f VALU wave-instructions per 64 paths" % pr["valu_wave_instructions_per_64_paths"], ""]
    tot = sum((e.get("hbm_read_bytes", 0) + e.get("hbm_write_bytes", 0)) for e in out["kernels"].values())
    out["hbm_bytes_per_frame"] = tot
    lines += ["HBM-side bytes per frame, all tracing kernels: %.4g GB (algorithmic: %.4g MB)" % (tot * 1e-9, rl.get("algorithmic_bytes", 0) * 1e-6), ""]
json.dump(out, open("profiles/%s_counters.json" % tag, "w"), indent=1, sort_keys=True)
open("profiles/%s_counters.md" % tag, "w").write("\n".join(lines))
print("\n".join(lines))

"""Condenses gpurun_out/profile_<tag>/ (tools/collect_profile.sh) into profiles/<tag>_*.{csv,md}."""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/profile_%s" % tag
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), "profiles/%s_bench_kernel_stats.csv" % tag)
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), "profiles/%s_bench_under_rocprof.json" % tag)
def collect(kernel):
    """counter sums over the dispatches whose kernel name contains `kernel` (each pass is one bench run of one launch)"""
    v, b = {}, None
    for name in ("pmc_valu", "pmc_lds", "pmc_grbm", "pmc_fetch", "pmc_write"):
        if not os.path.exists(os.path.join(src, name, "p_counter_collection.csv")):
            continue
        with open(os.path.join(src, name, "p_counter_collection.csv")) as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"]:
                    continue
                v[row["Counter_Name"]] = v.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        b = json.load(open(os.path.join(src, name + ".json")))
    return v, b


vals, bench = collect("pt_megakernel")
kms = bench["roofline"]["kernel_ms"]  # the megakernel's own time (roofline.kernel)
lines = ["# %s — SQ counters of pt_megakernel (one launch: 1080p, 256 paths/pixel, depth 50)" % tag, "",
         "Collected by tools/collect_profile.sh in separate --pmc passes (no tracing).", "", "| counter | value |", "|---|---|"]
for k in sorted(vals):
    lines.append("| %s | %.6g |" % (k, vals[k]))
wc = vals["SQ_WAVE_CYCLES"]
waves = vals["SQ_WAVES"]
derived = {
    "VALU instructions per wave": vals["SQ_INSTS_VALU"] / waves,
    "VALUUtilization = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64) (active lanes per VALU instruction)": vals["SQ_THREAD_CYCLES_VALU"] / (vals["SQ_ACTIVE_INST_VALU"] * 64),
    "VALUBusy (gfx94x formula) = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); sums the issue windows of the ~4 resident waves, so it exceeds 1 when the VALU is saturated": vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (vals["GRBM_GUI_ACTIVE"] / 8),
    "shader clock, GHz (GRBM_GUI_ACTIVE / 8 / kernel time)": vals["GRBM_GUI_ACTIVE"] / 8 / (kms * 1e-3) * 1e-9,
    "VALU active share of wave cycles (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)": vals["SQ_ACTIVE_INST_VALU"] / wc,
    "any-instruction active share of wave cycles": vals["SQ_ACTIVE_INST_ANY"] / wc,
    "resident waves per SIMD (SQ_WAVE_CYCLES*4 / (1024 SIMDs * kernel cycles at the measured ~2.28 GHz))": wc * 4 / (1024 * kms * 1e-3 * 2.28e9),
    "LDS bank-conflict share of LDS-array cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)": vals["SQ_LDS_BANK_CONFLICT"] / max(vals["SQ_LDS_IDX_ACTIVE"], 1),
    "issue-stall share (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": vals["SQ_WAIT_INST_ANY"] / wc,
    "waitcnt/barrier share (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": vals["SQ_WAIT_ANY"] / wc,
    "VALU wave-instruction issue interval per SIMD, ns (kernel time * 1024 SIMDs / SQ_INSTS_VALU)": kms * 1e6 * 1024 / vals["SQ_INSTS_VALU"],
}
rl = bench["roofline"]
if "wave_steps_per_launch" in rl:
    steps = rl["wave_steps_per_launch"] * rl.get("launches", 1)
    per_step = vals["SQ_INSTS_VALU"] / steps
    derived["VALU wave-instructions per ray step (SQ_INSTS_VALU / wave steps of the launch)"] = per_step
    if str(rl.get("scene_is_hit", "")).startswith("sphere tree"):
        # (a ray step of the tree variants ends when few lanes are still walking: steps are shorter and more than rays / 64)
        derived["ray steps per 64 rays of the launch"] = steps * 64.0 / max(rl["rays_per_launch"] * rl.get("launches", 1), 1.0)
        derived["VALU wave-instructions per 64 rays"] = vals["SQ_INSTS_VALU"] / max(rl["rays_per_launch"] * rl.get("launches", 1) / 64.0, 1.0)
    else:
        scan = rl["shared_ordinate_tests"] * 7 + (rl["spheres"] - rl["shared_ordinate_tests"]) * 9
        derived["of which the filter scan (7 per test in shared-ordinate words, 9 otherwise)"] = scan
        derived["VALU wave-instructions per ray step outside the scan"] = per_step - scan
    derived["SALU wave-instructions per ray step"] = vals["SQ_INSTS_SALU"] / steps
    derived["LDS wave-instructions per ray step"] = vals["SQ_INSTS_LDS"] / steps
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    # rocprofv3 derived metrics, in KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
    # (double it for wide coalesced reads); other widths are uncalibrated, so both figures are orders of magnitude
    derived["HBM-side read KB per launch (FETCH_SIZE as reported; x2 for the gfx950 correction)"] = vals["FETCH_SIZE"]
    derived["HBM-side write KB per launch (WRITE_SIZE)"] = vals["WRITE_SIZE"]
    derived["HBM-side GB/s of the launch ((2*FETCH_SIZE + WRITE_SIZE) KB / kernel time) against ~8000 GB/s peak"] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / (kms * 1e-3) * 1e-9
lines += ["", "| derived | value |", "|---|---|"]
for k, v in derived.items():
    lines.append("| %s | %.4g |" % (k, v))
lines += ["", "megakernel ms of the counted launch: %.2f; its algorithmic roofline fraction in that run: %.3f; of the whole render (all tracing kernels, every ray priced as the reference's scan): %.3f"
          % (kms, rl["frac"], rl["render"]["algorithmic_frac"])]
pv, _ = collect("pt_primary<")
if pv:
    paths = rl["render"]["paths"]
    lines += ["", "## pt_primary (camera rays against their pixel's candidate list + first radiance step) in the same launch", "",
              "| counter | value |", "|---|---|"]
    for k in sorted(pv):
        lines.append("| %s | %.6g |" % (k, pv[k]))
    lines += ["", "| derived | value |", "|---|---|",
              "| VALU wave-instructions per 64 paths | %.4g |" % (pv["SQ_INSTS_VALU"] / max(paths / 64.0, 1.0)),
              "| VALUUtilization (active lanes per VALU instruction) | %.4g |" % (pv["SQ_THREAD_CYCLES_VALU"] / (pv["SQ_ACTIVE_INST_VALU"] * 64)),
              "| primary pass ms (lists + pt_primary, HIP events) | %.4g |" % rl["render"]["primary_pass_ms"]]
open("profiles/%s_megakernel_counters.md" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

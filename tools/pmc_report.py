import re, sqlite3, sys
db, log = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
vals = {}
for name, v, n in c.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like '%megakernel%' group by counter_name"):
    vals[name] = v / n
rays = None
for line in open(log):
    m = re.search(r"rays ([0-9.e+]+)", line)
    if m: rays = float(m.group(1)); kms = float(re.search(r"kernel ([0-9.]+) ms", line).group(1))
steps = rays / 64.0
print("per dispatch:", {k: "%.3e" % v for k, v in vals.items()})
print("rays %.3e kernel %.1f ms  wave-steps(min) %.3e" % (rays, kms, steps))
print("VALU/wave-step %.0f  SALU/wave-step %.0f" % (vals["SQ_INSTS_VALU"] / steps, vals["SQ_INSTS_SALU"] / steps))
wc = vals["SQ_WAVE_CYCLES"]
print("wave-cycle split: active %.2f  wait_inst %.2f  wait_any %.2f" % (vals["SQ_ACTIVE_INST_ANY"] / wc, vals["SQ_WAIT_INST_ANY"] / wc, vals["SQ_WAIT_ANY"] / wc))
print("VALU issue cycles per SIMD: %.3e (x4 quad) over kernel cycles@2.4GHz %.3e" % (vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024, kms * 1e-3 * 2.4e9))

"""Static instruction mix of a kernel between its cycle-counter probes (s_memtime: the RPROF / SPROF points of the solve and setup
kernels), from the compiler's own assembly -- what an iteration's phases ISSUE, independent of any run.
    python tools/isa_mix.py daqp_amd/csrc/reg_kernel.hip 'k_ldp_regILi3ELi25ELb1E' [min instructions per region]
Classes: f64 = v_fma/v_fmac/v_mul/v_add_f64, acc = v_accvgpr_read/write (M lives in accumulation registers: every use is a copy),
lane = v_readlane / v_writelane / v_readfirstlane, lds = ds_*, vmem = global/flat/scratch, mfma, valu = every other vector instruction."""
import collections
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 120
with tempfile.NamedTemporaryFile(suffix=".s") as tf:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", src, "-o", tf.name],
                          stderr=subprocess.DEVNULL)
    text = open(tf.name).read().splitlines()
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
body = text[start:end]


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc"
    if re.match(r"v_(fma|fmac|mul|add|min|max)_f64", op): return "f64"
    if re.match(r"v_(readlane|writelane|readfirstlane)", op): return "lane"
    if op.startswith("ds_"): return "lds"
    if re.match(r"(global|flat|scratch|buffer)_", op): return "vmem"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return "other"


marks = [i for i, l in enumerate(body) if "s_memtime" in l] + [len(body)]
print(f"{text[start][:-1]}: {len(body)} lines, {len(marks) - 1} probes")
print("%7s %7s | %6s %6s %6s %6s %6s %6s %6s %6s | %6s" % ("from", "to", "f64", "acc", "lane", "valu", "salu", "lds", "vmem", "mfma", "all"))
for a, b in zip([0] + marks[:-1], marks):
    c = collections.Counter()
    for l in body[a:b]:
        t = l.strip()
        if t and t[0] not in ";." and not t.endswith(":"):
            c[cls(t.split()[0])] += 1
    n = sum(c.values())
    if n >= least:
        print("%7d %7d | %6d %6d %6d %6d %6d %6d %6d %6d | %6d" % (a, b, c["f64"], c["acc"], c["lane"], c["valu"], c["salu"], c["lds"], c["vmem"], c["mfma"], n))

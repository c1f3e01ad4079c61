"""profiles/r*_pmc_summary.json from the per-config PMC aggregates of tools/pmc_json.py:
   python tools/pmc_config_summary.py <out.json> pmc_raw_C2.json pmc_raw_C3.json pmc_raw_C4.json pmc_raw_C5.json [kernel_stats.csv]
(kernel_stats.csv: the rocprofv3 --kernel-trace --stats table of the same round profile -- the average duration of each setup kernel, so that a
per-launch record divides a kernel's bytes by THAT kernel's time)
Per config: the solve launch's HBM bytes (FETCH_SIZE in KiB units, doubled for 16-byte coalesced reads on gfx950 as
MI355X_MICROARCH.md prescribes, + WRITE_SIZE) and the issue-side counters that say what binds the kernel; the same for
the setup launch.  bench.py quotes `traffic_bytes_per_launch` and `binding` of the newest summary."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, library_stamp

out = sys.argv[1]
# the solve LAUNCH of C2 / C5 is two kernels: the image kernel (fp32 image of M, two waves per SIMD) and, right behind it, the full-register
# kernel for the problems it handed over -- their counters are summed ("+")
SOLVE = {"C2": r"k_ldp_reg<3, 25, true, 2>+k_ldp_reg<3, 25, true, 0>", "C3": r"k_ldp_reg<1, [68]", "C4": r"k_ldp_wg<4, false, true>+k_ldp_wg<4, false, false>", "C5": r"k_ldp_reg<3, 25, true, 2>+k_ldp_reg<3, 25, true, 0>"}
# C4's setup is three launches, each with its own record
SETUP_LAUNCHES = {"C4": [r"k_fact_wg", r"k_setup_m", r"k_setup<true, 4, true, false>"]}
SETUP = {"C2": r"k_setup_blk<4, 56||k_setup_fast<56", "C3": r"k_setup_tiny|k_setup_fast<16", "C4": r"k_setup<true", "C5": r"k_setup_blk<4, 56||k_setup_fast<56"}
N_SIMD, F_CLK = 1024, 2.4e9        # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak (MI355X_MICROARCH.md); SQ_* cycle counters tick every 4 cycles


def pick(raw, pats):
    for pat in pats.split("||"):       # alternatives in order of preference (the ordered setup kernel also runs, empty, behind k_setup_blk)
        names, tot = [], {}
        for part in pat.split("+"):    # kernels of ONE launch sequence: counters summed
            for k, v in raw.items():
                if re.search(part, k):
                    names.append(k)
                    for c, x in v.items():
                        tot[c] = tot.get(c, 0.0) + x["mean"]
                    break
        if names:
            return " + ".join(names), tot
    return None, {}


def kernel_ms(stats_csv):
    """{kernel name (as pmc_json.py prints it): average duration in ms} from a rocprofv3 kernel_stats.csv"""
    import csv
    out_ = {}
    if not stats_csv:
        return out_
    for r in csv.DictReader(open(stats_csv)):
        name = re.sub(r"^void daqp_amd::|\(.*$", "", r["Name"])
        out_[name] = float(r["AverageNs"]) * 1e-6
    return out_


def describe(c):
    g = lambda k: c.get(k, 0.0)
    d = {}
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        d["hbm_read_bytes"] = g("FETCH_SIZE") * 1024 * 2
        d["hbm_written_bytes"] = g("WRITE_SIZE") * 1024
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        d["issue"] = {"active_inst_any_over_wave_cycles": g("SQ_ACTIVE_INST_ANY") / wc, "wait_any_over_wave_cycles": g("SQ_WAIT_ANY") / wc,
                      "wait_inst_any_over_wave_cycles": g("SQ_WAIT_INST_ANY") / wc, "active_inst_valu_over_wave_cycles": g("SQ_ACTIVE_INST_VALU") / wc,
                      "wait_inst_lds_over_wave_cycles": g("SQ_WAIT_INST_LDS") / wc, "active_inst_lds_over_wave_cycles": g("SQ_ACTIVE_INST_LDS") / wc}
        d["instructions"] = {k: g("SQ_INSTS_" + k) for k in ("VALU", "SALU", "LDS", "VMEM")}
        d["waves"] = g("SQ_WAVES")
        if g("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_over_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    return d


# the build of the library these counters were taken with: bench.py quotes a summary only when the loaded library carries the same stamp
res = {"_stamp": library_stamp()}
stats = kernel_ms(next((a for a in sys.argv[2:] if a.endswith(".csv")), None))
for path in [a for a in sys.argv[2:] if not a.endswith(".csv")]:
    cfg = re.search(r"(C\d)", os.path.basename(path)).group(1)
    raw = json.load(open(path))
    ks, cs = pick(raw, SOLVE[cfg])
    kt, ct = pick(raw, SETUP[cfg])
    s, t = describe(cs), describe(ct)
    entry = {"batch": CONFIGS[cfg]["per_gpu"], "solve_kernel": ks, "setup_kernel": kt, "solve": s, "setup": t}
    if cfg in SETUP_LAUNCHES:
        entry["setup_launches"] = []
        for pat in SETUP_LAUNCHES[cfg]:
            kn, cn = pick(raw, pat)
            if kn:
                entry["setup_launches"].append(dict(describe(cn), kernel=kn, avg_ms_kernel_trace=stats.get(kn)))
    if "hbm_read_bytes" in s:
        entry["traffic_bytes_per_launch"] = s["hbm_read_bytes"] + s["hbm_written_bytes"]
    if "issue" in s:
        i = s["issue"]
        insts = sum(s["instructions"].values())
        # the roof that binds these kernels: instruction issue.  The VALU pipe is the one resource a SIMD's waves cannot share:
        # busy_valu = cycles in which a VALU instruction was issuing, summed over waves; with every SIMD's vector pipe busy back to
        # back on this instruction mix the launch would take busy_valu / (SIMDs x clock).  busy_any counts every instruction class
        # (scalar, LDS, memory issue overlap with VALU only across waves): the floor of a design with ONE wave per SIMD.
        busy = 4.0 * cs.get("SQ_ACTIVE_INST_ANY", 0.0)
        busy_valu = 4.0 * cs.get("SQ_ACTIVE_INST_VALU", 0.0)
        entry["issue"] = {"wave_instructions": insts, "cycles_per_instruction": busy / insts if insts else None,
                          "attainable_ms": busy_valu / (N_SIMD * F_CLK) * 1e3,
                          "one_wave_per_simd_floor_ms": busy / (N_SIMD * F_CLK) * 1e3,
                          "note": "attainable = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x 2.4 GHz): every SIMD's vector pipe issuing back to back on the measured "
                                  "instruction mix; one_wave_per_simd_floor = the same with SQ_ACTIVE_INST_ANY (nothing overlaps inside one wave)"}
        entry["binding"] = {"resource": "instruction issue + dependent-operation latency (one wave per SIMD holds the iterate in registers)" if cfg != "C4"
                            else "per-CU memory pipeline (scan / row-cache reads at ~30 B/clk/CU) + the master wave's substitution chains",
                            "frac": i["active_inst_any_over_wave_cycles"], "frac_is": "SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES of the solve launch",
                            "wait_any_frac": i["wait_any_over_wave_cycles"], "valu_frac": i["active_inst_valu_over_wave_cycles"],
                            "wave_instructions_per_qp": insts / entry["batch"]}
    res[cfg] = entry
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

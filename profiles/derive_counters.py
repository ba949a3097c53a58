"""profiles/derive_counters.py OUTDIR -- the derived figures of profiles/rNN/README.md from one run of tools/profile_round.sh:
per config the kernel ms (HIP events of bench.py under the profiler), cycles, VALU count, cycles per VALU instruction per
SIMD, active lanes, waves per SIMD, LDS conflicts, HBM bytes and GB/s.  Prints JSON.  Formulas as in the README."""
import json, re, sys
out = sys.argv[1]
def counters(path):
    d = {}
    for ln in open(path):
        m = re.match(r"(\w+)\s+n=\s*\d+\s+mean=([\d.e+-]+)", ln)
        if m: d[m.group(1)] = float(m.group(2))
    return d
def kernel_ms(tag):
    return json.loads(open(f"{out}/kt_{tag}.json").read().strip().splitlines()[-1])["kernel_ms"]
rows = {}
for name, summ, tag in (("config3", "config3_pmc_summary.txt", "config3"), ("config2", "config2_pmc_summary.txt", "config2"),
                        ("config5_band_of_8", "config5_band8_pmc_summary.txt", "config5_band_of_8")):
    c = counters(f"{out}/{summ}")
    ms = kernel_ms(tag)
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    b = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    rows[name] = dict(ms=ms, cyc=cyc, valu=c["SQ_INSTS_VALU"], cpv=cyc * 1024 / c["SQ_INSTS_VALU"], util=2 / (cyc * 1024 / c["SQ_INSTS_VALU"]),
                      lanes=c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]), wps=4 * c["SQ_WAVE_CYCLES"] / (cyc * 1024),
                      lds=(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else 0.0,
                      fetch_kb=c["FETCH_SIZE"], write_kb=c["WRITE_SIZE"], bytes=b, gbs=b / (ms * 1e-3) / 1e9,
                      salu=c["SQ_INSTS_SALU"], branch=c["SQ_INSTS_BRANCH"], lds_insts=c["SQ_INSTS_LDS"])
print(json.dumps(rows, indent=1))

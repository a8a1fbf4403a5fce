"""Per-kernel JSON summary of a set of rocprofv3 counter passes (one directory per --pmc group, each also holding the
kernel trace of that pass).  Units and corrections as MI355X_MICROARCH.md prescribes: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* /
SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, GRBM_GUI_ACTIVE sums the 8 XCDs; FETCH_SIZE / WRITE_SIZE are
KB, FETCH_SIZE reads half of a wide coalesced stream on gfx950 (x 2 = upper bound; reads narrower than 16 B per lane are
uncalibrated and are reported raw as well).

    python3 benchmarks/pmc_kernels.py <title> <name-substring>[,<name-substring>...] <pass-dir>... > profiles/roundN/x_pmc.json
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    title, names, dirs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    vals = {n: collections.defaultdict(list) for n in names}
    dur = {n: [] for n in names}
    meta = {n: {} for n in names}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    for n in names:
                        if n in r["Kernel_Name"]:
                            vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
                            meta[n] = {"grid_threads": int(r.get("Grid_Size", 0) or 0), "workgroup": int(r.get("Workgroup_Size", 0) or 0),
                                       "vgpr": int(r.get("VGPR_Count", 0) or 0), "lds_bytes": int(r.get("LDS_Block_Size", 0) or 0)}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    for n in names:
                        if n in r["Kernel_Name"]:
                            dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {"title": title, "passes": [os.path.basename(os.path.normpath(d)) for d in dirs], "kernels": []}
    for n in names:
        c = {k: sum(v) / len(v) for k, v in vals[n].items()}
        if not c:
            continue
        us = sum(dur[n]) / len(dur[n]) / 1e3 if dur[n] else None
        k = {"kernel": n, "launches_profiled": max(len(v) for v in vals[n].values()), "avg_duration_us_under_pmc": us,
             "counters_per_launch": c}
        k.update(meta[n])
        dv = {}
        if "GRBM_GUI_ACTIVE" in c:
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            simd_cycles = cycles * 1024
            if us:
                dv["clock_ghz"] = cycles / (us * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                dv["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
            if "SQ_ACTIVE_INST_VALU" in c:
                dv["valu_busy_fraction"] = c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles
            if "SQ_WAVE_CYCLES" in c:
                dv["waves_resident_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / simd_cycles
            if "SQ_ACTIVE_INST_LDS" in c:
                dv["lds_issue_busy_fraction"] = c["SQ_ACTIVE_INST_LDS"] * 4 / simd_cycles
            if "SQ_ACTIVE_INST_SCA" in c:
                dv["salu_busy_fraction"] = c["SQ_ACTIVE_INST_SCA"] * 4 / simd_cycles
        if "SQ_WAVE_CYCLES" in c:
            for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
                if w in c:
                    dv[w.lower() + "_share_of_wave_cycles"] = c[w] / c["SQ_WAVE_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
            dv["lds_bank_conflict_share_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            fetch, write = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
            dv.update({"fetch_bytes_raw": fetch, "fetch_bytes_x2_upper_bound": 2 * fetch, "write_bytes": write,
                       "hbm_traffic_bytes": 2 * fetch + write})
            if us:
                dv["hbm_gb_per_s"] = (2 * fetch + write) / (us * 1e-6) / 1e9
        k["derived"] = dv
        out["kernels"].append(k)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

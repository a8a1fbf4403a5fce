"""MFMA utilisation of the attention kernels from a rocprofv3 counter pass.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
        --output-format csv -d /tmp/pmc -o attn -- python3 benchmarks/attention_pmc_driver.py
    (separate passes, same program, for --pmc FETCH_SIZE and --pmc WRITE_SIZE: -o fetch / -o write)
    python3 benchmarks/pmc_attention.py /tmp/pmc/attn_counter_collection.csv /tmp/pmc/attn_kernel_trace.csv \
        [/tmp/pmc/fetch_counter_collection.csv /tmp/pmc/write_counter_collection.csv] > profiles/roundN/attention_pmc_v2.json

FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream (MI355X_MICROARCH.md, HBM),
so `hbm_traffic_bytes` = 2 x FETCH_SIZE + WRITE_SIZE.

SQ_VALU_MFMA_BUSY_CYCLES counts cycles (summed over SIMDs) in which the matrix pipe is busy; GRBM_GUI_ACTIVE counts
active cycles summed over the 8 XCDs.  utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)."""
import collections
import csv
import json
import sys


def main():
    counters, trace = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    dur = collections.defaultdict(list)
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            if "attn_" in n and "anonymous" in n:
                grid = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)))
                wg = r.get("Workgroup_Size") or str(int(r.get("Workgroup_Size_X", 0)) * int(r.get("Workgroup_Size_Y", 1)) * int(r.get("Workgroup_Size_Z", 1)))
                key = (n[n.index("attn_"):].split("(")[0], grid, wg)
                dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in [counters] + extra:
        with open(path) as fh:
            for r in csv.DictReader(fh):
                n = r["Kernel_Name"]
                if "attn_" in n and "anonymous" in n:
                    key = (n[n.index("attn_"):].split("(")[0], r["Grid_Size"], r["Workgroup_Size"])
                    vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = []
    for key, cs in sorted(vals.items()):
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        d = dur.get(key, [])
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            avg["hbm_traffic_bytes"] = 2 * 1024 * avg["FETCH_SIZE"] + 1024 * avg["WRITE_SIZE"]
        rec = {"kernel": key[0], "grid_threads": int(key[1]), "workgroup": int(key[2]), "launches": len(next(iter(cs.values()))),
               "avg_duration_us_under_pmc": round(sum(d) / len(d) / 1e3, 1) if d else None, "counters": avg}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
            simd_cycles = avg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            rec["mfma_utilisation"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4)
            if "split3" in key[0]:       # bf16 matrix pipe: 2.5 PFLOP/s dense when busy every cycle at the peak clock
                rec["executed_bf16_tflops_at_peak_clock"] = round(rec["mfma_utilisation"] * 2500.0, 1)
            else:
                rec["equivalent_fp32_mfma_tflops"] = round(rec["mfma_utilisation"] * 157.3, 1)
            if "SQ_ACTIVE_INST_VALU" in avg:
                rec["valu_busy"] = round(avg["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, 4)
            if d:
                rec["clock_ghz"] = round(avg["GRBM_GUI_ACTIVE"] / 8.0 / (sum(d) / len(d)), 3)
        out.append(rec)
    print(json.dumps({"note": __doc__.strip().splitlines()[-2] + " " + __doc__.strip().splitlines()[-1], "kernels": out}, indent=1))


if __name__ == "__main__":
    main()

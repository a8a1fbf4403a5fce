"""profiles/roundN/chamfer_mfma_pmc.json from rocprofv3 counter passes of benchmarks/chamfer_only.py (algorithm mfma).

    cd /tmp && export TMPDIR=/tmp
    for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
               "FETCH_SIZE" "WRITE_SIZE"; do
        rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_cm_$i -o p -- python3 benchmarks/chamfer_only.py 4 16384 16384 mfma
    done
    python3 benchmarks/pmc_chamfer_mfma.py gpurun_out/pmc_cm_* > profiles/roundN/chamfer_mfma_pmc.json

Separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass); FETCH_SIZE is in KB and
reports half of a wide coalesced read stream on gfx950, hence x 2.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    per = {"chamfer_mfma_kernel": collections.defaultdict(list), "chamfer_prep_kernel": collections.defaultdict(list)}
    dur = collections.defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    for k in per:
                        if k in r["Kernel_Name"]:
                            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "*kernel_trace.csv")):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    for k in per:
                        if k in r["Kernel_Name"]:
                            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    avg = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}
    m = avg["chamfer_mfma_kernel"]
    p = avg.get("chamfer_prep_kernel", {})
    B, n = 32, 16384
    pairs = 2.0 * B * n * n
    dur_us = {k: sum(v) / len(v) / 1e3 for k, v in dur.items()}
    fetch = (m.get("FETCH_SIZE", 0.0) + p.get("FETCH_SIZE", 0.0)) * 1024
    write = (m.get("WRITE_SIZE", 0.0) + p.get("WRITE_SIZE", 0.0)) * 1024
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0
    out = {
        "kernel": ("chamfer_prep_kernel + chamfer_mfma_kernel" if p else "chamfer_mfma_kernel (coordinates split in the kernel: no prep pass)")
                  + " (B=32, n=m=16384, one launch = both directions)",
        "command": "rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 benchmarks/chamfer_only.py 4 16384 16384 mfma "
                   "(one pass per counter group: SQ/GRBM, FETCH_SIZE, WRITE_SIZE)",
        "avg_duration_us_under_pmc": dur_us,
        "per_launch": avg,
        "derived": {
            "fetch_bytes_raw": fetch, "fetch_bytes_gfx950_corrected_x2": 2 * fetch, "write_bytes": write,
            "hbm_traffic_bytes_corrected": 2 * fetch + write,
            "algorithmic_bytes": 20.0 * B * 2 * n,
            "note_traffic": ("prep + search kernels together.  The prep pass writes 48 bytes of bf16 operand pieces per "
                             "candidate (50 MB for both clouds) and the search reads them back through L2 (each XCD holds the "
                             "clouds its workgroups work on); dist/idx are 8.4 MB.") if p else
                            ("search kernel alone: it reads 12 bytes per candidate per (cloud, query tile) through its XCD's L2 and "
                             "splits them itself; dist/idx are 8.4 MB."),
            "mfma_instructions": pairs / 1024 * 2,
            "mfma_busy_fraction": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024),
            "valu_busy_fraction": m["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * 1024),
            "valu_wave_instructions": m["SQ_INSTS_VALU"], "valu_instr_per_pair": m["SQ_INSTS_VALU"] * 64 / pairs,
            "clock_ghz": cycles / (dur_us["chamfer_mfma_kernel"] * 1e3),
            "waves_resident_per_simd": m["SQ_WAVE_CYCLES"] * 4 / (cycles * 1024),
            "note_pipes": "SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_32x32x16_bf16 per SIMD; SIMD-cycles = "
                          "GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs; SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles.",
        },
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

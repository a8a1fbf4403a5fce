"""Per-kernel averages of a rocprofv3 --pmc pass (counter_collection.csv), optionally joined with kernel durations.

    python3 benchmarks/pmc_summary.py <counter_collection.csv> [kernel_trace.csv] [name filter]
"""
import collections
import csv
import sys


def main():
    counters = sys.argv[1]
    trace = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".csv") else None
    flt = sys.argv[-1] if len(sys.argv) > 2 and not sys.argv[-1].endswith(".csv") else ""
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(counters) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            if flt in n:
                vals[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    if trace:
        with open(trace) as fh:
            for r in csv.DictReader(fh):
                n = r["Kernel_Name"]
                if flt in n:
                    dur[n[:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for n, cs in vals.items():
        d = dur.get(n)
        print(n, f"dispatches={len(next(iter(cs.values())))}", f"avg_us={sum(d) / len(d) / 1e3:.1f}" if d else "")
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()

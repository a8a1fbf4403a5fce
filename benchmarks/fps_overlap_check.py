"""Does the side-stream farthest point sampling really run underneath the first kNN / edge-conv stage?

Reads a rocprofv3 kernel trace of bench.py and reports, for every fps_kernel launch of the last steps, which kernels
ran (partly) at the same time and for how long the sampling was the only kernel on the GPU.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o step -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline
    python3 benchmarks/fps_overlap_check.py /tmp/prof/.../step_kernel_trace.csv
"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    fps = [i for i, r in enumerate(rows) if "fps_kernel" in r[2]][-6:]
    for i in fps:
        s, e, name, q = rows[i]
        others = [(max(s, a), min(e, b), n, qq) for a, b, n, qq in rows if n is not name and b > s and a < e and (a, b) != (s, e)]
        covered = 0
        cur = s
        for a, b, n, qq in sorted(others):
            if b > cur:
                covered += b - max(a, cur)
                cur = b
        print(f"fps_kernel on queue {q}: {(e - s) / 1e3:8.1f} us, with another kernel running for {covered / 1e3:8.1f} us of it "
              f"({len(others)} kernels: {sorted(set(n.split('(')[0][-40:] for _, _, n, _ in others))[:6]})")


if __name__ == "__main__":
    main()

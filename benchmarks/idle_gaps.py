"""GPU idle time inside one steady-state train step, from a rocprofv3 kernel trace of bench.py: the union of all kernel
intervals (all queues) between two optimizer updates, the idle remainder, and where the largest gaps are.

    python3 benchmarks/idle_gaps.py <kernel_trace.csv> [marker substring]
"""
import csv
import sys


def main():
    marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_update_kernel"
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    for a, b in zip(marks[-4:-1], marks[-3:]):
        step = rows[a + 1:b + 1]
        t0, t1 = rows[a][1], rows[b][1]
        busy, cur, gaps = 0, t0, []
        prev = rows[a][2]
        for s, e, n in step:
            if s > cur:
                gaps.append((s - cur, prev, n))
            if e > cur:
                busy += e - max(s, cur)
                cur = e
                prev = n
        print(f"step: wall {(t1 - t0) / 1e6:7.3f} ms, some kernel running {busy / 1e6:7.3f} ms, idle {(t1 - t0 - busy) / 1e6:6.3f} ms "
              f"in {len(gaps)} gaps ({sum(1 for g in gaps if g[0] > 20000)} longer than 20 us)")
        # where: idle time by tenths of the step
        tenth = [0.0] * 10
        cur = t0
        for s, e, n in step:
            if s > cur:
                tenth[min(9, int(10 * (cur - t0) / (t1 - t0)))] += (s - cur) / 1e6
            cur = max(cur, e)
        print("   idle ms per tenth of the step:", " ".join(f"{x:5.2f}" for x in tenth))
        for g, p, n in sorted(gaps, reverse=True)[:6]:
            print(f"   {g / 1e3:7.1f} us between {p.split('(')[0][-50:]} and {n.split('(')[0][-50:]}")


if __name__ == "__main__":
    main()

"""Steady-state per-step kernel breakdown from a rocprofv3 kernel trace of bench.py.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o step -- python3 bench.py --steps 10 --warmup 5
    python3 benchmarks/trace_breakdown.py /tmp/prof/step_kernel_trace.csv --steps 8 > profiles/roundN/step_breakdown.txt

The warm-up steps contain TunableOp / MIOpen search kernels, so the whole-run --stats summary is
not a picture of a train step.  This tool keeps only the last `--steps` steps, delimited by the last
launch of each step (the optimizer's final multi-tensor kernel), and groups kernels by family.
"""
import argparse
import collections
import csv
import re

FAMILIES = [
    ("ptops: chamfer", r"chamfer_"),
    ("ptops: knn", r"knn_kernel"),
    ("ptops: top-k rows", r"topk_rows_kernel"),
    ("ptops: fps", r"fps_kernel"),
    ("ptops: edge-conv", r"edge_[a-z_]+kernel"),
    ("ptops: gather/index", r"(?<!scatter_)gather_|index_points"),      # not ATen's _scatter_gather_elementwise_kernel
    ("ptops: attention fwd", r"attn_fwd_"),
    ("ptops: attention bwd", r"attn_bwd_|attn_dq_reduce|attn_dkv_reduce"),
    ("ptops: layernorm", r"ln_fwd_kernel|ln_bwd_kernel"),
    ("ptops: colsum", r"colsum_"),
    ("ptops: batchnorm rows", r"bn_[a-z_]+kernel"),
    ("ptops: GEMM (fp32 on the bf16 pipe)", r"gemm_split3|wgrad_reduce"),
    ("attention bwd (aten)", r"fmha|attention|AttentionBackward|efficient_attention"),
    ("GEMM (hipBLASLt/rocBLAS/CK)", r"^Cijk_|gemm|Gemm"),
    ("conv (MIOpen)", r"conv|Conv|igemm|miopen"),
    ("norm", r"layer_norm|LayerNorm|group_norm|GroupNorm|RowwiseMoments|batch_norm|BatchNorm"),
    ("reduce", r"reduce_kernel|Reduce"),
    ("optimizer", r"multi_tensor_apply|adamw_"),
    ("copy / cat / fill", r"CatArray|copy|Copy|fill|Fill|memcpy|memset"),
    ("sort / scatter / index", r"sort|Sort|scatter|index_|gather"),
    ("elementwise", r"elementwise|vectorized"),
]


def family(name):
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--marker", default="adamw_update_kernel", help="kernel launched a fixed number of times per step")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--timeline", action="store_true",
                    help="also: how much of the wall is covered by >= 1 / >= 2 kernels, the idle gaps, busy time per queue")
    a = ap.parse_args()

    rows = []
    with open(a.trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    # markers per step: smallest period such that the kernel count between successive groups is constant
    per = 1
    for cand in (1, 2, 3, 4, 6, 8):
        if len(marks) < cand * (a.steps + 1):
            break
        tail = marks[-cand * (a.steps + 1):][::cand]
        gaps = {tail[i + 1] - tail[i] for i in range(len(tail) - 1)}
        if len(gaps) == 1:
            per = cand
            break
    tail = marks[-per * (a.steps + 1):][::per]
    lo, hi = tail[0], tail[-1]
    win = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / a.steps / 1e6
    fam_t, fam_n = collections.Counter(), collections.Counter()
    ker_t, ker_n = collections.Counter(), collections.Counter()
    for s, e, n, _q in win:
        f = family(n)
        fam_t[f] += e - s
        fam_n[f] += 1
        short = re.sub(r"<.*", "", n)[:90]
        ker_t[short] += e - s
        ker_n[short] += 1
    busy = sum(fam_t.values()) / a.steps / 1e6
    print(f"steps analysed: {a.steps} (marker '{a.marker}' x{per}/step, {len(win) // a.steps} launches/step)")
    print(f"wall per step {wall:.2f} ms, kernel-busy per step {busy:.2f} ms")
    print(f"{'family':34s} {'ms/step':>9s} {'%busy':>7s} {'launches/step':>14s}")
    for f, t in fam_t.most_common():
        print(f"{f:34s} {t / a.steps / 1e6:9.3f} {100 * t / sum(fam_t.values()):7.1f} {fam_n[f] / a.steps:14.1f}")
    print(f"\ntop {a.top} kernels")
    for k, t in ker_t.most_common(a.top):
        print(f"{t / a.steps / 1e6:9.3f} ms  x{ker_n[k] / a.steps:7.1f}  {k}")
    if a.timeline:
        timeline(win, rows[lo][0], rows[hi][0], a.steps)


def timeline(win, t0, t1, steps):
    """Coverage of the wall by running kernels: parallel graph branches show as time with >= 2 kernels in flight, launch
    overhead as idle gaps between the end of one kernel and the start of the next."""
    ev = []
    for s, e, _n, _q in win:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, cover = 0, t0, collections.Counter()
    for t, d in ev:
        cover[min(depth, 3)] += max(0, min(t, t1) - last)
        last = max(last, min(t, t1))
        depth += d
    print("\ntimeline per step (ms): " + ", ".join(
        f"{k if k < 3 else '>=3'} in flight {v / steps / 1e6:.2f}" for k, v in sorted(cover.items())))
    # idle gaps: maximal intervals with nothing in flight
    gaps, end = [], None
    for s, e, n, _q in win:
        if end is not None and s > end[0]:
            gaps.append((s - end[0], end[1], n))
        if end is None or e > end[0]:
            end = (e, n)
    buckets = collections.Counter()
    for g, _a, _b in gaps:
        key = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else ">=20us"
        buckets[key] += g
        buckets[key + " n"] += 1
    print("idle gaps per step: " + ", ".join(
        f"{k}: {buckets[k] / steps / 1e6:.2f} ms in {buckets[k + ' n'] / steps:.0f}" for k in ("<2us", "2-5us", "5-20us", ">=20us")))
    print("largest idle gaps (us, after -> before):")
    for g, before, after in sorted(gaps, reverse=True)[:12]:
        print(f"{g / 1e3:9.1f}  {re.sub(r'<.*', '', before)[:60]}  ->  {re.sub(r'<.*', '', after)[:60]}")
    per_q = collections.Counter()
    for s, e, _n, q in win:
        per_q[q] += e - s
    print("kernel time per queue (ms/step): " + ", ".join(f"{q or '?'}: {t / steps / 1e6:.2f}" for q, t in per_q.most_common()))


if __name__ == "__main__":
    main()

"""Is the replay of the graphed train step bound by the HOST's launch of the graph?

For one model the graphed step is timed twice: the host time until `step(...)` returns (launch of the graph + the small
copies around it, no synchronisation) and the time per step with one synchronisation at the end.  Run once per setting of
DEBUG_CLR_GRAPH_PACKET_CAPTURE (the runtime reads it at start-up):

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python3 benchmarks/graph_launch_cost.py img     # what the package sets
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python3 benchmarks/graph_launch_cost.py img     # the runtime's default

With packet capture on, memset nodes do not replay reliably (imgadapointr_amd/__init__.py), so that run's RESULTS are not
checked (the probe is overridden, verify off): only its timing is of interest.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
setting = os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402

import bench  # noqa: E402
from imgadapointr_amd.utils import graph_runner  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "img"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    if setting != "0":
        graph_runner._memset_probe[(dev.type, dev.index)] = True          # timing only
    model, cfg, spec, step, batch, B = bench.make_workload(which, dev)
    g = graph_runner.GraphedTrainStep(step, warmup=3, verify=False)
    partial, gt, extra = batch
    for _ in range(8):
        g(partial, gt, extra)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t = time.perf_counter()
        g(partial, gt, extra)
        host.append(time.perf_counter() - t)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    first = [round(x * 1e3, 3) for x in host[:6]]       # right after a synchronisation: no back-pressure from a full queue
    host.sort()
    out = {"model": which, "packet_capture": setting, "graphed": bool(g.enabled and g.replays >= steps), "steps": steps,
           "ms_per_step": round(total / steps * 1e3, 3), "host_ms_per_call_median": round(host[len(host) // 2] * 1e3, 3),
           "host_ms_per_call_min": round(host[0] * 1e3, 3), "enqueue_all_ms_per_step": round(t_enq / steps * 1e3, 3),
           "host_ms_first_calls": first,
           "nodes_note": "host time ~ step time => the launch of the graph is the bound"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

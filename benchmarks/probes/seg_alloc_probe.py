"""Is the seg train step's forward bit-reproducible when only the allocator's state differs (another model alive)?
If not: which module output differs first (checksums by forward hooks at the step where the losses part)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

import parity_checks as pc  # noqa: E402
from imgadapointr_amd.tools import builder, runner  # noqa: E402

import struct  # noqa: E402


def rng_offset():
    st = torch.cuda.get_rng_state()
    return struct.unpack("<q", bytes(st[-8:].tolist()))[0], struct.unpack("<q", bytes(st[-16:-8].tolist()))[0]


OFFSETS = []
B = 2
CASE = sys.argv[1] if len(sys.argv) > 1 else "seg"


def batch(i):
    g = torch.Generator().manual_seed(100 + i)
    partial = (torch.rand(B, 2048, 3, generator=g) - 0.5).cuda()
    gt = (torch.rand(B, 16384, 3, generator=g) - 0.5).cuda()
    extra = [torch.rand(B, 3, 224, 224, generator=g).cuda()]
    if CASE == "seg":
        cls = torch.zeros(B, 16)
        cls[torch.arange(B), torch.arange(B) % 16] = 1.0
        extra.append(cls.cuda())
    return partial, gt, tuple(extra)


def run(n_steps, junk=None, graphed=False):
    from imgadapointr_amd.utils.graph_runner import GraphedTrainStep
    model, cfg, spec = pc.build_trainstep_model(CASE, "cuda")
    model.train()
    step = runner.TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
    sums = []
    if graphed:
        return run_graphed(model, step, spec, n_steps)

    def hook(name):
        def f(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for j, o in enumerate(outs):
                if isinstance(o, torch.Tensor) and o.is_floating_point():
                    sums[-1].append((f"{name}[{j}]", o.detach().double().sum().item(), o.detach().double().abs().sum().item()))
        return f
    for name, mod in model.named_modules():
        if name:
            mod.register_forward_hook(hook(name))
    model.base_model.segmentator.dp1.register_forward_pre_hook(lambda m, a: OFFSETS.append(("plain dp1", rng_offset())))
    losses = []
    for i in range(n_steps):
        if junk is not None and i == junk[0]:
            junk[1]()
        sums.append([])
        torch.manual_seed(1000 + i)
        partial, gt, extra = batch(i)
        OFFSETS.append(("plain step start", i, rng_offset()))
        l = step(partial, gt, extra, spec["epoch"])
        losses.append((l[0].item(), l[1].item()))
    return (model, step), losses, sums


def run_graphed(model, step, spec, n_steps):
    """As the test does; module-output checksums of the eager steps that follow the capture (the check's A and B)."""
    from imgadapointr_amd.utils.graph_runner import GraphedTrainStep
    gs = GraphedTrainStep(step, warmup=3, verify=True)
    sums = []
    model.base_model.segmentator.dp1.register_forward_pre_hook(
        lambda m, a: None if torch.cuda.is_current_stream_capturing() else OFFSETS.append(("graphed-run eager dp1", rng_offset())))

    def hook(name):
        def f(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for j, o in enumerate(outs):
                if isinstance(o, torch.Tensor) and o.is_floating_point():
                    sums[-1].append((f"{name}[{j}]", o.detach().double().sum().item(), o.detach().double().abs().sum().item()))
        return f
    orig = gs._capture

    def capture_then_hook(*a):
        e = orig(*a)
        for name, mod in model.named_modules():
            if name:
                mod.register_forward_hook(hook(name))
        return e
    gs._capture = capture_then_hook
    real_step = gs.step

    class Counting:
        def __getattr__(self, k):
            return getattr(real_step, k)

        def __setattr__(self, k, v):
            setattr(real_step, k, v)

        def __call__(self, *a, **kw):
            sums.append([])
            OFFSETS.append(("graphed inner step start", rng_offset(), "capturing", torch.cuda.is_current_stream_capturing()))
            return real_step(*a, **kw)
    gs.step = Counting()
    losses = []
    for i in range(n_steps):
        torch.manual_seed(1000 + i)
        partial, gt, extra = batch(i)
        l = gs(partial, gt, extra, spec["epoch"])
        losses.append((l[0].item(), l[1].item()))
    print("   graphed: enabled", gs.enabled, "replays", gs.replays, gs.verified, flush=True)
    return (model, gs), losses, sums


def main():
    steps = 5
    if len(sys.argv) > 2 and sys.argv[2] == "graph":
        with pc._Pinned():
            keep0, base, sums0 = run(steps)
            keep1, lg, sums_g = run(steps, graphed=True)       # the eager model stays alive, as in the test
            for o in OFFSETS:
                print("   ", o)
            print("graphed (verify)   :", ["%.1e" % abs(a[0] - b[0]) for a, b in zip(base, lg)], flush=True)
            # sums_g: one list per eager call of the inner step: 3 warm-up steps (no hooks yet: empty), the captured call (empty:
            # hooks are installed after it), then the check's A and B
            filled = [s for s in sums_g if s]
            print("   eager calls with checksums:", len(filled), flush=True)
            for tag, s in zip(("A", "B"), filled):
                for (n0, a0, b0), (n1, a1, b1) in zip(sums0[3], s):
                    if n0 != n1 or a0 != a1 or b0 != b1:
                        print(f"{tag}: first differing module output vs the plain run's step 3: {n0} / {n1}: {a0!r} vs {a1!r}", flush=True)
                        break
                else:
                    print(f"{tag}: every module output checksum equals the plain run's", flush=True)
        return
    with pc._Pinned():
        keep0, base, sums0 = run(steps)
        del keep0
        torch.cuda.empty_cache()
        _, again, _ = run(steps)
        print("fresh second run   :", ["%.1e" % abs(a[0] - b[0]) for a, b in zip(base, again)], flush=True)
        keep1, l1, sums1 = run(steps)                         # kept alive below
        print("third run          :", ["%.1e" % abs(a[0] - b[0]) for a, b in zip(base, l1)], flush=True)
        _, l2, sums2 = run(steps)                             # with another model + optimizer state alive
        print("with a model alive :", ["%.1e" % abs(a[0] - b[0]) for a, b in zip(base, l2)], flush=True)
        hold = []

        def fragment():                                       # change the allocator's state in the middle of a run
            hold.extend(torch.empty(int(x), device="cuda") for x in (3e5, 7e6, 1.3e6, 5e7, 999, 2.5e6) for _ in range(7))
            del hold[::2]
        _, l3, sums3 = run(steps, junk=(3, fragment))
        print("fragmented at step3:", ["%.1e" % abs(a[0] - b[0]) for a, b in zip(base, l3)], flush=True)
        for tag, s in (("alive", sums2), ("fragmented", sums3)):
            done = False
            for i in range(steps):
                for (n0, a0, b0), (n1, a1, b1) in zip(sums0[i], s[i]):
                    if n0 != n1 or a0 != a1 or b0 != b1:
                        print(f"{tag}: first differing module output at step {i}: {n0} / {n1}: {a0!r} vs {a1!r}", flush=True)
                        done = True
                        break
                if done:
                    break
            if not done:
                print(f"{tag}: every module output checksum equal", flush=True)


if __name__ == "__main__":
    main()

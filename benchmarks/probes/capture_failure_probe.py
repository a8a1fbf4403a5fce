"""What state does a FAILED stream capture leave behind in this PyTorch / ROCm stack (generator, allocator, stream)?"""
import traceback

import torch


def attempt(name, fn):
    try:
        fn()
        print(f"  {name}: ok", flush=True)
        return True
    except Exception as exc:      # noqa: BLE001
        print(f"  {name}: {type(exc).__name__}: {str(exc).splitlines()[0][:150]}", flush=True)
        return False


def failing_capture(kind):
    dev = torch.device("cuda:0")
    x = torch.rand(64, 64, device=dev)
    g = torch.cuda.CUDAGraph()
    before = torch.cuda.current_stream(dev)
    try:
        with torch.cuda.graph(g):
            y = torch.rand(64, 64, device=dev) + x          # registers the generator with the capture
            if kind == "item":
                y.sum().item()
            elif kind == "pin":
                torch.empty(1 << 20).pin_memory()
            elif kind == "sync":
                torch.cuda.synchronize()
            elif kind == "raise":
                raise RuntimeError("python error inside the capture")
    except Exception as exc:      # noqa: BLE001
        print(f"capture[{kind}] failed: {type(exc).__name__}: {str(exc).splitlines()[0][:120]}", flush=True)
    print(f"  current stream restored: {torch.cuda.current_stream(dev) == before}; capturing now: "
          f"{torch.cuda.is_current_stream_capturing()}", flush=True)
    return g, before


def after(repair, g, before):
    dev = torch.device("cuda:0")
    if repair:
        torch.cuda.set_stream(before)
        gen = torch.cuda.default_generators[0]
        attempt("repair generator", lambda: gen.graphsafe_set_state(gen.clone_state()))
        attempt("repair allocator", lambda: torch._C._cuda_endAllocateToPool(0, g.pool()))
        attempt("graph.reset", g.reset)
    attempt("synchronize", torch.cuda.synchronize)
    attempt("eager rand", lambda: torch.rand(8, device=dev).sum().item())
    attempt("empty_cache", torch.cuda.empty_cache)

    def recapture():
        z = torch.zeros(64, device=dev)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            out = torch.rand(64, device=dev) + z
        g2.replay()
        a = out.clone()
        g2.replay()
        torch.cuda.synchronize()
        assert not torch.equal(a, out), "replays drew the same numbers"
    attempt("new capture with RNG + 2 replays", recapture)


def main():
    for kind in ("raise", "item", "sync", "pin"):
        for repair in (False, True):
            print(f"--- kind={kind} repair={repair}", flush=True)
            try:
                g, before = failing_capture(kind)
                after(repair, g, before)
                if not repair:
                    print("  (now repairing for the next case)")
                    after(True, g, before)
            except Exception:     # noqa: BLE001
                traceback.print_exc()


if __name__ == "__main__":
    main()

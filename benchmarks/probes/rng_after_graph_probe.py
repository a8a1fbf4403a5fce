"""Does an eager dropout after (capture, replay, set_rng_state) draw the mask an uninterrupted eager run draws?"""
import struct

import torch
import torch.nn.functional as F


def offset():
    st = torch.cuda.get_rng_state()
    return struct.unpack("<q", bytes(st[-8:].tolist()))[0], struct.unpack("<q", bytes(st[-16:-8].tolist()))[0]


def baseline():
    torch.manual_seed(7)
    n = torch.empty(2, 3, 64, device="cuda").normal_()
    print("  baseline offset after normal_", offset())
    x = torch.ones(1 << 16, device="cuda")
    return F.dropout(x, 0.4), F.dropout(x, 0.4)


def with_graph(replay, restore_after_capture, restore_after_replay):
    torch.manual_seed(7)
    n = torch.empty(2, 3, 64, device="cuda").normal_()
    s1 = torch.cuda.get_rng_state()
    print("  offset after normal_", offset())
    z = torch.ones(1 << 16, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = F.dropout(z, 0.4)
        y2 = F.dropout(z, 0.4)
    print("  offset after capture", offset())
    if restore_after_capture:
        torch.cuda.set_rng_state(s1)
    if replay:
        g.replay()
        torch.cuda.synchronize()
        print("  offset after replay", offset())
        if restore_after_replay:
            torch.cuda.set_rng_state(s1)
            print("  offset after restore", offset())
    x = torch.ones(1 << 16, device="cuda")
    a, b = F.dropout(x, 0.4), F.dropout(x, 0.4)
    return a, b, y.clone(), y2.clone()


def main():
    b0, b1 = baseline()
    for cfg in ((False, False, False), (False, True, False), (True, False, True), (True, True, True), (True, True, False)):
        print("replay=%s restore_after_capture=%s restore_after_replay=%s" % cfg)
        a, b, y, y2 = with_graph(*cfg)
        print("  eager masks equal the baseline's:", bool(torch.equal(a, b0)), bool(torch.equal(b, b1)),
              "| replayed masks equal the baseline's:", bool(torch.equal(y, b0)), bool(torch.equal(y2, b1)))


if __name__ == "__main__":
    main()

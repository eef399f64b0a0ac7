#!/usr/bin/env python3
"""The practical HBM roof of the box ("verify on box with a copy kernel"): device-to-device copies and a
read-only reduction over buffers far larger than the 256 MB Infinity Cache, timed with events.  torch is plumbing here
(device memory + a copy / sum kernel), not part of the product."""
import json

import torch


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    out = {"device": torch.cuda.get_device_name(0)}
    for gib in (1, 4):
        n = gib * (1 << 30) // 4
        src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
        dst = torch.empty_like(src)
        t = timed(lambda: dst.copy_(src), 20)
        out["copy_%dGiB_TBps" % gib] = 2 * n * 4 / t / 1e12  # read + write
        t = timed(lambda: src.sum(), 20)
        out["read_%dGiB_TBps" % gib] = n * 4 / t / 1e12
        t = timed(lambda: dst.fill_(1.0), 20)
        out["write_%dGiB_TBps" % gib] = n * 4 / t / 1e12
        del src, dst
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Stream-copy probe: the HBM bandwidth this box actually delivers (SURVEY.md sec.8d asks for the measured ceiling next to
the 8 TB/s specification that `roofline.peak` uses).  Device-to-device copies and fills of buffers far larger than the
256 MiB Infinity Cache, timed with events; prints one JSON object."""
import json
import torch

def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps

def main():
    out = {"device": torch.cuda.get_device_name(0)}
    for gib in (1, 4):
        n = gib << 30
        src = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(1)
        dst = torch.empty_like(src)
        t_copy = timed(lambda: dst.copy_(src))
        t_fill = timed(lambda: dst.fill_(3))
        t_read = timed(lambda: src.view(torch.int64).sum())
        out["%d_GiB" % gib] = {"copy_GBps_read_plus_write": round(2 * n / t_copy / 1e9, 1), "fill_GBps_write": round(n / t_fill / 1e9, 1),
                               "sum_GBps_read": round(n / t_read / 1e9, 1)}
        del src, dst
    print(json.dumps(out))

if __name__ == "__main__":
    main()

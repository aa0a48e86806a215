#!/usr/bin/env python
"""Sample rocm-smi (sclk / power) while a big conv GEMM runs back to back: what clock does the MFMA path sustain?"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if "sclk" in l or "ower" in l]
            samples.append((time.time(), keep))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.3)


def main():
    N, H, C, k = 112, 64, 320, 3
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(N, H, H, C, generator=g, device="cuda").half()
    pws = [ops.pack_weight(torch.randn(C, C, k, k, generator=g, device="cuda") * 0.02, None, torch.float16, "cuda") for _ in range(4)]
    outs = [torch.empty(N, H, H, C, device="cuda", dtype=torch.float16) for _ in range(4)]
    for i in range(4):
        ops.conv_gemm(x, pws[i], out=outs[i])
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    time.sleep(1.0)
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 6.0:
        for i in range(4):
            ops.conv_gemm(x, pws[i], out=outs[i])
        n += 4
        if n % 64 == 0:
            torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"busy phase: {n} launches, {ms * 1e3:.1f} us each, {2.0 * N * H * H * C * C * 9 / ms / 1e9:.0f} TFLOP/s", flush=True)
    time.sleep(1.0)
    global stop
    stop = True
    th.join()
    for t, keep in samples:
        print(f"{t - t0:6.2f}s", " | ".join(keep)[:230])


if __name__ == "__main__":
    main()

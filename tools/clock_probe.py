#!/usr/bin/env python
"""Sample rocm-smi (sclk / power) while a workload repeats: what clock does the chip sustain under it?

    python tools/clock_probe.py                       a big conv GEMM back to back (the MFMA path)
    python tools/clock_probe.py --attn                the 14-sample head_dim-40 self-attention of UNet level 0
    python tools/clock_probe.py --pipeline [--batch B]   the whole 50-step try-on loop (bench.py's workload)
"""
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


def fmt(line):
    """'GPU[0] : sclk clock level: 1: (2200Mhz)' -> 'sclk 2200Mhz'; 'GPU[0] : Current Socket Graphics Package Power (W): 1385.0' -> '1385.0 W'"""
    if "sclk" in line:
        return "sclk " + line.split("(")[-1].rstrip(")")
    if "ower (W)" in line:
        return line.rsplit(":", 1)[-1].strip() + " W"
    return ""


def attn_main():
    """--attn: the 14-sample head_dim-40 self-attention of UNet level 0, back to back."""
    N, h, S, d = 14, 8, 4096, 40
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(N, S, 3 * h * d, generator=g, device="cuda").half()
    C = h * d
    q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    out = ops.attention(q, k, v, h)
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    time.sleep(1.0)
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 5.0:
        for _ in range(8):
            ops.attention(q, k, v, h, out=out)
        n += 8
        if n % 64 == 0:
            torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"busy phase: {n} attention launches, {ms * 1e3:.1f} us each", flush=True)
    time.sleep(1.0)
    global stop
    stop = True
    th.join()
    for t, keep in samples:
        print(f"{t - t0:6.2f}s", " | ".join(k for k in (fmt(k) for k in keep) if k))


def pipeline_main(batch):
    """--pipeline [--batch B]: the whole 50-step try-on loop (bench.py's workload), images back to back."""
    import bench
    dev = torch.device("cuda", 0)
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
    lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, batch, dev)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5, num_inference_steps=50,
              output_type="pt", cond_noise=cn)
    pipe(**kw)
    pipe(**kw)
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    time.sleep(1.0)
    t0 = time.time()
    n = 0
    while time.time() - t0 < 6.0:
        pipe(**kw)
        n += 1
    torch.cuda.synchronize()
    print(f"busy phase: {n} calls of batch {batch}, {(time.time() - t0) / n * 1e3:.1f} ms each", flush=True)
    time.sleep(1.0)
    global stop
    stop = True
    th.join()
    for t, keep in samples:
        print(f"{t - t0:6.2f}s", " | ".join(k for k in (fmt(k) for k in keep) if k))


def main():
    if "--attn" in sys.argv:
        return attn_main()
    if "--pipeline" in sys.argv:
        return pipeline_main(int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1)
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
        print(f"{t - t0:6.2f}s", " | ".join(k for k in (fmt(k) for k in keep) if k))


if __name__ == "__main__":
    main()

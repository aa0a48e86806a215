#!/usr/bin/env python
"""HBM-side traffic of the VAE decode of BASELINE configs[4] (768 x 768, bf16, 4 images), op by op, from rocprofv3 --pmc passes.

    python3 tools/vae_decode_pmc.py run <ops.json>                 one warm decode, then one decode with a marker launch (incr_kernel)
                                                                   after every op; writes the op list (shapes, algorithmic bytes)
    python3 tools/vae_decode_pmc.py sum <FETCH csv> <WRITE csv> <ops.json> <out.json>

Run `run` once plain (for ops.json) and once under each of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/collect_vae_decode_pmc.sh).
gfx950 corrections as MI355X_MICROARCH.md prescribes: read bytes = 2 x FETCH_SIZE x 1024 (16-byte-per-lane streaming reads are tallied
at half size), write bytes = WRITE_SIZE x 1024; counters sit at the L2 -> fabric boundary (MALL-served bytes included)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(out_json, res=768, B=4):
    import torch
    import bench
    from edgestyle_amd import ops, config as C, weights as W
    from edgestyle_amd.models import AutoencoderKL
    dev = torch.device("cuda", 0)
    vcfg = C.sd15_vae()
    vae = AutoencoderKL(W.random_state_dict(W.vae_shapes(vcfg), 0, "vae.", device=dev), vcfg, torch.bfloat16).to(dev)
    s = res // vcfg.scale
    z = torch.randn(B, s, s, vae.engine.lat_pad, device=dev).to(torch.bfloat16)
    vae.decode_nhwc(z, unscaled_latents=True)
    torch.cuda.synchronize()
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    real = (ops.conv_gemm, ops.group_norm, ops.attention)
    log = []

    def nb(t):
        return 0 if t is None else t.numel() * t.element_size()

    def conv(x, pw, **kw):
        out = real[0](x, pw, **kw)
        q = pw[0] if isinstance(pw, (list, tuple)) else pw
        log.append(dict(op="conv%dx%d" % (q.ksize, q.ksize), x=list(x.shape), out=list(out.shape), upsample=bool(kw.get("upsample")),
                        act_bytes=nb(x) + nb(kw.get("x2")) + nb(kw.get("residual")) + sum(nb(t) for t in (kw.get("tail") or ()) if t is not None) + nb(out),
                        weight_bytes=nb(q.w), gflop=2.0 * out.shape[0] * out.shape[1] * out.shape[2] * q.cout * q.kpad / 1e9))
        ops.incr(ctr)
        return out

    def gn(x, *a, **kw):
        out = real[1](x, *a, **kw)
        log.append(dict(op="group_norm", x=list(x.shape), out=list(out.shape), act_bytes=nb(x) + nb(kw.get("x2")) + nb(out), weight_bytes=0, gflop=0.0))
        ops.incr(ctr)
        return out

    def attn(q, k, v, *a, **kw):
        out = real[2](q, k, v, *a, **kw)
        log.append(dict(op="attention", x=list(q.shape), out=list(out.shape), act_bytes=nb(q) + nb(k) + nb(v) + nb(out), weight_bytes=0, gflop=0.0))
        ops.incr(ctr)
        return out
    ops.incr(ctr)                                   # opening marker
    ops.conv_gemm, ops.group_norm, ops.attention = conv, gn, attn
    try:
        vae.decode_nhwc(z, unscaled_latents=True)
    finally:
        ops.conv_gemm, ops.group_norm, ops.attention = real
    torch.cuda.synchronize()
    json.dump(dict(resolution=res, images=B, dtype="bf16", ops=log), open(out_json, "w"))
    print(f"{len(log)} ops recorded", flush=True)


def short(name):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", name)
    if m:
        return m.group(1)
    n = name[5:] if name.startswith("void ") else name
    n = n.replace("(anonymous namespace)::", "")
    return re.split(r"[<(]", n, 1)[0][:40]


def per_op(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    groups, cur = [], None
    for _, k, v in rows:
        if k == "incr_kernel":
            if cur is not None:
                groups.append(cur)
            cur = []
        elif cur is not None:
            cur.append((k, v))
    return groups                                     # dispatches between consecutive markers; the last marker closes the last op


def summarize(fetch_csv, write_csv, ops_json, out_json):
    meta = json.load(open(ops_json))
    ops = meta["ops"]
    F, Wr = per_op(fetch_csv, "FETCH_SIZE"), per_op(write_csv, "WRITE_SIZE")
    # the LAST len(ops) groups of each pass are the marked decode (the warm-up decode runs before the opening marker)
    F, Wr = F[-len(ops):], Wr[-len(ops):]
    assert len(F) == len(ops) and len(Wr) == len(ops), (len(F), len(Wr), len(ops))
    levels = {}
    rows = []
    for o, f, w in zip(ops, F, Wr):
        rd = 2 * 1024 * sum(v for _, v in f)
        wr = 1024 * sum(v for _, v in w)
        n, h, wd, c = o["out"] if len(o["out"]) == 4 else (o["out"][0], 0, 0, o["out"][-1])
        key = f"{c} ch @ {h}x{wd}" if h else f"tokens {o['out'][1]} x {c}"
        rows.append(dict(op=o["op"], out=o["out"], kernels=[k for k, _ in f], algorithmic_bytes=o["act_bytes"] + o["weight_bytes"],
                         read_bytes=int(rd), write_bytes=int(wr)))
        a = levels.setdefault(key, dict(ops=0, algorithmic_bytes=0, read_bytes=0, write_bytes=0, gflop=0.0))
        a["ops"] += 1
        a["algorithmic_bytes"] += o["act_bytes"] + o["weight_bytes"]
        a["read_bytes"] += int(rd)
        a["write_bytes"] += int(wr)
        a["gflop"] += o["gflop"]
    for a in levels.values():
        a["hbm_over_algorithmic"] = round((a["read_bytes"] + a["write_bytes"]) / max(a["algorithmic_bytes"], 1), 3)
        a["gflop"] = round(a["gflop"], 1)
    tot_alg = sum(r["algorithmic_bytes"] for r in rows)
    tot_hbm = sum(r["read_bytes"] + r["write_bytes"] for r in rows)
    out = dict(round=5, workload=f"AutoencoderKL.decode (PL:552-557) of {meta['images']} latents at {meta['resolution']}x{meta['resolution']}, {meta['dtype']}, "
                                 "eager launches, one marker launch after every op",
               method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; reads = 2 x FETCH_SIZE x 1024, writes = WRITE_SIZE x 1024 "
                      "(MI355X_MICROARCH.md, HBM section); L2->fabric side: MALL-served bytes included",
               total=dict(algorithmic_bytes=tot_alg, hbm_bytes=tot_hbm, ratio=round(tot_hbm / max(tot_alg, 1), 3)),
               by_output_level=levels, ops=rows)
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(dict(total=out["total"], by_output_level=levels), indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        summarize(*sys.argv[2:6])

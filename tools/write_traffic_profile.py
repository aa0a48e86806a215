#!/usr/bin/env python
"""gpurun_out/traffic/{launches,FETCH_SIZE,WRITE_SIZE}.json (tools/collect_traffic.sh) -> profiles/r01_gemm_pmc_traffic_b1.json
and profiles/r01_gemm_step_launches_b1.json."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = os.path.join(ROOT, "gpurun_out", "traffic")
L = json.load(open(os.path.join(T, "launches.json")))
F = json.load(open(os.path.join(T, "FETCH_SIZE.json")))
W = json.load(open(os.path.join(T, "WRITE_SIZE.json")))
alg = sum(x["geom"]["algorithmic_bytes"] for x in L) / len(L)


def slab_bytes(q):
    rows = -(-q["cout"] // q["bn"]) * q["bn"]
    return q["splitk"] * q["N"] * q["Hout"] * q["Wout"] * rows * 4


slab = sum(slab_bytes(x["geom"]) for x in L if x["geom"]["splitk"] > 1) / len(L)
f = F["conv_gemm_kernel"]["FETCH_SIZE"]["per_launch"]
w = W["conv_gemm_kernel"]["WRITE_SIZE"]["per_launch"]
out = {
    "round": 1,
    "workload": f"the {len(L)} es_conv_gemm launches of ONE denoising step at batch 1 in grouped/lockstep mode "
                "(profiles/r01_gemm_step_launches_b1.json: geometry, tile, split-K and weight-group counts of every launch), "
                "replayed stand-alone by tools/gemm_step_traffic.py (tools/collect_traffic.sh)",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (TCC slot limit), eager launches (PMC "
              "collection segfaults through a hipGraph replay / the full pipeline on ROCm 7.2); gfx950 correction per "
              "MI355X_MICROARCH.md §HBM: read bytes = 2 * FETCH_SIZE * 1024 (wide 16-B/lane streaming loads are tallied at half "
              "size), write bytes = WRITE_SIZE * 1024.  The counters sit at the L2 (TCC) -> fabric boundary: bytes served by "
              "the 256 MB MALL are included.",
    "conv_gemm_kernel": {
        "launches": len(L), "grouped_launches": sum(1 for x in L if x["geom"].get("group_n")),
        "splitk_launches": sum(1 for x in L if x["geom"]["splitk"] > 1),
        "FETCH_SIZE_KiB_per_launch_raw": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(w, 1),
        "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024),
        "algorithmic_bytes_per_launch": int(alg),
        "splitk_slab_write_bytes_per_launch": int(slab),
        "note": "algorithmic = activations once + every weight set once + output (+ residual) per launch.  Excess reads: "
                "weights re-fetched once per XCD (8 private L2s) and 3x3 halo rows re-read across M tiles; a tile_m-fastest "
                "XCD chunk order that keeps each weight byte on one XCD was measured SLOWER on these shapes "
                "(tools/xcd_order_bench.py) - the re-fetches are served from MALL, not HBM, and do not bound the kernel.",
    },
    "splitk_reduce_kernel": {
        "launches": F["splitk_reduce_kernel"]["FETCH_SIZE"]["launches"],
        "FETCH_SIZE_KiB_per_launch_raw": round(F["splitk_reduce_kernel"]["FETCH_SIZE"]["per_launch"], 1),
        "WRITE_SIZE_KiB_per_launch": round(W["splitk_reduce_kernel"]["WRITE_SIZE"]["per_launch"], 1),
    },
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_gemm_pmc_traffic_b1.json"), "w"), indent=1)
json.dump(L, open(os.path.join(ROOT, "profiles", "r01_gemm_step_launches_b1.json"), "w"))
print(json.dumps(out["conv_gemm_kernel"], indent=1))

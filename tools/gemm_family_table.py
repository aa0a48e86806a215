#!/usr/bin/env python
"""profiles/rNN_gemm_family_table.txt from the per-launch dumps (bench.py ES_DUMP_GEMM=1 -> profiles/rNN_gemm_step_launches_*.json):
GEMM time of one step by family (k = 1 against k = 3) and by the kernel that ran the launch.   python3 tools/gemm_family_table.py r05"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
print("# GEMM launches of one denoising step by family, from in-kernel stamps of a hipGraph-replayed step (bench.py ES_DUMP_GEMM=1, profiles/%s_gemm_step_launches_*.json;" % R)
print("# stamps read 5-12 % high): k = 1 (1x1 convolutions, linear layers) against k = 3, and by the kernel that ran them (gemm = conv_gemm_kernel 128 | 64-pixel tiles,")
print("# gemm8p = the 256-pixel phase-interleaved tiles, linear_xs = the row-stationary short-K kernel).  tools/gemm_family_table.py; VERDICT r4 item 1.")
for tag, name in (("b1", "batch 1 (BASELINE configs[1])"), ("b8", "batch 8 (configs[2])"), ("768_b4", "768x768 bf16 batch 4 (configs[4])")):
    p = os.path.join(ROOT, "profiles", f"{R}_gemm_step_launches_{tag}.json")
    if not os.path.exists(p):
        continue
    L = json.load(open(p))
    tot = sum(l["seconds"] for l in L)
    print(f"{name}: {len(L)} launches, {tot * 1e3:.2f} ms")
    fam = {}
    for l in L:
        g = l["geom"]
        M = g["N"] * g["Hout"] * g["Wout"]
        K = (g["C1"] + g["C2"]) * g["k"] ** 2 + g.get("ctail", 0)
        fl = 2.0 * M * K * g["cout"]
        kern = "linear_xs" if g.get("kernel") == "linear_xs" else ("gemm8p" if g["bn"] in (320, 256) else "gemm")
        for key in ((g["k"], "all"), (g["k"], kern)):
            a = fam.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1; a[1] += l["seconds"]; a[2] += fl
    for k in (1, 3):
        for kern in ("all", "gemm", "gemm8p", "linear_xs"):
            if (k, kern) in fam:
                c, t, fl = fam[(k, kern)]
                print(f"   k={k} {kern:<12} {c:4d} launches  {t * 1e3:7.3f} ms  {fl / t / 1e12:5.0f} TFLOP/s  {t / tot * 100:4.1f} %")

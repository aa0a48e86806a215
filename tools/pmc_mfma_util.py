#!/usr/bin/env python
"""Matrix-core utilisation per kernel of a denoising step from a rocprofv3 --pmc pass over the REAL pipeline (eager launches).

    python3 tools/pmc_mfma_util.py <counter_collection.csv> <out.json>

Counters: SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over all SIMDs of the chip; = 16 per
v_mfma_f32_16x16x32 f16/bf16) and GRBM_GUI_ACTIVE (cycles the dispatch kept the GPU busy; rocprofv3 sums the 8 XCDs' counters).
MFMA utilisation of a kernel = sum MFMA_BUSY / (sum GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the fraction of SIMD-cycles with the
matrix pipe busy (a lower bound for short kernels: the GUI_ACTIVE window of a dispatch includes ~6 us of counter start / stop;
`mfma_busy_ghz` = busy SIMD-cycles per SIMD per us of kernel time = utilisation x clock avoids that: divide by the ~2.25 GHz the
loop sustains).  Cross-check: MFMA_BUSY x 1024 FLOP = the FLOPs the matrix pipes were issued (16 busy cycles per 16x16x32 MFMA).  Every step ends with one incr_kernel launch; steps 5..44 of the 50 are averaged."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", name)
    if m:
        return m.group(1)
    n = name[5:] if name.startswith("void ") else name
    n = n.replace("(anonymous namespace)::", "")
    return re.split(r"[<(]", n, 1)[0][:40]


def main(path, out_json):
    rows = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = rows[int(r["Dispatch_Id"])]
        d["k"] = short(r["Kernel_Name"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    steps, cur = [], []
    for _, d in sorted(rows.items()):
        if d["k"] == "incr_kernel":
            steps.append(cur)
            cur = []
        else:
            cur.append(d)
    sel = steps[5:45] if len(steps) >= 45 else steps[1:]
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for st in sel:
        for d in st:
            a = agg[d["k"]]
            a[0] += 1
            a[1] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            a[2] += d.get("GRBM_GUI_ACTIVE", 0.0)
            a[3] += d.get("ns", 0.0)
    n = max(len(sel), 1)
    out = {"steps_averaged": len(sel), "method": __doc__.split("\n\n")[2].replace("\n", " "), "kernels": {}}
    tot_busy = tot_act = 0.0
    for k, (cnt, busy, act, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        if act <= 0:
            continue
        util = busy / (act / 8.0 * 256 * 4)
        out["kernels"][k] = {"launches_per_step": round(cnt / n, 1), "gui_active_cycles_per_step": round(act / 8.0 / n), "mfma_busy_per_step": round(busy / n),
                             "mfma_util": round(util, 4), "us_per_step_under_pmc": round(ns / n / 1e3, 1),
                             # independent of the GUI_ACTIVE window (which includes each dispatch's counter start / stop, ~6 us):
                             # busy SIMD-cycles per SIMD per microsecond of kernel time = utilisation x clock in GHz
                             "mfma_busy_ghz": round(busy / 1024.0 / ns, 4) if ns else None,
                             "mfma_flop_issued_per_step": round(busy / n * 1024.0)}
        tot_busy += busy
        tot_act += act
    out["whole_step_mfma_util"] = round(tot_busy / (tot_act / 8.0 * 256 * 4), 4) if tot_act else None
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])

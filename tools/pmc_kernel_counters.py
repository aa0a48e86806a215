#!/usr/bin/env python
"""Per-kernel sums of arbitrary rocprofv3 --pmc counters over the steps of the REAL pipeline (eager launches), with ratios to the
first counter given.

    python3 tools/pmc_kernel_counters.py <counter_collection.csv> <out.json> <COUNTER_A> [COUNTER_B ...]

Every step ends with one incr_kernel launch; steps 5..44 of the 50 are averaged.  SQ wave-state counters (MI355X_MICROARCH.md,
PMC slots): SQ_WAIT_ANY = wave parked at s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue stall (MFMA read-after-write, busy pipe),
SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to ~SQ_WAVE_CYCLES (all in quad-cycles)."""
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


def main(path, out_json, *counters):
    rows = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = rows[int(r["Dispatch_Id"])]
        d["k"] = short(r["Kernel_Name"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    steps, cur = [], []
    for _, d in sorted(rows.items()):
        if d["k"] == "incr_kernel":
            steps.append(cur)
            cur = []
        else:
            cur.append(d)
    sel = steps[5:45] if len(steps) >= 45 else steps[1:]
    n = max(len(sel), 1)
    agg = defaultdict(lambda: defaultdict(float))
    for st in sel:
        for d in st:
            a = agg[d["k"]]
            a["launches"] += 1
            a["ns"] += d["ns"]
            for c in counters:
                a[c] += d.get(c, 0.0)
    out = {"steps_averaged": len(sel), "counters": list(counters), "kernels": {}}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        base = a[counters[0]] or 1.0
        out["kernels"][k] = {"launches_per_step": round(a["launches"] / n, 1), "us_per_step_under_pmc": round(a["ns"] / n / 1e3, 1),
                             **{c: round(a[c] / n) for c in counters}, **{c + "/" + counters[0]: round(a[c] / base, 4) for c in counters[1:]}}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in list(out["kernels"].items())[:12]:
        print(k, {kk: vv for kk, vv in v.items() if "/" in kk or kk == "us_per_step_under_pmc"})


if __name__ == "__main__":
    main(*sys.argv[1:])

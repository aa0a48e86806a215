#!/usr/bin/env python
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short table (demangled-ish names, top N) for profiles/."""
import csv
import re
import sys


def short(name: str) -> str:
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(DF16_|DF16b)?(.*?)E+v", name)
    if m:
        dt = {"DF16_": "f16", "DF16b": "bf16", None: ""}[m.group(2)]
        args = re.findall(r"L[ib](\d+)E", m.group(3))
        return f"{m.group(1)}<{','.join([dt] + args)}>"
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void at::native::", "torch::", name)
    return name[:90]


def main(path, top=30):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# source: {path}\n# total kernel time: {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
    print("kernel,calls,total_ms,avg_us,pct,min_us,max_us")
    for r in rows[:top]:
        print(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
              f"{float(r['Percentage']):.2f},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)

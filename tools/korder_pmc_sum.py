"""Average every counter of a rocprofv3 counter_collection.csv per kernel (conv_gemm kernels only), K order read off the name."""
import csv, re, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d = rows[int(r["Dispatch_Id"])]
    d["k"] = r["Kernel_Name"]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = defaultdict(lambda: defaultdict(list))
for _, d in sorted(rows.items()):
    if "conv_gemm" not in d["k"]:
        continue
    chunk = "Lb1EEEv" in d["k"] or re.search(r", true>\(", d["k"]) is not None
    key = ("8p " if "conv_gemm8p" in d["k"] else "128 ") + ("chunk-major" if chunk else "tap-major")
    for c, v in d.items():
        if c != "k":
            agg[key][c].append(v)
for key, a in sorted(agg.items()):
    print(key, {c: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for c, v in a.items()}, f"({len(a['ns'])} dispatches, first dropped)")

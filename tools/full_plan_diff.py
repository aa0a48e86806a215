#!/usr/bin/env python
"""es_load_weights against the Python host at SD1.5 width WITHOUT a GPU: both builders run on the library's dry recorder
(tests/helpers.py python_dry_context; es_load_weights(device = -2)) and must record the same calls with the same arguments in
all five plans and put the same bytes behind every weight / bias / column-sum / norm / fusion-parameter pointer.
(tests/test_load_weights_cpu.py does this at reduced width in the CPU test tier; this is the full-width run: ~20 GB of host
memory, a few minutes.)

    python tools/full_plan_diff.py [--batch 1,8] [--no-constants]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd.native import NativeContext  # noqa: E402
from tests.helpers import full_weights, python_dry_context, diff_plans, plan_constants  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", default="1")
    ap.add_argument("--no-constants", action="store_true")
    a = ap.parse_args()
    t = time.time()
    ucfg, vcfg, ws = full_weights(seed=0)
    print(f"weights: {time.time() - t:.0f} s", flush=True)
    ok = True
    for B in [int(b) for b in a.batch.split(",")]:
        t = time.time()
        lib, pctx, keep = python_dry_context(ws, ucfg, vcfg, B, True, 50, rank=32)
        t1 = time.time()
        nat = NativeContext(ws, ucfg, vcfg, batch_size=B, num_inference_steps=50, device=-1 if a.no_constants else -2)
        print(f"batch {B}: python host {t1 - t:.0f} s, es_load_weights {time.time() - t1:.0f} s, arena {lib.es_ctx_arena_bytes(nat.ctx) / 2 ** 30:.2f} GiB", flush=True)
        for which in range(L.PLAN_COUNT):
            d = diff_plans(lib, pctx, nat.ctx, which)
            line = f"  plan {which}: {lib.es_ctx_plan_size(pctx, which)} calls, " + ("identical" if d is None else d)
            if d is None and not a.no_constants:
                ca, cb = plan_constants(lib, pctx, which), plan_constants(lib, nat.ctx, which)
                bad = [i for i, (x, y) in enumerate(zip(ca, cb)) if x != y]
                line += f"; {len(ca)} constant blobs, {sum(map(len, ca)) / 2 ** 20:.0f} MiB, {len(bad)} differ"
                ok = ok and not bad and len(ca) == len(cb)
            ok = ok and d is None
            print(line, flush=True)
        nat.close()
        lib.es_ctx_destroy(pctx)
        del keep
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""Per-phase cycles of the attention K/V loop (tool-only -DES_ATTN_STAMPS build of attention.hip).

    build here:  see the ES_ATTN_STAMPS note in attention.hip (-> edgestyle_amd/lib/dbg/libes_attn_stamps.so)
    run on GPU:  ES_HIP_LIB=edgestyle_amd/lib/dbg/libes_attn_stamps.so python tools/attn_stamps.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import lib, ops  # noqa: E402

NAMES = ["loop", "barrier", "K reads+QK", "softmax", "V reads+PV", "staging"]


def main():
    lib.load()
    raw = ctypes.CDLL(lib.LIB_PATH)
    g = torch.Generator(device="cuda").manual_seed(0)
    for N, h, S, d in [(16, 8, 4096, 40), (2, 8, 4096, 40), (16, 8, 1024, 80)]:
        C = h * d
        qkv = torch.randn(N, S, 3 * C, generator=g, device="cuda").half()
        for _ in range(3):
            ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], h)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        assert raw.es_attn_debug_read(buf) == 0
        tiles = S // 64
        vals = [buf[i] / tiles for i in range(6)]
        print(f"N={N} S={S} d={d}: cycles per K/V tile (wave 0 of one block): " +
              ", ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, vals)) + f" | total {sum(vals):.0f}")


def main_pp():
    """attention40pp_kernel (head_dim 40 self-attention, the ping-pong kernel): cycles per K/V tile of wave 0 (group 0) and wave 4 (group 1) of
    one workgroup in [V-phase work | wait at the barrier behind it | M-phase work | wait at the barrier behind it]."""
    lib.load()
    raw = ctypes.CDLL(lib.LIB_PATH)
    g = torch.Generator(device="cuda").manual_seed(0)
    for N, h, S, d in [(14, 8, 4096, 40), (112, 8, 4096, 40), (2, 8, 4096, 40)]:
        C = h * d
        qkv = torch.randn(N, S, 3 * C, generator=g, device="cuda").half()
        for _ in range(3):
            ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], h)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        assert raw.es_attn_debug_read(buf) == 0
        tiles = S // 64
        for grp in (0, 1):
            v = [buf[8 + grp * 4 + i] / tiles for i in range(4)]
            print(f"N={N} S={S} d={d} group {grp}: V-phase work {v[0]:.0f} | wait {v[1]:.0f} | M-phase work {v[2]:.0f} | wait {v[3]:.0f} | tile {sum(v):.0f} cycles", flush=True)


if __name__ == "__main__":
    if "--pp" in sys.argv:
        main_pp()
    else:
        main()

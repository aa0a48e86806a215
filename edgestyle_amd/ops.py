"""Thin Python wrappers over the C ABI: torch only allocates device memory and supplies the current HIP stream.

Internal activation layout is NHWC ([N,H,W,C], dtype fp16/bf16); every wrapper launches on
`torch.cuda.current_stream()` so the calls can be captured into a hipGraph (`torch.cuda.graph`).
"""
import ctypes as C
import math
import os as _os
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import lib as L

_DT = {torch.float16: L.ES_F16, torch.bfloat16: L.ES_BF16}
LANE = 0            # scratch-buffer namespace: concurrent chains on different HIP streams must not share scratch


class lane:
    """with ops.lane(i): ... — kernels launched inside use scratch buffers (split-K workspace, GroupNorm partials)
    private to lane i, so independent chains can run concurrently on separate streams."""

    def __init__(self, i: int):
        self.i = i

    def __enter__(self):
        global LANE
        self.prev, LANE = LANE, self.i

    def __exit__(self, *a):
        global LANE
        LANE = self.prev


XCD_ORDER = -1      # tuning knob: -1 auto, 0 tile_n fastest, 1 tile_m fastest
DEEP_RING = _os.environ.get("ES_DEEP_RING", "1") == "1"     # 4-stage LDS ring for launches of <= 1 workgroup per CU
FORCE_BN = 0        # tuning knob: 0 = per-launch choice between the legal N tiles
FORCE_STAGES = 0    # tuning knob (tools/gemm_bench.py): 0 = kernel picks the LDS ring depth
FORCE_WAVES = 0     # tuning knob: 0 = planner picks 4 or 8 waves per 128-pixel workgroup
EIGHT_WAVES = _os.environ.get("ES_EIGHT_WAVES", "1") == "1"
SMALL_TILE = _os.environ.get("ES_SMALL_TILE", "1") == "1"    # 64x64 tile for tiny launches
PROFILE = None      # set to a Profiler by bench.py: every es_conv_gemm launch gets an in-kernel timing slot


class Profiler:
    """Per-launch durations as executed (also inside a hipGraph replay, where host-side events cannot look):
    each workgroup of an instrumented launch atomically mins its start / maxes its end s_memrealtime stamp
    (100 MHz constant clock) into a device slot; duration = (max end - min start) * 10 ns."""

    def __init__(self, device, capacity: int = 8192, stamps: bool = True):
        self.slots = torch.empty((capacity, 2), dtype=torch.int64, device=device)
        self.meta = []
        self.stamps = stamps        # False: only record the descriptors (launch list), no in-kernel atomics
        self.descs = []             # a copy of every es_gemm_desc launched while this profiler was active
        self.reset()

    def reset(self):
        self.slots[:, 0] = 1 << 62
        self.slots[:, 1] = 0

    def next(self, meta) -> int:
        i = len(self.meta)
        if i >= self.slots.shape[0]:
            raise L.EdgeStyleHipError("Profiler capacity exceeded")
        self.meta.append(meta)
        return self.slots[i].data_ptr() if self.stamps else 0

    def replay_gemms(self, only_ksize: int = 0):
        """Launch the recorded es_conv_gemm descriptors again on the current stream, un-instrumented (the production
        kernel: no stamp atomics).  The buffers they point at must still be alive (keep the graph that owns them).
        only_ksize = 3: only the 3x3 convolutions."""
        lib = L.load()
        for d in self.descs:
            if only_ksize and (isinstance(d, L.XsDesc) or int(d.ksize) != only_ksize):
                continue
            fn = lib.es_linear_xs if isinstance(d, L.XsDesc) else lib.es_conv_gemm
            L.check(fn(C.byref(d), _stream()), "GEMM replay")

    def results(self, first: int = 0):
        s = self.slots[first:len(self.meta)].cpu().tolist()
        return [(m, (e - b) * 1e-8) for m, (b, e) in zip(self.meta[first:], s) if e > 0]
BK = 64
BM = 128


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise L.EdgeStyleHipError(f"unsupported dtype {t.dtype}: the HIP path computes in fp16 or bf16")


# ----------------------------------------------------------------------------------------------------------------
# weight packing (host side, once at load)
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class PackedWeight:
    w: torch.Tensor                 # [rows_padded, Kpad] dtype, K = (ky,kx,c) tap-major
    bias: Optional[torch.Tensor]    # [rows_padded] fp32
    cout: int                       # true GEMM N
    cin: int                        # padded input channels (multiple of 8)
    ksize: int
    bn: int
    geglu: bool = False
    ln_colsum: Optional[torch.Tensor] = None   # fp32 [rows_padded]: LayerNorm folded into this linear layer (pack_weight_ln)
    ln_eps: float = 1e-5
    ctail: int = 0                  # channels of the 1x1 tail sources appended along K (pack_weight_tail)
    korder: int = 0                 # K order of the k*k*cin main part: 0 tap-major (ky,kx,c), 1 chunk-major (c/64,ky,kx,c%64)

    @property
    def rows_padded(self):
        return self.w.shape[0]

    @property
    def kpad(self):
        return self.w.shape[1]


def choose_bn(cout: int) -> int:
    return 160 if (cout % 160 == 0 and cout % 128 != 0) else 128


# K order of the 3x3 convolutions (es_gemm_desc.korder).  Chunk-major - the nine taps of a 64-channel chunk back to back - makes
# eight of a workgroup's nine shifted reads of the same activation lines L2 hits: the level-0 batch-8 launch fetches 0.39 GB
# through the fabric instead of 2.6 GB (FETCH_SIZE, profiles/r04_korder_pmc.txt).  It is nevertheless 5-10 % SLOWER in every
# sustained A/B (profiles/r04_korder_bench.txt), even with eight of the nine activation DMAs removed altogether
# (profiles/r04_patchsim_ablation.txt), so it is OFF by default: ES_CHUNK_MAJOR=1 (read by csrc/builder.hip too) turns it on.
CHUNK_MAJOR = _os.environ.get("ES_CHUNK_MAJOR", "0") == "1"


def choose_korder(k: int, cin_padded: int) -> int:
    """es_gemm_desc.korder of a convolution's packed weights (csrc/builder.hip: the same rule)."""
    return 1 if (CHUNK_MAJOR and k == 3 and cin_padded % BK == 0) else 0


def _chunk_major(w: torch.Tensor, k: int, cin: int) -> torch.Tensor:
    """[Cout, k*k*cin] tap-major -> chunk-major."""
    cout = w.shape[0]
    return w.reshape(cout, k * k, cin // BK, BK).permute(0, 2, 1, 3).reshape(cout, k * k * cin)


def pack_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], dtype, device, geglu: bool = False,
                cin_pad: Optional[int] = None, cout_pad: Optional[int] = None) -> PackedWeight:
    """weight: [Cout, Cin, k, k] (conv) or [Cout, Cin] (linear) fp32, on any device -> packed tensor on `device`.

    GEGLU (ff.net.0.proj, out = 2*inner): rows are re-ordered in blocks of 32 = [16 hidden | 16 gate] so that
    the GEMM epilogue finds hidden and gate of the same output column in adjacent MFMA fragments.
    `cout_pad` appends zero output channels (used to hand 4-channel latents on as 8-channel NHWC tensors).
    """
    if weight.dim() == 2:
        weight = weight[:, :, None, None]
    weight = weight.to(device=device, dtype=torch.float32)
    cout, cin, k, _ = weight.shape
    cp = cin_pad or ((cin + 7) // 8 * 8)
    w = weight.permute(0, 2, 3, 1)                              # [Cout, k, k, Cin]
    if cp != cin:
        w = torch.nn.functional.pad(w, (0, cp - cin))
    w = w.reshape(cout, k * k * cp)
    korder = choose_korder(k, cp)
    if korder:
        w = _chunk_major(w, k, cp)
    b = None if bias is None else bias.to(device=device, dtype=torch.float32)
    if geglu:
        inner = cout // 2
        assert inner % 16 == 0
        idx = torch.arange(cout, device=device)
        blk, within = idx // 32, idx % 32
        src = torch.where(within < 16, blk * 16 + within, inner + blk * 16 + (within - 16))
        w = w[src]
        if b is not None:
            b = b[src]
    cout_eff = cout_pad or cout
    bn = 128 if geglu else choose_bn(cout_eff)
    rows = (cout_eff + bn - 1) // bn * bn
    ktrue = w.shape[1]
    kpad = (ktrue + BK - 1) // BK * BK
    wp = torch.zeros(rows, kpad, dtype=dtype, device=device)
    wp[:cout, :ktrue] = w.to(dtype)
    bp = None
    if b is not None:
        bp = torch.zeros(rows, dtype=torch.float32, device=device)
        bp[:cout] = b
    return PackedWeight(wp, bp, cout_eff, cp, k, bn, geglu, korder=korder)


def pack_weight_tail(weight: torch.Tensor, tail_weight: torch.Tensor, bias: Optional[torch.Tensor], dtype, device) -> PackedWeight:
    """conv (weight [Cout,Cin,k,k]) and a 1x1 conv over OTHER tensors of the output's size (tail_weight [Cout,Ct] or
    [Cout,Ct,1,1]) as one GEMM: K = (ky,kx,c) taps of the conv, then the tail channels (es_gemm_desc.t1/t2).
    ResnetBlock2D: conv2(h) + conv_shortcut(x); `bias` is the SUM of the two biases.  64-aligned channels only."""
    weight = weight.to(device=device, dtype=torch.float32)
    tw = tail_weight.to(device=device, dtype=torch.float32).reshape(tail_weight.shape[0], -1)
    cout, cin, k, _ = weight.shape
    if cin % BK or tw.shape[1] % BK or tw.shape[0] != cout:
        raise L.EdgeStyleHipError("pack_weight_tail: conv and tail channels must be multiples of 64, equal Cout")
    korder = choose_korder(k, cin)
    wm = weight.permute(0, 2, 3, 1).reshape(cout, k * k * cin)
    w = torch.cat([_chunk_major(wm, k, cin) if korder else wm, tw], 1)
    bn = choose_bn(cout)
    rows = (cout + bn - 1) // bn * bn
    wp = torch.zeros(rows, w.shape[1], dtype=dtype, device=device)
    wp[:cout] = w.to(dtype)
    bp = None
    if bias is not None:
        bp = torch.zeros(rows, dtype=torch.float32, device=device)
        bp[:cout] = bias.to(device=device, dtype=torch.float32)
    return PackedWeight(wp, bp, cout, cin, k, bn, False, ctail=tw.shape[1], korder=korder)


def pack_weight_ln(weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                   eps: float, dtype, device, geglu: bool = False) -> PackedWeight:
    """Linear(LayerNorm(x)) as ONE launch on the raw x (es_gemm_desc.ln_colsum):
        LN(x) W^T + b = rstd (x (W*gamma)^T - mean colsum(W*gamma)) + (W beta + b)
    The kernel gathers every row's mean / rstd from the activation tiles it stages anyway.  weight: [Cout, Cin]."""
    w = weight.to(device=device, dtype=torch.float32)
    if w.dim() == 4:
        w = w[:, :, 0, 0]
    g = gamma.to(device=device, dtype=torch.float32)
    b = beta.to(device=device, dtype=torch.float32)
    # the two reductions in double, rounded once (order-independent: es_load_weights computes the same values on the host)
    fb = w.double() @ b.double()
    if bias is not None:
        fb = fb + bias.to(device=device, dtype=torch.float64)
    pw = pack_weight(w * g[None, :], fb.float(), dtype, device, geglu=geglu)
    pw.ln_colsum = pw.w.double().sum(dim=1).float().contiguous()          # of the ROUNDED weights the MFMAs multiply
    pw.ln_eps = float(eps)
    return pw


# ----------------------------------------------------------------------------------------------------------------
# split-K workspace (grown outside graph capture by a warm-up pass)
# ----------------------------------------------------------------------------------------------------------------
_workspace = {}
_workspace_retired = []


def _get_workspace(nbytes: int, device) -> torch.Tensor:
    device = (device, LANE)
    ws = _workspace.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise L.EdgeStyleHipError("split-K workspace too small during graph capture; run one eager warm-up first")
        if ws is not None:
            _workspace_retired.append(ws)      # graphs captured earlier still write their slabs there: never freed
        ws = torch.empty(max(nbytes // 4, 1 << 22), dtype=torch.float32, device=device[0])
        _workspace[device] = ws
    return ws


# Launch planner.  A launch is described by (bn, splitk, stages); its cost model below is in units of one K-step of a
# 128-wide tile on a CU shared by two workgroups (~0.77 us) and was fitted on tools/gemm_tune.py sweeps over the
# shapes of a batch-1 denoising step (total within 1.1 % of the per-shape best configuration, 6.4 % below the
# previous tiles-vs-target rules):
#   * 512 workgroups are resident at once (256 CUs x 2); a launch takes ceil(WGs / 512) rounds of one workgroup's time;
#   * a 160-wide tile costs 1.25x a 128-wide one per K-step; a workgroup alone on its CU runs ~10 % faster;
#   * split-K pays a reduce launch (~12 units) plus the fp32 slab round trip, needs >= 12 K-steps per slice and is
#     never worth it for K <= 640.
#   * the big tile (256 px x 320 couts, one workgroup per CU) does the work of two 160-wide workgroups in ~8 % less
#     time (half the L2->LDS bytes) and is charged FRACTIONAL rounds; since round 4 (lean epilogue) it is offered from 2048 pixels up,
#     where it wins as a split-K launch of ~256 workgroups (the 16 x 16 decoder level at batch 8: 1.3-1.5x, tools/plan_fit_bench.py).
PLAN_T160, PLAN_ALONE, PLAN_TFIX, PLAN_RED_FIX, PLAN_SLAB_BYTES_PER_UNIT = 1.25, 0.9, 2.0, 12.0, 4.0e6
PLAN_T320, PLAN_BIG_MIN_M, PLAN_T320_FIX, PLAN_BIG_MIN_NK = 2.3, 2048, 3.0, 16
# the 256 x 256 tile of the LayerNorm-folded / GEGLU layers against the 128-wide tile on the same layers (csrc/builder.hip plan_gemm)
PLAN_T256, PLAN_T256_FIX, PLAN_T256_GEGLU, PLAN_LN_TK, PLAN_T256_MARGIN = 2.6, 6.0, 7.0, 1.55, 0.93
#   * the 64x64 tile (tiny launches): a K-step costs 0.5 units with the CU to itself and 0.5 + 0.16 (w - 1)^2 with w
#     workgroups per CU (measured 0.87 at w = 2.5, 2.0 at w = 3.75); x1.5 for K > 2560, where MFMA throughput starts to
#     matter and the small tile reads LDS twice as often per FLOP;
#     1024 resident workgroups (32 KB of LDS each); offered up to 16k pixels.
PLAN_T64_ALONE, PLAN_T64, PLAN_T64_LONG, PLAN_SMALL_MAX_M = 0.5, 0.16, 1.5, 16384
# the 256 x 320 phase-interleaved tile (csrc/gemm_conv8p.hip): 1.03-1.15x the 128-pixel tile on launches of >= 1 round of
# 256 workgroups with K >= 1024 (tools/gemm8p_bench.py: 1.07-1.30 PFLOP/s on the batch-8 3x3 launches)
BIG_TILE = _os.environ.get("ES_BIG_TILE", "1") == "1"
BIG_TILE_256 = _os.environ.get("ES_BIG_TILE_256", "1") != "0"     # its 256-wide form for the LayerNorm-folded / GEGLU linear layers
PLAN_SLAB_BYTES_PER_UNIT = float(_os.environ.get("ES_PLAN_SLAB", PLAN_SLAB_BYTES_PER_UNIT))
PLAN_RED_FIX = float(_os.environ.get("ES_PLAN_REDFIX", PLAN_RED_FIX))
PLAN_MIN_SLICE, PLAN_NK_NOSPLIT, PLAN_RESIDENT = 12, 10, 512


_PLAN_TUNING = any(k in _os.environ for k in ("ES_PLAN_SLAB", "ES_PLAN_REDFIX"))


def plan_gemm(M: int, rows_padded: int, kpad: int, geglu: bool = False, bns=(160, 128, 64), allow_split: bool = True):
    """(bn, splitk, stages) of an es_conv_gemm launch: ONE planner for both hosts - the library's (es_plan_gemm_choice,
    csrc/builder.hip), which es_load_weights uses itself.  `plan_gemm_reference` below is the Python copy it replaced: kept for
    the tuning tools (cost-model knobs from the environment) and as a guard - tests/test_load_weights_cpu.py sweeps both."""
    if _PLAN_TUNING:
        return plan_gemm_reference(M, rows_padded, kpad, geglu, bns, allow_split)
    arr = (C.c_int * len(bns))(*[int(b) for b in bns])
    bn, sk, st = C.c_int(), C.c_int(), C.c_int()
    if L.load().es_plan_gemm_choice(int(M), int(rows_padded), int(kpad), int(bool(geglu)), arr, len(bns), int(bool(allow_split)),
                                    C.byref(bn), C.byref(sk), C.byref(st)) != 0:
        raise L.EdgeStyleHipError(f"plan_gemm: rows_padded {rows_padded} fits no N tile")
    stages = st.value
    if stages == 4 and not (DEEP_RING and LANE == 0):       # tool knobs: no 4-deep ring
        stages = 2
    return bn.value, sk.value, stages


def plan_gemm_reference(M: int, rows_padded: int, kpad: int, geglu: bool = False, bns=(160, 128, 64), allow_split: bool = True):
    """The Python copy of the library's planner (one more round as a guard; tuning tools): (bn, splitk, stages) with the
    lowest modelled time among the legal N tiles (ties go to the wider tile).  bn = 256 (the 256 x 256 phase-interleaved tile,
    offered for the LayerNorm-folded / GEGLU linear layers whose N is a multiple of 256) is decided AFTER the choice among the
    other tiles, on a model fitted on those layers (csrc/builder.hip plan_gemm, tools/ln256_bench.py)."""
    if 256 in bns:
        others = tuple(b for b in bns if b != 256)
        nk = kpad // BK
        other = None
        if others:
            try:
                other = plan_gemm_reference(M, rows_padded, kpad, geglu, others, allow_split)
            except L.EdgeStyleHipError:
                other = None
        if rows_padded % 256:
            if other is None:
                raise L.EdgeStyleHipError(f"plan_gemm: rows_padded {rows_padded} fits no N tile")
            return other
        t256 = ((M + 255) // 256) * (rows_padded // 256)
        c256 = float(-(-t256 // 256)) * (nk * PLAN_T256 + PLAN_T256_FIX + (PLAN_T256_GEGLU if geglu else 0.0))
        take = other is None
        if other is not None and M >= PLAN_BIG_MIN_M and nk >= PLAN_BIG_MIN_NK and other[1] == 1 and other[0] != 64:
            to = ((M + BM - 1) // BM) * (rows_padded // other[0])
            co = float(-(-to // PLAN_RESIDENT)) * (nk * PLAN_LN_TK * (PLAN_T160 if other[0] == 160 else 1.0) + PLAN_T256_FIX)
            take = c256 < PLAN_T256_MARGIN * co
        return (256, 1, 2) if take else other
    if geglu:
        return 128, 1, 2
    nk = kpad // BK
    best = None
    for bn in bns:
        if rows_padded % bn:
            continue
        if bn == 320 and (M < PLAN_BIG_MIN_M or nk < PLAN_BIG_MIN_NK) and len(bns) > 1:
            continue
        if bn == 64 and M > PLAN_SMALL_MAX_M and len(bns) > 1:
            continue
        bm = {320: 256, 64: 64}.get(bn, BM)
        resident = {320: PLAN_RESIDENT // 2, 64: 2 * PLAN_RESIDENT}.get(bn, PLAN_RESIDENT)
        tiles = ((M + bm - 1) // bm) * (rows_padded // bn)
        cands = [1]
        if allow_split and tiles < resident // 2 and nk > PLAN_NK_NOSPLIT:
            cands += list(range(2, min(nk // PLAN_MIN_SLICE, 32) + 1))
        for sk in cands:
            wgs = tiles * sk
            if bn == 320:
                tk = PLAN_T320
            elif bn == 64:
                if sk > 1 and wgs > PLAN_RESIDENT:
                    continue
                w = wgs / (PLAN_RESIDENT // 2)                 # workgroups per CU
                tk = (PLAN_T64_ALONE + PLAN_T64 * max(0.0, w - 1.0) ** 2) * (1.0 if nk <= 40 else PLAN_T64_LONG)
            else:
                tk = (PLAN_T160 if bn == 160 else 1.0) * (PLAN_ALONE if wgs <= PLAN_RESIDENT // 2 else 1.0)
            # rounds of workgroups: whole ones for the two-per-CU tiles; the 256 x 320 tile (one workgroup per CU, long tiles) is
            # measured to cost its FRACTIONAL number of rounds - 896 tiles = 3.5 rounds run in 3.5 tile times, the CUs of a part-filled
            # round run faster (tools/plan_fit_bench.py, round 4) - and no less than 0.9 of a round
            rounds = max(wgs / 256.0, 0.9) if bn == 320 else float(-(-wgs // resident))
            t = rounds * ((nk / sk) * tk + (PLAN_T320_FIX if bn == 320 else PLAN_TFIX))
            if sk > 1:
                t += PLAN_RED_FIX + sk * M * rows_padded * 8.0 / PLAN_SLAB_BYTES_PER_UNIT
            if best is None or t < best[0]:
                best = (t, bn, sk, wgs)
    if best is None:
        raise L.EdgeStyleHipError(f"plan_gemm: rows_padded {rows_padded} fits no N tile")
    _, bn, sk, wgs = best
    # at most one workgroup per CU: a 4-deep LDS ring (3 K-steps of DMA in flight) hides the HBM round trip that the
    # 2-stage ring leaves exposed when no second workgroup shares the CU.  Not inside concurrent chains (LANE > 0,
    # ES_CHAIN_MODE=streams): there 2 stages = 72 KB LDS let workgroups of different chains share a CU.
    if bn == 64:
        stages = 4 if (DEEP_RING and LANE == 0 and wgs <= PLAN_RESIDENT and nk // sk >= 6) else 2
    else:
        stages = 4 if (bn != 320 and DEEP_RING and LANE == 0 and wgs <= PLAN_RESIDENT // 2 and nk // sk >= 6) else 2
    return bn, sk, stages


def choose_launch_bn(M: int, pw: "PackedWeight") -> int:
    return plan_gemm(M, pw.rows_padded, pw.kpad, pw.geglu)[0]


def choose_splitk(M: int, rows_padded: int, bn: int, kpad: int) -> int:
    return plan_gemm(M, rows_padded, kpad, bns=(bn,))[1]


def plan_launch(M: int, pw: "PackedWeight", bn: int):
    """(splitk, stages) for a fixed N tile."""
    _, sk, st = plan_gemm(M, pw.rows_padded, pw.kpad, pw.geglu, bns=(bn,))
    return sk, st


XS_ENABLED = _os.environ.get("ES_XS", "1") == "1"      # row-stationary short-K linear kernel (csrc/linear_xs.hip)
XS_RESIDUAL = _os.environ.get("ES_XS_RESIDUAL", "1") == "1"   # ... also for the K = 320 output projections that add a residual
XS_TARGET_WGS = 256
XS_FORCE_SLICES = 0
XS_MIN_M = int(_os.environ.get("ES_XS_MIN_M", "8192"))   # 0: no size policy (tests exercise every shape)
_zero_bias = {}


def xs_shape_reference(M: int, pw: "PackedWeight", min_m: int) -> bool:
    """The Python copy of the library's es_linear_xs_eligible (kept one more round as a guard, and for min_m != the shipped 8192:
    tests exercise every instantiation with min_m = 0).  K = 320 | 640 linear layers with an output width of whole 128-byte
    lines: the to_q|k|v and GEGLU projections of the 64x64 and 32x32 levels."""
    if pw.ksize != 1 or pw.kpad not in (320, 640) or pw.cin != pw.kpad or pw.ctail:
        return False
    ch = 64 if pw.kpad == 320 else 32
    line = ch * (128 // (ch if pw.geglu else 2 * ch))        # GEMM columns per 128-byte output line
    if pw.cout % line or pw.cout < 4 * line:
        return False
    # where it wins (tools/xs_bench.py, batch-1 shapes): 1.4-1.6x on the 14-sample launches of both levels and 1.05-1.1x on
    # the decoder's wide K = 320 projections; it loses where a workgroup's share of N is a few chunks (the activation
    # rows are re-read per slice and the 40 KB stages no longer amortise): small M with K = 640, narrow N at small M
    if min_m and (M < min_m or (M < 4 * min_m and pw.cout < (960 if pw.kpad == 320 else 1920))):
        return False
    return True


def xs_eligible(M: int, pw: "PackedWeight", pws, group_n, hw: int) -> bool:
    """Does this plain linear launch go to es_linear_xs?  The shape / size rule is the library's (es_linear_xs_eligible: what
    es_load_weights applies itself); the grouping conditions belong to the caller's launch."""
    if not XS_ENABLED:
        return False
    if pws is not None:
        if len(pws) > 4 or any((n * hw) % 256 for n in group_n) or any((q.ln_colsum is None) != (pw.ln_colsum is None) for q in pws):
            return False
    if XS_MIN_M != 8192:
        return xs_shape_reference(M, pw, XS_MIN_M)
    return bool(L.load().es_linear_xs_eligible(int(M), int(pw.ksize), int(pw.kpad), int(pw.cin), int(pw.ctail), int(pw.cout), int(pw.geglu)))


def linear_xs(x: torch.Tensor, pw, M: int, out: torch.Tensor, group_rows=None, residual: Optional[torch.Tensor] = None,
              gn: Optional[dict] = None) -> torch.Tensor:
    """x: [M, K] contiguous; pw (or list of pw for a grouped launch with `group_rows` rows each) -> out [M, cstore]
    (+ residual [M, cstore] contiguous: K = 320 only).  gn = dict(part, gamma, beta, groups, nchunk, hw, eps): GroupNorm in front
    (es_xs_desc.gn_part; gamma / beta lists for a grouped launch)."""
    pws = None
    if isinstance(pw, (list, tuple)):
        pws = list(pw) if len(pw) > 1 else None
        pw = pw[0]
    K = pw.kpad
    ch = 64 if K == 320 else 32
    pline = 128 // (ch if pw.geglu else 2 * ch)             # chunks per 128-byte output line
    total = pw.cout // ch
    lines = total // pline
    rbs = (M + 255) // 256
    want = max(1, min(lines, XS_TARGET_WGS // rbs))
    if XS_FORCE_SLICES:                                      # tool knob (tools/xs_slices_bench.py)
        want = max(1, min(lines, XS_FORCE_SLICES))
    lps = -(-lines // want)                                  # lines per slice
    nslices = -(-lines // lps)
    d = L.XsDesc()
    d.x, d.out = x.data_ptr(), out.data_ptr()

    def bias_of(q):
        if q.bias is not None:
            return q.bias.data_ptr()
        key = (x.device, q.rows_padded)
        z = _zero_bias.get(key)
        if z is None:
            z = _zero_bias[key] = torch.zeros(q.rows_padded, dtype=torch.float32, device=x.device)
        return z.data_ptr()
    d.w, d.bias = pw.w.data_ptr(), bias_of(pw)
    d.M, d.K, d.Cout, d.rows_padded = M, K, pw.cout, pw.rows_padded
    d.ldo = out.shape[-1]
    d.geglu, d.ln, d.ln_eps = int(pw.geglu), int(pw.ln_colsum is not None), pw.ln_eps
    d.nslices, d.chunks_per_slice, d.dtype = nslices, lps * pline, _dt(x)
    d.residual = residual.data_ptr() if residual is not None else None
    if gn is not None:
        d.gn_part, d.gn_groups, d.gn_nchunk, d.gn_hw, d.gn_eps = gn["part"].data_ptr(), gn["groups"], gn["nchunk"], gn["hw"], gn["eps"]
        gam, bet = gn["gamma"], gn["beta"]
        if pws is not None:
            if len(gam) != len(pws) or len(bet) != len(pws):
                raise L.EdgeStyleHipError("grouped linear_xs: one GroupNorm parameter set per weight group")
            for g in range(len(pws)):
                d.gn_gamma_g[g], d.gn_beta_g[g] = gam[g].data_ptr(), bet[g].data_ptr()
        else:
            if isinstance(gam, (list, tuple)):
                gam, bet = gam[0], bet[0]
            d.gn_gamma, d.gn_beta = gam.data_ptr(), bet.data_ptr()
    if pws is not None:
        d.ngroups = len(pws)
        acc = 0
        for g, (q, n) in enumerate(zip(pws, group_rows)):
            if (q.rows_padded, q.kpad, q.cout, q.geglu) != (pw.rows_padded, pw.kpad, pw.cout, pw.geglu):
                raise L.EdgeStyleHipError("grouped linear_xs: weight geometry differs between groups")
            acc += n // 128
            d.mt_end[g] = acc
            d.w_g[g] = q.w.data_ptr()
            d.bias_g[g] = bias_of(q)
    if PROFILE is not None:
        d.prof = PROFILE.next((2.0 * M * pw.cout * K, 1, (M, pw.cout, K, 1, 1, 0),
                               dict(N=M, H=1, W=1, C1=K, C2=0, cout=pw.cout, k=1, stride=1, pad=0, upsample=False,
                                    geglu=pw.geglu, splitk=1, Hout=1, Wout=1, residual=residual is not None, temb=False, bn=0, stages=3,
                                    group_n=list(group_rows) if pws is not None else None, ctail=0, kernel="linear_xs",
                                    algorithmic_bytes=_algorithmic_bytes(x, None, pw, pws, out, residual))))
        dd = L.XsDesc()
        C.memmove(C.byref(dd), C.byref(d), C.sizeof(L.XsDesc))
        dd.prof = None
        PROFILE.descs.append(dd)
    L.check(L.load().es_linear_xs(C.byref(d), _stream()), "es_linear_xs")
    return out


# Wide residual stream (es_gemm_desc.residual_lo / out_lo): the tensors every block adds into travel as hi + lo, the low part riding
# on the tensor object as `._lo`.  "auto" (default): bf16 pipelines only - bf16 keeps 8 significant bits, and the rounding of the
# stream's sums, accumulating from block to block, was 5 dB of configs[4]'s error (profiles/r04_bf16_error_budget_2steps.txt);
# fp16 pipelines keep the single-tensor stream (11 bits; nothing measurable to gain, bandwidth to lose).  ES_WIDE_STREAM=0 | 1 forces.
WIDE_STREAM = _os.environ.get("ES_WIDE_STREAM", "auto")


# GroupNorm statistics handed over by the producing GEMM (es_gemm_desc.gn_part -> es_gn_desc.ext_chunks): the convolution whose
# output a GroupNorm normalises writes the per-(sample, 64-pixel block, group) sums from its epilogue; the GroupNorm is then one
# streaming pass (no statistics launch, no second read).  The table rides on the output tensor as `._gnp`.
# OFF by default (ES_GN_HANDOVER=1: at the 64 x 64 level, where the stand-alone GroupNorm is the two-launch form; "all": everywhere):
# built and measured in round 4, it LOSES - the epilogues' extra LDS passes and barriers cost more than the statistics launch they
# replace (batch 1: 476 vs 466 ms per image at the 64 x 64 level only, 482 everywhere; batch 8: 2623 vs 2553 ms;
# profiles/r04_gn_handover.txt) - the deeper levels' one-launch slab GroupNorm already reads its input once.
GN_HANDOVER = _os.environ.get("ES_GN_HANDOVER", "0") in ("1", "all")


def gn_handover(hw: int, c: int, groups: int) -> bool:
    """Does a GEMM whose [.., hw, c] output feeds a GroupNorm over `groups` groups hand the statistics over?  A rule of the SHAPE
    alone (grouped and per-net launches of a layer must agree): where the stand-alone GroupNorm is the two-launch form - slabs too
    large for a workgroup's registers, the 64 x 64 level - the hand-over removes its statistics launch and second read; the
    one-launch slab form of the deeper levels already reads its input once, and there the hand-over measured a net loss
    (profiles/r04_gn_handover.txt)."""
    if not GN_HANDOVER or c % 8 or hw % 64 or c % groups or c // groups > 64:
        return False
    return GN_HANDOVER_ALL or not L.load().es_group_norm_is_slab(hw, c, groups)


GN_HANDOVER_ALL = _os.environ.get("ES_GN_HANDOVER", "0") == "all"     # tool switch: also where the consumer is the slab form


def wide_stream(dtype) -> bool:
    return WIDE_STREAM == "1" or (WIDE_STREAM == "auto" and dtype == torch.bfloat16)


def carry_lo(dst: torch.Tensor, src: torch.Tensor, shape=None) -> torch.Tensor:
    """a view / reshape `dst` of the stream tensor `src` keeps its low part (viewed alike)"""
    lo = getattr(src, "_lo", None)
    if lo is not None:
        dst._lo = lo.reshape(dst.shape if shape is None else shape)
    return dst


def conv_gemm(x: torch.Tensor, pw: PackedWeight, *, stride: int = 1, pad: Optional[int] = None,
              upsample: bool = False, x2: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None,
              residual: Optional[torch.Tensor] = None, act: int = L.ACT_NONE, out_scale: float = 1.0,
              out_scale_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              out_hw=None, splitk: Optional[int] = None, stages: int = 0,
              group_n: Optional[Sequence[int]] = None, tail: Optional[Sequence[torch.Tensor]] = None,
              x_rep: int = 1, wide: bool = False, out_lo: Optional[torch.Tensor] = None, gn_groups: int = 0,
              gn_part: Optional[torch.Tensor] = None):
    """x: [N,H,W,C1] (+ x2 [N,H,W,C2]); returns [N,Hout,Wout,Cout] (Cout/2 for GEGLU).

    x_rep > 1: the launch covers x_rep * N samples, sample n reading x[n % N] (es_gemm_desc.x_nmod): one sample tensor
    feeding several groups of a grouped launch without a replicated copy.

    Grouped launch: `pw` is a list of PackedWeights of identical geometry and `group_n` the number of consecutive
    samples of x each of them applies to (sum = N): one launch instead of len(pw)."""
    pws = None
    if isinstance(pw, (list, tuple)):
        pws = list(pw)
        pw = pws[0]
        if len(pws) == 1:
            pws = None
    N, H, W, C1 = x.shape
    nsrc = N
    if x_rep > 1:
        if x2 is not None or tail:
            raise L.EdgeStyleHipError("conv_gemm: x_rep needs a single source")
        N = N * x_rep
    C2 = 0 if x2 is None else x2.shape[3]
    if C1 + C2 != pw.cin:
        raise L.EdgeStyleHipError(f"conv_gemm: input channels {C1}+{C2} != packed {pw.cin}")
    tails = [t for t in (tail or ()) if t is not None]
    if sum(t.shape[3] for t in tails) != pw.ctail or len(tails) > 2:
        raise L.EdgeStyleHipError(f"conv_gemm: tail channels {[t.shape[3] for t in tails]} != packed {pw.ctail}")
    k = pw.ksize
    if pad is None:
        pad = 1 if k == 3 else 0
    Hin, Win = (H * 2, W * 2) if upsample else (H, W)
    if out_hw is None:
        Hout = (Hin + 2 * pad - k) // stride + 1
        Wout = (Win + 2 * pad - k) // stride + 1
    else:
        Hout, Wout = out_hw
    act_i = L.ACT_GEGLU if pw.geglu else act
    cstore = pw.cout // 2 if pw.geglu else pw.cout
    if out is None:
        out = torch.empty((N, Hout, Wout, cstore), dtype=x.dtype, device=x.device)
    M = N * Hout * Wout
    # (a residual rides on es_linear_xs at K = 320 - Attention.to_out / proj_out of the 64 x 64 level - unless this launch carries the
    #  two-word residual stream of a bf16 pipeline, which only es_conv_gemm implements)
    xs_res = residual is None or (XS_RESIDUAL and pw.kpad == 320 and not pw.geglu and pw.ln_colsum is None and residual.is_contiguous()
                                  and residual.numel() == M * cstore and not (wide and wide_stream(x.dtype)))
    if (k == 1 and stride == 1 and not upsample and x2 is None and temb is None and xs_res and not tails
            and x_rep == 1 and act == L.ACT_NONE and out_scale == 1.0 and out_scale_dev is None and splitk is None and FORCE_BN == 0
            and x.is_contiguous() and out.is_contiguous()
            and xs_eligible(M, pw, pws, group_n, Hout * Wout)):
        linear_xs(x.reshape(M, C1), pws if pws is not None else pw, M, out.reshape(M, cstore),
                  None if pws is None else [n * Hout * Wout for n in group_n], residual=residual)
        return out
    big_ok = BIG_TILE and C1 % BK == 0 and C2 % BK == 0 and not pw.geglu and pw.ln_colsum is None and \
        (group_n is None or all((n * Hout * Wout) % 256 == 0 for n in group_n))
    small_ok = SMALL_TILE and C1 % BK == 0 and C2 % BK == 0 and not pw.geglu and FORCE_WAVES != 8
    # the 256 x 256 phase-interleaved tile (round 5): LayerNorm-folded and GEGLU linear layers whose N is a multiple of 256
    ln256_ok = BIG_TILE_256 and (pw.geglu or pw.ln_colsum is not None) and k == 1 and stride == 1 and not upsample and x2 is None \
        and not tails and temb is None and x_rep == 1 and C1 % BK == 0 and pw.rows_padded % 256 == 0 and pw.cout % 8 == 0 \
        and not (pw.geglu and residual is not None) and not wide and gn_groups == 0 and FORCE_WAVES != 8 \
        and (group_n is None or all((n * Hout * Wout) % 256 == 0 for n in group_n))
    cand = ((256,) if ln256_ok else ()) + ((320,) if big_ok else ()) + (160, 128) + ((64,) if small_ok else ())
    bn, auto_splitk, auto_stages = plan_gemm(M, pw.rows_padded, pw.kpad, pw.geglu,
                                             bns=cand if FORCE_BN == 0 else (FORCE_BN,), allow_split=pw.ln_colsum is None)
    if bn == 320 and FORCE_BN == 0 and (auto_splitk if splitk is None else splitk) == 1 and not L.load().es_conv_gemm8p_form_ok(
            int(act_i), int(pw.cout), int(temb is not None), int(Hout * Wout), int(residual is not None)):
        # the 256 x 320 tile does not implement this epilogue form (an activation, time-embedding rows that differ inside a 128-pixel
        # half or meet a residual): plan again without it, so that the tile planned, recorded and reported is the tile that runs
        bn, auto_splitk, auto_stages = plan_gemm(M, pw.rows_padded, pw.kpad, pw.geglu, bns=tuple(b for b in cand if b != 320),
                                                 allow_split=pw.ln_colsum is None)
    if splitk is None:
        splitk = auto_splitk
        stages = stages or auto_stages
    d = L.GemmDesc()
    d.x, d.x2, d.w = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), pw.w.data_ptr()
    d.bias = pw.bias.data_ptr() if pw.bias is not None else None
    d.temb = temb.data_ptr() if temb is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.out_scale_dev = out_scale_dev.data_ptr() if out_scale_dev is not None else None
    d.out = out.data_ptr()
    d.N, d.Hsrc, d.Wsrc, d.C1, d.C2 = N, H, W, C1, C2
    d.Hout, d.Wout, d.Cout = Hout, Wout, pw.cout
    d.rows_padded, d.Kpad = pw.rows_padded, pw.kpad
    d.ksize, d.stride, d.pad = k, stride, pad
    d.upsample = 1 if upsample else 0
    d.temb_stride = temb.stride(0) if temb is not None else 0
    d.act, d.splitk, d.bn, d.dtype, d.out_scale = act_i, splitk, bn, _dt(x), out_scale
    d.stages = stages or FORCE_STAGES
    d.korder = pw.korder
    if (gn_groups > 0 and gn_handover(Hout * Wout, cstore, gn_groups) and not pw.geglu and pw.ln_colsum is None
            and pw.cout // gn_groups <= (160 if bn == 320 else bn)):
        # the consumer of `out` is a GroupNorm over gn_groups groups: hand its statistics over from this launch's epilogue
        shape = (N, 2 * (Hout * Wout // 64), gn_groups, 2)
        part = gn_part if gn_part is not None else torch.empty(shape, dtype=torch.float32, device=x.device)
        if tuple(part.shape) != shape or part.dtype != torch.float32 or not part.is_contiguous():
            raise L.EdgeStyleHipError(f"conv_gemm: gn_part must be a contiguous fp32 {shape}")
        d.gn_part, d.gn_groups = part.data_ptr(), gn_groups
        out._gnp = (part, gn_groups)
    if wide and residual is not None and wide_stream(x.dtype) and cstore % 8 == 0 and not pw.geglu:
        # this launch adds into the residual stream: the sum over (residual hi + lo) in fp32, stored as hi + lo
        lo_in = getattr(residual, "_lo", None)
        if lo_in is not None:
            if lo_in.numel() != residual.numel() or not lo_in.is_contiguous():
                raise L.EdgeStyleHipError("conv_gemm: the residual's low part does not match it")
            d.residual_lo = lo_in.data_ptr()
        if out_lo is None:
            out_lo = torch.empty_like(out)
        elif out_lo.shape != out.shape or out_lo.dtype != out.dtype or not out_lo.is_contiguous():
            raise L.EdgeStyleHipError("conv_gemm: out_lo must match out")
        d.out_lo = out_lo.data_ptr()
        out._lo = out_lo
    # XCD chunk order: keep the larger operand's tiles together on one XCD (see conv_gemm_kernel)
    d.xcd_m_fastest = (1 if (pws is None and splitk == 1 and M <= 2048 and pw.w.numel() > x.numel() * x_rep + (x2.numel() if x2 is not None else 0)) else 0) \
        if XCD_ORDER < 0 else XCD_ORDER
    d.x_nmod = nsrc if x_rep > 1 else 0
    if FORCE_WAVES:
        d.waves = FORCE_WAVES
    elif EIGHT_WAVES and k == 1 and M <= 65536 and C1 % BK == 0 and C2 % BK == 0 and bn not in (64, 320, 256) \
            and not (int(d.stages) == 4 and bn != 128) and int(d.stages) != 3:
        # 1x1 convs / linears are short-K, latency-bound launches: two waves per SIMD on the same 128-pixel tile overlap
        # DMA issue, fragment reads and MFMAs (tools/gemm_tune.py: 3-15 % on every 1x1 shape of a batch-1 step, none on 3x3 or on the
        # memory-bound 1x1 launches of large batches)
        d.waves = 8
    if splitk > 1:
        ws = _get_workspace(splitk * M * pw.rows_padded * 4, x.device)
        d.workspace = ws.data_ptr()
    if pw.ln_colsum is not None:
        if (pws is not None and any(q.ln_colsum is None for q in pws)) or x2 is not None or k != 1:
            raise L.EdgeStyleHipError("LayerNorm-folded weights need a plain linear launch (all groups folded)")
        d.ln_colsum, d.ln_eps = pw.ln_colsum.data_ptr(), pw.ln_eps
    if tails:
        if any(t.shape[:3] != (N, Hout, Wout) or not t.is_contiguous() for t in tails):
            raise L.EdgeStyleHipError("conv_gemm: tail sources must be contiguous [N,Hout,Wout,C]")
        d.t1, d.Ct1 = tails[0].data_ptr(), tails[0].shape[3]
        if len(tails) == 2:
            d.t2, d.Ct2 = tails[1].data_ptr(), tails[1].shape[3]
    if pws is not None:
        hw = Hout * Wout
        gran = 256 if bn in (320, 256) else BM
        if len(pws) > 4 or len(group_n) != len(pws) or sum(group_n) != N or any((n * hw) % gran for n in group_n):
            raise L.EdgeStyleHipError(f"grouped conv_gemm: groups must cover N in whole {gran}-pixel tiles (<= 4 groups)")
        d.ngroups = len(pws)
        acc = 0
        for g, (q, n) in enumerate(zip(pws, group_n)):
            if (q.rows_padded, q.kpad, q.cout, q.cin, q.ksize, q.geglu, q.ctail, q.korder) != \
                    (pw.rows_padded, pw.kpad, pw.cout, pw.cin, pw.ksize, pw.geglu, pw.ctail, pw.korder):
                raise L.EdgeStyleHipError("grouped conv_gemm: weight geometry differs between groups")
            acc += n * hw // BM
            d.mt_end[g] = acc
            d.w_g[g] = q.w.data_ptr()
            d.bias_g[g] = q.bias.data_ptr() if q.bias is not None else None
            d.ln_colsum_g[g] = q.ln_colsum.data_ptr() if q.ln_colsum is not None else None
    if PROFILE is not None:           # bench.py roofline leg: in-kernel s_memrealtime stamps for this launch
        d.prof = PROFILE.next((2.0 * M * pw.cout * (k * k * (C1 + C2) + pw.ctail), k,
                               (M, pw.cout, k * k * (C1 + C2) + pw.ctail, stride, splitk, bn),
                               dict(N=N, H=H, W=W, C1=C1, C2=C2, cout=pw.cout, k=k, stride=stride, pad=pad,
                                    upsample=bool(upsample), geglu=pw.geglu, splitk=splitk, Hout=Hout, Wout=Wout,
                                    residual=residual is not None, temb=temb is not None, bn=bn,
                                    stages=int(d.stages), group_n=list(group_n) if pws is not None else None,
                                    ctail=pw.ctail,
                                    algorithmic_bytes=_algorithmic_bytes(x, x2, pw, pws, out, residual)
                                    + sum(t.numel() * t.element_size() for t in tails))))
        dd = L.GemmDesc()
        C.memmove(C.byref(dd), C.byref(d), C.sizeof(L.GemmDesc))
        dd.prof = None
        PROFILE.descs.append(dd)
    L.check(L.load().es_conv_gemm(C.byref(d), _stream()), "es_conv_gemm")
    return out


def _algorithmic_bytes(x, x2, pw, pws, out, residual) -> int:
    """HBM bytes a launch must move at least once: activations in, every weight set, output out, residual in."""
    n = x.numel() * x.element_size() + (x2.numel() * x2.element_size() if x2 is not None else 0)
    n += sum(q.w.numel() * q.w.element_size() for q in (pws if pws is not None else [pw]))
    n += out.numel() * out.element_size()
    if residual is not None:
        n += residual.numel() * residual.element_size()
    return int(n)


def linear(x: torch.Tensor, pw: PackedWeight, **kw) -> torch.Tensor:
    """x: [..., K] -> [..., Cout]; runs the same implicit-GEMM kernel with a 1x1 'image' of M pixels."""
    shp = x.shape
    M = x.numel() // shp[-1]
    out = kw.pop("out", None)
    res = kw.pop("residual", None)
    rep = kw.get("x_rep", 1)
    p0 = pw[0] if isinstance(pw, (list, tuple)) else pw
    cstore = p0.cout // 2 if p0.geglu else p0.cout
    y = conv_gemm(x.reshape(M, 1, 1, shp[-1]), pw,
                  residual=None if res is None else carry_lo(res.reshape(M * rep, 1, 1, cstore), res),
                  out=None if out is None else out.reshape(M * rep, 1, 1, cstore), **kw)
    if rep > 1:
        return carry_lo(y.reshape(shp[0] * rep, *shp[1:-1], cstore), y)
    return carry_lo(y.reshape(*shp[:-1], cstore), y)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q: [N,Sq,heads*d] view (any row stride), k/v: [N,Skv,heads*d] views -> [N,Sq,heads*d] contiguous."""
    N, Sq, Cq = q.shape
    Skv = k.shape[1]
    dh = Cq // heads
    if out is None:
        out = torch.empty((N, Sq, Cq), dtype=q.dtype, device=q.device)
    for t in (q, k, v, out):
        if t.stride(2) != 1:
            raise L.EdgeStyleHipError("attention: innermost stride must be 1")
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.N, d.heads, d.Sq, d.Skv, d.d = N, heads, Sq, Skv, dh
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
    d.bsq, d.bsk, d.bsv, d.bso = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    d.scale = scale if scale is not None else 1.0 / math.sqrt(dh)
    d.dtype = _dt(q)
    L.check(L.load().es_attention(C.byref(d), _stream()), "es_attention")
    return out


_gn_partials = {}


def group_norm(x: torch.Tensor, gamma, beta, groups: int, eps: float, silu: bool,
               x2: Optional[torch.Tensor] = None, group_n: Optional[Sequence[int]] = None) -> torch.Tensor:
    """x: [N,H,W,C1] (+x2 [N,H,W,C2]) -> normalised [N,H,W,C1+C2].  gamma/beta may be lists (one per group of
    `group_n` consecutive samples): one launch for several nets' GroupNorms."""
    N, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[3]
    out = torch.empty((N, H, W, C1 + C2), dtype=x.dtype, device=x.device)
    key = (x.device, N, groups, LANE)
    part = _gn_partials.get(key)
    if part is None:
        part = torch.empty(L.load().es_group_norm_partials_bytes(N, groups) // 4, dtype=torch.float32, device=x.device)
        _gn_partials[key] = part
    d = L.GnDesc()
    d.x, d.x2, d.out = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), out.data_ptr()
    if isinstance(gamma, (list, tuple)) and len(gamma) > 1:
        if len(gamma) > 4 or len(group_n) != len(gamma) or sum(group_n) != N:
            raise L.EdgeStyleHipError("grouped group_norm: bad group table")
        d.ngroups = len(gamma)
        acc = 0
        for g, n in enumerate(group_n):
            acc += n
            d.n_end[g] = acc
            d.gamma_g[g] = gamma[g].data_ptr()
            d.beta_g[g] = beta[g].data_ptr()
        d.partials = part.data_ptr()
    else:
        if isinstance(gamma, (list, tuple)):
            gamma, beta = gamma[0], beta[0]
        d.gamma, d.beta, d.partials = gamma.data_ptr(), beta.data_ptr(), part.data_ptr()
    gnp = getattr(x, "_gnp", None)
    if gnp is not None and x2 is None and gn_handover(H * W, C1, groups) and gnp[1] == groups \
            and tuple(gnp[0].shape) == (N, 2 * (H * W // 64), groups, 2):
        d.partials, d.ext_chunks = gnp[0].data_ptr(), 2 * (H * W // 64)      # the producer's statistics: one streaming pass
    d.N, d.HW, d.C1, d.C2, d.groups = N, H * W, C1, C2, groups
    d.eps, d.silu, d.dtype = eps, 1 if silu else 0, _dt(x)
    L.check(L.load().es_group_norm(C.byref(d), _stream()), "es_group_norm")
    return out


# Transformer2DModel.norm -> proj_in as ONE read of the tensor: a statistics pass (es_group_norm stats_only) and the projection on
# es_linear_xs, which normalises the rows it holds in registers (es_xs_desc.gn_part) - where the projection runs on that kernel anyway
# (the grouped launches of the two shallow levels, the UNet's own at batch 8), and at K = 320 from 8192 rows on, where the row-stationary
# kernel loses ~2 us to the tiled one but the apply pass it replaces costs 8-10 (profiles/r05_gn_proj_in.txt).  ES_GN_FOLD=0: two launches
# + projection as before.  The native builder applies the same rule (csrc/builder.hip gn_proj_in).
GN_FOLD = _os.environ.get("ES_GN_FOLD", "1") == "1"


def gn_fold_ok(M: int, hw: int, groups: int, pw: "PackedWeight", pws, group_n) -> bool:
    if not (GN_FOLD and XS_ENABLED) or GN_HANDOVER or hw % 256 or groups > 32:
        return False
    if pw.ksize != 1 or pw.kpad not in (320, 640) or pw.cin != pw.kpad or pw.ctail or pw.geglu or pw.ln_colsum is not None \
            or pw.kpad % groups or not xs_shape_reference(M, pw, 0):
        return False
    if M < (8192 if pw.kpad == 320 else 32768):
        return False
    if pws is not None and (len(pws) > 4 or any((n * hw) % 256 for n in group_n)
                            or any((q.rows_padded, q.kpad, q.cout, q.geglu, q.ln_colsum is None) != (pw.rows_padded, pw.kpad, pw.cout, pw.geglu, True)
                                   for q in pws)):
        return False
    return True


def gn_proj_in(x: torch.Tensor, gamma, beta, groups: int, eps: float, pw, group_n: Optional[Sequence[int]] = None) -> torch.Tensor:
    """x [N,H,W,C] -> proj_in(GroupNorm(x)) [N,H,W,Cout] (no activation in between).  gamma / beta / pw: lists for a grouped launch
    (group_n samples each)."""
    N, H, W, Cc = x.shape
    pws = list(pw) if isinstance(pw, (list, tuple)) and len(pw) > 1 else None
    p0 = pw[0] if isinstance(pw, (list, tuple)) else pw
    M, hw = N * H * W, H * W
    if not (x.is_contiguous() and gn_fold_ok(M, hw, groups, p0, pws, group_n)):
        kw = {} if group_n is None else dict(group_n=group_n)
        return conv_gemm(group_norm(x, gamma, beta, groups, eps, False, **kw), pw, **kw)
    key = (x.device, N, groups, LANE)
    part = _gn_partials.get(key)
    if part is None:
        part = torch.empty(L.load().es_group_norm_partials_bytes(N, groups) // 4, dtype=torch.float32, device=x.device)
        _gn_partials[key] = part
    d = L.GnDesc()
    d.x, d.partials = x.data_ptr(), part.data_ptr()
    d.N, d.HW, d.C1, d.C2, d.groups = N, hw, Cc, 0, groups
    d.eps, d.silu, d.dtype, d.stats_only = eps, 0, _dt(x), 1
    L.check(L.load().es_group_norm(C.byref(d), _stream()), "es_group_norm")
    out = torch.empty((N, H, W, p0.cout), dtype=x.dtype, device=x.device)
    gl = list(gamma) if isinstance(gamma, (list, tuple)) else [gamma]
    bl = list(beta) if isinstance(beta, (list, tuple)) else [beta]
    linear_xs(x.reshape(M, Cc), pws if pws is not None else p0, M, out.reshape(M, p0.cout),
              None if pws is None else [n * hw for n in group_n],
              gn=dict(part=part, gamma=gl, beta=bl, groups=groups, nchunk=L.load().es_group_norm_chunks(hw), hw=hw, eps=eps))
    return out


def layer_norm(x: torch.Tensor, gamma, beta, eps: float = 1e-5, group_rows: Optional[Sequence[int]] = None) -> torch.Tensor:
    Cc = x.shape[-1]
    M = x.numel() // Cc
    out = torch.empty_like(x)
    if isinstance(gamma, (list, tuple)) and len(gamma) > 1:
        if len(gamma) > 4 or len(group_rows) != len(gamma) or sum(group_rows) != M:
            raise L.EdgeStyleHipError("grouped layer_norm: bad group table")
        d = L.LnDesc()
        d.x, d.out = x.data_ptr(), out.data_ptr()
        acc = 0
        for g, n in enumerate(group_rows):
            acc += n
            d.row_end[g] = acc
            d.gamma_g[g] = gamma[g].data_ptr()
            d.beta_g[g] = beta[g].data_ptr()
        d.ngroups, d.M, d.C, d.eps, d.dtype = len(gamma), M, Cc, eps, _dt(x)
        L.check(L.load().es_layer_norm_grouped(C.byref(d), _stream()), "es_layer_norm_grouped")
        return out
    if isinstance(gamma, (list, tuple)):
        gamma, beta = gamma[0], beta[0]
    L.check(L.load().es_layer_norm(_ptr(x), _ptr(out), _ptr(gamma), _ptr(beta), M, Cc, eps, _dt(x), _stream()),
            "es_layer_norm")
    return out


def timestep_embedding(t: torch.Tensor, dim: int, dtype) -> torch.Tensor:
    """t: fp32 device [N] -> [N, dim]"""
    N = t.numel()
    out = torch.empty((N, dim), dtype=dtype, device=t.device)
    L.check(L.load().es_timestep_embedding(_ptr(t), _ptr(out), N, dim, _DT[dtype], _stream()), "es_timestep_embedding")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(a)
    L.check(L.load().es_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), _dt(a), _stream()), "es_add")
    _drop_riders(out)
    return out


def _drop_riders(t: torch.Tensor):
    """A tensor written in place no longer matches what rides on its Python object: the low word of the two-word residual stream
    (`._lo`) and the GroupNorm statistics its producer handed over (`._gnp`) describe the OLD contents."""
    for name in ("_lo", "_gnp"):
        if hasattr(t, name):
            delattr(t, name)


def nchw_to_nhwc(x: torch.Tensor, dtype, cpad: Optional[int] = None) -> torch.Tensor:
    """fp32 NCHW (contiguous, device) -> NHWC dtype with channels zero-padded to cpad."""
    N, Cc, H, W = x.shape
    cp = cpad or Cc
    x = x.contiguous().float()
    out = torch.empty((N, H, W, cp), dtype=dtype, device=x.device)
    L.check(L.load().es_nchw_f32_to_nhwc(_ptr(x), _ptr(out), N, Cc, H * W, cp, _DT[dtype], _stream()),
            "es_nchw_f32_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, channels: Optional[int] = None, scale: float = 1.0, shift: float = 0.0,
                 clamp01: bool = False) -> torch.Tensor:
    """NHWC dtype -> fp32 NCHW, optionally y = clamp(x*scale+shift, 0, 1) (image postprocess PL:570-572)."""
    N, H, W, Cs = x.shape
    Cc = channels or Cs
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
    L.check(L.load().es_nhwc_to_nchw_f32(_ptr(x), _ptr(out), N, Cc, H * W, Cs, scale, shift, 1 if clamp01 else 0,
                                         _dt(x), _stream()), "es_nhwc_to_nchw_f32")
    return out


def vae_sample(moments: torch.Tensor, noise_nchw: torch.Tensor, latent: int, lpad: int, scaling: float):
    N, H, W, _ = moments.shape
    z = torch.empty((N, H, W, lpad), dtype=moments.dtype, device=moments.device)
    L.check(L.load().es_vae_sample(_ptr(moments), _ptr(noise_nchw.contiguous().float()), _ptr(z), N, H * W, latent,
                                   lpad, scaling, _dt(moments), _stream()), "es_vae_sample")
    return z


def cfg_ddim_step(noise: torch.Tensor, latents: torch.Tensor, model_in: torch.Tensor, coef: torch.Tensor,
                  step_idx: torch.Tensor, guidance_scale: float, cfg: bool):
    """noise [2B|B,H,W,L] dtype; latents fp32 [B,H,W,L] (in place); model_in [2B|B,H,W,Ls] dtype (rewritten)."""
    B, H, W, Lc = latents.shape
    L.check(L.load().es_cfg_ddim_step(_ptr(noise), _ptr(latents), _ptr(model_in), _ptr(coef), _ptr(step_idx),
                                      guidance_scale, B, H * W, Lc, model_in.shape[3], 1 if cfg else 0, coef.shape[0],
                                      _dt(noise), _stream()), "es_cfg_ddim_step")


def cfg_unipc_step(noise, latents, last_sample, m0, m1, model_in, coef, step_idx, guidance_scale: float, cfg: bool):
    """UniPC counterpart of cfg_ddim_step; last_sample/m0/m1 are fp32 [B,H,W,L] state tensors (in place)."""
    B, H, W, Lc = latents.shape
    L.check(L.load().es_cfg_unipc_step(_ptr(noise), _ptr(latents), _ptr(last_sample), _ptr(m0), _ptr(m1),
                                       _ptr(model_in), _ptr(coef), _ptr(step_idx), guidance_scale, B, H * W, Lc,
                                       model_in.shape[3], 1 if cfg else 0, coef.shape[0], _dt(noise), _stream()),
            "es_cfg_unipc_step")


def memcpy(dst: torch.Tensor, src: torch.Tensor):
    """dst[:] = src, both contiguous device tensors of equal byte size (a C-ABI call: part of recorded plans)."""
    nb = src.numel() * src.element_size()
    if not (dst.is_contiguous() and src.is_contiguous()) or dst.numel() * dst.element_size() != nb:
        raise L.EdgeStyleHipError("memcpy: contiguous tensors of equal byte size")
    L.check(L.load().es_memcpy(_ptr(dst), _ptr(src), nb, _stream()), "es_memcpy")
    _drop_riders(dst)


def memcpy2d(dst_ptr: int, dpitch: int, src_ptr: int, spitch: int, width: int, height: int):
    """`height` rows of `width` bytes, pitches in bytes (src pitch 0 = broadcast one row is NOT supported: use height 1)."""
    L.check(L.load().es_memcpy2d(C.c_void_p(dst_ptr), dpitch, C.c_void_p(src_ptr), spitch, width, height, _stream()),
            "es_memcpy2d")


def fill_f32(dst: torch.Tensor, value: float):
    L.check(L.load().es_fill_f32(_ptr(dst), float(value), dst.numel(), _stream()), "es_fill_f32")


def latents_to_input(latents: torch.Tensor, model_in: torch.Tensor, cfg: bool):
    """latents fp32 [B,H,W,L] -> model_in [2B|B,H,W,Ls] compute dtype, channel-padded, both CFG halves (PL:443-447)."""
    B, H, W, Lc = latents.shape
    L.check(L.load().es_latents_to_input(_ptr(latents), _ptr(model_in), B, H * W, Lc, model_in.shape[3], 1 if cfg else 0,
                                         _dt(model_in), _stream()), "es_latents_to_input")


def incr(ctr: torch.Tensor):
    L.check(L.load().es_incr(_ptr(ctr), _stream()), "es_incr")


def gather_row(table: torch.Tensor, idx: torch.Tensor, out: torch.Tensor):
    """out[:] = table[idx[0]] (fp32), selected on the device"""
    L.check(L.load().es_gather_row(_ptr(table), _ptr(idx), _ptr(out), out.numel(), table.shape[0], _stream()),
            "es_gather_row")


_fusion_scratch = {}


def _fusion_desc(res, res_bs, params: dict, N: int, HW: int, Cc: int, scales, scales_dev, out, eps, slot: int,
                 addend: Optional[torch.Tensor] = None):
    t0 = res[0]
    key = (t0.device, N, slot)
    sc = _fusion_scratch.get(key)
    if sc is None:
        sc = torch.empty(L.load().es_fusion_scratch_bytes(N) // 4, dtype=torch.float32, device=t0.device)
        _fusion_scratch[key] = sc
    u = torch.empty((N, HW, Cc), dtype=t0.dtype, device=t0.device)
    if out is None:
        out = torch.empty((N, HW, Cc), dtype=t0.dtype, device=t0.device)
    d = L.FusionDesc()
    for i in range(6):
        d.res[i] = res[i].data_ptr()
        d.res_bs[i] = res_bs[i]
        d.res_scale[i] = float(scales[i])
    d.res_scale_dev = scales_dev.data_ptr() if scales_dev is not None else None
    for name in ("w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "w3", "b3"):
        setattr(d, name, params[name].data_ptr())
    d.scratch, d.u, d.out = sc.data_ptr(), u.data_ptr(), out.data_ptr()
    d.N, d.HW, d.C, d.eps, d.dtype = N, HW, Cc, eps, _dt(t0)
    if addend is not None:
        if addend.numel() != N * HW * Cc or not addend.is_contiguous() or addend.dtype != t0.dtype:
            raise L.EdgeStyleHipError("fusion addend must be a contiguous [N,HW,C] tensor of the compute dtype")
        d.addend = addend.data_ptr()
    return d, u, out


def fusion_block(res, res_bs, params: dict, N: int, HW: int, Cc: int, scales, scales_dev=None,
                 out: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """res: 6 tensors (or views) whose data_ptr() is sample 0 of net i; res_bs: 6 batch strides (elements)."""
    d, _u, out = _fusion_desc(res, res_bs, params, N, HW, Cc, scales, scales_dev, out, eps, 0)
    L.check(L.load().es_fusion_block(C.byref(d), _stream()), "es_fusion_block")
    return out


def fusion_blocks(blocks, N: int, scales, scales_dev=None, eps: float = 1e-5, addends=None, first_block: int = 0):
    """All fusion blocks of a step in three launches.  blocks: list of (res, res_bs, params, HW, Cc) as for
    fusion_block; returns the list of [N,HW,Cc] outputs.  addends: optional list of [N,HW,Cc] tensors added to the
    outputs (the UNet skip tensors: saves the 13 separate adds of PL:500-510)."""
    keep, outs = [], []
    for k0 in range(0, len(blocks), L.FUSION_MAX_BATCH):
        part = blocks[k0:k0 + L.FUSION_MAX_BATCH]
        arr = (L.FusionDesc * len(part))()
        for k, (res, res_bs, params, HW, Cc) in enumerate(part):
            d, u, out = _fusion_desc(res, res_bs, params, N, HW, Cc, scales, scales_dev, None, eps, first_block + k0 + k,
                                     None if addends is None else addends[k0 + k])
            arr[k] = d
            keep.append(u)
            outs.append(out)
        L.check(L.load().es_fusion_blocks(arr, len(part), _stream()), "es_fusion_blocks")
    return outs


def pack_fusion_params(sd: dict, prefix: str, dtype, device) -> dict:
    """Repack one ControlNetBlock (MC:23-53) pixel-major for es_fusion_block."""
    w1 = sd[f"{prefix}.first_conv.weight"].float()            # [3C, 2, 1, 1], out channel j = 3c + p
    c3 = w1.shape[0]
    c = c3 // 3
    g1 = sd[f"{prefix}.first_normalization.weight"].float()   # [3C, H, W]
    hw = g1.shape[1] * g1.shape[2]

    def plane3(t):   # [3C,H,W] -> [HW, C, 3]
        return t.reshape(c, 3, hw).permute(2, 0, 1).contiguous()

    def plane1(t):   # [C,H,W] -> [HW, C]
        return t.reshape(c, hw).permute(1, 0).contiguous()

    p = {
        "w1": w1.reshape(c, 3, 2).contiguous(),
        "b1": sd[f"{prefix}.first_conv.bias"].float().reshape(c, 3).contiguous(),
        "g1": plane3(g1).to(dtype), "be1": plane3(sd[f"{prefix}.first_normalization.bias"].float()).to(dtype),
        "w2": sd[f"{prefix}.second_conv.weight"].float().reshape(c, 3).contiguous(),
        "b2": sd[f"{prefix}.second_conv.bias"].float().contiguous(),
        "g2": plane1(sd[f"{prefix}.second_normalization.weight"].float()).to(dtype),
        "be2": plane1(sd[f"{prefix}.second_normalization.bias"].float()).to(dtype),
        "w3": sd[f"{prefix}.third_conv.weight"].float().reshape(c).contiguous(),
        "b3": sd[f"{prefix}.third_conv.bias"].float().contiguous(),
    }
    return {k: v.to(device) for k, v in p.items()}

"""Per-kernel parity: every C-ABI entry point vs a plain PyTorch fp32 restatement of the same op (the primitives the
oracle is made of), on seeded inputs.  Tolerances: fp16 storage + fp32 accumulate => rel err <= 2e-3 of the output
scale per op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-6))


def nhwc(x, dtype=torch.float16):
    return x.permute(0, 2, 3, 1).contiguous().to(device=DEV, dtype=dtype)


def q16(x, dtype=torch.float16):
    """round-trip through the storage dtype so the reference sees the same inputs"""
    return x.to(dtype).float()


CONV_CASES = [
    # N, Cin, Cout, H, k, stride, upsample
    (2, 64, 64, 16, 3, 1, False),
    (2, 320, 320, 16, 3, 1, False),       # bn=160 path
    (1, 128, 256, 8, 3, 2, False),        # stride 2
    (2, 64, 128, 8, 3, 1, True),          # fused nearest-2x upsample
    (2, 128, 64, 8, 1, 1, False),         # 1x1
    (1, 8, 64, 16, 3, 1, False),          # Cin=8 (padded 4->8 conv_in), non-aligned K
    (1, 16, 32, 32, 3, 2, False),         # cond-embedding shapes, non-aligned
    (1, 96, 256, 8, 3, 2, False),
    (2, 320, 4, 16, 3, 1, False),         # conv_out: Cout=4
    (1, 128, 3, 16, 3, 1, False),         # VAE conv_out: Cout=3 (scalar stores)
    (3, 64, 64, 5, 3, 1, False),          # ragged M (75 pixels)
    (1, 1280, 1280, 8, 3, 1, False),      # deep K, split-K path
]


@pytest.mark.parametrize("N,Cin,Cout,H,k,stride,up", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv_gemm(N, Cin, Cout, H, k, stride, up, dtype):
    from edgestyle_amd import ops
    if dtype == torch.bfloat16 and Cin > 320:
        pytest.skip("bf16 covered on the smaller cases")
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = q16(torch.randn(N, Cin, H, H, generator=g), dtype)
    w = q16(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    b = torch.randn(Cout, generator=g) * 0.1
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=k // 2)
    pw = ops.pack_weight(w, b, dtype, DEV)
    y = ops.conv_gemm(nhwc(x, dtype), pw, stride=stride, upsample=up)
    torch.cuda.synchronize()
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    assert y.shape == (N, ref.shape[2], ref.shape[3], Cout)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < tol


def test_conv_gemm_vae_asymmetric_pad():
    """VAE encoder downsample: F.pad(0,1,0,1) + conv stride 2 padding 0"""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(3)
    x = q16(torch.randn(1, 64, 16, 16, generator=g))
    w = q16(torch.randn(64, 64, 3, 3, generator=g) / 24)
    b = torch.randn(64, generator=g) * 0.1
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2, padding=0)
    y = ops.conv_gemm(nhwc(x), ops.pack_weight(w, b, torch.float16, DEV), stride=2, pad=0, out_hw=(8, 8))
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 3e-3


def test_conv_gemm_epilogue_concat_temb_residual_silu_scale():
    from edgestyle_amd import ops, lib
    g = torch.Generator().manual_seed(5)
    N, C1, C2, Cout, H = 2, 128, 64, 128, 8
    x1 = q16(torch.randn(N, C1, H, H, generator=g))
    x2 = q16(torch.randn(N, C2, H, H, generator=g))
    w = q16(torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt(9 * (C1 + C2)))
    b = torch.randn(Cout, generator=g) * 0.1
    temb = q16(torch.randn(N, 512, generator=g))
    res = q16(torch.randn(N, Cout, H, H, generator=g))
    off = 256
    ref = F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1) + temb[:, off:off + Cout, None, None]
    ref = F.silu(ref) * 0.7 + res
    pw = ops.pack_weight(w, b, torch.float16, DEV)
    tdev = temb.to(DEV, torch.float16)
    sdev = torch.tensor([0.7], device=DEV)
    for splitk in (1, 3):
        y = ops.conv_gemm(nhwc(x1), pw, x2=nhwc(x2), temb=tdev[:, off:], residual=nhwc(res), act=lib.ACT_SILU,
                          out_scale_dev=sdev, splitk=splitk)
        assert rel_err(y.permute(0, 3, 1, 2), ref) < 3e-3, splitk


@pytest.mark.parametrize("M,K,Nout", [(2, 320, 1280), (154, 768, 640), (512, 320, 960), (300, 1280, 320)])
def test_linear(M, K, Nout):
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(M)
    x = q16(torch.randn(M, K, generator=g))
    w = q16(torch.randn(Nout, K, generator=g) / math.sqrt(K))
    b = torch.randn(Nout, generator=g) * 0.1
    y = ops.linear(x.to(DEV, torch.float16), ops.pack_weight(w, b, torch.float16, DEV))
    assert rel_err(y, F.linear(x, w, b)) < 3e-3


def test_linear_geglu():
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(11)
    M, K, inner = 200, 320, 1280
    x = q16(torch.randn(M, K, generator=g))
    w = q16(torch.randn(2 * inner, K, generator=g) / math.sqrt(K))
    b = torch.randn(2 * inner, generator=g) * 0.1
    hidden, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ref = hidden * F.gelu(gate)
    y = ops.linear(x.to(DEV, torch.float16), ops.pack_weight(w, b, torch.float16, DEV, geglu=True))
    assert y.shape == (M, inner)
    assert rel_err(y, ref) < 3e-3


ATTN_CASES = [
    # N, heads, Sq, Skv, d
    (2, 8, 256, 256, 40),
    (2, 8, 200, 77, 40),       # cross-attention, ragged both sides
    (1, 8, 128, 128, 80),
    (1, 8, 64, 64, 160),
    (2, 4, 256, 77, 16),
    (2, 4, 64, 64, 32),
    (1, 4, 16, 16, 64),
    (1, 1, 256, 256, 512),     # VAE mid-block attention
    (1, 8, 1024, 1024, 40),
    # >= 512 blocks of 128 queries: the 32x32-score-tile kernel (head_dim 40 / 80)
    (8, 8, 1024, 1024, 40),
    (8, 8, 1000, 77, 40),      # ragged queries and keys
    (8, 8, 1024, 200, 80),
    (16, 8, 520, 136, 80),
    # >= 512 blocks of 256 queries at head_dim 40: two 32-query blocks per wave share every K / V fragment read
    (16, 8, 1024, 1024, 40),
    (16, 8, 1000, 77, 40),     # ragged queries (last wave: one block partly, one block entirely out of range) and keys
    (5, 32, 900, 333, 40),
]


@pytest.mark.parametrize("N,heads,Sq,Skv,d", ATTN_CASES)
def test_attention(N, heads, Sq, Skv, d):
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(Sq + d)
    C = heads * d
    q = q16(torch.randn(N, Sq, C, generator=g))
    k = q16(torch.randn(N, Skv, C, generator=g))
    v = q16(torch.randn(N, Skv, C, generator=g))
    qh, kh, vh = (t.view(N, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(N, Sq, C)
    y = ops.attention(q.to(DEV, torch.float16), k.to(DEV, torch.float16), v.to(DEV, torch.float16), heads)
    assert rel_err(y, ref) < 4e-3


@pytest.mark.parametrize("N,heads,Sq,Skv,d", [(2, 8, 256, 256, 40), (8, 8, 1024, 1024, 40), (16, 8, 1024, 1024, 40), (1, 8, 128, 128, 80)])
def test_attention_bf16(N, heads, Sq, Skv, d):
    """bf16 (BASELINE configs[4]); at head_dim 40 the running softmax reference rides in the pad element of the bf16 Q operand
    (8 mantissa bits): a spiking key forces it to move by large, inexactly representable steps."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(Sq + d + N)
    C = heads * d
    q = torch.randn(N, Sq, C, generator=g).bfloat16().float()
    k = torch.randn(N, Skv, C, generator=g)
    k[:, Skv // 2 + 3] *= 7.3
    k = k.bfloat16().float()
    v = torch.randn(N, Skv, C, generator=g).bfloat16().float()
    qh, kh, vh = (t.view(N, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(N, Sq, C)
    y = ops.attention(q.to(DEV, torch.bfloat16), k.to(DEV, torch.bfloat16), v.to(DEV, torch.bfloat16), heads)
    assert rel_err(y, ref) < 2e-2


@pytest.mark.parametrize("N,heads,Sq,Skv,d,dtype", [
    (14, 8, 4096, 77, 40, torch.float16),      # level-0 cross-attention of the lockstep encoder pass
    (2, 8, 1000, 77, 40, torch.float16),       # ragged query count, few samples (one block per strip)
    (3, 8, 1024, 77, 80, torch.float16),       # level 1
    (2, 8, 256, 96, 80, torch.bfloat16),       # all three key tiles full
    (2, 4, 200, 33, 40, torch.bfloat16),       # one key into the second tile, four heads
    (1, 6, 64, 5, 40, torch.float16),          # heads not a multiple of four, a handful of keys
])
def test_cross_attention_kv_resident_kernel(N, heads, Sq, Skv, d, dtype):
    """attention_kvres_kernel (round 5): launches with Skv <= 96 and head_dim 40 | 80 - the text-token cross-attention of the 64 x 64 and
    32 x 32 levels (diffusers Attention under CL:205-238 / PL:500-510, 77 keys) - keep K and V in registers and run a single-pass
    softmax.  Against the fp32 reference, against the tiled kernels it replaces (es_attention_set_kvres(0)), with a spiking key, with
    q / k / v that are column slices of wider buffers (the k|v projection of set_context is one [N, 77, 2C] tensor)."""
    from edgestyle_amd import ops, lib
    g = torch.Generator().manual_seed(Sq + d + Skv)
    C = heads * d
    q = torch.randn(N, Sq, C + 8, generator=g)
    kv = torch.randn(N, Skv, 2 * C, generator=g)
    kv[:, Skv // 2, :C] *= 6.0
    q, kv = q.to(dtype).float(), kv.to(dtype).float()
    qh, kh, vh = (t.reshape(N, -1, heads, d).transpose(1, 2) for t in (q[:, :, :C], kv[:, :, :C], kv[:, :, C:]))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(N, Sq, C)
    dq, dkv = q.to(DEV, dtype), kv.to(DEV, dtype)
    L = lib.load()
    prev = L.es_attention_set_kvres(2)
    try:
        y1 = ops.attention(dq[:, :, :C], dkv[:, :, :C], dkv[:, :, C:], heads)
        L.es_attention_set_kvres(0)
        y0 = ops.attention(dq[:, :, :C], dkv[:, :, :C], dkv[:, :, C:], heads)
    finally:
        L.es_attention_set_kvres(prev)
    tol = 4e-3 if dtype == torch.float16 else 2e-2
    assert rel_err(y1, ref) < tol and rel_err(y0, ref) < tol
    assert rel_err(y1, y0.float().cpu()) < tol
    assert bool(torch.isfinite(y1).all())


def test_attention_forced_rescale_and_strided_qkv():
    """online-softmax rescale branch: a late key dominates; q/k/v are column slices of one fused [N,S,3C] buffer"""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(0)
    N, heads, S, d = 1, 2, 192, 40
    C = heads * d
    qkv = torch.randn(N, S, 3 * C, generator=g)
    qkv[:, 150, C:2 * C] *= 8.0            # key 150 (third tile) spikes
    qkv = q16(qkv)
    q, k, v = qkv.split(C, dim=-1)
    qh, kh, vh = (t.reshape(N, S, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(N, S, C)
    dq = qkv.to(DEV, torch.float16)
    y = ops.attention(dq[:, :, :C], dq[:, :, C:2 * C], dq[:, :, 2 * C:], heads)
    assert rel_err(y, ref) < 4e-3


def test_attention32_head_dim_80_variant():
    """The head_dim-80 instantiation of the 32x32-tile kernel is opt-in (ES_ATTN32=2, read once per process): run the
    comparison in a child process with the switch set."""
    import os
    import subprocess
    import sys
    code = (
        "import torch, torch.nn.functional as F\n"
        "from edgestyle_amd import ops\n"
        "g = torch.Generator().manual_seed(3)\n"
        "N, heads, Sq, Skv, d = 8, 8, 1000, 200, 80\n"
        "C = heads * d\n"
        "q = torch.randn(N, Sq, C, generator=g).half().float(); k = torch.randn(N, Skv, C, generator=g).half().float()\n"
        "v = torch.randn(N, Skv, C, generator=g).half().float(); k[:, 150] *= 5.0\n"
        "qh, kh, vh = (t.view(N, -1, heads, d).transpose(1, 2) for t in (q, k, v))\n"
        "ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(N, Sq, C)\n"
        "y = ops.attention(q.cuda().half(), k.cuda().half(), v.cuda().half(), heads).float().cpu()\n"
        "err = float((y - ref).abs().max() / ref.abs().max())\n"
        "assert err < 4e-3, err\n"
        "print('ok', err)\n")
    env = dict(os.environ, ES_ATTN32="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_attention40_ping_pong_kernel_matches_previous_kernel_and_fp32(tmp_path):
    """The head_dim-40 ping-pong kernel (8 waves, two groups one barrier out of phase: attention40pp_kernel, the default for
    self-attention launches of >= 256 workgroups) in both forms (ES_ATTN_PP = 1: 32 queries per wave, 2: 64) against the
    32x32-tile kernel it replaces (ES_ATTN_PP = 0) and the fp32 reference - including the lazy-rescale path (spiking keys
    in a late tile and in the second tile), ragged query counts, and a PEAKY softmax: scores spanning +-30 (one key per
    query dominates by e^30), where the scale*log2(e) fold into the fp16 Q operand costs the most.  The env switch is read
    once per process, hence child processes."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "from edgestyle_amd import ops\n"
        "g = torch.Generator().manual_seed(11)\n"
        "outs = {}\n"
        "for name, (N, S) in dict(a=(16, 1024), b=(3, 4096), c=(2, 1000)).items():\n"
        "    heads, d = 8, 40\n"
        "    C = heads * d\n"
        "    qkv = torch.randn(N, S, 3 * C, generator=g); qkv[:, 700, C:2 * C] *= 6.0; qkv[:, 70, C:C + d] *= 9.0\n"
        "    if name == 'b':\n"
        "        qkv[:, :, :C] *= 3.0; qkv[:, :, C:2 * C] *= 3.0      # |score| up to ~30 after the 1/sqrt(40) scale\n"
        "    dq = qkv.half().cuda()\n"
        "    Sk = S if name != 'c' else 960\n"
        "    outs[name] = ops.attention(dq[:, :, :C], dq[:, :Sk, C:2 * C], dq[:, :Sk, 2 * C:], heads).cpu()\n"
        "torch.save(outs, sys.argv[1])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for pp in ("0", "1", "2"):
        f = str(tmp_path / f"y{pp}.pt")
        r = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=dict(os.environ, ES_ATTN_PP=pp),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        res[pp] = torch.load(f, weights_only=True)
    g = torch.Generator().manual_seed(11)
    for name, (N, S) in dict(a=(16, 1024), b=(3, 4096), c=(2, 1000)).items():
        heads, d = 8, 40
        C = heads * d
        qkv = torch.randn(N, S, 3 * C, generator=g)
        qkv[:, 700, C:2 * C] *= 6.0
        qkv[:, 70, C:C + d] *= 9.0
        if name == "b":
            qkv[:, :, :C] *= 3.0
            qkv[:, :, C:2 * C] *= 3.0
        qkv = q16(qkv)
        Sk = S if name != "c" else 960
        qh, kh, vh = (t.reshape(N, -1, heads, d).transpose(1, 2) for t in (qkv[:, :, :C], qkv[:, :Sk, C:2 * C], qkv[:, :Sk, 2 * C:]))
        if name == "b":
            sc = (qh[0, 0, :256] @ kh[0, 0].T) / d ** 0.5
            assert float(sc.max()) > 25.0 and float(sc.min()) < -25.0          # the scores do span +-30
        ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(N, S, C)
        # peaky: Q is pre-multiplied by scale * log2(e) and re-rounded to fp16, an error of 2^-11 |score| in the exponent
        # (measured 9.7e-3 of the output's max with scores up to +-30; PyTorch's own fp16 SDPA rounds the raw scores to
        # fp16, which is coarser).  Bar 1.5e-2 there, 4e-3 on ordinary data.
        bar = 1.5e-2 if name == "b" else 4e-3
        for pp in ("0", "1", "2"):
            assert rel_err(res[pp][name], ref) < bar, (name, pp, rel_err(res[pp][name], ref))
        assert rel_err(res["1"][name], res["0"][name]) < 4e-3 and rel_err(res["2"][name], res["0"][name]) < 4e-3


@pytest.mark.parametrize("d,N", [(40, 8), (80, 8), (40, 16)])
def test_attention32_forced_rescale(d, N):
    """32x32-tile kernel: a late key dominates some queries only (lazy rescale on the mixed 32-query / 16-query layouts);
    N = 16 at head_dim 40 runs the two-blocks-per-wave form, where one block of a wave rescales and the other may not"""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(d)
    heads, S = 8, 1024
    C = heads * d
    qkv = torch.randn(N, S, 3 * C, generator=g)
    qkv[:, 700, C:2 * C] *= 6.0
    qkv[:, 901, C:C + d] *= 10.0            # head 0 only, later tile
    qkv = q16(qkv)
    q, k, v = qkv.split(C, dim=-1)
    qh, kh, vh = (t.reshape(N, S, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(N, S, C)
    dq = qkv.to(DEV, torch.float16)
    y = ops.attention(dq[:, :, :C], dq[:, :, C:2 * C], dq[:, :, 2 * C:], heads)
    assert rel_err(y, ref) < 4e-3


@pytest.mark.parametrize("N,C1,C2,H,silu,eps", [(2, 320, 0, 16, True, 1e-5), (2, 1280, 640, 8, True, 1e-5),
                                                (1, 64, 0, 32, False, 1e-6), (2, 1280, 1280, 8, True, 1e-5),
                                                (1, 128, 0, 64, True, 1e-6),
                                                # one-launch slab path: 4 / 2 / 1 groups per block, concat inside a slab
                                                (2, 320, 0, 32, True, 1e-5), (2, 640, 0, 16, True, 1e-5),
                                                (3, 960, 0, 16, True, 1e-5), (2, 1280, 1280, 16, True, 1e-5),
                                                (2, 320, 320, 12, False, 1e-5),
                                                # 1024-thread slab blocks (level 0), and the two-launch path
                                                (2, 320, 0, 64, True, 1e-5), (2, 320, 320, 64, True, 1e-5),
                                                (2, 640, 320, 32, True, 1e-5), (1, 640, 320, 64, True, 1e-5)])
def test_group_norm(N, C1, C2, H, silu, eps):
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(C1 + C2)
    C = C1 + C2
    x = q16(torch.randn(N, C, H, H, generator=g) * 2 + 0.5)
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    x1 = nhwc(x[:, :C1])
    x2 = nhwc(x[:, C1:]) if C2 else None
    y = ops.group_norm(x1, gamma.to(DEV), beta.to(DEV), 32, eps, silu, x2=x2)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 3e-3


@pytest.mark.parametrize("M,C", [(300, 320), (77, 640), (64, 1280), (5, 64)])
def test_layer_norm(M, C):
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(C)
    x = q16(torch.randn(M, C, generator=g) * 3 + 1)
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    y = ops.layer_norm(x.to(DEV, torch.float16), gamma.to(DEV), beta.to(DEV))
    assert rel_err(y, F.layer_norm(x, (C,), gamma, beta)) < 3e-3


@pytest.mark.parametrize("C,S,N", [(64, 16, 2), (320, 8, 2), (1280, 8, 1), (320, 64, 2)])
def test_fusion_block_vs_reference_restatement(C, S, N):
    """es_fusion_block == interleave_tensors + ControlNetBlock (MC:23-63, 479-501) from the oracle"""
    from edgestyle_amd import ops
    from oracle import sd15_oracle as O
    g = torch.Generator().manual_seed(C + S)
    p = "blk"
    sd = {
        f"{p}.first_conv.weight": torch.randn(3 * C, 2, 1, 1, generator=g) * 0.7,
        f"{p}.first_conv.bias": torch.randn(3 * C, generator=g) * 0.1,
        f"{p}.first_normalization.weight": q16(1 + 0.1 * torch.randn(3 * C, S, S, generator=g)),
        f"{p}.first_normalization.bias": q16(0.1 * torch.randn(3 * C, S, S, generator=g)),
        f"{p}.second_conv.weight": torch.randn(C, 3, 1, 1, generator=g) * 0.6,
        f"{p}.second_conv.bias": torch.randn(C, generator=g) * 0.1,
        f"{p}.second_normalization.weight": q16(1 + 0.1 * torch.randn(C, S, S, generator=g)),
        f"{p}.second_normalization.bias": q16(0.1 * torch.randn(C, S, S, generator=g)),
        f"{p}.third_conv.weight": torch.randn(C, 1, 1, 1, generator=g),
        f"{p}.third_conv.bias": torch.randn(C, generator=g) * 0.1,
    }
    res = [q16(torch.randn(N, C, S, S, generator=g)) for _ in range(6)]
    scales = [1.0, 0.5, 1.0, 2.0, 1.0, 0.0]
    ref = O.controlnet_block(sd, p, O.interleave_tensors([r * s for r, s in zip(res, scales)]))
    params = ops.pack_fusion_params(sd, p, torch.float16, DEV)
    # nets 1,3,5 live in one batched buffer [3N, HW, C] like the batched openpose pass
    pose = torch.cat([nhwc(res[1]), nhwc(res[3]), nhwc(res[5])]).reshape(3 * N, S * S, C)
    r = [nhwc(res[0]), pose[0:], nhwc(res[2]), pose[N:], nhwc(res[4]), pose[2 * N:]]
    y = ops.fusion_block(r, [S * S * C] * 6, params, N, S * S, C, scales)
    y = y.reshape(N, S, S, C).permute(0, 3, 1, 2)
    assert rel_err(y, ref) < 4e-3


@pytest.mark.parametrize("level", list(range(13)))
def test_fusion_block_vs_the_references_own_outputs(level):
    """es_fusion_block against outputs of the REFERENCE's own ControlNetBlock(interleave_tensors(...)) (MC:23-63, 479-501):
    tests/golden/ref_fusion.safetensors, written by tests/golden/make_golden_ref_fusion.py from the reference's source text -
    all 13 (channels, size) pairs of MC:73-102 at batch 2.  PINNED parity for the kernel the reference owns: sampled elements
    within 4e-3 of the tensor's scale (fp16 storage), full-tensor abs-sum within 1e-3."""
    import os
    from safetensors.torch import load_file
    from edgestyle_amd import ops
    from tests import helpers as H
    gold = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fusion.safetensors"))
    C, S = H.REF_FUSION_LEVELS[level]
    N = 2
    sd, res = H.ref_fusion_case(level)
    params = ops.pack_fusion_params({"b." + k: v for k, v in sd.items()}, "b", torch.float16, DEV)
    r = [nhwc(x).reshape(N, S * S, C) for x in res]
    y = ops.fusion_block(r, [S * S * C] * 6, params, N, S * S, C, [1.0] * 6)
    y = y.reshape(N, S, S, C).permute(0, 3, 1, 2).float().cpu()
    want = gold[f"level{level}_sample"]
    assert float((H.ref_fusion_sample(y) - want).abs().max()) <= 4e-3 * float(want.abs().max())
    sums = gold[f"level{level}_sums"]
    assert abs(float(y.double().abs().sum()) - float(sums[1])) <= 1e-3 * float(sums[1])


@pytest.mark.parametrize("C,S,N", [(320, 64, 2), (1280, 8, 2)])
def test_fusion_block_residuals_with_a_large_common_mean(C, S, N):
    """Real zero-conv outputs carry offsets: residuals = randn * 0.3 + 5 and same-sign first_conv weights make the
    LayerNorm input z ~ 10 +- 0.4 over up to 3.9 M elements (variance 600x below mean^2) - the fp32 E[x^2] - E[x]^2
    statistics of csrc/fusion.hip must survive that."""
    from edgestyle_amd import ops
    from oracle import sd15_oracle as O
    g = torch.Generator().manual_seed(C * 7 + S)
    p = "blk"
    sd = {
        f"{p}.first_conv.weight": 1.0 + 0.05 * torch.randn(3 * C, 2, 1, 1, generator=g),
        f"{p}.first_conv.bias": torch.randn(3 * C, generator=g) * 0.05,
        f"{p}.first_normalization.weight": q16(1 + 0.1 * torch.randn(3 * C, S, S, generator=g)),
        f"{p}.first_normalization.bias": q16(0.1 * torch.randn(3 * C, S, S, generator=g)),
        f"{p}.second_conv.weight": 0.5 + 0.05 * torch.randn(C, 3, 1, 1, generator=g),
        f"{p}.second_conv.bias": 4.0 + torch.randn(C, generator=g) * 0.05,
        f"{p}.second_normalization.weight": q16(1 + 0.1 * torch.randn(C, S, S, generator=g)),
        f"{p}.second_normalization.bias": q16(0.1 * torch.randn(C, S, S, generator=g)),
        f"{p}.third_conv.weight": torch.randn(C, 1, 1, 1, generator=g),
        f"{p}.third_conv.bias": torch.randn(C, generator=g) * 0.1,
    }
    res = [q16(torch.randn(N, C, S, S, generator=g) * 0.3 + 5.0) for _ in range(6)]
    scales = [1.0] * 6
    ref = O.controlnet_block(sd, p, O.interleave_tensors(res))
    params = ops.pack_fusion_params(sd, p, torch.float16, DEV)
    r = [nhwc(t).reshape(N, S * S, C) for t in res]
    y = ops.fusion_block(r, [S * S * C] * 6, params, N, S * S, C, scales)
    y = y.reshape(N, S, S, C).permute(0, 3, 1, 2)
    assert rel_err(y, ref) < 4e-3, rel_err(y, ref)


def test_fusion_blocks_batched_equals_per_block():
    """es_fusion_blocks (all blocks of a step in three launches) == es_fusion_block per block, bit for bit."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(77)
    N, blocks = 2, []
    for C, S in [(64, 16), (64, 16), (128, 8), (320, 8), (64, 32)]:
        params = {"w1": torch.randn(C, 3, 2, generator=g), "b1": torch.randn(C, 3, generator=g) * 0.1,
                  "g1": 1 + 0.1 * torch.randn(S * S, C, 3, generator=g), "be1": 0.1 * torch.randn(S * S, C, 3, generator=g),
                  "w2": torch.randn(C, 3, generator=g), "b2": torch.randn(C, generator=g) * 0.1,
                  "g2": 1 + 0.1 * torch.randn(S * S, C, generator=g), "be2": 0.1 * torch.randn(S * S, C, generator=g),
                  "w3": torch.randn(C, generator=g), "b3": torch.randn(C, generator=g) * 0.1}
        params = {k: v.to(DEV, torch.float16 if k in ("g1", "be1", "g2", "be2") else torch.float32).contiguous()
                  for k, v in params.items()}
        res = [torch.randn(N, S * S, C, generator=g).to(DEV, torch.float16) for _ in range(6)]
        blocks.append((res, [S * S * C] * 6, params, S * S, C))
    scales = [1.0, 0.5, 1.0, 2.0, 1.0, 0.25]
    single = [ops.fusion_block(r, bs, p, N, hw, c, scales) for r, bs, p, hw, c in blocks]
    batched = ops.fusion_blocks(blocks, N, scales)
    for a, b in zip(single, batched):
        assert torch.equal(a, b)
    # addend = the UNet skip tensor: one fused pass == block output followed by es_add
    adds = [torch.randn(N, hw, c, generator=g).to(DEV, torch.float16) for _, _, _, hw, c in blocks]
    summed = ops.fusion_blocks(blocks, N, scales, addends=adds)
    for a, b, d in zip(single, summed, adds):
        assert torch.equal(ops.add(a, d), b)


def test_timestep_embedding():
    from edgestyle_amd import ops
    from oracle import sd15_oracle as O
    t = torch.tensor([981.0, 1.0, 500.0])
    y = ops.timestep_embedding(t.to(DEV), 320, torch.float16)
    ref = O.timestep_sinusoid(t, 320)
    assert float((y.float().cpu() - ref).abs().max()) < 2e-3


def test_cfg_ddim_step_matches_oracle_scheduler():
    from edgestyle_amd import ops
    from edgestyle_amd.schedulers import DDIMScheduler
    from oracle import sd15_oracle as O
    g = torch.Generator().manual_seed(9)
    B, H = 2, 8
    sched = O.DDIM()
    ts = sched.set_timesteps(10)
    lat = torch.randn(B, 4, H, H, generator=g)
    eps = q16(torch.randn(2 * B, 4, H, H, generator=g))
    gs = 7.5
    e = eps[:B] + gs * (eps[B:] - eps[:B])
    step = 3
    ref = sched.step(e, int(ts[step]), lat)
    mine = DDIMScheduler()
    mine.set_timesteps(10)
    assert mine.timesteps.tolist() == ts.tolist()
    coef = mine.coef_table().to(DEV)
    latd = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
    model_in = torch.zeros(2 * B, H, H, 8, dtype=torch.float16, device=DEV)
    idx = torch.tensor([step], dtype=torch.int32, device=DEV)
    ops.cfg_ddim_step(nhwc(eps), latd, model_in, coef, idx, gs, True)
    assert rel_err(latd.permute(0, 3, 1, 2), ref) < 1e-5
    assert rel_err(model_in[:B, :, :, :4].permute(0, 3, 1, 2), ref) < 2e-3
    assert torch.equal(model_in[:B], model_in[B:]) and float(model_in[..., 4:].abs().max()) == 0.0
    ops.incr(idx)
    assert int(idx.item()) == step + 1


def test_layout_conversions_and_add_and_vae_sample():
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 8, 8, generator=g)
    y = ops.nchw_to_nhwc(x.to(DEV), torch.float16, cpad=8)
    assert y.shape == (2, 8, 8, 8) and float(y[..., 4:].abs().max()) == 0
    assert rel_err(y[..., :4].permute(0, 3, 1, 2), x) < 1e-3
    z = ops.nhwc_to_nchw(y, channels=4, scale=0.5, shift=0.5, clamp01=True)
    assert rel_err(z, (q16(x) / 2 + 0.5).clamp(0, 1)) < 1e-3
    a = torch.randn(64, 40, generator=g)
    b = torch.randn(64, 40, generator=g)
    s = ops.add(a.to(DEV, torch.float16), b.to(DEV, torch.float16))
    assert rel_err(s, q16(a) + q16(b)) < 1e-3
    mom = q16(torch.randn(2, 8, 8, 8, generator=g))            # NCHW moments
    noise = torch.randn(2, 4, 8, 8, generator=g)
    mean, logvar = mom.chunk(2, dim=1)
    ref = (mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * noise) * 0.18215
    zz = ops.vae_sample(nhwc(mom), noise.to(DEV), 4, 8, 0.18215)
    assert rel_err(zz[..., :4].permute(0, 3, 1, 2), ref) < 2e-3 and float(zz[..., 4:].abs().max()) == 0


def test_errors_are_loud():
    from edgestyle_amd import ops, lib
    x = torch.zeros(1, 4, 4, 12, dtype=torch.float16, device=DEV)    # 12 channels: not a multiple of 8
    pw = ops.pack_weight(torch.zeros(8, 16, 1, 1), None, torch.float16, DEV)
    with pytest.raises(lib.EdgeStyleHipError):
        ops.conv_gemm(x, pw)
    with pytest.raises(lib.EdgeStyleHipError):
        ops.attention(torch.zeros(1, 8, 36, dtype=torch.float16, device=DEV),
                      torch.zeros(1, 8, 36, dtype=torch.float16, device=DEV),
                      torch.zeros(1, 8, 36, dtype=torch.float16, device=DEV), heads=1)   # d=36 unsupported


def test_conv_gemm_small_tile_equals_default_tile():
    """bn=64 (64x64 tile for tiny launches) reproduces the 128-pixel tile bit for bit: concat + temb + residual + SiLU,
    ragged M, split-K, 2/4-stage rings, grouped weights, 1x1."""
    from edgestyle_amd import ops, lib
    g = torch.Generator().manual_seed(44)

    def both(fn):
        ops.FORCE_BN = 64
        try:
            a = fn()
        finally:
            ops.FORCE_BN = 0
        return a, fn()

    N, C1, C2, Cout, H = 3, 128, 64, 320, 10             # M = 300
    x1 = torch.randn(N, H, H, C1, generator=g).to(DEV, torch.float16)
    x2 = torch.randn(N, H, H, C2, generator=g).to(DEV, torch.float16)
    pw = ops.pack_weight(torch.randn(Cout, C1 + C2, 3, 3, generator=g) / 40, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
    temb = torch.randn(N, Cout, generator=g).to(DEV, torch.float16)
    res = torch.randn(N, H, H, Cout, generator=g).to(DEV, torch.float16)
    for splitk, stages in ((1, 2), (3, 2), (1, 4), (2, 4)):
        a, b = both(lambda: ops.conv_gemm(x1, pw, x2=x2, temb=temb, residual=res, act=lib.ACT_SILU, splitk=splitk, stages=stages))
        assert torch.equal(a, b), (splitk, stages)
    xl = torch.randn(200, 1280, generator=g).to(DEV, torch.bfloat16)
    pl = ops.pack_weight(torch.randn(1280, 1280, generator=g) / 36, torch.randn(1280, generator=g) * 0.1, torch.bfloat16, DEV)
    a, b = both(lambda: ops.linear(xl, pl))
    assert torch.equal(a, b)
    counts, Hg = [2, 4, 2], 8                             # 64-pixel samples: groups of 128 / 256 / 128 pixels
    xg = torch.randn(sum(counts), Hg, Hg, C1, generator=g).to(DEV, torch.float16)
    pws = [ops.pack_weight(torch.randn(320, C1, 3, 3, generator=g) / 34, torch.randn(320, generator=g) * 0.1,
                           torch.float16, DEV) for _ in counts]
    a, b = both(lambda: ops.conv_gemm(xg, pws, group_n=counts))
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_linear_with_folded_layer_norm(dtype):
    """Linear(LayerNorm(x)) as one launch on the raw x (es_gemm_desc.ln_colsum) vs F.layer_norm + F.linear, on every
    tile variant that can run it (128x128|160, 8 waves, 64x64), plain / GEGLU, grouped, ragged M, x with a large mean."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(66)
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    ops.XS_ENABLED = False                                # this test is about the tiled kernel's fold; linear_xs has its own
    for M, C, Cout, geglu in [(300, 320, 960, False), (200, 640, 640, False), (257, 320, 2560, True), (130, 1280, 1280, False)]:
        x = q16(torch.randn(M, C, generator=g) * 1.5 + 0.7 + torch.randn(M, 1, generator=g), dtype)
        gamma = 1 + 0.2 * torch.randn(C, generator=g)
        beta = 0.1 * torch.randn(C, generator=g)
        w = torch.randn(Cout, C, generator=g) / math.sqrt(C)
        b = torch.randn(Cout, generator=g) * 0.1
        y = F.linear(F.layer_norm(x, (C,), gamma, beta, 1e-5), w, b)
        if geglu:
            h, gate = y.chunk(2, dim=-1)
            y = h * F.gelu(gate)
        pw = ops.pack_weight_ln(w, b, gamma, beta, 1e-5, dtype, DEV, geglu=geglu)
        xd = x.to(DEV, dtype)
        outs = [ops.linear(xd, pw)]
        for knob, val in (("FORCE_WAVES", 8), ("FORCE_BN", 64)):
            if knob == "FORCE_BN" and geglu:
                continue
            setattr(ops, knob, val)
            try:
                outs.append(ops.linear(xd, pw))
            finally:
                setattr(ops, knob, 0)
        for o in outs:
            assert rel_err(o, y) < tol, (M, C, Cout, geglu)
    # grouped: three weight sets with their own LayerNorm parameters over 2 + 4 + 2 samples of 64 tokens... 128-row groups
    C, Cout, counts = 320, 960, [256, 512, 256]
    xg = q16(torch.randn(sum(counts), C, generator=g) * 2 + 0.3, dtype)
    pws, refs, a = [], [], 0
    for n in counts:
        gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        w, b = torch.randn(Cout, C, generator=g) / math.sqrt(C), torch.randn(Cout, generator=g) * 0.1
        pws.append(ops.pack_weight_ln(w, b, gamma, beta, 1e-5, dtype, DEV))
        refs.append(F.linear(F.layer_norm(xg[a:a + n], (C,), gamma, beta, 1e-5), w, b))
        a += n
    yg = ops.linear(xg.to(DEV, dtype), pws, group_n=counts)
    ops.XS_ENABLED = True
    assert rel_err(yg, torch.cat(refs)) < tol


def test_conv_gemm_eight_wave_tile_equals_four_wave_tile():
    """waves=8 (the 128-pixel tile on two waves per SIMD) reproduces the 4-wave kernel bit for bit: 3x3 with concat +
    temb + residual + SiLU and split-K, 1x1 linear, GEGLU, 2- and 4-stage rings, both N tiles, bf16."""
    from edgestyle_amd import ops, lib
    g = torch.Generator().manual_seed(33)

    def both(fn):
        ops.FORCE_WAVES = 8
        try:
            a = fn()
        finally:
            ops.FORCE_WAVES = 0
        return a, fn()

    for dtype in (torch.float16, torch.bfloat16):
        N, C1, C2, H = 3, 128, 64, 12                    # M = 432 (ragged)
        for Cout in (256, 320):                          # bn = 128 / 160
            x1 = torch.randn(N, H, H, C1, generator=g).to(DEV, dtype)
            x2 = torch.randn(N, H, H, C2, generator=g).to(DEV, dtype)
            pw = ops.pack_weight(torch.randn(Cout, C1 + C2, 3, 3, generator=g) / 40, torch.randn(Cout, generator=g) * 0.1, dtype, DEV)
            temb = torch.randn(N, Cout, generator=g).to(DEV, dtype)
            res = torch.randn(N, H, H, Cout, generator=g).to(DEV, dtype)
            for splitk, stages in ((1, 2), (3, 2), (1, 4)):
                if stages == 4 and Cout == 320:
                    continue
                a, b = both(lambda: ops.conv_gemm(x1, pw, x2=x2, temb=temb, residual=res, act=lib.ACT_SILU, splitk=splitk, stages=stages))
                assert torch.equal(a, b), (dtype, Cout, splitk, stages)
    xl = torch.randn(300, 320, generator=g).to(DEV, torch.float16)
    pl = ops.pack_weight(torch.randn(1280, 320, generator=g) / 18, torch.randn(1280, generator=g) * 0.1, torch.float16, DEV)
    a, b = both(lambda: ops.linear(xl, pl))
    assert torch.equal(a, b)
    pg = ops.pack_weight(torch.randn(2560, 320, generator=g) / 18, torch.randn(2560, generator=g) * 0.1, torch.float16, DEV, geglu=True)
    a, b = both(lambda: ops.linear(xl, pg))
    assert torch.equal(a, b)


def test_grouped_launches_equal_separate_launches():
    """One grouped launch over the batch-concatenated activations == per-net launches (bit for bit): conv/linear with
    temb + residual, GroupNorm, LayerNorm."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(12)
    counts, H, C, Cout = [2, 6, 4, 2], 8, 64, 128         # 128 pixels per 2 samples: groups tile in 128-pixel units
    N = sum(counts)
    x = torch.randn(N, H, H, C, generator=g).to(DEV, torch.float16)
    res = torch.randn(N, H, H, Cout, generator=g).to(DEV, torch.float16)
    temb = torch.randn(N, 256, generator=g).to(DEV, torch.float16)
    pws = [ops.pack_weight(torch.randn(Cout, C, 3, 3, generator=g) / 24, torch.randn(Cout, generator=g) * 0.1,
                           torch.float16, DEV) for _ in counts]
    yg = ops.conv_gemm(x, pws, temb=temb[:, 64:], residual=res, group_n=counts)
    a = 0
    for pw, n in zip(pws, counts):
        ys = ops.conv_gemm(x[a:a + n], pw, temb=temb[a:a + n, 64:], residual=res[a:a + n])
        assert torch.equal(yg[a:a + n], ys)
        a += n
    gam = [(1 + 0.1 * torch.randn(C, generator=g)).to(DEV) for _ in counts]
    bet = [(0.1 * torch.randn(C, generator=g)).to(DEV) for _ in counts]
    ng = ops.group_norm(x, gam, bet, 32, 1e-5, True, group_n=counts)
    tok = x.reshape(N, H * H, C)
    lg = ops.layer_norm(tok, gam, bet, group_rows=[n * H * H for n in counts])
    a = 0
    for i, n in enumerate(counts):
        assert torch.equal(ng[a:a + n], ops.group_norm(x[a:a + n], gam[i], bet[i], 32, 1e-5, True))
        assert torch.equal(lg[a:a + n], ops.layer_norm(tok[a:a + n].contiguous(), gam[i], bet[i]))
        a += n
    with pytest.raises(Exception):
        ops.conv_gemm(x, pws, group_n=[3, 5, 4, 2])      # 3 samples x 64 px is not a whole number of 128-px tiles


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_ln_geglu_big_tile_equals_small_tile(dtype):
    """bn = 256 (round 5): the 256 x 256 form of the phase-interleaved tile that the LayerNorm-folded and GEGLU linear layers of the
    C = 1280 level run on (diffusers BasicTransformerBlock norm1 -> attn1.to_q|k|v, norm2 -> attn2.to_q, norm3 -> ff.net.0 under
    CL:205-238) must reproduce the 128-wide tile bit for bit - same MFMA order per accumulator, the row statistics of the fold summed
    on the matrix core in the same order, the GEGLU product in the same fp32 expression - and match the fp32 reference: LayerNorm +
    GEGLU, LayerNorm alone, GEGLU alone, a plain layer with a residual, a grouped launch, ragged M."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(77)
    K = 1280

    def both(fn):
        ops.FORCE_BN = 256
        try:
            big = fn()
        finally:
            ops.FORCE_BN = 0
        prev = ops.BIG_TILE_256
        ops.BIG_TILE_256 = False
        try:
            small = fn()
        finally:
            ops.BIG_TILE_256 = prev
        return big, small
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    for M, N, geglu, ln in [(2304, 2560, True, True), (2304, 768, False, True), (1000, 512, True, False), (4096, 1280, False, True)]:
        x = torch.randn(M, K, generator=g).to(dtype).float()
        w = (torch.randn(N, K, generator=g) / math.sqrt(K))
        b = torch.randn(N, generator=g) * 0.1
        gam, bet = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
        pw = ops.pack_weight_ln(w, b, gam, bet, 1e-5, dtype, DEV, geglu=geglu) if ln else ops.pack_weight(w, b, dtype, DEV, geglu=geglu)
        xd = x.to(DEV, dtype)
        big, small = both(lambda: ops.linear(xd, pw))
        assert torch.equal(big, small), (M, N, geglu, ln)
        h = F.layer_norm(x, (K,), gam, bet, 1e-5) if ln else x
        y = h @ w.t() + b
        if geglu:
            y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
        assert rel_err(big, y) < tol, (M, N, geglu, ln, rel_err(big, y))
    # LayerNorm-folded layer with a residual, and a grouped LayerNorm + GEGLU launch ([2, 6, 4, 2] x 256 rows)
    M, N = 3584, 1280
    x = torch.randn(M, K, generator=g).to(DEV, dtype)
    res = torch.randn(M, N, generator=g).to(DEV, dtype)
    pw = ops.pack_weight_ln(torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g) * 0.1, 1 + 0.1 * torch.randn(K, generator=g),
                            0.1 * torch.randn(K, generator=g), 1e-5, dtype, DEV)
    big, small = both(lambda: ops.linear(x, pw, residual=res))
    assert torch.equal(big, small)
    counts = [2, 6, 4, 2]
    pws = [ops.pack_weight_ln(torch.randn(2560, K, generator=g) / math.sqrt(K), torch.randn(2560, generator=g) * 0.1, 1 + 0.1 * torch.randn(K, generator=g),
                              0.1 * torch.randn(K, generator=g), 1e-5, dtype, DEV, geglu=True) for _ in counts]
    xg = torch.randn(sum(counts) * 256, K, generator=g).to(DEV, dtype)
    big, small = both(lambda: ops.linear(xg, pws, group_n=[n * 256 for n in counts]))
    assert torch.equal(big, small)
    a = 0
    for pw_, n in zip(pws, counts):
        ops.FORCE_BN = 256
        try:
            sep = ops.linear(xg[a * 256:(a + n) * 256], pw_)
        finally:
            ops.FORCE_BN = 0
        assert torch.equal(big.reshape(-1, 1280)[a * 256:(a + n) * 256], sep)
        a += n


def test_conv_gemm_big_tile_equals_small_tile():
    """bn=320 (256-pixel x 320-cout tile, 8 waves, streamed W fragments, two-pass epilogue) must reproduce the
    128-pixel tile bit for bit (same K order per accumulator) and match F.conv2d: concat + temb + residual + SiLU,
    ragged M, split-K, grouped weights, 1x1, stride 2."""
    from edgestyle_amd import ops, lib
    g = torch.Generator().manual_seed(21)

    def both(fn):
        ops.FORCE_BN = 320
        try:
            big = fn()
        finally:
            ops.FORCE_BN = 0
        return big, fn()

    N, C1, C2, Cout, H = 3, 128, 64, 640, 12            # M = 432: one full + one ragged 256-pixel tile, 2 N tiles
    x1 = q16(torch.randn(N, C1, H, H, generator=g))
    x2 = q16(torch.randn(N, C2, H, H, generator=g))
    w = q16(torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt(9 * (C1 + C2)))
    b = torch.randn(Cout, generator=g) * 0.1
    temb = q16(torch.randn(N, Cout, generator=g))
    res = q16(torch.randn(N, Cout, H, H, generator=g))
    ref = F.silu(F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1) + temb[:, :, None, None]) + res
    pw = ops.pack_weight(w, b, torch.float16, DEV)
    tdev = temb.to(DEV, torch.float16)
    for splitk in (1, 2):
        big, small = both(lambda: ops.conv_gemm(nhwc(x1), pw, x2=nhwc(x2), temb=tdev, residual=nhwc(res),
                                                act=lib.ACT_SILU, splitk=splitk))
        assert torch.equal(big, small), splitk
        assert rel_err(big.permute(0, 3, 1, 2), ref) < 3e-3
    # The epilogue forms the 256 x 320 tile implements itself (round 4; the combination above - an activation, or a time embedding
    # beside a residual - is handed to the 128 x 160 tile by es_conv_gemm): bias only / + residual on a ragged M, and the time
    # embedding as one row per 128-pixel half (H*W % 128 == 0: ragged last tile of half a tile), out_scale on top
    for kw, want in ((dict(), F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1)),
                     (dict(residual=nhwc(res)), F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1) + res),
                     (dict(residual=nhwc(res), out_scale=0.37), F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1) * 0.37 + res)):
        big, small = both(lambda: ops.conv_gemm(nhwc(x1), pw, x2=nhwc(x2), splitk=1, **kw))
        assert torch.equal(big, small), list(kw)
        assert rel_err(big.permute(0, 3, 1, 2), want) < 3e-3, list(kw)
    xt = q16(torch.randn(N, C1, 8, 16, generator=g))                            # 3 x 128 pixels: M = 384 = one tile and a half
    wt_ = q16(torch.randn(Cout, C1, 3, 3, generator=g) / math.sqrt(9 * C1))
    pwt_ = ops.pack_weight(wt_, b, torch.float16, DEV)
    big, small = both(lambda: ops.conv_gemm(nhwc(xt), pwt_, temb=tdev, splitk=1))
    assert torch.equal(big, small)
    assert rel_err(big.permute(0, 3, 1, 2), F.conv2d(xt, wt_, b, padding=1) + temb[:, :, None, None]) < 3e-3
    # the two-word residual stream on the big tile: first word pair out of a one-word residual (bf16)
    xb, rb = nhwc(x1, torch.bfloat16), nhwc(res, torch.bfloat16)
    pwb = ops.pack_weight(q16(w[:, :C1], torch.bfloat16), b, torch.bfloat16, DEV)
    big, small = both(lambda: ops.conv_gemm(xb, pwb, residual=rb, wide=True, splitk=1))
    assert torch.equal(big, small) and torch.equal(big._lo, small._lo) and float(big._lo.float().abs().max()) > 0
    # stride 2 and 1x1, Cout = 320 (a single N tile)
    w2 = q16(torch.randn(320, C1, 3, 3, generator=g) / math.sqrt(9 * C1))
    pw2 = ops.pack_weight(w2, None, torch.float16, DEV)
    big, small = both(lambda: ops.conv_gemm(nhwc(x1), pw2, stride=2))
    assert torch.equal(big, small) and rel_err(big.permute(0, 3, 1, 2), F.conv2d(x1, w2, None, stride=2, padding=1)) < 3e-3
    w3 = q16(torch.randn(960, C1, 1, 1, generator=g) / math.sqrt(C1))
    pw3 = ops.pack_weight(w3, b[:1].repeat(960), torch.float16, DEV)
    big, small = both(lambda: ops.conv_gemm(nhwc(x1), pw3))
    assert torch.equal(big, small)
    # grouped: 3 weight sets over 2 + 4 + 2 samples of 16x16 (256 pixels each)
    counts, Hg = [2, 4, 2], 16
    xg = torch.randn(sum(counts), Hg, Hg, C1, generator=g).to(DEV, torch.float16)
    pws = [ops.pack_weight(torch.randn(320, C1, 3, 3, generator=g) / 34, torch.randn(320, generator=g) * 0.1,
                           torch.float16, DEV) for _ in counts]
    big, small = both(lambda: ops.conv_gemm(xg, pws, group_n=counts))
    assert torch.equal(big, small)
    # tail sources (conv2 + conv_shortcut as one launch) on the big tile, slices that start inside the tail
    Ct1, Ct2 = 64, 128
    t1 = q16(torch.randn(N, Ct1, H, H, generator=g))
    t2 = q16(torch.randn(N, Ct2, H, H, generator=g))
    wc = q16(torch.randn(320, C1, 3, 3, generator=g) / math.sqrt(9 * C1))
    wt = q16(torch.randn(320, Ct1 + Ct2, generator=g) / math.sqrt(Ct1 + Ct2))
    pwt = ops.pack_weight_tail(wc, wt, b[:320], torch.float16, DEV)
    reft = F.conv2d(x1, wc, b[:320], padding=1) + F.conv2d(torch.cat([t1, t2], 1), wt[:, :, None, None])
    for splitk in (1, 7, 21):                             # K = 9 * 128 + 192 = 21 K-tiles: one per slice at 21
        big, small = both(lambda: ops.conv_gemm(nhwc(x1), pwt, tail=(nhwc(t1), nhwc(t2)), splitk=splitk))
        assert torch.equal(big, small), splitk
        assert rel_err(big.permute(0, 3, 1, 2), reft) < 3e-3, splitk
    with pytest.raises(Exception):                        # 128-pixel group boundaries cannot use 256-pixel tiles
        ops.FORCE_BN = 320
        try:
            ops.conv_gemm(xg[:, :8, :8].contiguous(), pws, group_n=counts)   # 64-pixel samples: groups of 128/256/128 px
        finally:
            ops.FORCE_BN = 0


@pytest.mark.parametrize("N,H,C,Cout,Ct1,Ct2,splitk", [
    (2, 16, 128, 256, 64, 0, None),       # planner's choice
    (2, 16, 128, 256, 128, 64, None),     # two tail sources (decoder: shortcut over the skip concat)
    (1, 8, 64, 128, 64, 64, 11),          # K = 9*64 + 128 = 11 K-steps, one per slice: the last two slices START in the tail
    (2, 8, 256, 320, 192, 0, 5),          # slices that straddle the 3x3 taps / tail boundary, 160-wide N tile
])
def test_conv_gemm_tail_sources(N, H, C, Cout, Ct1, Ct2, splitk):
    """conv3x3(h) + conv1x1(cat(t1, t2)) as ONE launch (ResnetBlock2D conv2 + conv_shortcut, es_gemm_desc.t1/t2)."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + C + Ct1 + Ct2)
    h = q16(torch.randn(N, C, H, H, generator=g))
    t1 = q16(torch.randn(N, Ct1, H, H, generator=g))
    t2 = q16(torch.randn(N, Ct2, H, H, generator=g)) if Ct2 else None
    w3 = q16(torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5)
    w1 = q16(torch.randn(Cout, Ct1 + Ct2, 1, 1, generator=g) / (Ct1 + Ct2) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    tcat = t1 if t2 is None else torch.cat([t1, t2], 1)
    ref = F.conv2d(h, w3, None, padding=1) + F.conv2d(tcat, w1, None) + b[None, :, None, None]
    pw = ops.pack_weight_tail(w3, w1, b, torch.float16, DEV)
    y = ops.conv_gemm(nhwc(h), pw, tail=(nhwc(t1), None if t2 is None else nhwc(t2)), splitk=splitk)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 3e-3
    with pytest.raises(Exception):
        ops.conv_gemm(nhwc(h), pw, tail=(nhwc(t1)[..., :32],))


def test_conv_gemm_tail_sources_grouped_matches_per_group():
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(77)
    C, Cout, Ct, H = 64, 128, 64, 16
    pws, hs, ts = [], [], []
    for n in (2, 1):
        pws.append(ops.pack_weight_tail(q16(torch.randn(Cout, C, 3, 3, generator=g) / 24), q16(torch.randn(Cout, Ct, generator=g) / 8),
                                        torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV))
        hs.append(nhwc(q16(torch.randn(n, C, H, H, generator=g))))
        ts.append(nhwc(q16(torch.randn(n, Ct, H, H, generator=g))))
    sep = torch.cat([ops.conv_gemm(h, p, tail=(t,), splitk=1) for h, p, t in zip(hs, pws, ts)], 0)
    grp = ops.conv_gemm(torch.cat(hs, 0), pws, tail=(torch.cat(ts, 0),), group_n=[2, 1], splitk=1)
    assert torch.equal(sep, grp)


def test_two_group_launches_group_norm_and_conv():
    """ngroups == 2 (the group tables have four slots; unused ones must not catch any tile / sample)."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(5)
    C, H, counts = 64, 16, [1, 2]
    x = torch.randn(sum(counts), H, H, C, generator=g).to(DEV, torch.float16)
    gam = [(1 + 0.1 * torch.randn(C, generator=g)).to(DEV) for _ in counts]
    bet = [(0.1 * torch.randn(C, generator=g)).to(DEV) for _ in counts]
    ng = ops.group_norm(x, gam, bet, 32, 1e-5, True, group_n=counts)
    a = 0
    for i, n in enumerate(counts):
        assert torch.equal(ng[a:a + n], ops.group_norm(x[a:a + n].contiguous(), gam[i], bet[i], 32, 1e-5, True))
        a += n
    pws = [ops.pack_weight(torch.randn(128, C, 3, 3, generator=g) / 24, torch.randn(128, generator=g) * 0.1, torch.float16, DEV)
           for _ in counts]
    yg = ops.conv_gemm(x, pws, group_n=counts, splitk=1)
    a = 0
    for i, n in enumerate(counts):
        assert torch.equal(yg[a:a + n], ops.conv_gemm(x[a:a + n].contiguous(), pws[i], splitk=1))
        a += n


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_linear_xs_row_stationary_kernel(dtype):
    """es_linear_xs (csrc/linear_xs.hip: activations in registers, weights streamed through an LDS ring) vs torch and vs
    the tiled kernel, on every instantiation: K = 320 | 640, plain | GEGLU, LayerNorm | none, N split over slices
    (small M), ragged M, grouped weights, a bias-free projection (to_q|k|v)."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(99)
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    ops.XS_MIN_M = 0                                      # no size policy here: every instantiation must be exercised
    cases = [  # M, C, Cout, geglu, ln, bias
        (57344 // 8, 320, 960, False, True, False),      # to_q|k|v: 28 row blocks x 9 slices
        (2048, 320, 2560, True, True, True),             # decoder GEGLU: 8 row blocks x 20 slices
        (1000, 640, 1920, False, True, False),           # ragged M, K = 640
        (512, 640, 5120, True, True, True),              # K = 640 GEGLU (4 stages per output line)
        (300, 320, 320, False, False, True),             # proj_in-like plain layer, ragged
        (256, 640, 640, False, False, True),
        (40, 320, 1280, True, False, True),              # fewer rows than one wave pair
    ]
    for M, C, Cout, geglu, ln, bias in cases:
        x = q16(torch.randn(M, C, generator=g) * 1.5 + 0.7 + torch.randn(M, 1, generator=g), dtype)
        gamma = 1 + 0.2 * torch.randn(C, generator=g)
        beta = 0.1 * torch.randn(C, generator=g)
        w = torch.randn(Cout, C, generator=g) / math.sqrt(C)
        b = torch.randn(Cout, generator=g) * 0.1 if bias else None
        xin = F.layer_norm(x, (C,), gamma, beta, 1e-5) if ln else x
        y = F.linear(xin, w, b)
        if geglu:
            h, gate = y.chunk(2, dim=-1)
            y = h * F.gelu(gate)
        if ln:
            pw = ops.pack_weight_ln(w, b, gamma, beta, 1e-5, dtype, DEV, geglu=geglu)
        else:
            pw = ops.pack_weight(w, b, dtype, DEV, geglu=geglu)
        xd = x.to(DEV, dtype)
        assert ops.xs_eligible(M, pw, None, None, 1)
        ops.XS_ENABLED = False
        try:
            tiled = ops.linear(xd, pw)
        finally:
            ops.XS_ENABLED = True
        got = ops.linear(xd, pw)
        assert rel_err(got, y) < tol, (M, C, Cout, geglu, ln, rel_err(got, y))
        assert rel_err(got, tiled) < tol, (M, C, Cout, geglu, ln)
    # grouped: four weight sets over [2, 6, 4, 2] x 256 rows (the lockstep encoder's group table), GEGLU
    C, Cout, counts = 320, 2560, [512, 1536, 1024, 512]
    xg = q16(torch.randn(sum(counts), C, generator=g) * 2 + 0.3, dtype)
    pws, refs, a = [], [], 0
    for n in counts:
        gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        w, b = torch.randn(Cout, C, generator=g) / math.sqrt(C), torch.randn(Cout, generator=g) * 0.1
        pws.append(ops.pack_weight_ln(w, b, gamma, beta, 1e-5, dtype, DEV, geglu=True))
        hh, gate = F.linear(F.layer_norm(xg[a:a + n], (C,), gamma, beta, 1e-5), w, b).chunk(2, dim=-1)
        refs.append(hh * F.gelu(gate))
        a += n
    yg = ops.linear(xg.to(DEV, dtype), pws, group_n=counts)
    again = ops.linear(xg.to(DEV, dtype), pws, group_n=counts)
    assert rel_err(yg, torch.cat(refs)) < tol
    assert torch.equal(yg, again)                         # deterministic: same launch twice, bit for bit
    # residual (es_xs_desc.residual; round 4): out = x W^T + bias + residual at K = 320 - Attention.to_out / proj_out of the 64 x 64
    # level - many stages per workgroup (every counted wait of the steady state), few (the tails), N slices, ragged M, grouped
    # (M > 32768: one slice per row block - 5 or 20 stages per workgroup, the steady state of the counted waits; below that the N range
    #  is split and a workgroup runs one or two stages)
    for M, Cout, counts in ((2048, 320, None), (700, 1280, None), (40, 320, None), (256 * 14, 320, [512, 1536, 1024, 512]), (8192, 640, None),
                            (40000, 320, None), (33000, 1280, None), (256 * 160, 640, [8192, 16384, 8192, 8192])):
        x = q16(torch.randn(M, 320, generator=g) * 1.5, dtype)
        res = q16(torch.randn(M, Cout, generator=g) * 2.0, dtype)
        ws = [(torch.randn(Cout, 320, generator=g) / math.sqrt(320), torch.randn(Cout, generator=g) * 0.1) for _ in (counts or [M])]
        pws = [ops.pack_weight(w, b, dtype, DEV) for w, b in ws]
        rows = counts or [M]
        ref = torch.cat([F.linear(x[sum(rows[:i]):sum(rows[:i + 1])], w, b) for i, (w, b) in enumerate(ws)]) + res
        kw = dict(group_n=counts) if counts else {}
        pw = pws if counts else pws[0]
        xd, rd = x.to(DEV, dtype), res.to(DEV, dtype)
        ops.XS_RESIDUAL = False
        try:
            tiled = ops.linear(xd, pw, residual=rd, **kw)
        finally:
            ops.XS_RESIDUAL = True
        prof = ops.PROFILE
        class Rec:                                          # which kernel ran: the profiler hook sees the descriptor kind
            descs, metas = [], []
            def next(self, meta):
                self.metas.append(meta); return None
        ops.PROFILE = Rec()
        try:
            got = ops.linear(xd, pw, residual=rd, **kw)
            assert ops.PROFILE.metas[-1][3].get("kernel") == "linear_xs" and ops.PROFILE.metas[-1][3]["residual"]
        finally:
            ops.PROFILE = prof
        again = ops.linear(xd, pw, residual=rd, **kw)
        torch.cuda.synchronize()
        assert rel_err(got, ref) < tol, (M, Cout, rel_err(got, ref))
        assert rel_err(got, tiled) < tol and torch.equal(got, again)
    ops.XS_MIN_M = 8192


@pytest.mark.parametrize("N,H,C1,C2,Cout,stride,up,Ct,splitk,bn", [
    (2, 16, 128, 0, 256, 1, False, 0, None, 0),      # planner's choice, plain 3x3
    (2, 16, 64, 128, 320, 1, False, 0, None, 160),   # skip concat: chunks run over (x | x2)
    (1, 16, 128, 0, 256, 2, False, 0, None, 128),    # stride 2
    (2, 8, 64, 0, 128, 1, True, 0, None, 128),       # nearest-2x upsample: per-row parities place the taps
    (2, 8, 128, 0, 128, 1, False, 64, 5, 128),       # 1x1 tail behind the chunks, split-K slices that start mid-chunk / in the tail
    (3, 5, 64, 0, 64, 1, False, 0, None, 64),        # ragged M on the 64 x 64 tile
    (8, 32, 128, 0, 320, 1, False, 64, None, 320),   # the 256 x 320 tile, tail sources
    (8, 32, 64, 64, 320, 2, False, 0, 3, 320),       # the 256 x 320 tile: concat, stride 2, split-K
    (4, 16, 64, 0, 320, 1, True, 0, None, 320),      # the 256 x 320 tile with the fused upsample
])
def test_conv_gemm_chunk_major_k_order(N, H, C1, C2, Cout, stride, up, Ct, splitk, bn):
    """es_gemm_desc.korder = 1 (opt-in, ES_CHUNK_MAJOR=1): weights packed k = (c / 64, ky, kx, c % 64) and the loaders of both
    GEMM kernels walking K in that order - the same convolution as the tap-major default, to rounding (the fp32 accumulation
    order differs), and each against torch's conv2d."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(N * 100 + C1 + C2 + Cout + stride)
    C = C1 + C2
    x = q16(torch.randn(N, C, H, H, generator=g))
    w = q16(torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = torch.randn(Cout, generator=g) * 0.1
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=1)
    t = wt = None
    if Ct:
        t = q16(torch.randn(N, Ct, H, H, generator=g))
        wt = q16(torch.randn(Cout, Ct, 1, 1, generator=g) / math.sqrt(Ct))
        ref = ref + F.conv2d(t, wt, None)
    xs = nhwc(x)
    x1, x2 = (xs[..., :C1].contiguous(), xs[..., C1:].contiguous()) if C2 else (xs, None)
    outs = {}
    prev = ops.CHUNK_MAJOR, ops.FORCE_BN
    try:
        ops.FORCE_BN = bn
        for ko in (0, 1):
            ops.CHUNK_MAJOR = bool(ko)
            pw = ops.pack_weight_tail(w, wt, b, torch.float16, DEV) if Ct else ops.pack_weight(w, b, torch.float16, DEV)
            assert pw.korder == ko
            outs[ko] = ops.conv_gemm(x1, pw, x2=x2, stride=stride, upsample=up, splitk=splitk, tail=(nhwc(t),) if Ct else None)
    finally:
        ops.CHUNK_MAJOR, ops.FORCE_BN = prev
    torch.cuda.synchronize()
    for ko in (0, 1):
        assert rel_err(outs[ko].permute(0, 3, 1, 2), ref) < 3e-3, ko
    assert rel_err(outs[1], outs[0]) < 1e-3


@pytest.mark.parametrize("N,H,C,Cout,k,bn,splitk", [
    (2, 16, 128, 128, 3, 128, None),      # ResnetBlock conv2 + x
    (2, 16, 320, 320, 1, 160, None),      # to_out + tokens (8-wave tile when the planner picks it)
    (8, 32, 128, 320, 3, 320, None),      # the 256 x 320 tile
    (1, 8, 256, 64, 3, 64, None),         # the 64 x 64 tile
    (1, 8, 640, 128, 3, 128, 5),          # split-K: the reduce kernel adds and splits
])
def test_conv_gemm_wide_residual_stream(N, H, C, Cout, k, bn, splitk):
    """es_gemm_desc.residual_lo / out_lo (ops.conv_gemm(wide=True) on a bf16 pipeline): the residual add sees hi + lo of the
    stream, sums in fp32 and writes hi + lo back - out + out_lo equals round(conv) + (residual + residual_lo) to bf16's SECOND
    8 bits (2^-15 of the value), where the single-tensor stream is good for 2^-8; and out alone is that sum rounded once."""
    from edgestyle_amd import ops
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(N + C + Cout + k)
    x = q16(torch.randn(N, C, H, H, generator=g), dt)
    w = q16(torch.randn(Cout, C, k, k, generator=g) / math.sqrt(C * k * k), dt)
    b = torch.randn(Cout, generator=g) * 0.1
    res_wide = torch.randn(N, Cout, H, H, generator=g) * 3.0                     # a stream value with 16 significant bits
    res_hi = q16(res_wide, dt)
    res_lo = q16(res_wide - res_hi, dt)
    branch = q16(F.conv2d(x, w, b, padding=k // 2), dt)                          # the epilogue rounds the branch first
    want = branch + res_hi + res_lo
    pw = ops.pack_weight(w, b, dt, DEV)
    r = nhwc(res_hi, dt)
    r._lo = nhwc(res_lo, dt)
    prev = ops.FORCE_BN
    try:
        ops.FORCE_BN = bn
        y = ops.conv_gemm(nhwc(x, dt), pw, residual=r, wide=True, splitk=splitk)
        y1 = ops.conv_gemm(nhwc(x, dt), pw, residual=r, splitk=splitk)           # single-tensor stream: hi only
    finally:
        ops.FORCE_BN = prev
    torch.cuda.synchronize()
    assert hasattr(y, "_lo") and not hasattr(y1, "_lo")
    hi, lo = y.float().cpu().permute(0, 3, 1, 2), y._lo.float().cpu().permute(0, 3, 1, 2)
    scale = float(want.abs().max())
    # the branch itself carries the GEMM's fp32-accumulation-order noise (~1e-3 of ITS scale, rounded to bf16 either way):
    # compare against the kernel's own branch, recovered from the single-tensor result within bf16 resolution
    assert float((hi + lo - want).abs().max()) <= 2.5e-2 * float(branch.abs().max()) + 2 ** -14 * scale
    # hi is the sum rounded ONCE: hi == round(hi + lo) except where the sum sat on a tie (lo = half an ulp: either neighbour)
    assert float((hi != q16(hi + lo, dt)).float().mean()) < 2e-3
    assert float(lo.abs().max()) <= 2 ** -8 * scale
    # and the wide sum is closer to the exact one than the single-tensor sum by construction
    e_wide = float((hi + lo - want).abs().mean())
    e_one = float((y1.float().cpu().permute(0, 3, 1, 2) - want).abs().mean())
    assert e_wide < 0.5 * e_one, (e_wide, e_one)


@pytest.mark.parametrize("N,H,C,Cout,k,bn,splitk", [
    (2, 16, 128, 320, 3, 160, None),      # 10-channel groups inside one 160-wide tile
    (2, 16, 128, 640, 3, 128, None),      # 20-channel groups straddle the 128-wide tiles: the two-entry scheme
    (8, 32, 128, 320, 3, 320, None),      # the 256 x 320 tile (two epilogue passes of 128 pixels x 320 couts)
    (1, 8, 256, 1280, 3, 64, None),       # 40-channel groups over 64-wide tiles
    (1, 8, 640, 1280, 3, 128, 5),         # split-K: the statistics come from the reduce kernel
    (3, 8, 64, 64, 1, 64, None),          # tiny width: 2-channel groups
    (2, 64, 64, 320, 3, 160, None),       # the 64 x 64 level (the shape the shipped policy hands over)
    (2, 64, 128, 320, 3, 160, 3),         # ... from a split-K launch (the UNet decoder at batch 1)
])
def test_group_norm_statistics_handed_over_by_the_producer(N, H, C, Cout, k, bn, splitk):
    """es_gemm_desc.gn_part -> es_gn_desc.ext_chunks: the convolution's epilogue writes the per-(sample,
    64-pixel block, group) sums of its output; the GroupNorm that reads it is one streaming pass.  Same normalised tensor as the
    stand-alone GroupNorm of the same convolution output (statistics summed in another order: fp32 rounding apart), both against
    torch; the convolution's own output is bit-identical with and without the hand-over."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(N + C + Cout + k + bn)
    x = q16(torch.randn(N, C, H, H, generator=g))
    w = q16(torch.randn(Cout, C, k, k, generator=g) / math.sqrt(C * k * k))
    b = torch.randn(Cout, generator=g) * 0.5
    gam, bet = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV), (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    pw = ops.pack_weight(w, b, torch.float16, DEV)
    prev = ops.FORCE_BN, ops.GN_HANDOVER_ALL, ops.GN_HANDOVER
    try:
        ops.FORCE_BN, ops.GN_HANDOVER_ALL, ops.GN_HANDOVER = bn, True, True      # opt-in feature (ES_GN_HANDOVER), every shape
        y0 = ops.conv_gemm(nhwc(x), pw, splitk=splitk)
        y1 = ops.conv_gemm(nhwc(x), pw, splitk=splitk, gn_groups=32)
        assert torch.equal(y0, y1) and hasattr(y1, "_gnp") and not hasattr(y0, "_gnp")
        n0 = ops.group_norm(y0, gam, bet, 32, 1e-5, True)      # stand-alone statistics
        n1 = ops.group_norm(y1, gam, bet, 32, 1e-5, True)      # the producer's
    finally:
        ops.FORCE_BN, ops.GN_HANDOVER_ALL, ops.GN_HANDOVER = prev
    torch.cuda.synchronize()
    part = y1._gnp[0].double().cpu()                            # [N, 2 * HW/64, 32, 2]
    yc = y1.double().cpu().reshape(N, H * H, 32, Cout // 32)
    assert torch.allclose(part[..., 0].sum(1), yc.sum((1, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(part[..., 1].sum(1), (yc * yc).sum((1, 3)), rtol=1e-4, atol=1e-2)
    ref = F.silu(F.group_norm(y1.float().permute(0, 3, 1, 2), 32, gam, bet, 1e-5))
    assert rel_err(n1.permute(0, 3, 1, 2), ref) < 3e-3 and rel_err(n0.permute(0, 3, 1, 2), ref) < 3e-3
    assert rel_err(n1, n0) < 2e-3


def test_group_norm_hand_over_does_not_depend_on_the_tile_or_the_grouping():
    """The statistics a producer hands over are summed in ONE order whatever tile, split-K or grouping the launch uses: the
    same convolution through the 64-, 128-, 160-wide and 256 x 320 tiles and through a split-K launch gives bit-identical partial
    tables (their outputs are bit-identical for splitk 1 already), and a grouped launch equals its per-net launches."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(5)
    N, H, C, Cout = 8, 32, 128, 640
    x = nhwc(q16(torch.randn(N, C, H, H, generator=g)))
    pws = [ops.pack_weight(q16(torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C)), torch.randn(Cout, generator=g) * 0.3,
                           torch.float16, DEV) for _ in range(2)]
    prev = ops.FORCE_BN, ops.GN_HANDOVER_ALL, ops.GN_HANDOVER
    tabs, outs = {}, {}
    try:
        ops.GN_HANDOVER_ALL = ops.GN_HANDOVER = True
        for bn in (64, 128, 320):
            ops.FORCE_BN = bn
            y = ops.conv_gemm(x, pws[0], splitk=1, gn_groups=32)
            tabs[bn], outs[bn] = y._gnp[0].clone(), y.clone()
        ops.FORCE_BN = 128
        grp = ops.conv_gemm(x, pws, group_n=[4, 4], splitk=1, gn_groups=32)
        sep = [ops.conv_gemm(x[i * 4:(i + 1) * 4].contiguous(), pws[i], splitk=1, gn_groups=32) for i in range(2)]
        gam, bet = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
        n_grp = ops.group_norm(grp, [gam, gam], [bet, bet], 32, 1e-5, True, group_n=[4, 4])
        n_sep = torch.cat([ops.group_norm(s, gam, bet, 32, 1e-5, True) for s in sep])
    finally:
        ops.FORCE_BN, ops.GN_HANDOVER_ALL, ops.GN_HANDOVER = prev
    torch.cuda.synchronize()
    assert torch.equal(outs[64], outs[128]) and torch.equal(outs[128], outs[320])
    # per (sample, block, group) the two entries may be split differently between N tiles: their SUM is what the consumer adds up,
    # and it is made of the same per-channel sums - compare that sum
    for bn in (128, 320):
        a = tabs[64].view(N, -1, 2, 32, 2)
        b = tabs[bn].view(N, -1, 2, 32, 2)
        assert torch.allclose(a.sum(2), b.sum(2), rtol=2e-6, atol=1e-4)
    assert torch.equal(grp, torch.cat(sep)) and torch.equal(grp._gnp[0], torch.cat([s._gnp[0] for s in sep]))
    assert torch.equal(n_grp, n_sep)


def test_oversize_launches_are_cut_into_runs_of_whole_samples():
    """es_conv_gemm / es_linear_xs run a launch whose operands outgrow the kernels' 32-bit buffer offsets (2 GiB: 12+ try-ons per
    call at 512 x 512) as several launches over runs of whole samples.  With the limit lowered (es_set_operand_limit) small launches
    are cut the same way: bit-identical to the uncut launch - grouped 3x3 with temb + residual, split-K, stride 2, nearest-2x
    upsample, 1x1 tail sources, a GEGLU linear layer with the LayerNorm fold over rows, linear_xs grouped / with a residual,
    and a source shared modulo x_nmod."""
    from edgestyle_amd import ops, lib
    L = lib.load()
    g = torch.Generator().manual_seed(2024)
    counts, H, C, Cout = [2, 6, 4, 2], 8, 64, 128
    N = sum(counts)
    x = torch.randn(N, H, H, C, generator=g).to(DEV, torch.float16)
    x2 = torch.randn(N, H, H, C, generator=g).to(DEV, torch.float16)
    res = torch.randn(N, H, H, Cout, generator=g).to(DEV, torch.float16)
    temb = torch.randn(N, 256, generator=g).to(DEV, torch.float16)
    pws = [ops.pack_weight(torch.randn(Cout, C, 3, 3, generator=g) / 24, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV) for _ in counts]
    pw2 = ops.pack_weight(torch.randn(Cout, 2 * C, 3, 3, generator=g) / 34, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
    pwt = ops.pack_weight_tail(torch.randn(Cout, C, 3, 3, generator=g) / 24, torch.randn(Cout, 2 * C, 1, 1, generator=g) / 11,
                               torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
    M, K = 2048, 1280
    tok = torch.randn(M, K, generator=g).to(DEV, torch.float16)
    pwl = ops.pack_weight_ln(torch.randn(512, K, generator=g) / 36, torch.randn(512, generator=g) * 0.1, 1 + 0.1 * torch.randn(K, generator=g),
                             0.1 * torch.randn(K, generator=g), 1e-5, torch.float16, DEV, geglu=True)
    ehs = torch.randn(2, 77, 768, generator=g).to(DEV, torch.float16)
    pkv = ops.pack_weight(torch.randn(640, 768, generator=g) / 28, None, torch.float16, DEV)
    Mx = 4096
    xs_x = torch.randn(Mx, 320, generator=g).to(DEV, torch.float16)
    xs_r = torch.randn(Mx, 320, generator=g).to(DEV, torch.float16)
    xs_ln = [ops.pack_weight_ln(torch.randn(960, 320, generator=g) / 18, torch.randn(960, generator=g) * 0.1, 1 + 0.1 * torch.randn(320, generator=g),
                                0.1 * torch.randn(320, generator=g), 1e-5, torch.float16, DEV) for _ in range(2)]
    xs_o = ops.pack_weight(torch.randn(320, 320, generator=g) / 18, torch.randn(320, generator=g) * 0.1, torch.float16, DEV)

    def run():
        outs = [ops.conv_gemm(x, pws, temb=temb[:, 64:], residual=res, group_n=counts),
                ops.conv_gemm(x, pw2, x2=x2, splitk=3),
                ops.conv_gemm(x, pws[0], stride=2),
                ops.conv_gemm(x[:, :4, :4].contiguous(), pws[1], upsample=True),
                ops.conv_gemm(x, pwt, tail=(x, x2)),
                ops.linear(tok, pwl)]
        old = ops.XS_MIN_M
        ops.XS_MIN_M = 0
        try:
            outs.append(ops.linear(xs_x, xs_ln, group_n=[1024, 3072]))
            outs.append(ops.linear(xs_x, xs_o, residual=xs_r))
        finally:
            ops.XS_MIN_M = old
        torch.cuda.synchronize()
        return outs
    whole = run()
    prev = L.es_set_operand_limit(3 * H * H * Cout * 2 + 1)         # three samples of the 8 x 8 x 128 tensors; 96 rows of the linear layers
    try:
        assert prev == 0x7FFFFFFF
        cut = run()
    finally:
        L.es_set_operand_limit(0)
    for i, (a, b) in enumerate(zip(whole, cut)):
        assert torch.equal(a, b), i
    # GroupNorm in front of linear_xs (es_xs_desc.gn_part): cut between whole samples, the statistics pointer moved along
    xg = torch.randn(8, 32, 32, 320, generator=g).to(DEV, torch.float16)
    gam, bet = (1 + 0.2 * torch.randn(320, generator=g)).to(DEV), (0.2 * torch.randn(320, generator=g)).to(DEV)
    assert ops.gn_fold_ok(8 * 1024, 1024, 32, xs_o, None, None)
    whole = ops.gn_proj_in(xg, gam, bet, 32, 1e-6, xs_o)
    L.es_set_operand_limit(2 * 1024 * 640 + 256 * 640 + 1)            # two samples of 1024 rows per cut
    try:
        assert torch.equal(ops.gn_proj_in(xg, gam, bet, 32, 1e-6, xs_o), whole)
    finally:
        L.es_set_operand_limit(0)
    # a source shared modulo x_nmod (the text states of a weight-sharing group) is cut at multiples of x_nmod ...
    whole = ops.linear(ehs, pkv, x_rep=3)
    L.es_set_operand_limit(200 * 640 * 2)
    try:
        assert torch.equal(ops.linear(ehs, pkv, x_rep=3), whole)
        L.es_set_operand_limit(100 * 640 * 2)
        with pytest.raises(Exception, match="exceed|larger than"):        # ... and says so when not even one period (or the source itself) fits
            ops.linear(ehs, pkv, x_rep=3)
    finally:
        L.es_set_operand_limit(0)


def test_clock_probe_reads_a_plausible_shader_clock():
    """es_clock_probe (measurement tool, tools/clock_in_kernel.py): one wave counts shader cycles against the 100 MHz real-time counter."""
    from edgestyle_amd import lib
    L = lib.load()
    out = torch.zeros(2, dtype=torch.int64, device=DEV)
    lib.check(L.es_clock_probe(out.data_ptr(), 2000, None), "es_clock_probe")        # 2 ms
    torch.cuda.synchronize()
    cyc, ticks = [int(v) for v in out.cpu()]
    assert 200000 <= ticks <= 400000, ticks                                           # 2 ms of 100 MHz ticks (+ the last sleep)
    assert 100.0 <= cyc / ticks * 100.0 <= 2600.0, (cyc, ticks)                       # MHz: between deep idle and the 2.4 GHz peak
    with pytest.raises(Exception):
        lib.check(L.es_clock_probe(None, 2000, None), "es_clock_probe")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,H,counts", [(320, 32, None), (320, 16, [2, 6, 4, 2]), (640, 32, [16, 8, 8]), (640, 16, None)])
def test_group_norm_in_front_of_the_row_stationary_projection(dtype, C, H, counts):
    """Transformer2DModel.norm -> proj_in as one read (ops.gn_proj_in: es_group_norm stats_only + es_linear_xs gn_part) against a plain
    fp32 GroupNorm + 1x1 convolution, and against the two-launch path it replaces (same arithmetic up to the rounding of the normalised
    value: the fused form computes x * (rstd gamma) + (beta - mean rstd gamma), the stand-alone one (x - mean) rstd gamma + beta)."""
    from edgestyle_amd import ops
    g = torch.Generator().manual_seed(C + H + (len(counts) if counts else 0))
    N = sum(counts) if counts else 36 if C == 320 else 34
    while N * H * H < (8192 if C == 320 else 32768):
        N *= 2
    if counts:
        N = sum(counts)
        if N * H * H < (8192 if C == 320 else 32768):
            counts = [c * 4 for c in counts]
            N = sum(counts)
    x = (torch.randn(N, C, H, H, generator=g) * 1.5 + 0.3 * torch.randn(N, C, 1, 1, generator=g))
    xq = q16(x, dtype)
    ng = len(counts) if counts else 1
    gam = [1 + 0.2 * torch.randn(C, generator=g) for _ in range(ng)]
    bet = [0.2 * torch.randn(C, generator=g) for _ in range(ng)]
    ws = [q16(torch.randn(C, C, 1, 1, generator=g) / math.sqrt(C), dtype) for _ in range(ng)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(ng)]
    pws = [ops.pack_weight(w, b, dtype, DEV) for w, b in zip(ws, bs)]
    gd, bd = [t.to(DEV) for t in gam], [t.to(DEV) for t in bet]
    xin = nhwc(xq, dtype)
    assert ops.gn_fold_ok(N * H * H, H * H, 32, pws[0], pws if counts else None, counts)
    if counts:
        y = ops.gn_proj_in(xin, gd, bd, 32, 1e-6, pws, group_n=counts)
    else:
        y = ops.gn_proj_in(xin, gd[0], bd[0], 32, 1e-6, pws[0])
    old = ops.GN_FOLD
    ops.GN_FOLD = False
    try:
        y2 = ops.gn_proj_in(xin, gd, bd, 32, 1e-6, pws, group_n=counts) if counts else ops.gn_proj_in(xin, gd[0], bd[0], 32, 1e-6, pws[0])
    finally:
        ops.GN_FOLD = old
    torch.cuda.synchronize()
    a = 0
    refs = []
    for i, n in enumerate(counts or [N]):
        h = F.group_norm(xq[a:a + n], 32, gam[i], bet[i], 1e-6)
        refs.append(F.conv2d(q16(h, dtype), ws[i], bs[i]))
        a += n
    ref = torch.cat(refs, 0)
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    assert rel_err(y.permute(0, 3, 1, 2), ref) < tol
    assert rel_err(y2.permute(0, 3, 1, 2), ref) < tol
    assert rel_err(y, y2) < tol

"""Shared test scaffolding: seeded random-init weight sets in the reference's on-disk key layout, and the
oracle-vs-HIP single-step check used by __graft_entry__.smoke()."""
import torch

from edgestyle_amd import config as C, weights as W


def make_weights(ucfg, vcfg, rank=4, seed=0):
    """UNet, openpose ControlNet, two ControlLoRA nets (saved keys only), fusion blocks, VAE — fp32 CPU."""
    return dict(
        unet=W.random_state_dict(W.unet_shapes(ucfg), seed, "unet."),
        openpose=W.random_state_dict(W.controlnet_shapes(ucfg), seed, "openpose."),
        lora0=W.random_state_dict(W.controllora_saved_shapes(ucfg, rank), seed, "controlnet_0."),
        lora1=W.random_state_dict(W.controllora_saved_shapes(ucfg, rank), seed, "controlnet_1."),
        fusion=W.random_state_dict(W.fusion_shapes(ucfg), seed, "fusion."),
        vae=W.random_state_dict(W.vae_shapes(vcfg), seed, "vae."),
    )


def quantize(sd, dtype=torch.float16):
    """Round weights through the storage dtype so oracle and HIP path see the same parameter values."""
    return {k: v.to(dtype).float() for k, v in sd.items()}


def oracle_nets(ws, ucfg):
    """The reference's 6-net list [agnostic-LoRA, pose, clothes-LoRA, pose, clothes-LoRA, pose] (TT:252-258) with
    LoRA nets tied to the UNet encoder (TT:259-261)."""
    from oracle import sd15_oracle as O
    n0 = O.tie_weights(ws["lora0"], ws["unet"])
    n1 = O.tie_weights(ws["lora1"], ws["unet"])
    op = ws["openpose"]
    return [(n0, ucfg), (op, ucfg), (n1, ucfg), (op, ucfg), (n1, ucfg), (op, ucfg)]


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-6))


def to_nhwc(x, device, dtype=torch.float16, cpad=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad != x.shape[-1]:
        x = torch.nn.functional.pad(x, (0, cpad - x.shape[-1]))
    return x.to(device=device, dtype=dtype)


def tiny_step_check(device="cuda:0", seed=0):
    """One full 6-cond multi-ControlNet + UNet step (== export_onnx.py:43-74) on the tiny config: HIP vs oracle.
    Returns max abs error of noise_pred."""
    from oracle import sd15_oracle as O
    from edgestyle_amd import engine as E
    ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=seed).items()}
    g = torch.Generator().manual_seed(42)
    N, s, c0 = 2, ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(N, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(N, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    scales = [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]
    t = 501
    ref = O.denoise_step(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), x, t, ehs, conds, scales)

    from edgestyle_amd.models import StepRunner
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.float16, device)
    out = runner.step_nchw(x.to(device), t, ehs.to(device), [c.to(device) for c in conds], scales)
    torch.cuda.synchronize()
    return float((out.float().cpu() - ref).abs().max())

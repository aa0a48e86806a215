#!/usr/bin/env python
"""Headline benchmark: 512x512 try-on images/sec @ 50 DDIM steps, 6-cond ControlNet (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W          # one process, cuda:0
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N --steps K --warmup W          # no launcher: starts the N ranks itself (self_launch)

One "step" = one pass of the hot path over one batch: condition embedding (3 VAE encodes + 3 openpose conv
stacks on the CFG-duplicated batch, PL:629-664) -> 50 x [6 ControlNet passes -> 13 fusion blocks -> UNet -> CFG ->
DDIM] -> VAE decode -> [0,1] image, inputs already resident in HBM.  Weights are seeded
random-init SD1.5-shaped tensors (no checkpoints exist offline), data is synthetic.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (implicit-GEMM conv/linear kernel, MFMA
bound), `cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample of denoising steps), `parity` (the
HIP step of the benchmarked configuration against that oracle on identical inputs), `throughput_mode` (BASELINE
configs[2], batch 8, with its own roofline) and `stress_mode` (BASELINE configs[4], 768x768 bf16 batch 4).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_at_sustained_clock", "sclk_mhz", "power_w", "clock_samples", "traffic",
                 "traffic_source", "mfma_util_pmc", "launches_per_step", "avg_launch_us", "gemm_time_per_step_ms", "conv3x3_only",
                 "families_stamped", "stamped", "how")


def host_cores() -> int:
    """Threads the CPU baseline may use: the scheduler affinity / cgroup quota of this process, capped at the GPU box's
    per-GPU CPU share (16).  Oversubscribing OpenMP threads on a quota-limited box stalls for minutes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("ES_BENCH_THREADS", "16"))))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def build_pipeline(device, dtype, seed=0, tiny=False, keep_cpu=False, resolution=512):
    from edgestyle_amd import config as C, weights as W
    from edgestyle_amd.models import (UNet2DConditionModel, ControlNetModel, ControlLoRAModel, AutoencoderKL,
                                      EdgeStyleMultiControlNetModel)
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
    ucfg, vcfg = (C.tiny_unet(), C.tiny_vae()) if tiny else (C.sd15_unet(), C.sd15_vae())
    if resolution != ucfg.sample_size * vcfg.scale:
        import dataclasses
        # outside the reference's domain (its fusion blocks are hard-wired to 512x512, MC:73-102): LN planes of the
        # new size are random-init like everything else (BASELINE config 5)
        ucfg = dataclasses.replace(ucfg, sample_size=resolution // vcfg.scale)
    rank = 4 if tiny else 32
    gen_dev = "cpu" if keep_cpu else device
    ws = dict(
        unet=W.random_state_dict(W.unet_shapes(ucfg), seed, "unet.", device=gen_dev),
        openpose=W.random_state_dict(W.controlnet_shapes(ucfg), seed, "openpose.", device=gen_dev),
        lora0=W.random_state_dict(W.controllora_saved_shapes(ucfg, rank), seed, "controlnet_0.", device=gen_dev),
        lora1=W.random_state_dict(W.controllora_saved_shapes(ucfg, rank), seed, "controlnet_1.", device=gen_dev),
        fusion=W.random_state_dict(W.fusion_shapes(ucfg), seed, "fusion.", device=gen_dev),
        vae=W.random_state_dict(W.vae_shapes(vcfg), seed, "vae.", device=gen_dev),
    )
    unet = UNet2DConditionModel(ws["unet"], ucfg, dtype).to(device)
    vae = AutoencoderKL(ws["vae"], vcfg, dtype).to(device)
    pose = ControlNetModel(ws["openpose"], ucfg, dtype).to(device)
    nets = []
    for key in ("lora0", "lora1"):
        n = ControlLoRAModel(ws[key], ucfg, dtype, lora_linear_rank=rank, uses_vae=True)
        n.set_autoencoder(vae)
        n.tie_weights(unet)                                   # TT:259-261
        nets.append(n.to(device))
    mc = EdgeStyleMultiControlNetModel([nets[0], pose, nets[1], pose, nets[1], pose], ucfg)   # TT:50, TT:252-258
    mc.load_state_dict(ws["fusion"])
    mc.to(device)
    pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=unet, controlnet=mc).to(device)
    return pipe, ws, ucfg, vcfg


def make_inputs(ucfg, vcfg, B, device, seed=42, first_index=0):
    """SURVEY.md §8d synthetic inputs, seed 42 (TT:274); image conds in [-1,1], pose conds in [0,1] (TT:29-48).
    Everything that belongs to ONE try-on (latents, prompt embeddings, VAE sampling noise of its conditions) is drawn
    from a generator seeded with seed + the image's GLOBAL index (first_index + position in the batch), so an image does
    not depend on the world size or on which rank serves it; the six condition images are shared."""
    s = ucfg.sample_size
    res = s * vcfg.scale
    g = torch.Generator().manual_seed(seed)
    imgs = []
    for i in range(6):
        u = torch.rand(1, 3, res, res, generator=g)
        imgs.append((u * 2 - 1 if i % 2 == 0 else u).to(device))
    lat = torch.empty(B, 4, s, s)
    pe = torch.empty(B, 77, ucfg.cross_attention_dim)
    ne = torch.empty(B, 77, ucfg.cross_attention_dim)
    cond_noise = [torch.empty(2 * B, 4, s, s) if i % 2 == 0 else None for i in range(6)]
    for j in range(B):
        gj = torch.Generator().manual_seed(seed + 1 + first_index + j)
        lat[j] = torch.randn(4, s, s, generator=gj)
        pe[j] = torch.randn(77, ucfg.cross_attention_dim, generator=gj) * 0.5
        ne[j] = torch.randn(77, ucfg.cross_attention_dim, generator=gj) * 0.5
        for i in range(0, 6, 2):                     # CFG-duplicated batch (CL:39): rows j (uncond half) and B + j
            cond_noise[i][j] = torch.randn(4, s, s, generator=gj)
            cond_noise[i][B + j] = torch.randn(4, s, s, generator=gj)
    return lat, pe.to(device), ne.to(device), imgs, cond_noise


class _FakePipeline:
    """Stand-in for the CPU rehearsal of this script's multi-rank control flow (tests/test_dist_gloo.py, `--fake-pipeline`):
    "decodes" image j as a constant plane holding the mean of its latents, so the gathered batch can be checked against
    the per-image seeds.  Never used for a measurement."""
    def __init__(self, ucfg, vcfg):
        self.res = ucfg.sample_size * vcfg.scale
        self.use_graph = False
        self.collect_timing, self.timing = False, {}

    def __call__(self, latents=None, **kw):
        import types
        img = latents.float().mean(dim=(1, 2, 3)).reshape(-1, 1, 1, 1).expand(-1, 3, 8, 8).contiguous()
        return types.SimpleNamespace(images=img)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


class ClockSampler:
    """sclk / package power while a region runs: a thread polls `rocm-smi --showclocks --showpower` (the probe of
    tools/clock_probe.py, ~3 samples per second; it costs host CPU only - the timed regions replay hipGraphs).  The chip sits at
    its package power limit under the MFMA-dense kernels (MI355X_MICROARCH.md, DVFS give-back), so a fraction of the 2.4 GHz peak
    means little without the clock the part actually held: `summary()` gives the medians, the first sample dropped."""
    def __init__(self, period=0.3):
        import threading
        self.period, self.samples, self._stop, self._th = period, [], threading.Event(), None

    def _poll(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                sclk = power = None
                for ln in out.splitlines():
                    if "sclk" in ln:
                        m = re.search(r"\((\d+)Mhz\)", ln)
                        sclk = float(m.group(1)) if m and sclk is None else sclk
                    elif "ower (W)" in ln and power is None:
                        try:
                            power = float(ln.rsplit(":", 1)[-1])
                        except ValueError:
                            pass
                if sclk is not None or power is not None:
                    self.samples.append((time.time(), sclk, power))
            except Exception:           # no rocm-smi on this host: the fields stay null
                return
            self._stop.wait(self.period)

    def start(self):
        import threading
        self.samples, self._stop = [], threading.Event()
        self._t0 = time.time()
        # not under a profiler: rocprofv3 preloads its library into every child, which initialises the GPU there before `rocm-smi` (a
        # Python script) is exec'ed - the GPU boxes refuse an exec from a process that has touched the device; the fields stay null
        if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            self._th = None
            return self
        self._th = threading.Thread(target=self._poll, daemon=True)
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=15)
        return self.summary()

    def summary(self):
        import statistics
        sm = [x for x in self.samples if x[0] >= self._t0 + 0.25] or self.samples      # (a sample taken as the region starts shows the idle clock)
        clk = [x[1] for x in sm if x[1] is not None]
        pw = [x[2] for x in sm if x[2] is not None]
        return {"sclk_mhz": round(statistics.median(clk), 0) if clk else None, "power_w": round(statistics.median(pw), 0) if pw else None,
                "samples": len(sm)}


def gemm_roofline(pipe, traffic_profile="r05_gemm_pmc_traffic_b1.json", replay_iters=20):
    """Price the implicit-GEMM kernel against the dense fp16 MFMA peak with ALGORITHMIC flops (2*M*Cout*k*k*Cin) over
    the es_conv_gemm launches of one captured denoising step (the unit replayed 50x per image = 96 % of the image's
    FLOPs).  Two clocks, both reported:
      * `avg_launch_us` / `achieved`: in-kernel s_memrealtime stamps (min workgroup start .. max workgroup end) taken
        while the step replays as a hipGraph - per-launch resolution (conv3x3_only comes from it), but the stamp mode
        adds two global atomics per workgroup, so it reads a little high;
      * `replay`: the same launch list run by the production kernels (no stamps), GEMM launches only, captured as one
        graph and timed by HIP events around `replay_iters` replays on the launching stream - what rocprofv3's
        per-kernel average (profiles/) must agree with."""
    sampler = ClockSampler()
    res = pipe.profile_one_step(gemm_replay_iters=replay_iters, around_replay=(lambda expected_ms: sampler.start(), sampler.stop))
    clk = sampler.summary()
    tot_f = sum(m[0] for m, _ in res)
    tot_t = sum(t for _, t in res)
    f3 = sum(m[0] for m, _ in res if m[1] == 3)
    t3 = sum(t for m, t in res if m[1] == 3)
    shapes = {}
    if os.environ.get("ES_DUMP_GEMM"):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", os.environ.get("ES_DUMP_GEMM_NAME", "gemm_step_launches.json")), "w") as f:
            json.dump([dict(geom=m[3], seconds=t) for m, t in res], f)
    for (fl, k, shp, _g), t in res:
        a = shapes.setdefault(shp, [0, 0.0, 0.0])
        a[0] += 1; a[1] += t; a[2] += fl
    if os.environ.get("ES_DUMP_GEMM"):
        log("GEMM shapes of one step by total time: (M, Cout, K, stride, splitk, bn) calls total_us avg_us TFLOP/s")
        for shp, (c, t, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:45]:
            log(f"  {shp} {c} {t * 1e6:.0f} {t / c * 1e6:.1f} {fl / t / 1e12:.0f}")
    n = len(res)
    ach = tot_f / tot_t / 1e12
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", traffic_profile)
    if os.path.exists(tp):   # HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
        try:                  # gfx950 corrections applied) over the same launch list replayed stand-alone; see DESIGN.md §7
            prof = json.load(open(tp)).get("conv_gemm_kernel", {})
            alg_now = sum(m[3].get("algorithmic_bytes", 0) for m, _ in res) / max(n, 1)
            # only valid for the workload it was collected on (same launch list: count and algorithmic bytes)
            if prof.get("launches") == n and abs(prof.get("algorithmic_bytes_per_launch", 0) - alg_now) <= 0.01 * alg_now:
                traffic = prof.get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{traffic_profile} (builder-collected rocprofv3 --pmc passes, not measured in this run)"
        except Exception:
            traffic = None
    # matrix-core utilisation of the GEMM kernels by PMC (SQ_VALU_MFMA_BUSY_CYCLES per dispatch, builder-collected on the same
    # workload: tools/collect_mfma_util.sh); rides along like `traffic`, never measured inside this run
    mfma_util = None
    mp = os.path.join(ROOT, "profiles", traffic_profile.replace("gemm_pmc_traffic", "mfma_util"))
    if traffic is not None and os.path.exists(mp):
        try:
            mk = json.load(open(mp)).get("kernels", {})
            mfma_util = {k: mk[k]["mfma_util"] for k in ("conv_gemm_kernel", "conv_gemm8p_kernel", "linear_xs_kernel") if k in mk}
            mfma_util["source"] = "profiles/" + os.path.basename(mp)
        except Exception:
            mfma_util = None
    n_xs = sum(1 for m, _ in res if m[3].get("kernel") == "linear_xs")
    stamped = {"achieved": round(ach, 2), "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "avg_launch_us": round(tot_t * 1e6 / max(n, 1), 2),
               "gemm_time_per_step_ms": round(tot_t * 1e3, 3),
               "how": "in-kernel s_memrealtime stamps (min workgroup start .. max workgroup end, two extra atomics per workgroup) "
                      "on every GEMM launch of one hipGraph-replayed step: per-launch resolution, reads 5-12 % high"}
    rp = getattr(pipe, "last_gemm_replay_ms", None)
    rp3 = getattr(pipe, "last_conv3_replay_ms", None)
    out = {"bound": "mfma",
           "kernel": "conv_gemm_kernel / conv_gemm8p_kernel (implicit-GEMM conv3x3/1x1/linear) + linear_xs_kernel (row-stationary short-K linear)",
           "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": traffic, "traffic_source": traffic_src,
           "mfma_util_pmc": mfma_util, "launches_per_step": n, "linear_xs_launches": n_xs,
           "algorithmic_gflop_per_launch": round(tot_f / max(n, 1) / 1e9, 3),
           "algorithmic_bytes_per_launch": int(sum(m[3].get("algorithmic_bytes", 0) for m, _ in res) / max(n, 1)),
           "conv3x3_only": ({"achieved": round(f3 / (rp3 * 1e-3) / 1e12, 2), "frac": round(f3 / (rp3 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                             "launches": sum(1 for m, _ in res if m[1] == 3), "time_per_step_ms": round(rp3, 3),
                             "how": "the 3x3 convolutions of one step alone (with their split-K reduces) as the production "
                                    "kernels run them: one hipGraph, HIP events around the replays",
                             "stamped_achieved": round(f3 / t3 / 1e12, 2) if t3 else None} if rp3 else
                            {"achieved": round(f3 / t3 / 1e12, 2) if t3 else None,
                             "frac": round(f3 / t3 / 1e12 / MFMA_PEAK_TFLOPS, 4) if t3 else None,
                             "launches": sum(1 for m, _ in res if m[1] == 3), "how": "from the stamps"}),
           "stamped": stamped}
    if rp:
        # the headline pair: production kernels, HIP events on the launching stream
        out.update({"achieved": round(tot_f / (rp * 1e-3) / 1e12, 2), "frac": round(tot_f / (rp * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "avg_launch_us": round(rp * 1e3 / max(n, 1), 2), "gemm_time_per_step_ms": round(rp, 3),
                    "how": f"algorithmic FLOPs of the {n} GEMM launches of one denoising step / their time as the production "
                           f"kernels run them (no stamps): the launches (+ their split-K reduces) captured as one hipGraph, HIP "
                           f"events around {replay_iters} replays on the launching stream"})
    else:
        out.update({k: stamped[k] for k in ("achieved", "frac", "avg_launch_us", "gemm_time_per_step_ms", "how")})
    # the ceiling the kernels actually ran under: the clock and package power held DURING the event-timed replays (`frac` itself
    # stays against the 2.4 GHz dense peak)
    out["sclk_mhz"], out["power_w"], out["clock_samples"] = clk["sclk_mhz"], clk["power_w"], clk["samples"]
    out["frac_at_sustained_clock"] = (round(out["achieved"] / (MFMA_PEAK_TFLOPS * clk["sclk_mhz"] / 2400.0), 4)
                                      if clk["sclk_mhz"] and out.get("achieved") else None)
    # (checked against the chip's own counters: one idle wave beside the replays reads the same clock as rocm-smi; the workgroups of the
    #  256 x 320 convolution tile themselves run at 1.79-1.90 GHz, single launch or back to back - profiles/r05_clock_in_kernel.txt)
    out["clock_source"] = "rocm-smi --showclocks --showpower polled while the GEMM launch list replays (medians); peak clock 2400 MHz"
    # per family: the 3x3 convolutions against everything else (1x1 convolutions, linear layers, linear_xs), from the stamps
    fam = {}
    for (fl, k, _shp, _g), t in res:
        a = fam.setdefault("k3" if k == 3 else "k1", [0, 0.0, 0.0])
        a[0] += 1; a[1] += t; a[2] += fl
    out["families_stamped"] = {k: {"launches": c, "ms": round(t * 1e3, 3), "tflops": round(fl / t / 1e12, 1) if t else None}
                               for k, (c, t, fl) in sorted(fam.items())}
    return out


def cpu_baseline_and_parity(pipe, ws, ucfg, B, steps_total, tiny):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference diffusers pipeline) on this host's cores: full
    6-cond denoising steps at the benchmark batch (one untimed warm-up step, then the timed sample), extrapolated to the
    50-step loop.  The FIRST oracle step doubles as the checker of the benchmarked configuration: the HIP step (same
    grouped launches the timed region replays) runs on the same inputs and weights and its error is reported."""
    from oracle import sd15_oracle as O
    from tests.helpers import oracle_nets
    cores = host_cores()
    torch.set_num_threads(cores)
    # fp16-rounded weights on both sides: the HIP path stores fp16, the oracle computes with the same values in fp32
    cws = {k: {kk: vv.half().float().cpu() for kk, vv in v.items()} for k, v in ws.items() if k != "vae"}
    g = torch.Generator().manual_seed(1)
    N, s, c0 = 2 * B, ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(N, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(N, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    nets = oracle_nets(cws, ucfg)
    n_sample = 1 if tiny else 2
    with torch.no_grad():
        ref = O.denoise_step(cws["unet"], ucfg, cws["fusion"], nets, x, 501, ehs, conds, [1.0] * 6)   # warm-up + checker
        t0 = time.time()
        for i in range(n_sample):
            O.denoise_step(cws["unet"], ucfg, cws["fusion"], nets, x, 481 - 20 * i, ehs, conds, [1.0] * 6)
        dt = (time.time() - t0) / n_sample
    dev = pipe.device
    runner = pipe._runner
    mode_ok = runner.mode == "grouped" and runner._grouped_encoder(N).groupable(
        (s >> (len(ucfg.block_out_channels) - 1)) ** 2)
    got = runner.step_nchw(x.to(dev), 501, ehs.to(dev), [c.to(dev) for c in conds], [1.0] * 6).float().cpu()
    err = (got - ref).abs()
    parity = {"what": f"one HIP denoising step (6 ControlNets + fusion + UNet, CFG batch {N}, "
                      f"{'grouped lockstep launches' if mode_ok else 'per-net launches'}) vs the CPU oracle on identical "
                      "inputs and fp16-rounded weights",
              "max_abs": round(float(err.max()), 6), "rel_to_max": round(float(err.max() / ref.abs().max()), 6),
              "rms_rel": round(float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), 6),
              "bar_max_abs": 2e-2, "ok": bool(float(err.max()) <= 2e-2)}
    base = {"value": round(B / (dt * steps_total), 6), "unit": "images/s", "cores": cores, "kind": "port",
            "cpu": cpu_model(),
            "sample": f"{n_sample} of {steps_total} denoising steps after one untimed warm-up step (6 ControlNets + fusion "
                      f"+ UNet, CFG batch {N}) = {dt:.1f} s each on {cores} threads, x{steps_total}; condition embedding "
                      "and VAE decode excluded"}
    return base, parity


def call_phases(pipe, one, sync=None):
    """HIP-event times of the phases of ONE pipeline call (pipeline.collect_timing): condition embedding / per-call preparation /
    the denoising loop / VAE decode + post-processing.  One extra call after the timed region (events + a device sync per call).
    `one` must be rank-local: no collective."""
    pipe.collect_timing = True
    try:
        one()
        (sync or torch.cuda.synchronize)()
        t = dict(getattr(pipe, "timing", {}) or {})
    finally:
        pipe.collect_timing = False
    return {k: round(float(v), 3) for k, v in t.items()}


def vae_decode_accounting(pipe, B, decode_ms):
    """Algorithmic bytes of one VAE decode (every activation read once and written once, op by op as the decode is
    launched: convolutions with their residual / shortcut sources, GroupNorm, the mid-block attention; weights counted apart)
    and the rate the measured decode phase moves them at, against the 8 TB/s HBM3E peak."""
    from edgestyle_amd import ops
    acc = {"act": 0, "w": 0, "launches": 0, "flops": 0.0}
    real = (ops.conv_gemm, ops.group_norm, ops.attention)

    def nb(t):
        return 0 if t is None else t.numel() * t.element_size()

    def conv(x, pw, **kw):
        out = real[0](x, pw, **kw)
        pws = list(pw) if isinstance(pw, (list, tuple)) else [pw]
        acc["act"] += nb(x) + nb(kw.get("x2")) + nb(kw.get("residual")) + sum(nb(t) for t in (kw.get("tail") or ()) if t is not None) + nb(out)
        acc["w"] += sum(nb(q.w) for q in pws)
        acc["flops"] += 2.0 * out.shape[0] * out.shape[1] * out.shape[2] * pws[0].cout * pws[0].kpad
        acc["launches"] += 1
        return out

    def gn(x, *a, **kw):
        out = real[1](x, *a, **kw)
        acc["act"] += nb(x) + nb(kw.get("x2")) + nb(out)
        acc["launches"] += 1
        return out

    def attn(q, k, v, *a, **kw):
        out = real[2](q, k, v, *a, **kw)
        acc["act"] += nb(q) + nb(k) + nb(v) + nb(out)
        acc["launches"] += 1
        return out
    s = pipe.unet.cfg.sample_size
    z = torch.zeros(B, s, s, pipe.vae.engine.lat_pad, dtype=pipe.dtype, device=pipe.device)
    ops.conv_gemm, ops.group_norm, ops.attention = conv, gn, attn
    try:
        pipe.vae.decode_nhwc(z, unscaled_latents=True)
        torch.cuda.synchronize()
    finally:
        ops.conv_gemm, ops.group_norm, ops.attention = real
    out = {"ms": round(decode_ms, 3) if decode_ms else None, "images": B, "launches": acc["launches"],
           "algorithmic_bytes": int(acc["act"]), "weight_bytes": int(acc["w"]), "algorithmic_gflop": round(acc["flops"] / 1e9, 1),
           "peak": 8.0, "unit": "TB/s",
           "how": "bytes = every activation tensor of the decode read once and written once, op by op (SURVEY 8d); ms = the decode phase "
                  "of one pipeline call (graph replay + [0,1] post-processing + the copy out of the graph's buffer), HIP events"}
    if decode_ms:
        out["achieved"] = round(acc["act"] / (decode_ms * 1e-3) / 1e12, 3)
        out["frac"] = round(out["achieved"] / 8.0, 4)
        out["tflops"] = round(acc["flops"] / (decode_ms * 1e-3) / 1e12, 1)
    return out


def native_abi_leg(pipe, ws, ucfg, vcfg, B, T, dtype, dev_index, lat, pe, ne, imgs, cn, want_img, iters=3, graphs=2):
    """The benchmarked request served through the C ABI alone: es_load_weights builds the context from the raw
    state dicts (no model walk in Python), then es_prepare_conds + es_denoise_loop + es_vae_decode on raw device pointers,
    the whole loop as one hipGraph.  Reported beside the headline value (same kernels, same launch lists: what changes is who
    issues them); `bitwise_equal_to_pipeline` compares its image with the timed region's."""
    from edgestyle_amd.native import NativeContext
    dev = torch.device("cuda", dev_index)
    t0 = time.perf_counter()
    nat = NativeContext(ws, ucfg, vcfg, batch_size=B, guidance=True, num_inference_steps=T, dtype=dtype, device=dev_index)
    t_build = time.perf_counter() - t0          # (the bench's weights live on the GPU: the builder reads them back itself)
    try:
        nat.set_alphas_cumprod(pipe.scheduler.alphas_cumprod)
        nat.set_options(use_graphs=graphs)
        im = [i.to(dev, torch.float32).repeat_interleave(B, dim=0).contiguous() if i.shape[0] == 1 else i.to(dev, torch.float32) for i in imgs]
        nz = [None if z is None else z.to(dev, torch.float32).contiguous() for z in cn]
        ehs = torch.cat([ne, pe]).to(dev, dtype).contiguous()
        x0 = lat.permute(0, 2, 3, 1).contiguous().to(dev, torch.float32)
        ts = pipe.scheduler.set_timesteps(T).tolist()
        out = torch.empty((B, 3, ucfg.sample_size * vcfg.scale, ucfg.sample_size * vcfg.scale), dtype=torch.float32, device=dev)

        def one():
            x = x0.clone()
            nat.prepare_conds(im, nz)
            nat.denoise_loop(x, ehs, 7.5, ts)
            nat.vae_decode(x, out)
        one()
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / iters
        return {"workload": "the same request through the C ABI alone: context built by es_load_weights from raw state dicts, "
                            "es_prepare_conds + es_denoise_loop (" + {2: "whole loop as one hipGraph", 1: "one hipGraph per plan", 0: "launch by launch"}[graphs]
                            + ") + es_vae_decode on raw device pointers",
                "value": round(B / t, 4), "unit": "images/s", "ms_per_step": round(t * 1e3, 1), "steps": iters, "warmup": 2,
                "build_s": round(t_build, 1),
                "arena_gib": round(nat.lib.es_ctx_arena_bytes(nat.ctx) / 2 ** 30, 2),
                "calls_per_denoising_step": nat.plan_size(2),
                "bitwise_equal_to_pipeline": bool(torch.equal(out, want_img.to(dev, torch.float32)[:B]))}
    finally:
        nat.close()


def self_launch(n: int, argv) -> int:
    """Start `n` ranks of this script (one per GPU) as CHILD processes of a parent that has not initialised the GPU, the way
    the driver's own command does it: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1
    --master-port P bench.py <argv>`.  stdout (rank 0's one JSON line) and stderr pass straight through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:9])} ...")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None,
                    help="images per GPU per step; default 1 on one GPU (BASELINE configs[1]) and 8 per GPU when --gpus > 1 "
                         "(BASELINE configs[3]: 64 independent try-ons over 8 GPUs)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--resolution", type=int, default=512, help="512 (the metric) or 768 (BASELINE config 5 stress)")
    ap.add_argument("--tiny", action="store_true", help="width-reduced config (plumbing check only, not the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches (needed under rocprofv3 --pmc, which cannot "
                                                            "collect counters through a hipGraph replay)")
    ap.add_argument("--no-stress-mode", action="store_true",
                    help="skip the extra 768x768 bf16 batch-4 (BASELINE configs[4]) measurement")
    ap.add_argument("--no-throughput-mode", action="store_true",
                    help="skip the extra batch-8 (BASELINE configs[2]) measurement reported beside the headline value")
    ap.add_argument("--throughput-sweep", default="16",
                    help="more batch sizes measured beside configs[2] (2 timed calls each, comma-separated; \"\": none).  Above 11 try-ons "
                         "per call some operands outgrow the kernels' 32-bit buffer offsets (2 GiB); es_conv_gemm / es_linear_xs run those "
                         "launches as runs of whole samples (include/edgestyle_hip.h, es_set_operand_limit)")
    ap.add_argument("--no-native-abi", action="store_true",
                    help="skip the extra leg that serves the same request through the C ABI alone (es_load_weights context)")
    ap.add_argument("--native-graphs", type=int, default=2, choices=(0, 1, 2),
                    help="how the native-ABI leg replays its plans: 2 = the whole loop as ONE hipGraph (default), 1 = one hipGraph per "
                         "plan (a step per launch), 0 = launch by launch.  Under rocprofv3 use 1: the profiler's queue interception "
                         "cannot take one graph launch of ~14 k kernel nodes of a batch-8 loop (DESIGN.md, round-4 note)")
    ap.add_argument("--fake-pipeline", action="store_true",
                    help="CPU rehearsal of the multi-rank control flow (gloo, no GPU, no kernels): NOT a measurement")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher in front of us.  This parent never touches a GPU (no torch.cuda call,
        # no HIP call); it starts one fresh rank per GPU through torch.distributed.run, forwards rank 0's JSON line and
        # exits with the children's code.  Under an existing launcher (WORLD_SIZE set) the code below runs as before.
        sys.exit(self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv)))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal knobs (a 1-GPU box cannot host two RCCL ranks): ES_DIST_BACKEND=gloo ES_FORCE_DEVICE=0 runs the
    # multi-rank control flow of this script with every rank on one GPU; the driver's runs use neither
    dev_index = int(os.environ.get("ES_FORCE_DEVICE", local_rank))
    fake = args.fake_pipeline
    if not fake:
        torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if fake else os.environ.get("ES_DIST_BACKEND", "nccl"))   # nccl == RCCL over xGMI
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cpu") if fake else torch.device("cuda", dev_index)
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    if args.batch is None:
        args.batch = 1 if world == 1 else 8           # BASELINE configs[1] on one GPU, configs[3] (8 try-ons per GPU) on N

    import faulthandler
    faulthandler.dump_traceback_later(600, repeat=True, file=sys.stderr)     # where are we, if something stalls
    log(f"rank {rank}/{world}: building weights + packing on {device}")
    if fake:
        from edgestyle_amd import config as C
        ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
        pipe, ws = _FakePipeline(ucfg, vcfg), None
        args.no_roofline = args.no_cpu_baseline = args.no_throughput_mode = args.no_stress_mode = args.no_native_abi = True
    else:
        pipe, ws, ucfg, vcfg = build_pipeline(device, dtype, tiny=args.tiny, resolution=args.resolution)
    B = args.batch
    if args.no_graph:
        pipe.use_graph = False
    from edgestyle_amd.dist import gather_images
    # rank r serves the global images [r * B, (r + 1) * B): per-image seeds, world-size independent
    lat, pe, ne, imgs, cn = make_inputs(ucfg, vcfg, B, device, seed=42, first_index=rank * B)

    out = {}

    def run_local():
        return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
                    num_inference_steps=args.ddim_steps, output_type="pt", cond_noise=cn)

    def one():
        out["img"] = gather_images(run_local().images, world)          # the single RCCL gather of the path

    def sync():
        if not fake:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        sync()

    # (the first call captures the step graph, the second - same configuration - the whole-loop graph: at least two
    #  untimed calls, so that no capture falls into the timed region whatever W was asked for)
    for i in range(max(args.warmup, 2) if args.steps else args.warmup):
        one()
        sync()
        log(f"warmup {i} done")
    barrier()
    sampler = ClockSampler().start() if (rank == 0 and not fake) else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    barrier()
    dt = time.perf_counter() - t0
    clock = sampler.stop() if sampler is not None else None
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)       # MAX over ranks
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    log(f"timed region done: {dt:.3f} s for {args.steps} step(s)")
    if rank == 0:
        img = out["img"]
        assert img.shape[0] == B * world and bool(torch.isfinite(img).all()), "non-finite output image"
        line = {
            "metric": "512x512 try-on images/sec @ 50 DDIM steps (6-cond ControlNet)",
            "value": round(world * B * args.steps / dt, 4), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": max(args.warmup, 2) if args.steps else args.warmup,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16",
            "data": "synthetic (seeded random-init SD1.5-shaped weights, random conds/latents/prompt embeds, seed 42)",
            "config": {"workload": "", "images_per_gpu": B, "ddim_steps": args.ddim_steps, "parallelism": f"dp{world} (independent images, one RCCL gather)"},
        }
        if world > 1:
            line["rccl_ranks"] = torch.distributed.get_world_size()
            line["dist_backend"] = torch.distributed.get_backend()         # "nccl" is RCCL on ROCm
        cfg_idx = 4 if args.resolution != 512 else (3 if world > 1 else (1 if B == 1 else 2))
        what = f"{world}xMI355X data-parallel, {world * B} independent try-ons ({B} per GPU, per-image seeds), one RCCL gather of the decoded images; " if world > 1 else ""
        line["config"]["workload"] = ("TINY plumbing config" if args.tiny else
                                      f"BASELINE configs[{cfg_idx}]: {what}full 6-cond edgestyle_multicontrolnet + controllora, "
                                      f"{args.resolution}x{args.resolution}, {args.ddim_steps} DDIM steps, CFG 7.5, batch={B}/GPU, "
                                      "hipGraph-captured step, cond embedding + VAE decode included")
        if fake:
            line["data"] = "FAKE pipeline (CPU rehearsal of the multi-rank control flow: not a measurement)"
            line["config"]["workload"] = "REHEARSAL (no kernels run): " + line["config"]["workload"]
            line["rehearsal"] = {"gathered_image_means": [round(float(v), 6) for v in img[:, 0, 0, 0]],
                                 "expected": [round(float(make_inputs(ucfg, vcfg, 1, device, 42, j)[0].mean()), 6) for j in range(world * B)]}
        if clock is not None:
            line["clock"] = dict(clock, source="rocm-smi --showclocks --showpower polled during the timed region (medians); peak clock 2400 MHz")
        if args.steps:
            # `run_local`, not `one`: everything from here on runs on rank 0 ALONE - a collective here waits for ranks that are already
            # at the closing barrier (found by the two-rank rehearsal on a GPU box; the CPU rehearsal takes this path too)
            phases = call_phases(pipe, run_local, sync)
        if not fake and args.steps:
            line["phases_ms"] = dict(phases, how="HIP events at the phase boundaries of ONE extra pipeline call after "
                                     "the timed region: condition embedding / per-call preparation / denoising loop / VAE decode + post-processing")
        if not args.no_roofline:
            # enough replays of the step's GEMM launch list for the clock sampler to see the sustained state (~1.5 s)
            est = max(dt / max(args.steps, 1) / max(args.ddim_steps, 1) * 0.7, 1e-4)
            line["roofline"] = gemm_roofline(pipe, replay_iters=int(min(400, max(20, 1.5 / est))))
            log("roofline leg done")
        if not args.no_cpu_baseline and world == 1:
            # before the batch-8 leg: the parity step must run the very configuration the timed region replayed
            line["cpu_baseline"], line["parity"] = cpu_baseline_and_parity(pipe, ws, ucfg, B, args.ddim_steps, args.tiny)
            log(f"cpu baseline + parity done: {line['parity']}")
        if not args.no_native_abi and world == 1 and not args.tiny and args.resolution == 512 and not args.no_graph:
            line["native_abi"] = native_abi_leg(pipe, ws, ucfg, vcfg, B, args.ddim_steps, dtype, dev_index, lat, pe, ne, imgs, cn, img,
                                                graphs=args.native_graphs)
            log(f"native ABI leg done: {line['native_abi']}")
        if not args.no_throughput_mode and world == 1 and B != 8 and not args.tiny and args.resolution == 512:
            # BASELINE configs[2]: same path, 8 images per step (hipGraph-captured, throughput mode); reported beside
            # the headline value, never instead of it
            lat8, pe8, ne8, imgs8, cn8 = make_inputs(ucfg, vcfg, 8, device, seed=42)

            def one8():
                return pipe(prompt_embeds=pe8, negative_prompt_embeds=ne8, image=imgs8, latents=lat8, guidance_scale=7.5,
                            num_inference_steps=args.ddim_steps, output_type="pt", cond_noise=cn8).images
            one8()                                         # capture + warm clocks on this shape
            one8()
            torch.cuda.synchronize()
            n8 = 5                                        # (round 3: 2 - the +-4 % box spread made round-over-round deltas noise)
            t8 = time.perf_counter()
            for _ in range(n8):
                img8 = one8()
            torch.cuda.synchronize()
            t8 = (time.perf_counter() - t8) / n8
            assert bool(torch.isfinite(img8).all())
            line["throughput_mode"] = {"workload": "BASELINE configs[2]: same path, batch=8 per step", "value": round(8 / t8, 4),
                                       "unit": "images/s", "ms_per_step": round(t8 * 1e3, 1), "steps": n8, "warmup": 2}
            line["throughput_mode"]["phases_ms"] = call_phases(pipe, one8)
            if not args.no_roofline:
                r8 = gemm_roofline(pipe, traffic_profile="r05_gemm_pmc_traffic_b8.json", replay_iters=int(min(100, max(5, 1.5 / (t8 / args.ddim_steps * 0.65)))))
                line["throughput_mode"]["roofline"] = {k: r8[k] for k in ROOFLINE_KEYS if k in r8}
            log(f"throughput mode (batch 8): {8 / t8:.3f} images/s")
            if not args.no_native_abi and not args.no_graph:
                n8abi = native_abi_leg(pipe, ws, ucfg, vcfg, 8, args.ddim_steps, dtype, dev_index, lat8, pe8, ne8, imgs8, cn8, img8, iters=3,
                                       graphs=args.native_graphs)
                line["throughput_mode"]["native_abi"] = {k: n8abi[k] for k in ("value", "unit", "ms_per_step", "build_s", "arena_gib",
                                                                               "bitwise_equal_to_pipeline")}
                log(f"throughput mode through the C ABI alone: {n8abi['value']:.3f} images/s")
            del lat8, pe8, ne8, imgs8, cn8, img8
            # beside - never instead of - configs[2]: larger batches (2 timed calls each), so that the launch-granularity share of what
            # separates batch 8 from the conv path's ceiling is on record
            sweep = []
            for bs in [int(v) for v in args.throughput_sweep.split(",") if v.strip()]:
                try:
                    latS, peS, neS, imgsS, cnS = make_inputs(ucfg, vcfg, bs, device, seed=42)

                    def oneS():
                        return pipe(prompt_embeds=peS, negative_prompt_embeds=neS, image=imgsS, latents=latS, guidance_scale=7.5,
                                    num_inference_steps=args.ddim_steps, output_type="pt", cond_noise=cnS).images
                    oneS(); oneS()
                    torch.cuda.synchronize()
                    tS = time.perf_counter()
                    for _ in range(2):
                        imgS = oneS()
                    torch.cuda.synchronize()
                    tS = (time.perf_counter() - tS) / 2
                    assert bool(torch.isfinite(imgS).all())
                    sweep.append({"batch": bs, "value": round(bs / tS, 4), "unit": "images/s", "ms_per_step": round(tS * 1e3, 1), "steps": 2, "warmup": 2})
                    log(f"throughput sweep, batch {bs}: {bs / tS:.3f} images/s")
                    for k in [k for k in pipe._loops if k[0] == bs]:
                        del pipe._loops[k]                      # that batch size's graphs and buffers
                    del latS, peS, neS, imgsS, cnS, imgS
                    torch.cuda.empty_cache()
                except Exception as e:                          # a sweep point must never cost the line
                    sweep.append({"batch": bs, "error": repr(e)[:200]})
            if sweep:
                line["throughput_sweep"] = sweep
        if not args.no_stress_mode and world == 1 and not args.tiny and args.resolution == 512 and dtype == torch.float16:
            # BASELINE configs[4]: bf16, 768x768, batch 4 (outside the reference's own domain): three warmed,
            # timed pipeline calls on a second pipeline object; the 512 one is released first
            pipe._loops.clear()
            pipe._runner = None
            del pipe
            torch.cuda.empty_cache()
            pipe5, _ws5, ucfg5, vcfg5 = build_pipeline(device, torch.bfloat16, resolution=768)
            lat5, pe5, ne5, imgs5, cn5 = make_inputs(ucfg5, vcfg5, 4, device, seed=42)

            def one5():
                return pipe5(prompt_embeds=pe5, negative_prompt_embeds=ne5, image=imgs5, latents=lat5, guidance_scale=7.5,
                             num_inference_steps=args.ddim_steps, output_type="pt", cond_noise=cn5).images
            one5()
            one5()                                   # (second untimed call: captures the whole-loop graph)
            torch.cuda.synchronize()
            n5 = 3
            t5 = time.perf_counter()
            for _ in range(n5):
                img5 = one5()
            torch.cuda.synchronize()
            t5 = (time.perf_counter() - t5) / n5
            assert img5.shape == (4, 3, 768, 768) and bool(torch.isfinite(img5).all())
            line["stress_mode"] = {"workload": "BASELINE configs[4]: bf16, 768x768, 50 DDIM steps, batch=4, VAE decode included",
                                   "value": round(4 / t5, 4), "unit": "images/s", "ms_per_step": round(t5 * 1e3, 1),
                                   "steps": n5, "warmup": 2, "dtype": "bf16"}
            log(f"stress mode (768x768 bf16 batch 4): {4 / t5:.3f} images/s")
            ph5 = call_phases(pipe5, one5)
            line["stress_mode"]["phases_ms"] = ph5
            # the HBM-bound piece of this configuration: the VAE decode at 768x768 (128 channels at 768x768 = 151 MB per tensor and image)
            line["stress_mode"]["vae_decode"] = vae_decode_accounting(pipe5, 4, ph5.get("decode"))
            if not args.no_roofline:
                r5 = gemm_roofline(pipe5, traffic_profile="r05_gemm_pmc_traffic_768_b4.json",
                                   replay_iters=int(min(60, max(5, 1.5 / (t5 / args.ddim_steps * 0.65)))))
                line["stress_mode"]["roofline"] = {k: r5[k] for k in ROOFLINE_KEYS if k in r5}
            log(f"stress mode accounting done: decode {line['stress_mode']['vae_decode']}")
        faulthandler.cancel_dump_traceback_later()
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""DDIM scheduler with SD1.5's scheduler_config semantics (the [D] code behind model/edgestyle_pipeline.py:382-385 and
:520-522): scaled_linear betas 0.00085 -> 0.012 over 1000 train steps, `leading` spacing with steps_offset=1,
set_alpha_to_one=False, clip_sample=False, epsilon prediction, eta=0.

The arithmetic of `step` runs on the GPU inside es_cfg_ddim_step (fused with the CFG combine); this class owns
the host-side schedule and exports the per-step coefficient table that kernel indexes with a device step counter,
so the whole 50-step loop replays as one captured hipGraph per step with no host-side scalar updates.
"""
import torch


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]        # set_alpha_to_one=False
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config=None, **kw):
        cfg = dict(config) if isinstance(config, dict) else {}
        cfg.update(kw)
        keep = ("num_train_timesteps", "beta_start", "beta_end", "steps_offset")
        return cls(**{k: v for k, v in cfg.items() if k in keep})

    @property
    def config(self):
        """What `OtherScheduler.from_config(pipeline.scheduler.config)` (TT:273) reads: the SD1.5 scheduler_config."""
        return dict(num_train_timesteps=self.num_train_timesteps, beta_start=0.00085, beta_end=0.012,
                    beta_schedule="scaled_linear", steps_offset=self.steps_offset, timestep_spacing="leading",
                    set_alpha_to_one=False, clip_sample=False, prediction_type="epsilon")

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps) * ratio).round().flip(0).long() + self.steps_offset
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coef_table(self) -> torch.Tensor:
        """[steps, 4] fp32: sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev) for each timestep in order
        (a_prev: alphas_cumprod at t - num_train // steps = the next timestep of the list; alphas_cumprod[0] after the
        last).  Computed by the library's host-side helper es_ddim_coef_table - the very code es_denoise_loop uses - so the
        Python loop and the native loop see the same bits."""
        import ctypes as C
        from . import lib as L
        ts = self.timesteps.to(torch.float32).contiguous()
        ac = self.alphas_cumprod.to(torch.float32).contiguous()
        out = torch.empty((ts.numel(), 4), dtype=torch.float32)
        fp = C.POINTER(C.c_float)
        L.check(L.load().es_ddim_coef_table(C.cast(ac.data_ptr(), fp), ac.numel(), C.cast(ts.data_ptr(), fp), ts.numel(),
                                            C.cast(out.data_ptr(), fp)), "es_ddim_coef_table")
        return out


class UniPCMultistepScheduler:
    """The scheduler the reference's callers actually assign (`UniPCMultistepScheduler.from_config(
    pipeline.scheduler.config)`: test_text2image_pretrained_openpose.py:273, app.py:118, inference.py:538), with the
    SD1.5 scheduler config it inherits: scaled_linear betas 0.00085 -> 0.012, 1000 train steps, steps_offset 1,
    `leading` spacing; UniPC defaults solver_order 2, `bh2`, predict_x0, lower_order_final, epsilon prediction.

    Like DDIM here, the tensor arithmetic runs in one fused GPU kernel (es_cfg_unipc_step).  Every update of the
    multistep predictor/corrector is linear in {last_sample, m0, m1, x0}, so this class only derives the per-step
    scalar coefficients (float64 on the host) and exports them as a device table indexed by the step counter.
    """
    order = 1                     # the pipeline's progress accounting (PL:430, PL:537-539) sees a first-order loop
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 steps_offset: int = 1, solver_order: int = 2, timestep_spacing: str = "leading", **unused):
        if solver_order not in (1, 2):
            raise NotImplementedError("solver_order 1 or 2 (the reference uses the default, 2)")
        if timestep_spacing not in ("leading", "linspace"):
            raise ValueError(f"unsupported timestep_spacing {timestep_spacing}")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.solver_order, self.timestep_spacing = solver_order, timestep_spacing
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config=None, **kw):
        cfg = dict(config) if isinstance(config, dict) else {}
        cfg.update(kw)
        keep = ("num_train_timesteps", "beta_start", "beta_end", "steps_offset", "solver_order", "timestep_spacing")
        return cls(**{k: v for k, v in cfg.items() if k in keep})

    @property
    def config(self):
        return dict(num_train_timesteps=self.num_train_timesteps, steps_offset=self.steps_offset,
                    solver_order=self.solver_order, timestep_spacing=self.timestep_spacing)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        import numpy as np
        n, N = num_inference_steps, self.num_train_timesteps
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, N - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        else:
            ts = (np.arange(0, n + 1) * (N // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        ac = self.alphas_cumprod.numpy()
        sig = ((1 - ac) / ac) ** 0.5
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sigmas, [((1 - ac[0]) / ac[0]) ** 0.5]])
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        return self.timesteps

    # -- coefficient derivation (float64 scalars) ---------------------------------------------------------------
    def _asl(self, idx):
        import math
        s = float(self.sigmas[idx])
        a = 1.0 / math.sqrt(s * s + 1.0)
        return a, s * a, math.log(a) - math.log(s * a)

    @staticmethod
    def _rhos(rks, hh, order, corrector):
        import math
        import numpy as np
        h_phi_1 = math.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        B_h = math.expm1(hh)
        fact, R, b = 1, [], []
        for i in range(1, order + 1):
            R.append([rk ** (i - 1) for rk in rks])
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / fact
        R, b = np.array(R, dtype=np.float64), np.array(b, dtype=np.float64)
        if corrector:
            rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
        else:
            rhos = np.array([0.5]) if order == 2 else (np.linalg.solve(R[:-1, :-1], b[:-1]) if order > 2 else np.array([]))
        return rhos, h_phi_1, B_h

    def coef_table(self) -> torch.Tensor:
        """[steps, 12] fp32: alpha_t, sigma_t, use_c, c_last, c_m0, c_m1, c_x0, p_x, p_x0, p_m0, 0, 0."""
        T = len(self.timesteps)
        rows, lower_order_nums, this_order = [], 0, 1
        for i in range(T):
            a_i, s_i, l_i = self._asl(i)
            row = [a_i, s_i, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
            if i > 0:                                   # corrector with the order of the previous predictor
                order = this_order
                a_0, s_0, l_0 = self._asl(i - 1)
                h = l_i - l_0
                rks = [(self._asl(i - (k + 1))[2] - l_0) / h for k in range(1, order)] + [1.0]
                rhos, h_phi_1, B_h = self._rhos(rks, -h, order, True)
                A, Hh, Bc = s_i / s_0, a_i * h_phi_1, a_i * B_h
                if order == 1:
                    row[2:7] = [1.0, A, -Hh + Bc * rhos[0], 0.0, -Bc * rhos[0]]
                else:
                    row[2:7] = [1.0, A, -Hh + Bc * (rhos[0] / rks[0] + rhos[1]), -Bc * rhos[0] / rks[0], -Bc * rhos[1]]
            this_order = min(min(self.solver_order, T - i), lower_order_nums + 1)      # lower_order_final + warm-up
            a_t, s_t, l_t = self._asl(i + 1)
            h = l_t - l_i
            rks = [(self._asl(i - k)[2] - l_i) / h for k in range(1, this_order)] + [1.0]
            rhos, h_phi_1, B_h = self._rhos(rks, -h, this_order, False)
            A, Hh, Bp = s_t / s_i, a_t * h_phi_1, a_t * B_h
            if this_order == 1:
                row[7:10] = [A, -Hh, 0.0]
            else:
                row[7:10] = [A, -Hh + Bp * rhos[0] / rks[0], -Bp * rhos[0] / rks[0]]
            if lower_order_nums < self.solver_order:
                lower_order_nums += 1
            rows.append(row)
        return torch.tensor(rows, dtype=torch.float32).contiguous()

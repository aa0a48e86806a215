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
        return cls(**kw)

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
        """[steps, 4] fp32: sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev) for each timestep in order."""
        rows = []
        ratio = self.num_train_timesteps // self.num_inference_steps
        for t in self.timesteps.tolist():
            prev = t - ratio
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
            rows.append(torch.stack([a_t.sqrt(), (1 - a_t).sqrt(), a_p.sqrt(), (1 - a_p).sqrt()]))
        return torch.stack(rows).float().contiguous()

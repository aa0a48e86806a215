"""Try-on serving loop: the step AFTER the denoising path.

The reference serves one request at a time from a Gradio callback (`try_on`, app.py:151-182: six conditioning
images + a prompt -> `pipeline(...).images[0]`, generator re-seeded to 42 per request, app.py:162).  On an MI355X one
request leaves most of the chip idle in the UNet decoder (batch 8 delivers ~1.5x the images/s of batch 1 on the same
GPU: the current pair is in the bench line, `value` against `throughput_mode.value`), so the service here batches: requests that agree on (steps, guidance scale, control window, size) and arrive
within `max_wait_s` of each other run as ONE pipeline call of up to `max_batch` images, each with its own latents
drawn from its own seed — a request's image does not depend on what it was batched with (tests/test_host_cpu.py).

Host logic only (a queue, a worker thread, futures); the Gradio UI, the YOLO / OpenPose / SAM preprocessing and the
CLIP prompt picker of app.py:125-149 are out of scope.  Multi-GPU: one `TryOnService` per rank, the
front end deals requests round-robin (independent images) — no collective on the request path.
"""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch


@dataclass
class TryOnRequest:
    """One `try_on` call (app.py:151-182).  `images`: the six conditioning tensors [1,3,H,W] in the reference's order
    (agnostic, subject pose, clothes 1, pose 1, clothes 2, pose 2; images in [-1,1], poses in [0,1], TT:29-48)."""
    images: Sequence[torch.Tensor]
    prompt_embeds: torch.Tensor                 # [1,77,768]  (or pass `prompt` and let the pipeline's CLIP encode it)
    negative_prompt_embeds: torch.Tensor
    guidance_scale: float = 7.5
    num_inference_steps: int = 50
    seed: int = 42                              # app.py:162 re-seeds every request with 42
    control_guidance_start: float = 0.0
    control_guidance_end: float = 1.0
    future: Future = field(default_factory=Future, repr=False)
    t_submit: float = field(default_factory=time.perf_counter, repr=False)

    def key(self) -> Tuple:
        """Requests with equal keys can share a pipeline call (scalars of the call, tensor geometry)."""
        return (self.num_inference_steps, float(self.guidance_scale), float(self.control_guidance_start),
                float(self.control_guidance_end), tuple(tuple(i.shape[1:]) for i in self.images),
                tuple(self.prompt_embeds.shape[1:]))


def latents_for(seed: int, channels: int, h: int, w: int) -> torch.Tensor:
    """Initial latents of ONE request from its own CPU generator (the reference draws on the CPU generator too,
    PL:585-627): batching cannot change a request's noise."""
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn((1, channels, h, w), generator=g, dtype=torch.float32)


class TryOnService:
    """Batches `TryOnRequest`s into pipeline calls on a worker thread.

    `pipeline(**kwargs) -> obj with .images [B,3,H,W]` is the `StableDiffusionControlNetPipeline` call surface
    (prompt_embeds / negative_prompt_embeds / image / latents / guidance_scale / num_inference_steps /
    control_guidance_start / control_guidance_end / output_type="pt").  Allowed batch sizes are rounded DOWN to
    `batch_sizes` so that the pipeline only ever sees shapes whose hipGraph it has already captured."""

    def __init__(self, pipeline: Callable[..., Any], max_batch: int = 8, max_wait_s: float = 0.02,
                 batch_sizes: Sequence[int] = (1, 2, 4, 8), latent_channels: int = 4, vae_scale: int = 8,
                 cond_noise_fn: Optional[Callable[[int, List[int]], Any]] = None):
        if max_batch < 1 or not batch_sizes or min(batch_sizes) != 1:
            raise ValueError("max_batch >= 1 and batch_sizes must contain 1")
        self.pipeline = pipeline
        self.max_batch = max_batch
        self.max_wait_s = max_wait_s
        self.batch_sizes = sorted(b for b in set(batch_sizes) if b <= max_batch)
        self.latent_channels, self.vae_scale = latent_channels, vae_scale
        self.cond_noise_fn = cond_noise_fn
        self._q: "queue.Queue[Optional[TryOnRequest]]" = queue.Queue()
        self._pending: List[TryOnRequest] = []
        self._stop = threading.Event()
        self.stats = {"requests": 0, "calls": 0, "images": 0, "busy_s": 0.0}
        self._worker = threading.Thread(target=self._run, name="tryon-service", daemon=True)
        self._worker.start()

    # ------------------------------------------------------------------ client side
    def submit(self, req: TryOnRequest) -> Future:
        if len(req.images) != 6:
            raise ValueError("a try-on request carries six conditioning images (app.py:164-171)")
        if self._stop.is_set():
            raise RuntimeError("service is shut down")
        self._q.put(req)
        return req.future

    def try_on(self, images: Sequence[torch.Tensor], prompt_embeds, negative_prompt_embeds, scale: float = 7.5,
               steps: int = 50, seed: int = 42) -> torch.Tensor:
        """Blocking form with the argument order of app.py:151-160 (tensors instead of PIL images)."""
        return self.submit(TryOnRequest(images, prompt_embeds, negative_prompt_embeds, scale, steps, seed)).result()

    def shutdown(self, wait: bool = True):
        self._stop.set()
        self._q.put(None)
        if wait:
            self._worker.join()

    # ------------------------------------------------------------------ worker side
    def _take_batch(self) -> List[TryOnRequest]:
        """Oldest request first; everything compatible with it that is already queued or arrives within max_wait_s
        of the moment it reached the head of the line joins its batch (up to max_batch), the rest keeps its order."""
        while not self._pending:
            r = self._q.get()
            if r is None:
                return []
            self._pending.append(r)
        head = self._pending[0]
        deadline = time.perf_counter() + self.max_wait_s

        def compatible():
            return [r for r in self._pending if r.key() == head.key()]
        while len(compatible()) < self.max_batch:
            left = deadline - time.perf_counter()
            try:
                r = self._q.get(timeout=left) if left > 0 else self._q.get_nowait()
            except queue.Empty:
                break
            if r is None:
                self._stop.set()
                break
            self._pending.append(r)
        batch = compatible()[: self.max_batch]
        n = max(b for b in self.batch_sizes if b <= len(batch))
        batch = batch[:n]
        ids = {id(r) for r in batch}
        self._pending = [r for r in self._pending if id(r) not in ids]
        return batch

    def _call(self, batch: List[TryOnRequest]):
        r0 = batch[0]
        h, w = r0.images[0].shape[-2:]
        lat = torch.cat([latents_for(r.seed, self.latent_channels, h // self.vae_scale, w // self.vae_scale) for r in batch], 0)
        kwargs = dict(
            prompt_embeds=torch.cat([r.prompt_embeds for r in batch], 0),
            negative_prompt_embeds=torch.cat([r.negative_prompt_embeds for r in batch], 0),
            image=[torch.cat([r.images[k] for r in batch], 0) for k in range(6)],
            latents=lat, guidance_scale=r0.guidance_scale, num_inference_steps=r0.num_inference_steps,
            control_guidance_start=r0.control_guidance_start, control_guidance_end=r0.control_guidance_end,
            output_type="pt")
        if self.cond_noise_fn is not None:       # per-request VAE-condition sampling noise (CL:38-42), keyed by seed
            kwargs["cond_noise"] = self.cond_noise_fn(len(batch), [r.seed for r in batch])
        return self.pipeline(**kwargs).images

    def _run(self):
        while True:
            batch = self._take_batch()
            if not batch:
                if self._stop.is_set() and not self._pending:
                    # drain: fail whatever is still queued instead of leaving futures pending forever
                    while True:
                        try:
                            r = self._q.get_nowait()
                        except queue.Empty:
                            return
                        if r is not None:
                            r.future.set_exception(RuntimeError("service shut down"))
                continue
            t0 = time.perf_counter()
            try:
                imgs = self._call(batch)
                for i, r in enumerate(batch):
                    r.future.set_result(imgs[i:i + 1])
            except Exception as e:                 # one bad batch must not kill the loop
                for r in batch:
                    if not r.future.done():
                        r.future.set_exception(e)
            self.stats["busy_s"] += time.perf_counter() - t0
            self.stats["calls"] += 1
            self.stats["images"] += len(batch)
            self.stats["requests"] += len(batch)
            if self._stop.is_set() and not self._pending and self._q.empty():
                return

"""Counterpart of the reference's `test_text2image_pretrained_openpose.py` on the HIP path (TT:76-365).

Same flag names as TT:76-212 / test_inference.sh; same flow as TT:215-365: load UNet / VAE / openpose ControlNet /
EdgeStyle multi-ControlNet (load_pattern [0,None,1,None,1,None]), tie the LoRA nets to the UNet, UniPC scheduler,
seed 42, six guidance scales linspace(1,7,6), 50 steps, 3x3 JPEG grid of [subject, target, target2, 6 results].

Differences, all forced by what exists offline: the CLIP prompt picker `BestEmbeddings` (model/utils.py:647-684,
edgestyle_amd/prompts.py) needs a local CLIP checkpoint (`--clip_model_name_or_path`) — without one pass `--prompt`;
`--random_init DIR` writes seeded random-init model directories in the
reference's on-disk layout first (no trained weights exist here) and uses random prompt embeddings when no
tokenizer/text_encoder directory is given.

    python -m edgestyle_amd.cli --random_init /tmp/es_models --tiny --source_path src --target_path tgt ...
"""
import argparse
import os
from typing import List, Optional

import numpy as np
import torch

RESOLUTION = 512          # TT:25
NUM_IMAGES = 6            # TT:27
CONTROLNET_PATTERN = [0, None, 1, None, 1, None]   # TT:50


def parse_args(argv: Optional[List[str]] = None):
    p = argparse.ArgumentParser(description="EdgeStyle try-on on MI355X (counterpart of test_text2image_pretrained_openpose.py)")
    p.add_argument("--pretrained_model_name_or_path", type=str, default=None)
    p.add_argument("--pretrained_vae_name_or_path", type=str, default=None)
    p.add_argument("--pretrained_openpose_name_or_path", type=str, default=None)
    p.add_argument("--controlnet_model_name_or_path", type=str, default=None)
    p.add_argument("--controllora_use_vae", action="store_true")
    p.add_argument("--use_agnostic_images", action="store_true")
    p.add_argument("--mixed_precision", type=str, default="fp16", choices=["no", "fp16", "bf16"])
    p.add_argument("--prompt", type=str, default="edgestyle")
    p.add_argument("--prompt_text_to_add", type=str, default="")
    p.add_argument("--negative_prompt", type=str, default="")
    p.add_argument("--clip_model_name_or_path", type=str, default=None,
                   help="local CLIP directory (transformers CLIPModel + CLIPProcessor): the prompt is then picked from the "
                        "first target's clothes image by BestEmbeddings like TT:52-55, 316; --prompt_vocab names a JSON "
                        "with the reference's colour / clothing-item lists")
    p.add_argument("--prompt_vocab", type=str, default=None)
    p.add_argument("--source_path", type=str, default=None)
    p.add_argument("--source_image_name", type=str, default="1.jpg")
    p.add_argument("--target_path", type=str, default=None)
    p.add_argument("--target_image_name", type=str, default="0.jpg")
    p.add_argument("--target_path2", type=str, default=None)
    p.add_argument("--target_image_name2", type=str, default="2.jpg")
    p.add_argument("--result_path", type=str, default=".")
    p.add_argument("--image_result_name", type=str, default="result.jpg")
    p.add_argument("--num_inference_steps", type=int, default=50)
    p.add_argument("--scheduler", type=str, default="unipc", choices=["unipc", "ddim"])
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--random_init", type=str, default=None, help="write seeded random-init model dirs here and use them")
    p.add_argument("--tiny", action="store_true", help="with --random_init: width-reduced 128x128 config (plumbing demo)")
    return p.parse_args(argv)


# TT:29-48 without torchvision: Resize(shorter side, bilinear) -> CenterCrop -> ToTensor (-> Normalize(0.5, 0.5))
def load_image(path: str, resolution: int, normalize: bool) -> torch.Tensor:
    from PIL import Image
    img = Image.open(path).convert("RGB")
    w, h = img.size
    s = resolution / min(w, h)
    img = img.resize((max(resolution, round(w * s)), max(resolution, round(h * s))), Image.BILINEAR)
    w, h = img.size
    l, t = (w - resolution) // 2, (h - resolution) // 2
    img = img.crop((l, t, l + resolution, t + resolution))
    x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    return (x - 0.5) / 0.5 if normalize else x


def image_grid(imgs, rows: int, cols: int):
    from PIL import Image
    assert len(imgs) == rows * cols          # TT:63-64
    w, h = imgs[0].size
    grid = Image.new("RGB", size=(cols * w, rows * h))
    for i, img in enumerate(imgs):
        grid.paste(img, box=(i % cols * w, i // cols * h))
    return grid


def add_text_to_image(image, text: str):
    from PIL import ImageDraw
    image = image.convert("RGB")
    ImageDraw.Draw(image).text((0, 0), text, (255, 255, 255))
    return image


def write_random_models(root: str, tiny: bool, seed: int = 0):
    """Seeded random-init checkpoints in the reference's directory layout (MC:213-282, CL:600-614)."""
    from . import config as C, weights as W
    ucfg, vcfg = (C.tiny_unet(), C.tiny_vae()) if tiny else (C.sd15_unet(), C.sd15_vae())
    rank = 4 if tiny else 32
    W.save_model_dir(os.path.join(root, "sd", "unet"), W.random_state_dict(W.unet_shapes(ucfg), seed, "unet."), ucfg.to_dict())
    W.save_model_dir(os.path.join(root, "vae"), W.random_state_dict(W.vae_shapes(vcfg), seed, "vae."), vcfg.to_dict())
    W.save_model_dir(os.path.join(root, "openpose"), W.random_state_dict(W.controlnet_shapes(ucfg), seed, "openpose."), ucfg.to_dict())
    cn = os.path.join(root, "EdgeStyle", "controlnet")
    W.save_model_dir(cn, W.random_state_dict(W.fusion_shapes(ucfg), seed, "fusion."))
    for idx in (0, 1):
        cfg = ucfg.to_dict()
        cfg.update(uses_vae=True, lora_linear_rank=rank)
        W.save_model_dir(os.path.join(cn, f"controlnet_{idx}"),
                         W.random_state_dict(W.controllora_saved_shapes(ucfg, rank), seed, f"controlnet_{idx}."), cfg)
    return dict(pretrained_model_name_or_path=os.path.join(root, "sd"), pretrained_vae_name_or_path=os.path.join(root, "vae"),
                pretrained_openpose_name_or_path=os.path.join(root, "openpose"), controlnet_model_name_or_path=cn), ucfg, vcfg


def main(args):
    from .models import (AutoencoderKL, ControlLoRAModel, ControlNetModel, EdgeStyleMultiControlNetModel,
                         UNet2DConditionModel)
    from .pipeline import StableDiffusionControlNetPipeline
    from .schedulers import DDIMScheduler, UniPCMultistepScheduler
    device = torch.device("cuda")                                   # TT:216 (no CPU fallback here)
    weight_dtype = {"no": torch.float16, "fp16": torch.float16, "bf16": torch.bfloat16}[args.mixed_precision]   # TT:218-222
    if args.random_init:
        paths, _, _ = write_random_models(args.random_init, args.tiny, seed=0)
        for k, v in paths.items():
            setattr(args, k, v)
        args.controllora_use_vae = True
    tokenizer = text_encoder = None
    tok_dir = os.path.join(args.pretrained_model_name_or_path, "tokenizer")
    if os.path.isdir(tok_dir):                                      # TT:224-232
        from transformers import AutoTokenizer, CLIPTextModel
        tokenizer = AutoTokenizer.from_pretrained(args.pretrained_model_name_or_path, subfolder="tokenizer", use_fast=False)
        text_encoder = CLIPTextModel.from_pretrained(args.pretrained_model_name_or_path, subfolder="text_encoder")
    vae = AutoencoderKL.from_pretrained(args.pretrained_vae_name_or_path or os.path.join(args.pretrained_model_name_or_path, "vae"))
    unet = UNet2DConditionModel.from_pretrained(args.pretrained_model_name_or_path, subfolder="unet", torch_dtype=weight_dtype)
    openpose = ControlNetModel.from_pretrained(args.pretrained_openpose_name_or_path, torch_dtype=weight_dtype)
    controlnet = EdgeStyleMultiControlNetModel.from_pretrained(                                   # TT:252-258
        args.controlnet_model_name_or_path, vae=vae if args.controllora_use_vae else None,
        controlnet_class=ControlLoRAModel, load_pattern=CONTROLNET_PATTERN,
        static_controlnets=[None, openpose, None, openpose, None, openpose], torch_dtype=weight_dtype)
    for net in controlnet.nets:                                                                   # TT:259-261
        if net is not openpose:
            net.tie_weights(unet)
    pipeline = StableDiffusionControlNetPipeline.from_pretrained(
        args.pretrained_model_name_or_path, vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet,
        controlnet=controlnet, safety_checker=None, torch_dtype=weight_dtype)
    pipeline.scheduler = UniPCMultistepScheduler.from_config(pipeline.scheduler.config) if args.scheduler == "unipc" \
        else DDIMScheduler()                                                                      # TT:273
    generator = torch.Generator().manual_seed(args.seed)                                          # TT:274
    pipeline = pipeline.to(device)

    res = unet.cfg.sample_size * vae.cfg.scale
    tp2 = args.target_path2 or args.target_path

    def img(root, kind, name, normalize):
        return load_image(os.path.join(root, kind, name), res, normalize)
    from PIL import Image
    shown = [Image.open(os.path.join(args.source_path, "subject", args.source_image_name)).convert("RGB").resize((res, res)),
             Image.open(os.path.join(args.target_path, "subject", args.target_image_name)).convert("RGB").resize((res, res)),
             Image.open(os.path.join(tp2, "subject", args.target_image_name2)).convert("RGB").resize((res, res))]
    vae_in = args.controllora_use_vae                      # IMAGES_TRANSFORMS ([-1,1]) vs CONDITIONING (TT:334-352)
    conds = [img(args.source_path, "agnostic" if args.use_agnostic_images else "head", args.source_image_name, vae_in),
             img(args.source_path, "openpose", args.source_image_name, False),
             img(args.target_path, "clothes", args.target_image_name, vae_in),
             img(args.target_path, "openpose", args.target_image_name, False),
             img(tp2, "clothes", args.target_image_name2, vae_in),
             img(tp2, "openpose", args.target_image_name2, False)]
    kw = {}
    if tokenizer is None:
        g = torch.Generator().manual_seed(args.seed + 1)
        d = unet.cfg.cross_attention_dim
        kw = dict(prompt_embeds=torch.randn(1, 77, d, generator=g) * 0.5, negative_prompt_embeds=torch.randn(1, 77, d, generator=g) * 0.5)
    else:
        prompt = args.prompt
        if args.clip_model_name_or_path:                                                          # TT:52-55, TT:316
            from transformers import CLIPModel, CLIPProcessor
            from .prompts import BestEmbeddings
            clip = CLIPModel.from_pretrained(args.clip_model_name_or_path).eval()
            proc = CLIPProcessor.from_pretrained(args.clip_model_name_or_path)
            be = (BestEmbeddings.from_vocab_file(clip, proc, args.prompt_vocab) if args.prompt_vocab
                  else BestEmbeddings(clip, proc))
            prompt = be([Image.open(os.path.join(args.target_path, "clothes", args.target_image_name)).convert("RGB")])[0]
        kw = dict(prompt=prompt + " " + args.prompt_text_to_add, negative_prompt=args.negative_prompt)
    for gs in np.linspace(1.0, 7.0, NUM_IMAGES):                                                  # TT:318, 326-359
        out = pipeline(guidance_scale=float(gs), image=conds, num_inference_steps=args.num_inference_steps,
                       generator=generator, **kw).images[0]
        shown.append(add_text_to_image(out, f"Guidance scale: {gs:.2f}"))
    os.makedirs(args.result_path, exist_ok=True)
    out_path = os.path.join(args.result_path, args.image_result_name)
    image_grid(shown, 3, len(shown) // 3).save(out_path)                                          # TT:362-365
    return out_path


if __name__ == "__main__":
    print(main(parse_args()))

"""CLIP prompt picker of the reference's callers: `BestEmbeddings` (model/utils.py:647-684, used at
test_text2image_pretrained_openpose.py:52-55 and :316).

Host-side helper, not part of the denoising hot path: it scores a list of colour words and a list of clothing-item words
against the garment image(s) with a CLIP model (`logits_per_image` of a transformers CLIPModel + CLIPProcessor pair, both
passed in by the caller) and builds the prompt "edgestyle, <2 best colours>, <2 best items>".

The reference hard-codes its two vocabularies (model/utils.py:183-644, 270 + 190 entries); they are data of the reference
and are not reproduced here: pass them (`BestEmbeddings(model, processor, colors=..., clothing_items=...)`, or
`BestEmbeddings.from_vocab_file(model, processor, "vocab.json")` with {"colors": [...], "clothing_items": [...]}) to get
the reference's prompts word for word.  The defaults below are short generic lists so that the class works out of the box.
"""
import json
from typing import Any, List, Optional, Sequence

import torch

DEFAULT_COLORS = ["black", "white", "gray", "red", "orange", "yellow", "green", "blue", "navy", "purple", "pink", "brown",
                  "beige", "cream", "teal", "maroon", "olive", "gold", "silver", "turquoise"]
DEFAULT_CLOTHING_ITEMS = ["t-shirt", "shirt", "blouse", "sweater", "hoodie", "cardigan", "jacket", "coat", "blazer",
                          "dress", "skirt", "jeans", "trousers", "shorts", "leggings", "suit", "vest", "tank top",
                          "polo shirt", "jumpsuit"]


class BestEmbeddings:
    def __init__(self, model, processor, colors: Optional[Sequence[str]] = None,
                 clothing_items: Optional[Sequence[str]] = None, prefix: str = "edgestyle"):
        self.model, self.processor = model, processor
        self.colors = list(colors) if colors is not None else list(DEFAULT_COLORS)
        self.clothing_items = list(clothing_items) if clothing_items is not None else list(DEFAULT_CLOTHING_ITEMS)
        self.prefix = prefix

    @classmethod
    def from_vocab_file(cls, model, processor, path: str, **kw):
        with open(path) as f:
            v = json.load(f)
        return cls(model, processor, colors=v["colors"], clothing_items=v["clothing_items"], **kw)

    def __call__(self, images) -> List[str]:
        """One prompt per image: "<prefix>, c1, c2, i1, i2" (MU:652-664)."""
        best_colors = self.find_best(self.colors, images)
        best_items = self.find_best(self.clothing_items, images)
        return [self.prefix + ", " + ", ".join(c + i) for c, i in zip(best_colors, best_items)]

    @torch.no_grad()
    def find_best(self, items: Sequence[str], images, n: int = 2) -> List[List[str]]:
        """The n highest-probability entries of `items` for every image, by softmax over CLIP's image-text logits
        (MU:666-684)."""
        items = list(items)
        inputs = self.processor(text=items, images=images, return_tensors="pt", padding=True)
        dev = getattr(self.model, "device", None)
        if dev is not None and hasattr(inputs, "to"):
            inputs = inputs.to(dev)
        outputs: Any = self.model(**inputs)
        probs = outputs.logits_per_image.softmax(dim=1)
        best = torch.argsort(probs, dim=1, descending=True)[:, :n]
        return [[items[int(j)] for j in row] for row in best]

#!/usr/bin/env python
"""Fixtures for the arithmetic the REFERENCE ITSELF owns on the hot path - `ControlNetBlock` (MC:23-63) and the two
`interleave_*` functions (MC:479-514) - produced by the reference's own code, in the build container only.

`model/edgestyle_multicontrolnet.py` cannot be imported (its module-level imports need diffusers, which is not installed and
cannot be), but these three definitions use `torch` / `torch.nn` only.  This script parses the file with `ast`, takes exactly
those three definitions, drops their type annotations (one names a diffusers class; annotations do not compute), and executes
them in a namespace that holds `torch` and `nn` - no module import, no stand-in for anything.  Nothing of the reference travels:
what is committed is tensors (outputs) - inputs and parameters are regenerated from seeds by tests/helpers.py::ref_fusion_case.

  ref_fusion.safetensors   for the 13 (channels, size) pairs of MC:73-102 at batch 2: ControlNetBlock(interleave_tensors(six
                           residuals)) sampled (every 4th channel, 8 x 8 pixels) + full-tensor sum / abs-sum (float64), and two small
                           interleave cases in full

    python tests/golden/make_golden_ref_fusion.py            (needs /root/reference; ~1 minute)
"""
import ast
import os
import sys

import torch
from torch import nn
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import helpers as H                            # noqa: E402

REF = "/root/reference/model/edgestyle_multicontrolnet.py"
WANT = ("ControlNetBlock", "interleave_tensors", "interleave_tensors_from_list_of_lists")


class _DropAnnotations(ast.NodeTransformer):
    def visit_FunctionDef(self, node):
        self.generic_visit(node)
        node.returns = None
        for a in node.args.args + node.args.kwonlyargs + node.args.posonlyargs:
            a.annotation = None
        return node


def reference_definitions():
    tree = ast.parse(open(REF).read(), REF)
    picked = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in WANT]
    assert sorted(n.name for n in picked) == sorted(WANT), [n.name for n in picked]
    mod = ast.fix_missing_locations(_DropAnnotations().visit(ast.Module(body=picked, type_ignores=[])))
    ns = {"torch": torch, "nn": nn}
    exec(compile(mod, REF, "exec"), ns)
    return ns


def main():
    ns = reference_definitions()
    Block, interleave, interleave_lists = (ns[k] for k in WANT)
    out = {}
    torch.set_num_threads(8)
    with torch.no_grad():
        for i, (C, S) in enumerate(H.REF_FUSION_LEVELS):
            sd, res = H.ref_fusion_case(i)
            blk = Block(C, (S, S), 6)
            missing = blk.load_state_dict(sd, strict=True)
            assert not missing.missing_keys and not missing.unexpected_keys
            y = blk(interleave(res)).double()
            out[f"level{i}_sample"] = H.ref_fusion_sample(y).float().contiguous()
            out[f"level{i}_sums"] = torch.stack([y.sum(), y.abs().sum()])
            print(f"level {i}: C={C} S={S} sum {float(y.sum()):.6f} abs {float(y.abs().sum()):.3f}", flush=True)
        a, b = H.ref_interleave_cases()
        out["interleave_a"] = interleave(a).contiguous()
        lists = interleave_lists([a, b])
        out["interleave_lists_0"], out["interleave_lists_1"] = lists[0].contiguous(), lists[1].contiguous()
    save_file(out, os.path.join(HERE, "ref_fusion.safetensors"))
    print("wrote ref_fusion.safetensors", sum(v.numel() * v.element_size() for v in out.values()) >> 10, "KiB")


if __name__ == "__main__":
    main()

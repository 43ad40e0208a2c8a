"""Golden vectors for the arithmetic of the reference's training step, produced by the reference's OWN statements.

The EDM noising (/root/reference/train_svd.py:964-972), the conditioning dropout and channel concat (:992-1017) `target = latents` (:1020) and the
weighted MSE (:1025-1036) are inline statements of `main()`; the module cannot be imported here (diffusers / accelerate are absent).  This
script parses the file, lifts exactly those statement ranges out of `main()`'s AST and executes them -- unmodified -- in a
namespace that supplies the tensors they expect, then stores inputs' seeds and the values they computed.  Nothing of the reference's
source is written to the repo.  Usage (this container only):  python tests/golden/make_golden_step_math.py
"""
import ast
import os
from types import SimpleNamespace

import torch
from safetensors.torch import save_file

REF = "/root/reference/train_svd.py"
RANGES = {"noising": (964, 972), "dropout_concat": (992, 1017), "target": (1020, 1020), "loss": (1025, 1036)}
# (bsz, frames, h, w, embed dim, conditioning_dropout_prob, seed)
CASES = [(2, 3, 8, 6, 16, None, 0), (4, 2, 6, 8, 16, 0.1, 1), (8, 2, 4, 4, 8, 0.1, 2), (8, 1, 4, 4, 8, 0.3, 3)]


def lifted(tree, lo, hi):
    """Outermost statements of the file lying entirely inside lines [lo, hi]."""
    inside = [n for n in ast.walk(tree) if isinstance(n, ast.stmt) and n.lineno >= lo and n.end_lineno <= hi]
    top = [n for n in inside if not any(m is not n and m.lineno <= n.lineno and n.end_lineno <= m.end_lineno and
                                        any(c is n for c in ast.walk(m)) for m in inside)]
    return compile(ast.Module(body=sorted(top, key=lambda n: n.lineno), type_ignores=[]), REF, "exec")


def case_inputs(bsz, T, h, w, D, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    return dict(latents=0.7 * torch.randn(bsz, T, 4, h, w, generator=g), noise=torch.randn(bsz, T, 4, h, w, generator=g),
                conditional_latents=torch.randn(bsz, 4, h, w, generator=g), encoder_hidden_states=torch.randn(bsz, D, generator=g),
                model_pred=torch.randn(bsz, T, 4, h, w, generator=g))


def main():
    src = open(REF).read()
    tree = ast.parse(src)
    code = {k: lifted(tree, *r) for k, r in RANGES.items()}
    fns = {"torch": torch}
    exec(compile(ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "rand_log_normal"],
                            type_ignores=[]), REF, "exec"), fns)
    out = {}
    for i, (bsz, T, h, w, D, prob, seed) in enumerate(CASES):
        ns = dict(fns)
        ns.update(case_inputs(bsz, T, h, w, D, seed))
        ns.update(bsz=bsz, accelerator=SimpleNamespace(device=torch.device("cpu")), args=SimpleNamespace(conditioning_dropout_prob=prob),
                  generator=torch.Generator().manual_seed(2000 + seed))
        model_pred = ns.pop("model_pred")
        torch.manual_seed(3000 + seed)                       # rand_log_normal draws from the global generator
        exec(code["noising"], ns)
        exec(code["dropout_concat"], ns)
        exec(code["target"], ns)
        ns["model_pred"] = model_pred                        # stands in for the UNet call at :1021-1022
        exec(code["loss"], ns)
        for k in ("sigmas", "noisy_latents", "timesteps", "inp_noisy_latents", "encoder_hidden_states", "loss"):
            out[f"case{i}.{k}"] = ns[k].detach().clone().contiguous()
    # `_get_add_time_ids` (:878-898), a function nested in main(): executed with a stand-in for the two model attributes it reads
    nested = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "_get_add_time_ids"]
    ns = {"torch": torch, "unet": SimpleNamespace(config=SimpleNamespace(addition_time_embed_dim=256),
                                                  add_embedding=SimpleNamespace(linear_1=SimpleNamespace(in_features=768)))}
    exec(compile(ast.Module(body=nested, type_ignores=[]), REF, "exec"), ns)
    out["add_time_ids"] = ns["_get_add_time_ids"](7, 127, torch.tensor(0.0625), torch.float32, 3)     # call site :981-987
    ns["unet"].add_embedding.linear_1.in_features = 512
    try:
        ns["_get_add_time_ids"](7, 127, 0.02, torch.float32, 1)
        raise SystemExit("the reference's config check did not fire")
    except ValueError:
        pass
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_math.safetensors")
    save_file(out, path)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: tuple(v.shape) for k, v in out.items() if k.startswith("case1.")})


if __name__ == "__main__":
    main()

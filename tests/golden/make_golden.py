"""Generates tests/golden/tiny_step.safetensors: one seeded train step of the CPU oracle (oracle/) on the tiny topology.

The reference holds no golden vectors, tests or fixtures and diffusers/peft cannot be imported in the build
container, so this fixture freezes the ORACLE: it guards the restatement against accidental drift (tests/test_oracle.py re-derives
it on CPU) and gives the GPU parity test a second, committed anchor.  Its "full" half is tied to the reference's own code by
tests/golden/make_golden_unet_toplevel.py, which assembles the same step from the reference's UNet class (over the oracle's blocks)
and the reference's loop-body statements and finds it bit-equal (tests/test_oracle_toplevel.py).
Contents: the inputs are re-generated from the seeds (make_synthetic_batch is deterministic); stored are the prediction, the
loss and the L2 norm of every trainable gradient, for the full-parameter (config 2) and the LoRA r = 8 (config 5) step.
usage: python tests/golden/make_golden.py"""
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import e2e_checks  # noqa: E402
from oracle.unet import TINY_CONFIG  # noqa: E402

SPEC = dict(B=1, T=4, h=16, w=16, seed=3, lr=1e-4)


def main():
    torch.manual_seed(0)
    out = {}
    for tag, r in (("full", 0), ("lora8", 8)):
        ref = e2e_checks.oracle_step(TINY_CONFIG, SPEC["B"], SPEC["T"], SPEC["h"], SPEC["w"], seed=SPEC["seed"], lr=SPEC["lr"],
                                     cross_dim=TINY_CONFIG["cross_attention_dim"], lora_r=r)
        out[f"{tag}.pred"] = ref["pred"].contiguous()
        out[f"{tag}.loss"] = torch.tensor([ref["loss"]], dtype=torch.float64)
        names = sorted(ref["grads"])
        out[f"{tag}.grad_norms"] = torch.tensor([float(ref["grads"][n].double().norm()) for n in names], dtype=torch.float64)
        with open(os.path.join(HERE, f"tiny_step_{tag}_grad_names.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
    save_file(out, os.path.join(HERE, "tiny_step.safetensors"),
              metadata={"spec": repr(SPEC), "torch": torch.__version__, "generator": "tests/golden/make_golden.py"})
    print({k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()

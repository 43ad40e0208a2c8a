"""CPU-oracle references of the parity cases that take minutes and tens of GB (reference config 4's upper levels), computed ONCE on any
host with the memory and stored under tests/golden/_big/ (git-ignored; travels to the GPU box with the gpurun snapshot), so that the
GPU box compares against them without spending GPU-minutes on CPU work.  Inputs and weights are seeded; only the oracle's outputs
(loss, prediction, gradients, updated parameters) are stored, plus a fingerprint of the seeded weights.

    SVDX_SAVE_BIG_REF=1 python tests/golden/make_big_refs.py [L0] [L1]

Then, on the GPU:  SVDX_BIG_PARITY=1 python -m pytest tests/test_e2e_gpu.py -m gpu -k "c4_level and (L0 or L1)"
Without the stored file the test computes the oracle step itself (same result, slower)."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    os.environ["SVDX_SAVE_BIG_REF"] = "1"
    import e2e_checks
    from oracle import unet as ou
    ou.RECOMPUTE = True           # resnet / transformer modules under torch.utils.checkpoint: same values, a fraction of the memory
    want = sys.argv[1:] or ["L0"]
    for name, (C, heads, h, w) in e2e_checks.C4_LEVELS.items():
        if name.split()[0] not in want:
            continue
        cfg = e2e_checks.level_config(C, heads, num_frames=25)
        t0 = time.time()
        ref = e2e_checks.oracle_step_cached(f"{name} T=25 seed=11", cfg, 1, 25, h, w, 11, 1e-4, cfg["cross_attention_dim"])
        print(f"{name}: loss {ref['loss']:.7f}, {len(ref['grads'])} gradients, {time.time() - t0:.0f} s, "
              f"{'loaded ' + ref['cached'] if 'cached' in ref else 'computed and stored'}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    main()

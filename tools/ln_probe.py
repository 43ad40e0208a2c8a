"""One kernel family in isolation for rocprofv3 --kernel-trace --stats (developer probe): python tools/ln_probe.py <rows> <C> <affine 0|1>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svd_xtend_amd import kernels as K  # noqa: E402

rows, C, affine = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev, dt = torch.device("cuda"), torch.float16
k = K.backend()
x, dy, y = (torch.randn(rows, C, device=dev).to(dt) for _ in range(3))
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
st = torch.empty(rows, 2, device=dev)
k.ln_fwd(x, gamma, beta, y, st, rows, C, 1e-5)
scr = torch.empty(K.LN_PARTIAL_ROWS * 2 * C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
for _ in range(20):
    if affine:
        k.ln_bwd(dy, x, st, gamma, None, y, dg, db, rows, C, scratch=scr)
    else:
        k.ln_bwd(dy, x, st, gamma, None, y, None, None, rows, C)
torch.cuda.synchronize()

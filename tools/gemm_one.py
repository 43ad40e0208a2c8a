"""Run one svdx_gemm shape a few times (for rocprofv3 --pmc).  usage: gemm_one.py M N K variant [conv n h w cin]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from svd_xtend_amd import kernels as K  # noqa: E402

M, N, Kd, v = (int(x) for x in sys.argv[1:5])
be = K.backend()
dev = torch.device("cuda")
dt = torch.float16
gather = None
if len(sys.argv) > 5 and sys.argv[5] == "conv":
    n, h, w, ci = (int(x) for x in sys.argv[6:10])
    A = torch.randn(n * h * w, ci, device=dev).to(dt)
    lda = ci
    gather = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=h, wo=w, cin=ci, stride=1, lda=ci)
else:
    A = torch.randn(M, Kd, device=dev).to(dt)
    lda = Kd
B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(dt)
C = torch.zeros(M, N, device=dev, dtype=dt)
# rotate over several output/input buffers so the 256 MiB MALL does not hide HBM traffic
As = [A.clone() for _ in range(6)]
Cs = [C.clone() for _ in range(6)]
for i in range(12):
    be.gemm(As[i % 6], B, Cs[i % 6], M, N, Kd, lda, Kd, N, gather=gather, variant=v)
torch.cuda.synchronize()

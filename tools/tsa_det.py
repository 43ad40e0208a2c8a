"""Run-to-run reproducibility of the fused temporal self-attention (svdx_tsa_fwd): the same inputs into output buffers pre-filled with
different values must give identical bits, at the small widths of the test topology and at the benched level shapes.  The tool that
found the wide-store data hazard; the assertion form is tests/test_kernels_gpu.py.
    python tools/tsa_det.py"""
import sys, torch
sys.path.insert(0, '/root/repo')
from svd_xtend_amd import kernels as K
k = K.backend()
dev = torch.device('cuda')
def run(B, T, HW, heads, fill, seed=0, dt=torch.float16):
    C = heads * 64; M = B * T * HW
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C, generator=g).to(dt).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    wqkv = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(dt).to(dev)
    wo = (torch.randn(C, C, generator=g) * C ** -0.5).to(dt).to(dev)
    bo, cvec = (0.1 * torch.randn(C, generator=g)).to(dev), torch.randn(B, C, generator=g).to(dev)
    outs = [torch.full((M, C), fill, dtype=dt, device=dev), torch.full((M, 2), fill, device=dev), torch.full((M, 3 * C), fill, dtype=dt, device=dev),
            torch.full((M, C), fill, dtype=dt, device=dev), torch.full((M, C), fill, dtype=dt, device=dev)]
    n1, st, qkv, o, h1 = outs
    k.tsa_fwd(x, gamma, beta, 1e-5, wqkv, wo, bo, cvec, C, T * HW, 0, n1, st, qkv, o, h1, B, T, HW, C, heads, 0.125)
    torch.cuda.synchronize()
    return [t.clone() for t in outs]
for shape in [(1, 4, 256, 1), (2, 3, 384, 1), (1, 4, 64, 2), (2, 3, 96, 2), (1, 14, 2560, 5), (1, 14, 640, 5)]:
    P = K.tsa_pixels_per_band(shape[1], shape[2])
    a = run(*shape, fill=0.0); names = ['n1', 'stats', 'qkv', 'o', 'h1']
    bad = []
    for rep in range(4):
        b = run(*shape, fill=float('nan') if rep % 2 else 7.0)
        for n, u, v in zip(names, a, b):
            if not torch.equal(torch.nan_to_num(u.float(), nan=123.0), torch.nan_to_num(v.float(), nan=123.0)):
                d = (torch.nan_to_num(u.float(), nan=123.0) - torch.nan_to_num(v.float(), nan=123.0)).abs()
                bad.append((rep, n, int((d > 0).sum()), float(d.max())))
    print(shape, 'P', P, 'rows', P * shape[1], 'nan_in_out', [bool(torch.isnan(t.float()).any()) for t in a], 'MISMATCH' if bad else 'bit-identical', bad[:6])

print("---- where")
shape = (1, 4, 64, 2)
a = run(*shape, fill=0.0)
for rep in range(12):
    b = run(*shape, fill=5.0)
    d = (a[3].float() - b[3].float()).abs()
    if float(d.max()) > 0:
        idx = (d > 0).nonzero()
        rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
        r = rows[0]
        # does the wrong chunk equal some other row's data?  or the fill value?
        vals = b[3][r, cols[0]:cols[0] + 8].float().tolist()
        print(rep, 'rows', rows, 'cols', cols[0], '..', cols[-1], 'got', [round(v, 3) for v in vals[:4]], 'want', [round(v, 3) for v in a[3][r, cols[0]:cols[0] + 4].float().tolist()])
        # search for the got-chunk elsewhere in the correct output
        chunk = b[3][r, cols[0]:cols[0] + 8]
        C = a[3].shape[1]
        hits = [(rr, cc) for rr in range(a[3].shape[0]) for cc in range(0, C, 8) if torch.equal(a[3][rr, cc:cc + 8], chunk)]
        print('   chunk found in the correct output at', hits[:4])

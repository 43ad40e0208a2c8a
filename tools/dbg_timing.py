"""developer aid: replay-timing of the graph-captured step under different warm-up / chunking patterns (one box, one process each)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svd_xtend_amd.train import GraphedStep, Trainer
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--warm", type=int, default=0)
ap.add_argument("--chunk", type=int, default=10)
ap.add_argument("--nchunks", type=int, default=6)
ap.add_argument("--eager", type=int, default=2)
ap.add_argument("--two-graphs", action="store_true")
ap.add_argument("--loss-print", action="store_true")
ap.add_argument("--apply", action="store_true")
ap.add_argument("--post", default="")
ap.add_argument("--long", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
with torch.device(dev):
    model = UNetSpatioTemporalConditionModel()
bench.init_weights_(model, seed=1234)
tr = Trainer(model, dtype=torch.float16, lr=1e-5)
batch = bench.make_batch(1, 14, 40, 64, model.config.cross_attention_dim, seed=123, dev=dev)
if a.apply:
    from svd_xtend_amd import ops
    tr.rt.tuner = None
    for f in (getattr(ops, n, None) for n in dir(ops)):
        if hasattr(f, "cache_clear"):
            f.cache_clear()
for _ in range(a.eager): tr.step(batch)
torch.cuda.synchronize()
g = GraphedStep(tr, batch); g(); torch.cuda.synchronize()
if a.loss_print:
    print("loss", float(tr.last_loss()))
if a.post == "sync_stream":
    torch.cuda.current_stream().synchronize()
elif a.post == "kernel":
    y = tr.loss_slot + 1; torch.cuda.synchronize()
elif a.post == "d2h":
    y = tr.loss_slot.cpu()
elif a.post == "d2h_nb":
    y = tr.loss_slot.to("cpu", non_blocking=True); torch.cuda.synchronize()
elif a.post == "h2d":
    tr.loss_slot.copy_(torch.zeros(1)); torch.cuda.synchronize()
elif a.post == "inplace":
    tr.loss_slot.add_(0); torch.cuda.synchronize()
elif a.post == "alloc":
    y = torch.empty(1, device=dev); torch.cuda.synchronize()
elif a.post == "alloc_big":
    y = torch.empty(1 << 28, device=dev); torch.cuda.synchronize()
elif a.post == "svdx":
    tr.rt.k.zero_spans(tr.g_flat, torch.tensor([[0, 4]], dtype=torch.int32, device=dev), 1); torch.cuda.synchronize()
elif a.post == "zeros_new":
    y = torch.zeros(1, device=dev); torch.cuda.synchronize()
elif a.post == "other_plus1":
    z = torch.ones(1, device=dev); torch.cuda.synchronize(); y = z + 1; torch.cuda.synchronize()
elif a.post == "clone":
    y = tr.loss_slot.clone(); torch.cuda.synchronize()
elif a.post == "last_loss":
    y = tr.last_loss(); torch.cuda.synchronize()
elif a.post == "last_loss_nosync":
    y = tr.last_loss()
elif a.post == "big_plus1":
    y = tr.g_flat[:1 << 20] + 1; torch.cuda.synchronize()
elif a.post == "event":
    e = torch.cuda.Event(); e.record(); e.synchronize()
if a.two_graphs:
    for _ in range(a.eager): tr.step(batch)
    torch.cuda.synchronize()
    g = GraphedStep(tr, batch); g(); torch.cuda.synchronize()
for _ in range(a.warm): g()
torch.cuda.synchronize()
out = []
for _ in range(a.nchunks):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.chunk):
        g()
        if a.post == "each_step":
            y = tr.last_loss()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3 / a.chunk)
if a.long:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.long): g()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3 / a.long)
print(vars(a), " ".join(f"{x:.2f}" for x in out), f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)

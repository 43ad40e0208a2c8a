"""developer aid (round 4): which ordinary launch between two hipGraph replays of the step leaves the NEXT replays with a non-finite loss?
One process per case: capture, replay (loss must be finite), ACTION, three more replays, read the loss slot with a plain D2H copy."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svd_xtend_amd.train import GraphedStep, Trainer
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
case = sys.argv[1]
tiny = "--tiny" in sys.argv
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = dict(block_out_channels=(64, 128, 128, 128), addition_time_embed_dim=32, projection_class_embeddings_input_dim=96, cross_attention_dim=64,
           num_attention_heads=(1, 2, 2, 2)) if tiny else {}
with torch.device(dev):
    model = UNetSpatioTemporalConditionModel(**cfg)
bench.init_weights_(model, seed=1234)
tr = Trainer(model, dtype=torch.float16, lr=1e-5)
batch = bench.make_batch(1, 14, 40, 64, model.config.cross_attention_dim, seed=123, dev=dev)
for _ in range(2): tr.step(batch)
torch.cuda.synchronize()
pre = torch.zeros(1, device=dev)
z = torch.ones(1, device=dev)
side = torch.cuda.Stream()
torch.cuda.synchronize()
g = GraphedStep(tr, batch); g(); torch.cuda.synchronize()
def loss(): return float(tr.loss_slot.cpu())
single = case in ("noread", "sync_only", "read_other", "read_pflat", "read_batch", "h2d_other", "read_pinned")
l0 = loss() if not single else float("nan")
if case == "none" or case == "noread": pass
elif case == "read_batch": v = float(batch["sigmas"].cpu())
elif case == "h2d_other": pre.copy_(torch.ones(1))
elif case == "read_pinned":
    hp = torch.empty(1, pin_memory=True); hp.copy_(tr.loss_slot, non_blocking=True); torch.cuda.synchronize()
elif case == "sync_only": torch.cuda.synchronize()
elif case == "read_other": v = float(tr.opt_state[0].cpu())
elif case == "read_pflat": v = float(tr.p_flat[:1].cpu())
elif case == "plus1": y = z + 1
elif case == "div_out": torch.div(tr.loss_slot, 1.0, out=pre)
elif case == "plus1_out": torch.add(z, 1, out=pre)
elif case == "mul2": y = z * 2
elif case == "exp": y = torch.exp(z)
elif case == "inplace": z.add_(1)
elif case == "svdx_add":
    a = torch.ones(64, device=dev, dtype=torch.float16); b = torch.ones(64, device=dev, dtype=torch.float16); c = torch.empty_like(a)
    torch.cuda.synchronize(); tr.rt.k.add(a, b, c, 64)
elif case == "mm":
    a = torch.ones(64, 64, device=dev, dtype=torch.float16); torch.cuda.synchronize(); y = a @ a
elif case == "plus1_side":
    with torch.cuda.stream(side): y = z + 1
elif case == "plus1_big":
    zz = torch.ones(1 << 20, device=dev); torch.cuda.synchronize(); y = zz + 1
elif case == "plus1_sync_before":
    torch.cuda.synchronize(); y = z + 1; torch.cuda.synchronize()
elif case == "item": v = tr.last_loss().item()
torch.cuda.synchronize()
ls = []
nrep = 3
for i in range(nrep):
    g(); torch.cuda.synchronize()
    if not single or i == nrep - 1:
        ls.append(loss())
st = tr.opt_state.cpu().tolist()
ck = float(tr.p_flat.double().abs().sum().cpu())
print(f"{case:18s} psum {ck:.6f} loss before {l0:.5f}  after {['%.5f' % x for x in ls]}  opt_steps {st[0]:.0f} loss_scale {st[1]:g}", flush=True)

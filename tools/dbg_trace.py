"""developer aid: kernel list of a graph replay before and after a host<->device copy (run under rocprofv3 --kernel-trace)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svd_xtend_amd.train import GraphedStep, Trainer
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
with torch.device(dev):
    model = UNetSpatioTemporalConditionModel()
bench.init_weights_(model, seed=1234)
tr = Trainer(model, dtype=torch.float16, lr=1e-5)
batch = bench.make_batch(1, 14, 40, 64, model.config.cross_attention_dim, seed=123, dev=dev)
for _ in range(2): tr.step(batch)
torch.cuda.synchronize()
pre = torch.zeros(1, device=dev)
g = GraphedStep(tr, batch); g(); torch.cuda.synchronize()
mark = torch.zeros(256, device=dev)
def marker(): mark.fill_(1.0); torch.cuda.synchronize()      # a recognisable ATen fill kernel between phases... (itself an eager launch: keep it OUT of the phases)
g(); torch.cuda.synchronize()
g(); torch.cuda.synchronize()
pre.copy_(torch.ones(1)); torch.cuda.synchronize()            # the H2D copy
g(); torch.cuda.synchronize()
g(); torch.cuda.synchronize()
print("loss", float(tr.loss_slot.cpu()))

"""svdx_allreduce_grads + train.DirectAllReduce: the gradient sum over the ranks of a node as a direct reduce-scatter + all-gather over
peer-mapped buffers (replaces the all-reduce DistributedDataParallel runs inside accelerator.backward,
/root/reference/train_svd.py:815 + :1044, when RCCL would ring it).  The kernel itself is checked in tests/kernel_checks.py
(`check_optim`: GPU, simulator); here: the cross-process form -- IPC handle exchange, barrier protocol, Trainer wiring -- with the
ranks as PROCESSES sharing the one GPU of the test box (gloo carries the barriers: RCCL refuses two ranks on one device)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
gpu = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svd_xtend_amd.train import DirectAllReduce
    dev = torch.device("cuda", 0)
    buf = torch.zeros(n, dtype=torch.float32, device=dev)
    d = DirectAllReduce(buf)
    ok = True
    for rnd in range(3):                                 # fresh data every round: a stale read of a peer's previous contents would show
        parts = [torch.randn(n, generator=torch.Generator().manual_seed(1000 * rnd + q)) for q in range(world)]
        want = parts[0].clone()
        for q in range(1, world):
            want = want + parts[q]                       # rank order, as the kernel adds
        buf.copy_(parts[rank])
        if rnd == 2:
            d.start().wait()
        else:
            d.all_reduce()
        torch.cuda.synchronize()
        ok = ok and torch.equal(buf.cpu(), want)
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@gpu
@pytest.mark.parametrize("world,n", [(2, 4 * 100003), (4, 4 * 2501)])
def test_direct_allreduce_across_processes_on_one_gpu(world, n):
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
        assert dict(out) == {r: True for r in range(world)}, dict(out)


def _trainer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.step import edm_inputs, make_synthetic_batch
    from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
    from svd_xtend_amd.train import Trainer
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    dev = torch.device("cuda", 0)
    res = {}
    for direct in (False, True):
        orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
        scaled_init_(orc, 0)
        m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
        m.load_state_dict(orc.state_dict(), strict=True)
        tr = Trainer(m.to(dev), dtype=torch.float16, lr=1e-3)
        tr.use_direct_allreduce(direct)
        for step in range(2):
            b = make_synthetic_batch(1, 2, 16, 16, 100 + 10 * step + rank, cross_dim=64)
            unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
            tr.step({k: v.to(dev) for k, v in dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy,
                                                   target=b["latents"], sigmas=b["sigmas"]).items()})
        torch.cuda.synchronize()
        res[direct] = (tr.p_flat.cpu().clone(), float(tr.last_loss()))
    # two ranks: a + b is the same float whichever library adds it -- the direct sum walks RCCL's (here: gloo's) trajectory bit for bit
    out[rank] = (torch.equal(res[False][0], res[True][0]), res[False][1] == res[True][1], res[True][0].double().sum().item())
    dist.barrier()
    dist.destroy_process_group()


@gpu
def test_trainer_with_direct_allreduce_equals_the_collective_library():
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_trainer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        o = dict(out)
        assert o[0][:2] == (True, True) and o[1][:2] == (True, True), o
        assert o[0][2] == o[1][2], o                    # replicas identical

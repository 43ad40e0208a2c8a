"""Data-parallel path at 4 and 8 ranks on CPU (gloo over 127.0.0.1, emulated kernels): what the driver's 8-GPU node will run, minus
RCCL.  Per world size, with gradient accumulation 2 and rank-distinct data:

  * construction broadcasts rank 0's trainables (ranks are built from DIFFERENT seeds here, as DDP's constructor would face);
  * the three reduction schedules -- per-block buckets under the last backward sweep, ONE collective after it, ONE collective with
    side work (the next micro-batch's conditioners: north_star's schedule) queued beside it -- leave every rank with bitwise identical
    parameters; the two single-collective runs agree bit for bit, the bucket run to fp32 summation order;
  * nothing is reduced before the last micro-batch (accelerator.accumulate semantics, /root/reference/train_svd.py:941), the single
    schedules issue exactly one collective per optimizer step, the bucket schedule covers the flat buffer exactly once;
  * the loss slot carries the mean over ranks and micro-batches through the same collective (replaces accelerator.gather, :1039);
  * the result equals ONE process accumulating all world x 2 micro-batches, up to fp32 summation order."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ACCUM = 2


def _setup():
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emul
    from svd_xtend_amd import kernels
    kernels._set_backend_for_tests(emul.EmuBackend())


def _make(seed):
    from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, seed)
    m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    m.load_state_dict(orc.state_dict(), strict=True)
    return m


def _batch(seed):
    from oracle.step import edm_inputs, make_synthetic_batch
    b = make_synthetic_batch(1, 2, 8, 8, seed, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
    return dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy, target=b["latents"],
                sigmas=b["sigmas"])


def _micro_batches(step, rank):
    return [_batch(1000 + 100 * step + 10 * rank + a) for a in range(ACCUM)]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    _setup()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svd_xtend_amd.train import Trainer
    calls = []                                            # (elements, micro-batch index at call time) of every all_reduce
    real = dist.all_reduce

    def counting(t, *a, **kw):
        calls.append((t.numel(), cur["tr"].micro))
        return real(t, *a, **kw)
    dist.all_reduce = counting
    cur = {}
    res = {}
    for mode in ("buckets", "single", "single+side"):
        tr = Trainer(_make(rank), dtype=torch.float32, lr=1e-3, grad_accum=ACCUM)       # rank-specific init: the broadcast must fix it
        cur["tr"] = tr
        gathered = [torch.empty_like(tr.p_flat) for _ in range(world)]
        dist.all_gather(gathered, tr.p_flat)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "construction must leave rank 0's trainables everywhere"
        tr.overlap = mode == "buckets"
        order = []
        for step in range(2):
            calls.clear()
            tr.zero_grad()
            local = []                                    # this rank's micro-batch losses: increments of the (not yet reduced) loss slot
            for b in _micro_batches(step, rank):
                before = float(tr.loss_slot)
                tr.forward_backward(**b)
                local.append(float(tr.loss_slot) - before)
                if tr.micro < ACCUM:
                    assert not calls, "gradients are only reduced on the last micro-batch"
            if mode == "single+side":
                tr.finish_grads(side_work=lambda: order.append(("side", len(calls))))
            else:
                tr.finish_grads()
            tr.optimizer_step()
            order.append(("opt", len(calls)))
            if mode == "buckets":
                assert len(calls) >= len(tr._buckets) and sum(n for n, _ in calls) == tr.g_flat.numel(), \
                    "the buckets and the rest spans cover the flat buffer exactly once"
            else:
                assert [n for n, _ in calls] == [tr.g_flat.numel()], "ONE collective per optimizer step"
            assert all(m >= ACCUM - 1 for _, m in calls)    # during the last micro-batch's sweep (buckets) or after it
        if mode == "single+side":                         # the side work is queued after the collective started, before the optimizer
            assert order == [("side", 1), ("opt", 1), ("side", 1), ("opt", 1)], order
        # the loss slot rode through the sum: mean over ranks and micro-batches of the last step's local losses
        mine = torch.tensor([sum(local)], dtype=torch.float64)
        real(mine)
        assert abs(float(tr.last_loss()) - float(mine) / (world * ACCUM)) <= 1e-5 * abs(float(mine)), (float(tr.last_loss()), float(mine))
        res[mode] = tr.p_flat.clone()
    dist.all_reduce = real
    assert torch.equal(res["single"], res["single+side"])                    # the same collective, with or without work queued beside it
    # buckets vs one collective: a ring reduction adds the ranks in an order that depends on an element's position in the reduced
    # tensor, so beyond 2 ranks the two schedules agree to fp32 summation order, not bit for bit
    d = (res["buckets"] - res["single"]).abs()
    assert float(d.mean()) < 1e-7 and float(d.max()) < 2.5e-3, (float(d.mean()), float(d.max()))
    torch.save(res["single"], os.path.join(out, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [4, 8])
def test_wide_allreduce_schedules_agree_and_equal_grad_accumulation(tmp_path, world):
    port = 31500 + (os.getpid() + world) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ps = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert all(torch.equal(p, ps[0]) for p in ps)         # replicas identical after the reduced steps
    _setup()
    from svd_xtend_amd.train import Trainer
    tr = Trainer(_make(0), dtype=torch.float32, lr=1e-3, grad_accum=ACCUM * world)
    for step in range(2):
        tr.step([b for r in range(world) for b in _micro_batches(step, r)])
    d = (tr.p_flat - ps[0]).abs()
    # identical up to fp32 summation order (sum over ranks vs in-place accumulation); AdamW's m/sqrt(v) amplifies that only where
    # the gradient itself is at rounding level
    assert float(d.mean()) < 1e-7 and float(d.max()) < 2.5e-3, (float(d.mean()), float(d.max()))

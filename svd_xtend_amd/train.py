"""The training step of /root/reference/train_svd.py:941-1049 on the MI355X-native UNet.

`Trainer` owns what accelerate + torch.optim own in the reference: the trainable-set selection
(train_svd.py:761-766), one flat float master/grad/Adam buffer, the fp16 loss-scale state machine
(GradScaler semantics, on device), AdamW (train_svd.py:767-773) and the data-parallel gradient mean
(intended DDP semantics) as ONE all-reduce of the flat gradient buffer over RCCL.
Everything on the timed path is libsvdx kernels; torch supplies memory, streams and torch.distributed.
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import kernels as K
from .ops import flatten_trainables
from .unet import UNetSpatioTemporalConditionModel

ALIGN = 64  # floats


def select_trainable(model: torch.nn.Module, substr: str = "temporal_transformer_block") -> List[str]:
    """train_svd.py:761-766 (names containing `temporal_transformer_block`); with LoRA adapters injected
    (train_svd_lora.py:655-671: everything frozen, then `add_adapter`) the adapters are the trainable set."""
    has_lora = any(".lora_" in name for name, _ in model.named_parameters())
    names = []
    for name, p in model.named_parameters():
        p.requires_grad = (".lora_" in name) if has_lora else (substr in name)
        if p.requires_grad:
            names.append(name)
    return names


def build_adam_tiles(params, offsets, wt_map, device) -> torch.Tensor:
    """Tile table of svdx_adamw_tiled: every trainable parameter is cut into <= 64 x 64 tiles of its [rows, cols] view
    (matrices as they are, everything else as one row); weights with a transposed 16-bit twin (wt_map: id -> (offset, pitch))
    carry its location so the optimizer kernel writes it.  int32 [n_tiles, 6] = off, ld, rows, cols, wt_off, ldwt."""
    rows_out = []
    for p, off in zip(params, offsets):
        wt = wt_map.get(id(p))
        if p.ndim == 2 and (wt is not None or p.shape[1] % 4 == 0):
            R, C = p.shape
        else:                            # vectors, and matrices whose rows are not 16-byte multiples (a LoRA factor [out, r] with
            R, C = 1, p.numel()          # r = 1, 2, 3, ...: train_svd_lora.py accepts any --rank): one flat row, no transposed twin
        if C % 4 or off % 4:
            raise ValueError(f"trainable parameter of shape {tuple(p.shape)} at flat offset {off}: the optimizer kernel needs "
                             "16-byte multiples (every SVD / LoRA parameter is, or flattens to one)")
        r0 = torch.arange(0, R, 64)
        c0 = torch.arange(0, C, 64)
        rr, cc = torch.meshgrid(r0, c0, indexing="ij")
        rr, cc = rr.reshape(-1), cc.reshape(-1)
        t = torch.empty(rr.numel(), 6, dtype=torch.int64)
        t[:, 0] = off + rr * C + cc
        t[:, 1] = C
        t[:, 2] = torch.clamp(R - rr, max=64)
        t[:, 3] = torch.clamp(C - cc, max=64)
        if wt is None:
            t[:, 4], t[:, 5] = -1, 0
        else:
            if R % 4 or wt[0] % 4 or wt[1] % 4:
                raise ValueError(f"transposed twin of a {tuple(p.shape)} weight is not 16-byte aligned")
            t[:, 4] = wt[0] + cc * wt[1] + rr
            t[:, 5] = wt[1]
        rows_out.append(t)
    tiles = torch.cat(rows_out, 0)
    assert int(tiles.max()) < 2 ** 31
    return tiles.to(torch.int32).contiguous().to(device)


class _StreamWork:
    """`.wait()` of a collective that ran on a side stream: the current stream waits for it (the shape of torch.distributed's Work)."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class DirectAllReduce:
    """The gradient sum over the ranks of ONE node as svdx_allreduce_grads: a direct reduce-scatter + all-gather over peer-mapped
    buffers (include/svdx.h; SURVEY.md 5: xGMI is point-to-point, the direct exchange drives all seven links of a GPU at once where a
    ring drives one -- ~2.6 ms against ~18 ms for 1.59 GB on 8 ranks).  The fallback for an RCCL that picks a ring.

    Construction (collective over the group): every rank exports its buffer as an IPC handle (torch.multiprocessing's reduction of
    a CUDA tensor -- hipIpcGetMemHandle underneath; HSA_ENABLE_IPC_MODE_LEGACY=0 as everywhere on this stack), gathers the others'
    and maps them; peer access is switched on by one small device-to-device copy per peer.  Needs every rank's GPU visible to every
    process (torchrun's default: no per-rank HIP_VISIBLE_DEVICES mask).

    Ordering between ranks: three cross-rank barriers per sum (gradients complete -> reduce-scatter -> all-gather -> buffers free).
    Under the RCCL backend a barrier is a one-element all-reduce -- stream-ordered, the host never blocks; under any other backend
    (the gloo rehearsals) it is `synchronize()` + `dist.barrier()`."""

    def __init__(self, buf: torch.Tensor, process_group=None, kernels=None):
        from torch.multiprocessing.reductions import reduce_tensor
        self.pg = process_group
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        if self.world > K.MAX_PEERS:
            raise RuntimeError(f"DirectAllReduce: {self.world} ranks (one node, at most {K.MAX_PEERS})")
        if not (buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous() and buf.numel() % 4 == 0 and buf.data_ptr() % 16 == 0):
            raise RuntimeError("DirectAllReduce: a contiguous, 16-byte aligned float32 CUDA buffer of a multiple of 4 elements")
        self.buf, self.n, self.k = buf, buf.numel(), kernels or K.backend()
        self.stream_ordered = dist.get_backend(process_group) == "nccl"
        fn, args = reduce_tensor(buf)
        got = [None] * self.world
        dist.all_gather_object(got, (fn, args, int(buf.device.index)), group=process_group)
        self.peers, err = [], None
        try:
            for q, (f, a, dev_q) in enumerate(got):
                if q == self.rank:
                    self.peers.append(buf)
                    continue
                if dev_q >= torch.cuda.device_count():
                    raise RuntimeError(f"rank {q} holds GPU {dev_q}, which this process cannot see ({torch.cuda.device_count()} visible)")
                if dev_q != buf.device.index and not torch.cuda.can_device_access_peer(buf.device.index, dev_q):
                    raise RuntimeError(f"GPU {buf.device.index} has no peer access to GPU {dev_q}")
                t = f(*a)                                        # the peer's buffer, mapped (a tensor on ITS device)
                if t.numel() != self.n:
                    raise RuntimeError("ranks hold buffers of different sizes")
                torch.empty(4, dtype=torch.float32, device=buf.device).copy_(t[:4])  # first touch switches peer access on (this device -> that one)
                self.peers.append(t)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 -- decided together below: a rank that raised alone would leave the others in a collective
            err = e
        ok = torch.tensor([0.0 if err is not None else 1.0], device=buf.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=process_group)
        if float(ok) != 1.0:
            self.peers = []
            raise RuntimeError(f"DirectAllReduce: peer mapping failed on {'this rank: ' + repr(err) if err is not None else 'another rank'}")
        self.ptrs = [t.data_ptr() for t in self.peers]
        self._tok = torch.zeros(1, dtype=torch.float32, device=buf.device)
        self.side = torch.cuda.Stream(device=buf.device)

    def _xbarrier(self) -> None:
        if self.stream_ordered:
            dist.all_reduce(self._tok, group=self.pg)
        else:
            torch.cuda.synchronize()
            dist.barrier(group=self.pg)

    def all_reduce(self) -> None:
        """buf <- sum over ranks of buf, on the current stream."""
        k, peers = self.k, (self.ptrs if hasattr(self.k, "lib") else self.peers)       # the test emulation takes the tensors themselves
        k.allreduce_grads(peers, self.rank, self.n, -1)
        self._xbarrier()
        k.allreduce_grads(peers, self.rank, self.n, 0)
        self._xbarrier()
        k.allreduce_grads(peers, self.rank, self.n, 1)
        self._xbarrier()

    def start(self) -> _StreamWork:
        """The same on a side stream, ordered after what the current stream holds; `.wait()` orders the current stream after it.
        (Without stream-ordered barriers the host blocks inside: there is nothing to overlap, the sum runs in place.)"""
        if not self.stream_ordered:
            self.all_reduce()
            return _StreamWork(None)
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.all_reduce()
        return _StreamWork(self.side)


class Trainer:
    def __init__(self, model: UNetSpatioTemporalConditionModel, dtype: torch.dtype = torch.float16,
                 lr: float = 1e-5, betas=(0.9, 0.999), weight_decay: float = 1e-2, eps: float = 1e-8,
                 init_scale: float = 65536.0, growth_interval: int = 2000, process_group=None,
                 grad_accum: int = 1, lora_param_dtype: Optional[str] = None):
        """lora_param_dtype: None -- the trainable parameters and AdamW's moments are fp32 whatever the activation dtype (this library's
        default; for LoRA under bf16 a measured improvement on the reference's recipe, profiles/r5_lora_dtype_deviation.json).
        "reference" -- the recipe of /root/reference/train_svd_lora.py:666-674 itself: with --mixed_precision bf16 the UNet is cast to bf16
        BEFORE add_adapter, so adapters, gradients and torch.optim.AdamW's state are bf16 tensors.  The parameters are rounded to bf16 once
        here and every optimizer step runs torch's op sequence with bf16 rounding (svdx_adamw* param_mode 1, include/svdx.h).  Only
        meaningful with dtype = bfloat16 (under fp16 the reference upcasts the adapters to fp32: cast_training_params, :672-674)."""
        if lora_param_dtype not in (None, "reference"):
            raise ValueError(f"lora_param_dtype={lora_param_dtype!r}: None or 'reference'")
        if lora_param_dtype == "reference" and dtype != torch.bfloat16:
            raise ValueError("lora_param_dtype='reference' reproduces the bf16 recipe: pass dtype=torch.bfloat16")
        self.param_mode = K.PARAMS_BF16_REFERENCE if lora_param_dtype == "reference" else K.PARAMS_F32
        self.model, self.dtype = model, dtype
        self.lr, self.betas, self.wd, self.eps = lr, betas, weight_decay, eps
        self.growth_interval = growth_interval
        self.grad_accum = grad_accum
        self.pg = process_group
        self.direct: Optional[DirectAllReduce] = None    # use_direct_allreduce(): svdx_allreduce_grads instead of RCCL for the one-collective schedules
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        dev = next(model.parameters()).device
        self.dev = dev

        self.names = select_trainable(model)
        model._flat = flatten_trainables(model, ALIGN)
        self.params, self.offsets, off, self.p_flat, self.g_flat = model._flat
        self.n_flat = off
        # one extra aligned slot at the tail carries the loss through the gradient all-reduce (replaces the
        # separate accelerator.gather of train_svd.py:1039-1040)
        self.n_total = off + ALIGN
        self.m_flat = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.v_flat = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.loss_slot = self.g_flat[off:off + 1]
        # opt_state: 0 step, 1 loss_scale, 2 growth_tracker, 3 found_inf, 4 inv_scale, 5 bc1, 6 bc2, 7 skip
        self.dynamic = dtype == torch.float16
        st = [0.0] * K.OPT_STATE_ALLOC
        st[1], st[4], st[5], st[6], st[8] = (init_scale if self.dynamic else 1.0), 1.0, 1.0, 1.0, 1.0   # slots 9..15: schedule
        self.opt_state = torch.tensor(st, dtype=torch.float32, device=dev)
        self.micro = 0
        self.schedule = dict(name="constant", num_warmup_steps=0, num_training_steps=0, num_cycles=0.0, power=1.0, lr_end=1e-7,
                             steps_per_step=1)
        src = (dist.get_global_rank(process_group, 0) if process_group is not None else 0) if self.world > 1 else 0
        if self.world > 1:
            # DistributedDataParallel's constructor broadcast (train_svd.py:815 through accelerate.prepare -> _sync_module_states): the
            # replicas start from rank 0's module state whatever each rank holds locally.  Frozen parameters and buffers first, before
            # they are packed for the kernels; the trainables travel as the flat buffer below.
            for t in list(model.parameters()) + list(model.buffers()):
                if not t.requires_grad:
                    dist.broadcast(t.data, src=src, group=process_group)
        if self.param_mode == K.PARAMS_BF16_REFERENCE:
            self.p_flat.copy_(self.p_flat.to(torch.bfloat16).to(torch.float32))      # the adapters are created as bf16 tensors there
        self._build_runtime()
        if self.world > 1:
            # ... and the trainables (peft's gaussian LoRA init uses the process-global generator: ranks do differ there)
            dist.broadcast(self.p_flat, src=src, group=process_group)
            self.weights_changed()

    def _build_runtime(self) -> None:
        """Pack the model for the kernels and derive every table that depends on the packed layout (also after `load_state`
        replaced the weights)."""
        model, dtype, dev = self.model, self.dtype, self.dev
        old = getattr(self, "rt", None)
        model.prepare(dtype)
        self.rt = model.rt
        if old is not None:              # a rebuild after load_state: keep what the caller tuned / set on the previous runtime
            for knob in ("tuner", "gemm_variant", "split_k", "fuse_geglu", "fuse_dual", "fuse_tsa"):
                setattr(self.rt, knob, getattr(old, knob))
        # AdamW walks a tile table so that it can also emit the transposed 16-bit twins the data-grad GEMMs read
        tiled = bool(self.params)
        self.adam_tiles = build_adam_tiles(self.params, self.offsets, self.rt.wt_map, dev) if tiled else None
        self.rt.adam_writes_wt = self.adam_tiles is not None
        # zero_grad(): gradients written by exactly one GEMM per step (rt.write_once: the big matrices, 99 % of the buffer) are
        # STORED by the first backward after zero_grad() instead of being zeroed and accumulated; only the slots that are summed
        # with atomics (biases, LayerNorm, skinny cross-attention weights, alignment gaps, the loss slot) are cleared, by one
        # span-table launch.  Saves a 1.6 GB memset and a 1.6 GB read per step.
        spans, pos = [], 0
        for off, p in sorted(zip(self.offsets, self.params), key=lambda t: t[0]):    # layout order (ops._layout_order)
            if id(p) in self.rt.write_once:
                if off > pos:
                    spans.append((pos, off))
                pos = off + p.numel()
        if self.n_total > pos:
            spans.append((pos, self.n_total))
        chunks = []
        for lo, hi in spans:
            lo4, hi4 = lo // 4 * 4, -(-hi // 4) * 4          # write-once matrices are multiples of 4 elements
            for c in range(lo4, hi4, 1 << 16):
                chunks.append((c, min(1 << 16, hi4 - c)))
        self.zero_spans = torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous() if self.rt.write_once else None
        # GradScaler's inf check folded into the gradient-writing kernels (single rank; Runtime.fold_finite).
        # finite_spans: the accumulated slots below n_flat (zero_spans without the loss slot at the tail, which is not a gradient)
        fin = [(c, min(n, self.n_flat - c)) for c, n in chunks if c < self.n_flat]
        self.finite_spans = torch.tensor(fin, dtype=torch.int32, device=dev).contiguous() if (self.rt.write_once and fin) else None
        self.rt.found_inf = self.opt_state[3:4] if (self.world == 1 and self.zero_spans is not None) else None
        # gradient buckets for overlapping the all-reduce with the backward sweep: one contiguous slice of g_flat per transformer
        # block (its trainables are adjacent in named_parameters order), reduced as soon as backward_rows leaves the block
        # default schedule of the gradient sum over ranks: ONE collective after the backward sweep (north_star's form).  True = one
        # all-reduce per transformer block started during the sweep; neither has met RCCL on hardware yet, and a second stream beside the
        # power-limited backward sweep bought nothing in every single-GPU experiment, so the simpler form is the default
        self.overlap = False
        self.exposed_events = None      # a list: finish_grads() records the exposed part of the collective (bench.py)
        self._pending = []              # (span, work) of the all-reduces started during THIS step's backward sweep (zero_grad clears)
        self._buckets = {}
        off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        for kind, m in model.steps:
            if kind != "attn":
                continue
            ps = [p for p in m.parameters() if p.requires_grad and id(p) in off_of]
            if ps:
                lo = min(off_of[id(p)] for p in ps)
                hi = max(off_of[id(p)] + p.numel() for p in ps)
                self._buckets[id(m)] = (lo, -(-hi // ALIGN) * ALIGN)
        spans = sorted(self._buckets.values())
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "gradient buckets must not overlap"
        # whatever the buckets leave out (alignment gaps, trainables outside transformer blocks, the loss slot at the tail)
        self._rest, pos = [], 0
        for lo, hi in spans + [(self.n_total, self.n_total)]:
            if lo > pos:
                self._rest.append((pos, lo))
            pos = max(pos, hi)

    # ---- one micro-step: fwd + loss + bwd, grads accumulate into g_flat ---------------------------------
    def forward_backward(self, unet_in, timesteps, ehs, added_time_ids, noisy_latents, target, sigmas):
        """unet_in [B,T,8,h,w]; noisy_latents/target float [B,T,4,h,w]; sigmas float [B].
        Adds this micro-batch's loss into the loss slot and its grads into the flat buffer."""
        self.forward_loss(unet_in, timesteps, ehs, added_time_ids, noisy_latents, target, sigmas)
        self.backward()

    def forward_loss(self, unet_in, timesteps, ehs, added_time_ids, noisy_latents, target, sigmas):
        """Forward sweep + EDM loss (added into the loss slot); keeps d(loss)/d(prediction) for `backward()`."""
        m, k = self.model, self.rt.k
        pred = m.forward_rows(unet_in, timesteps, ehs, added_time_ids)
        B, T, _, h, w = unet_in.shape
        dpred = self.rt.empty(B * T * h * w, m.cout_pad)
        k.zero(dpred)
        k.edm_loss(pred, m.out_channels, noisy_latents.contiguous(), target.contiguous(), sigmas.contiguous(),
                   self.loss_slot, dpred, B, T, m.out_channels, h * w, self.opt_state)
        self._dpred = dpred

    def backward(self, on_block=None, block_grads_final: bool = True):
        """Backward sweep of the last `forward_loss`.  On the last micro-batch of a multi-rank step every transformer block's
        gradient slice starts its all-reduce the moment the sweep leaves the block (`allreduce_grads` then only waits).
        `on_block(module)` replaces that hook (GraphedStep cuts its graph segments there); `block_grads_final=False` tells the sweep
        that the hook does not read the block's gradients, so the queued skinny gradient launches (Runtime.flush_deferred) may wait
        for the end of the sweep."""
        m = self.model
        dpred, self._dpred = self._dpred, None
        last = self.micro + 1 == self.grad_accum
        m.grads_ready_cb = on_block if on_block is not None else (
            self._reduce_bucket if (self.world > 1 and self.overlap and last) else None)
        m.grads_ready_flush = block_grads_final
        try:
            m.backward_rows(dpred)
        finally:
            m.grads_ready_cb = None
            m.grads_ready_flush = True
        self.micro += 1

    def _reduce_bucket(self, module) -> None:
        """Start the all-reduce of one transformer block's gradient slice while the backward sweep continues (the collective
        is ordered after the kernels already queued on the current stream, not after the ones that follow)."""
        span = self._buckets.get(id(module))
        if span is not None:
            lo, hi = span
            self._pending.append((span, dist.all_reduce(self.g_flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))

    def tune_gemms(self, batch: Dict[str, torch.Tensor], rounds: int = 1, max_steps: int = 160) -> int:
        """Measure tile shape / split-K for every GEMM problem of this model inside real forward+backward sweeps on `batch`
        (ops.GemmTuner) and freeze the fastest per problem.  Run once before timing or graph capture; no optimizer step is
        taken, gradients and the loss slot are cleared afterwards.  Returns the number of sweeps used."""
        from .ops import GemmTuner
        self.rt.tuner = GemmTuner(rounds)
        n = 0
        while n < max_steps:
            self.zero_grad()
            self.forward_loss(**batch)
            self.backward(on_block=lambda m: None, block_grads_final=False)      # no gradient all-reduce here: these sweeps are measurements, not steps
            self.micro = 0
            n += 1
            if self.rt.tuner.end_step():
                break
        if self.rt.tuner.active:
            self.rt.tuner.freeze()
        self.zero_grad()
        self.micro = 0
        return n

    def zero_grad(self):
        """After this call the next backward behaves as if every gradient were zero.  (Write-once gradients keep their stale
        values until that backward stores over them -- do not read `.grad` in between.)"""
        for _, w in self._pending:        # collectives of a step that was abandoned before allreduce_grads(): finish them first
            w.wait()
        self._pending = []
        if self.rt.fold_finite and self.rt.found_inf is not None:
            self.rt.k.zero(self.rt.found_inf)       # raised by the gradient-writing kernels of THIS step only (a sweep abandoned before its
                                                    # optimizer step must not make the next one skip)
        if self.zero_spans is None:
            self.rt.k.zero(self.g_flat)
        else:
            self.rt.k.zero_spans(self.g_flat, self.zero_spans, self.zero_spans.shape[0])
            self.rt.grads_fresh = True          # consumed by the next backward_rows (Trainer.backward or loss.backward())

    # ---- gradient mean over ranks: ONE collective on the flat buffer -----------------------------------
    def allreduce_grads(self, async_op: bool = False):
        """Finish the gradient sum over ranks: wait for the buckets started during the backward sweep and reduce what they did
        not cover (always the loss slot); without overlap, ONE collective on the whole flat buffer."""
        if self.world == 1:
            return None
        if not self._pending:
            if self.direct is not None:
                w = self.direct.start()
                return w if async_op else w.wait()
            return dist.all_reduce(self.g_flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        done = {span for span, _ in self._pending}
        for span in self._buckets.values():          # a block whose backward was pruned never reported: reduce it now
            if span not in done:
                self._pending.append((span, dist.all_reduce(self.g_flat[span[0]:span[1]], op=dist.ReduceOp.SUM, group=self.pg,
                                                            async_op=True)))
        for lo, hi in self._rest:
            self._pending.append(((lo, hi), dist.all_reduce(self.g_flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))
        for _, w in self._pending:
            w.wait()
        self._pending = []
        return None

    def use_direct_allreduce(self, on: bool = True, verify: bool = True) -> bool:
        """Route the one-collective schedules (`overlap = False`) through svdx_allreduce_grads (DirectAllReduce) instead of RCCL's
        all-reduce.  Collective: every rank of the group calls it.  The per-block buckets stay on RCCL.
        verify: before the path is adopted, two rounds of fresh random data go through it and through the library's all-reduce; every
        rank votes (MIN over the group) and on any disagreement -- or any rank's exception -- ALL ranks stay on the library
        (`self.direct_check` says why).  A stale read of a peer-mapped buffer would otherwise corrupt gradients silently.
        Returns whether the direct path is in use."""
        self.direct = None
        if not (on and self.world > 1):
            return False
        if getattr(self, "_direct_verified", None) is not None:      # mapped and verified before (every rank took the same decision then)
            self.direct = self._direct_verified
            return True
        ok, why, direct = 1.0, None, None
        try:
            direct = DirectAllReduce(self.g_flat, self.pg, self.rt.k)
        except Exception as e:  # noqa: BLE001 -- its constructor has already agreed with the other ranks (MIN) that the mapping failed
            self.direct_check = {"agrees_with_library": False, "error": repr(e)[:200]}
            return False
        if verify:
            keep = self.g_flat.clone()
            for rnd in range(2):                           # fresh data twice: a stale read of a peer's previous contents would show
                err = None
                gsrc = torch.randn(self.g_flat.numel(), device=self.g_flat.device,
                                   generator=torch.Generator(device=self.g_flat.device).manual_seed(31 * rnd + dist.get_rank(self.pg)))
                try:                                       # a rank-local failure must not leave the others alone in the collectives below
                    self.g_flat.copy_(gsrc)
                    direct.all_reduce()
                    got = self.g_flat.clone()
                except Exception as e:  # noqa: BLE001
                    err, got = repr(e)[:200], None
                dist.all_reduce(gsrc, op=dist.ReduceOp.SUM, group=self.pg)
                if err is not None:
                    ok, why = 0.0, err
                elif not torch.allclose(got, gsrc, rtol=1e-4, atol=1e-5):
                    ok, why = 0.0, f"round {rnd}: direct sum differs from the library's by {float((got - gsrc).abs().max()):.3e}"
            self.g_flat.copy_(keep)
            flag = torch.tensor([ok], device=self.g_flat.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            agreed = bool(flag.item() == 1.0)
            self.direct_check = {"agrees_with_library": agreed, "rounds": 2}
            if why is not None:
                self.direct_check["error"] = why
            if not agreed:
                return False
            self._direct_verified = direct
        self.direct = direct
        return True

    def finish_grads(self, side_work=None) -> None:
        """Complete the gradient sum over ranks, with `side_work()` issued on the compute stream while the collective is in flight.
        north_star's schedule: ONE all-reduce of the flat gradient buffer started when the backward sweep ends, the frozen
        conditioners of the NEXT micro-batch (VAE encode + CLIP embed, which the reference runs at the top of its next iteration,
        train_svd.py:948, :975) beside it, AdamW after the wait -- `trainer.overlap = False` + a `side_work`.  With the per-block
        buckets (`overlap = True`) most of the sum already ran under the backward sweep; the side work then covers the tail.
        When `self.exposed_events` is a list, the (side work queued, collective waited for) event pair of the step is appended: the
        time between them is the part of the collective nothing hid."""
        if self.world == 1:
            if side_work is not None:
                side_work()
            return
        if self._pending:                                 # bucket mode: start what the sweep did not cover, then the side work
            done = {span for span, _ in self._pending}
            for span in list(self._buckets.values()) + self._rest:
                if span not in done and span[1] > span[0]:
                    self._pending.append((span, dist.all_reduce(self.g_flat[span[0]:span[1]], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))
            works = [w for _, w in self._pending]
        elif self.direct is not None:
            works = [self.direct.start()]
        else:
            works = [dist.all_reduce(self.g_flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)]
        if side_work is not None:
            side_work()
        ev = None
        if self.exposed_events is not None and self.g_flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.exposed_events.append(ev)
        self._pending = []

    # ---- optimizer step (unscale + inf check + AdamW + re-pack), no host sync -----------------------------
    def optimizer_step(self):
        k = self.rt.k
        n = self.n_flat
        if self.rt.fold_finite and self.rt.found_inf is not None and not self.rt.unchecked_grads:
            # the write-once gradients were tested by the kernels that stored them (ops.gemm_tn_acc -> opt_state[3]); what is left are
            # the accumulated slots.  On several ranks the SUM over ranks has to be tested: the full pass, after the all-reduce.
            if self.finite_spans is not None:
                k.check_finite_spans(self.g_flat, self.finite_spans, self.finite_spans.shape[0], self.opt_state)
        else:
            k.check_finite(self.g_flat, n, self.opt_state)
        self.rt.unchecked_grads = False
        k.optim_prep(self.opt_state, self.betas[0], self.betas[1], 2.0, 0.5, self.growth_interval, int(self.dynamic))
        grad_mul = 1.0 / (self.world * self.grad_accum)
        if self.adam_tiles is not None:
            k.adamw_tiled(self.p_flat, self.g_flat, self.m_flat, self.v_flat, self.adam_tiles, self.adam_tiles.shape[0], self.lr,
                          self.betas[0], self.betas[1], self.eps, self.wd, grad_mul, self.opt_state, self.rt.w16_flat,
                          self.rt.wt16_flat, param_mode=self.param_mode)
        else:
            k.adamw(self.p_flat, self.g_flat, self.m_flat, self.v_flat, n, self.lr, self.betas[0], self.betas[1],
                    self.eps, self.wd, grad_mul, self.opt_state, self.rt.w16_flat, param_mode=self.param_mode)
        self.model.refresh_trainable(masters_changed_on_host=False)
        self.micro = 0

    def step(self, batch, side_work=None) -> None:
        """Whole optimizer step: zero, fwd/bwd of the `grad_accum` micro-batches (one dict, or a sequence of them -- config 4 of
        the reference runs gradient_accumulation_steps = 2), gradient sum over ranks (per-block buckets under the last backward
        sweep, or one collective under `side_work` -- see `finish_grads`), AdamW."""
        batches = [batch] if isinstance(batch, dict) else list(batch)
        if len(batches) != self.grad_accum:
            raise ValueError(f"expected {self.grad_accum} micro-batch(es), got {len(batches)}")
        self.zero_grad()
        for b in batches:
            self.forward_backward(**b)
        self.finish_grads(side_work)
        self.optimizer_step()

    # ---- learning-rate schedule, evaluated on the device from the optimizer's step counter --------------------
    def set_schedule(self, name: str, num_warmup_steps: int = 0, num_training_steps: int = 0, num_cycles: float = 0.0,
                     power: float = 1.0, lr_end: float = 1e-7, steps_per_step: int = 1, step_rules: Optional[str] = None) -> None:
        """Install lambda(step) of diffusers' get_scheduler (train_svd.py:807-813) into opt_state[9..15] (include/svdx.h).
        `steps_per_step`: scheduler steps per optimizer step (accelerate steps the wrapped scheduler num_processes times)."""
        if name not in K.SCHED_KINDS:
            raise ValueError(f"unknown lr schedule {name!r} (have {sorted(K.SCHED_KINDS)})")
        vals = [float(K.SCHED_KINDS[name]), float(num_warmup_steps), float(num_training_steps), float(num_cycles), float(power),
                float(lr_end) / float(self.lr), float(steps_per_step)]
        extra = [0.0] * (K.OPT_STATE_ALLOC - K.OPT_STATE_FLOATS)
        if name == "piecewise_constant":
            from .optimization import parse_step_rules
            bounds, mults, last = parse_step_rules(step_rules or "")
            if len(bounds) > K.SCHED_MAX_RULES:
                raise ValueError(f"piecewise_constant: at most {K.SCHED_MAX_RULES} step rules")
            vals[1] = float(len(bounds))
            for i, (b, m) in enumerate(zip(bounds, mults)):
                extra[2 * i], extra[2 * i + 1] = b, m
            extra[2 * len(bounds)] = last
        self.opt_state[9:16] = torch.tensor(vals, dtype=torch.float32).to(self.dev)
        self.opt_state[K.OPT_STATE_FLOATS:] = torch.tensor(extra, dtype=torch.float32).to(self.dev)
        self.schedule = dict(name=name, num_warmup_steps=num_warmup_steps, num_training_steps=num_training_steps,
                             num_cycles=num_cycles, power=power, lr_end=lr_end, steps_per_step=steps_per_step)
        if name == "piecewise_constant":
            self.schedule["step_rules"] = step_rules

    def weights_changed(self) -> None:
        """The float masters of the trainables were written by something other than `optimizer_step` (EMA copy_to / restore,
        a loaded checkpoint): refresh the 16-bit copies the kernels read, including the transposed twins AdamW maintains."""
        saved, self.rt.adam_writes_wt = self.rt.adam_writes_wt, False
        try:
            self.model.refresh_trainable(masters_changed_on_host=True)
        finally:
            self.rt.adam_writes_wt = saved

    # ---- resume (`accelerator.save_state` / `load_state`, train_svd.py:698-729, 900-924, 1088-1090) -----------------
    def save_state(self, output_dir: str, ema=None, scheduler=None) -> None:
        from .checkpoint import save_state
        save_state(self, output_dir, ema=ema, scheduler=scheduler)

    def load_state(self, input_dir: str, ema=None, scheduler=None) -> None:
        from .checkpoint import load_state
        load_state(self, input_dir, ema=ema, scheduler=scheduler)

    def last_loss(self) -> torch.Tensor:
        """Mean (unscaled) loss over ranks/micro-batches of the last reduced step (device scalar; read it
        before the next zero_grad)."""
        return self.loss_slot / (self.world * self.grad_accum)


class GraphedStep:
    """One optimizer step (all `grad_accum` micro-batches) replayed from hipGraphs, with the gradient all-reduce overlapped.

    The step is captured once, on fixed input tensors (one dict per micro-batch), as a CHAIN of graphs sharing one memory pool: the
    first holds zero_grad, the whole forward + backward of every micro-batch but the last, then the last one's forward sweep, loss
    and backward sweep down to the first transformer block; every later graph holds the backward sweep between two transformer
    blocks; one more holds the optimizer.  Replaying them in order is the whole step, and between two replays -- outside any graph,
    so RCCL is never captured -- the slice of the flat gradient buffer that the previous segment completed starts its all-reduce
    (async: it runs beside the next segments' kernels).  Gradients are only reduced on the last micro-batch, as with
    `accelerator.accumulate` (train_svd.py:941).  On one rank the collectives vanish and the chain is just the step."""

    def __init__(self, trainer: "Trainer", batch, cut_blocks: Optional[bool] = None, record_plan: bool = False):
        """cut_blocks: cut the chain at every transformer block (None: only when there is a collective to interleave, i.e. on
        several ranks with `trainer.overlap`; True lets one rank rehearse the multi-rank chain).
        record_plan: also keep the captured launches as a C-replayable plan (include/svdx.h svdx_plan_*): `replay_plan()` then runs the
        step through svdx_plan_replay -- ctypes into C, no torch graph -- on the tensors of this object's memory pool.  One rank only (a
        plan holds kernel launches, not collectives)."""
        if record_plan and trainer.world > 1:
            raise ValueError("a launch plan holds kernel launches only: record it on one rank")
        self.plan = None
        batches = [batch] if isinstance(batch, dict) else list(batch)
        if cut_blocks is None:
            cut_blocks = trainer.world > 1 and trainer.overlap
        if len(batches) != trainer.grad_accum:
            raise ValueError(f"expected {trainer.grad_accum} micro-batch(es), got {len(batches)}")
        self.tr = trainer
        self.opt_beside_side_work = True                 # __call__(side_work): the optimizer graph on a second stream beside the side work (one rank)
        self._opt_stream = None
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.spans: List[list] = []                      # per graph segment: the flat-buffer slices whose gradients it completed
        tr = trainer
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())

        def sweep(on_last_block):
            tr.zero_grad()
            for b in batches[:-1]:
                tr.forward_loss(**b)
                tr.backward(on_block=lambda m: None, block_grads_final=False)
            tr.forward_loss(**batches[-1])
            tr.backward(on_block=on_last_block, block_grads_final=cut_blocks)    # a segment cut is followed by the block's collective

        with torch.cuda.stream(s):                       # warm-up on the capture stream (allocator, lazy module state)
            sweep(lambda m: None)
            tr.allreduce_grads()                         # the warm-up pass is a real step: keep the replicas identical
            tr.optimizer_step()
            torch.cuda.synchronize()
            # thread_local: RCCL's watchdog thread may query events while we capture (world > 1)
            pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            self.graphs.append(g)
            if record_plan:
                tr.rt.k.plan_begin()                     # every libsvdx launch from here to the optimizer's last one, in order
            try:
                calls = lambda: getattr(tr.rt.k, "n_calls", None)       # launches issued so far (None: a backend that does not count)
                mark = [calls()]

                def cut(module):
                    if not cut_blocks:
                        return                           # nothing to interleave: the whole sweep stays one graph
                    span = tr._buckets.get(id(module))
                    if mark[0] is not None and calls() == mark[0] and self.spans:
                        # no launch since the previous cut (the sweep left two blocks back to back): an empty segment would be a
                        # wasted replay -- this block's slice joins the previous segment's collective instead
                        if span is not None:
                            self.spans[-1].append(span)
                        return
                    self.graphs[-1].capture_end()
                    self.spans.append([span] if span is not None else [])
                    g2 = torch.cuda.CUDAGraph()
                    g2.capture_begin(pool=pool, capture_error_mode="thread_local")
                    self.graphs.append(g2)
                    mark[0] = calls()
                sweep(cut)
            finally:
                empty = mark[0] is not None and calls() == mark[0] and len(self.graphs) > 1
                with warnings.catch_warnings():
                    if empty:                            # torch warns about ending an empty capture; this one is dropped, not replayed
                        warnings.simplefilter("ignore")
                    self.graphs[-1].capture_end()
                if empty:
                    self.graphs.pop()                    # nothing was launched after the last cut
            self.g_opt = torch.cuda.CUDAGraph()
            self.g_opt.capture_begin(pool=pool, capture_error_mode="thread_local")
            try:
                tr.optimizer_step()
            finally:
                self.g_opt.capture_end()
                if record_plan:
                    self.plan = tr.rt.k.plan_end()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def replay_plan(self) -> None:
        """The same optimizer step as `__call__()` on one rank, issued by svdx_plan_replay on the current stream instead of hipGraphLaunch."""
        if self.plan is None:
            raise RuntimeError("GraphedStep(..., record_plan=True) first")
        self.plan.replay()

    def __call__(self, side_work=None) -> None:
        """side_work: callable queued between the backward sweep and the optimizer, beside the gradient collective
        (`Trainer.finish_grads`) -- e.g. the replay of the next micro-batch's VAE-encode graph."""
        tr = self.tr
        multi = tr.world > 1
        for i, g in enumerate(self.graphs):
            g.replay()
            if multi and tr.overlap and i < len(self.spans):
                for lo, hi in self.spans[i]:             # the gradient slices the segment just completed
                    tr._pending.append(((lo, hi), dist.all_reduce(tr.g_flat[lo:hi], op=dist.ReduceOp.SUM, group=tr.pg, async_op=True)))
        if side_work is not None and not multi and self.opt_beside_side_work:
            # one rank: nothing separates the optimizer from the side work (the next clip's frozen conditioners: they neither read nor write
            # anything AdamW touches), and the two want different things of the chip -- AdamW streams 12.6 GB through HBM on no matrix pipe,
            # the VAE encoder's convolutions live on the matrix pipes and the LDS.  The optimizer graph runs on a second stream beside
            # the side work; the step ends when both have (round 6; TrainLoop(overlap_optimizer=False) / bench --serial-optimizer: in a row)
            cur = torch.cuda.current_stream()
            if self._opt_stream is None:
                self._opt_stream = torch.cuda.Stream(device=tr.dev)
            self._opt_stream.wait_stream(cur)
            with torch.cuda.stream(self._opt_stream):
                self.g_opt.replay()
            side_work()
            cur.wait_stream(self._opt_stream)
            return
        tr.finish_grads(side_work)
        self.g_opt.replay()


def edm_prepare(latents, noise, cond_latents, sigmas):
    """UNet inputs of train_svd.py:964-972, 1014-1017 from latents (host-side data prep of the train script).
    Returns unet_in [B,T,8,h,w], timesteps [B], noisy_latents."""
    T = latents.shape[1]
    s = sigmas.to(latents)[:, None, None, None, None]
    noisy = latents + noise * s
    timesteps = 0.25 * sigmas.to(torch.float32).log()
    inp = noisy / ((s ** 2 + 1) ** 0.5)
    cond = cond_latents.unsqueeze(1).repeat(1, T, 1, 1, 1)
    return torch.cat([inp, cond], dim=2), timesteps, noisy


def conditioning_dropout(random_p, encoder_hidden_states, conditional_latents, prob: float):
    """Classifier-free-guidance dropout of train_svd.py:992-1011 (host-side data prep, plain tensor ops): `random_p` [B] uniform in
    [0, 1); the image embedding [B, D] -> [B, 1, D], zeroed where p < 2 prob; the conditioning latents zeroed where prob <= p < 3 prob."""
    bsz = random_p.shape[0]
    ehs = encoder_hidden_states.unsqueeze(1) * (random_p >= 2 * prob).to(encoder_hidden_states.dtype).reshape(bsz, 1, 1)
    keep = 1 - ((random_p >= prob) & (random_p < 3 * prob)).to(conditional_latents.dtype)
    return ehs, keep.reshape(bsz, 1, 1, 1) * conditional_latents

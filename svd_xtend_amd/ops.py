"""Host-side operators of the SVD UNet hot path: each op owns its packed weights and has an explicit
`fwd` / `bwd` made of libsvdx kernel launches (svd_xtend_amd.kernels).  No torch.autograd, no ATen math on
activations: torch is used for allocation, views, one-off weight re-layout at load time and a few tiny-tensor corner
cases noted inline (padded LoRA rank, LoRA scale != 1 on the per-clip cross-attention vectors).

Layout: activations are [rows, C] row-major with rows = (b, t, y, x) -- the "(B*T, HW, C)" layout of
SURVEY.md section 7 -- so transformer linears need no permute, 2-D convs are implicit GEMMs over rows and
temporal ops address rows with stride HW.
"""
from __future__ import annotations

import contextlib
import functools
import os
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import kernels as K

GN_GROUPS = 32


def rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _layout_order(names: List[str]) -> List[int]:
    """Order in which the trainables are laid out in the flat buffer.  named_parameters() order, except that the LoRA factors
    of one attention module are regrouped as A_q, A_k, A_v, A_out, B_q, B_k, B_v, B_out: the stacked q/k/v factors are then
    contiguous blocks whose 16-bit twins (written by AdamW) are GEMM operands as they stand -- no per-step re-packing."""
    first, keys = {}, []
    for i, n in enumerate(names):
        if ".lora_A." in n or ".lora_B." in n:
            g = first.setdefault(n.split(".to_")[0], i)
            keys.append((g, 0 if ".lora_A." in n else 1, i))
        else:
            keys.append((i, 0, i))
    return sorted(range(len(names)), key=lambda i: keys[i])


def flatten_trainables(model: "nn.Module", align: int = 64):
    """Re-home every trainable parameter (and its .grad) as a view of one flat float buffer -- adjacent q/k/v weights then form
    one [3C, C] block for the fused weight-grad GEMM, and the whole set is one all-reduce / one AdamW launch.  Returns
    (params, offsets, n_flat, p_flat, g_flat) with params in named_parameters() order (the order of the reference's optimizer
    parameter list) and offsets following `_layout_order`; the buffers carry one extra aligned slot at the tail (the loss rides
    there through the gradient all-reduce)."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    params = [p for _, p in named]
    dev = params[0].device if params else next(model.parameters()).device
    offs, off = [0] * len(params), 0
    for i in _layout_order([n for n, _ in named]):
        offs[i] = off
        off = rup(off + params[i].numel(), align)
    p_flat = torch.zeros(off + align, dtype=torch.float32, device=dev)
    g_flat = torch.zeros(off + align, dtype=torch.float32, device=dev)
    for p, o in zip(params, offs):
        n = p.numel()
        p_flat[o:o + n].copy_(p.data.reshape(-1))
        p.data = p_flat[o:o + n].view(p.shape)
        p.grad = g_flat[o:o + n].view(p.shape)
    return params, offs, off, p_flat, g_flat


class Runtime:
    """Per-model execution context: backend, activation dtype, device, scratch allocation."""

    def __init__(self, dtype: torch.dtype, device: torch.device):
        self.dt = dtype
        self.dev = device
        self.k = K.backend()
        self.gemm_variant = 4      # 0 register-staged reference, 1 global_load_lds, 4 production (lean buffer_load-lds loop, 128x160 tiles)
        self.split_k = True
        self.fuse_geglu = True
        # temporal self-attention op (norm1 -> q/k/v -> attention over frames -> out-projection + residual) as one launch when
        # the level qualifies (csrc/tsa.hip).  This and the switches below are plain attributes (no environment variable selects a kernel):
        # tools/ab_inproc.py flips them between two captures of the step in one process -- every one is ON because it measured faster
        self.fuse_tsa = True
        self.fuse_dual = True       # False: the LoRA adapter term as its own accumulate launch
        self.lora_stack_da = True   # False: dA of fused q/k/v adapters as three TN GEMMs
        self.tuner = None           # GemmTuner (Trainer.tune_gemms): measured tile / split-K per GEMM problem
        # transposed 16-bit twins ([K,N], operand of the data-grad GEMM) of the trainable nn.Linear weights live in one arena so
        # that the tiled AdamW kernel can write them (wt_map: id(weight) -> (element offset of W^T[0, n0], row pitch))
        self.wt16_flat = None
        self.wt_pos = 0
        self.wt_map = {}
        self.adam_writes_wt = False  # set by the Trainer: LinearOp.refresh then has nothing to re-transpose
        # weight gradients produced by ONE GEMM per step can be stored instead of accumulated on the first micro-batch: no
        # zeroed destination, no read of it.  write_once: ids of the parameters whose gradient has that property.
        self.grad_overwrite = False
        self.grads_fresh = False     # Trainer.zero_grad(): the next backward sweep stores the write-once gradients
        self.write_once = set()
        self.arenas = [None, None]
        self.arena_cap = 1 << 20    # floats
        self.arena_cur, self.arena_pos = None, 0
        # launches of a few us whose results nothing in the sweep reads (skinny weight gradients of the cross-attention value path,
        # the affine-gradient reductions of LayerNorm) are queued and run as table-driven launches when the sweep -- or, with gradient
        # buckets, the transformer block -- ends (False: one launch each, as before round 3)
        self.batch_small = True
        # at one clip per rank the gradient of a temporal block's cross-attention vector IS colsum(d(h1)) = the bias gradient the
        # attn1.to_out weight-gradient GEMM already computes on the matrix pipe: no svdx_colsum pass
        self.dvec_from_dw = True
        self._q_nn, self._q_outer, self._q_ln, self._q_M, self._q_outer_dst = [], [], [], None, set()
        self._q_fin, self._q_fin_dst, self._q_fin_bytes = [], set(), 0
        # the reducing launches of the row-sliced weight-gradient GEMMs wait for one table-driven launch at the end of the sweep (or of
        # the transformer block, with gradient buckets)
        self.defer_grad_finalize = True
        # (round 5's L2 prefetch of the weight-gradient kernels cost 0.69 ms in the step -- profiles/r5_ab_tn.txt -- and left the library in round 6)
        # svdx_gemm_tn / svdx_grad_finalize_batch raise this one-float flag (the trainer's opt_state[3]) for the write-once gradients they
        # store, so the optimizer only has to test the accumulated slots (Trainer.optimizer_step); None: nobody folds, the full pass runs
        self.found_inf = None
        self.fold_finite = True     # False: the 1.59 GB svdx_check_finite pass of rounds 1-4
        self.unchecked_grads = False
        self.fin_queue_budget = 768 << 20      # bytes of float slabs the queue may keep alive before it flushes (c2: ~3 flushes per sweep)
        # GroupNorm statistics of a tensor come from the store loop of the GEMM that writes it (svdx_gemm_gn) instead of a pass of their
        # own over it
        self.fuse_gn_stats = True
        self.p_flat = None          # flat float master buffer of the trainables (ops.flatten_trainables)
        self.w16_flat = None        # same layout in the activation dtype, written by svdx_adamw / one cast per refresh

    # ---- measurement aid: named regions of the sweep (bench.py brackets them with events to report an OP made of several launches) ----
    on_region = None              # callable(name, info dict, begin: bool) or None

    @contextlib.contextmanager
    def region(self, name: str, **info):
        cb = self.on_region
        if cb is None:
            yield
            return
        cb(name, info, True)
        try:
            yield
        finally:
            cb(name, info, False)

    # ---- deferred skinny launches (the queued jobs hold their tensors alive until the flush) -------------------------------------
    def _q_rows(self, M: int) -> None:
        if self._q_M is not None and self._q_M != M:
            self.flush_deferred()
        self._q_M = M

    def defer_nn(self, job, M: int, stage: int = 0) -> None:
        """job of kernels.small_linear_batch(trans=1); a job may read what jobs of LOWER stages wrote (one launch per stage)"""
        self._q_rows(M)
        while len(self._q_nn) <= stage:
            self._q_nn.append([])
        self._q_nn[stage].append(job)

    def defer_outer(self, job, M: int) -> None:
        """job of kernels.outer_acc_batch; runs after every queued defer_nn stage.  The jobs of one table run concurrently and add
        into their destination without atomics, so a destination may be queued ONCE per table: a second job on the same gradient
        (tied weights, a second sweep queued before a flush) first flushes what is queued."""
        self._q_rows(M)
        dst = job[2].data_ptr()
        if dst in self._q_outer_dst:
            self.flush_deferred()
            self._q_M = M
        self._q_outer_dst.add(dst)
        self._q_outer.append(job)

    def defer_ln_reduce(self, job) -> None:
        """job of kernels.ln_param_reduce_batch"""
        self._q_ln.append(job)

    def defer_grad_finalize_job(self, job) -> None:
        """job of kernels.grad_finalize_batch (the reducing launch of a row-sliced weight-gradient GEMM).  A destination may be queued
        once per table (the jobs of a launch run concurrently): a second job on it first flushes what is queued."""
        dst = job[3].data_ptr()
        nbytes = job[0].numel() * job[0].element_size()
        # the queued float slabs stay alive until the flush (16-32 slices x the gradient's size at the 64x40 level, and the temporal
        # blocks of that level run first in the backward sweep, while every saved activation is still live): bound what the queue holds
        if dst in self._q_fin_dst or (job[6] is not None and job[6].data_ptr() in self._q_fin_dst) or \
                self._q_fin_bytes + nbytes > self.fin_queue_budget:
            self.flush_deferred()
        self._q_fin_bytes += nbytes
        self._q_fin_dst.add(dst)
        if job[6] is not None:
            self._q_fin_dst.add(job[6].data_ptr())
        self._q_fin.append(job)

    def flush_deferred(self) -> None:
        k, M = self.k, self._q_M
        # the reducing launch FIRST: with `dvec_from_dw` the column sums it leaves (d(cross-attention vector) = the bias gradient of
        # attn1.to_out's weight-gradient GEMM) are what the skinny gradient chain below reads
        if self._q_fin:
            k.grad_finalize_batch(self._q_fin)
        for jobs in self._q_nn:
            if jobs:
                k.small_linear_batch(jobs, M, 1)
        if self._q_outer:
            k.outer_acc_batch(self._q_outer, M)
        if self._q_ln:
            k.ln_param_reduce_batch(self._q_ln)
        self.drop_deferred()

    def drop_deferred(self) -> None:
        self._q_nn, self._q_outer, self._q_ln, self._q_M, self._q_outer_dst = [], [], [], None, set()
        self._q_fin, self._q_fin_dst, self._q_fin_bytes = [], set(), 0

    @property
    def deferred_pending(self) -> bool:
        return bool(any(self._q_nn) or self._q_outer or self._q_ln or self._q_fin)

    def begin_pass(self, which: int) -> None:
        """Start of a forward (0) or backward (1) sweep: re-zero that sweep's statistics arena with ONE memset (GroupNorm
        statistics are accumulated with atomics; ~200 per-op memsets per step otherwise).  Forward statistics are saved
        for the backward, so the two sweeps use separate arenas."""
        if self.arenas[which] is None:
            self.arenas[which] = torch.empty(self.arena_cap, dtype=torch.float32, device=self.dev)
        self.k.zero(self.arenas[which])
        self.arena_cur, self.arena_pos = which, 0

    def take_zeroed(self, n: int):
        """-> (tensor of n zeroed floats, prezeroed flag)"""
        n64 = rup(n, 64)
        arena = self.arenas[self.arena_cur] if self.arena_cur is not None else None
        if arena is not None and self.arena_pos + n64 <= self.arena_cap:
            t = arena[self.arena_pos:self.arena_pos + n]
            self.arena_pos += n64
            return t, 1
        return self.f32(n), 0

    def act_view(self, master: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """The activation-dtype twin of a contiguous float tensor living inside p_flat (None otherwise)."""
        if master is None or self.p_flat is None:
            return None
        off = (master.data_ptr() - self.p_flat.data_ptr()) // 4
        if master.untyped_storage().data_ptr() != self.p_flat.untyped_storage().data_ptr() or off < 0:
            return None
        return self.w16_flat[off:off + master.numel()]

    def empty(self, *shape, dtype=None) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype or self.dt, device=self.dev)

    def f32(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def zeros_f32(self, *shape) -> torch.Tensor:
        t = self.f32(*shape)
        self.k.zero(t)
        return t


def choose_split(rt: "Runtime", M: int, N: int, Kd: int, ldc: int, bn: int = 0) -> int:
    """Split-K factor for bn-wide two-stage output tiles (0: the kernel's default width for this N) when the grid cannot fill the chip."""
    bn = bn or (160 if N % 160 == 0 else 128)
    tiles = ((M + 127) // 128) * ((N + bn - 1) // bn)
    kt = Kd // 64
    split = 1
    if rt.split_k and tiles <= 192 and kt >= 16 and N % 4 == 0 and ldc % 4 == 0:
        # every split keeps >= 16 K-steps (measured: below that the float-slab round trip costs more than the idle CUs), except
        # for the 5x8-level grids of <= 64 tiles where even 4-step splits pay
        min_k = 4 if tiles <= 64 else 16
        split = max(1, min(512 // tiles, kt // min_k, 16))
        while split > 1 and (kt + split - 1) // split * (split - 1) >= kt:     # every split must own >= 1 K-tile
            split -= 1
    return split


# svdx_gemm tile variants (csrc/gemm.hip): rows x columns of the output tile, LDS stages of the K-loop, waves per workgroup
TILE_OF_VARIANT = {7: (128, 160, 2, 4), 6: (160, 160, 2, 4), 8: (128, 128, 2, 4),
                   16: (256, 160, 3, 8), 17: (256, 128, 3, 8), 18: (256, 256, 2, 8), 20: (128, 160, 4, 4), 21: (128, 128, 4, 4),
                   23: (192, 160, 3, 8), 22: (192, 128, 3, 8), 25: (96, 160, 4, 4), 24: (96, 128, 4, 4)}
# instantiated in csrc/gemm.hip and offered to the in-situ tuner (bench.py --tune), but not to the cost model:
#   27 / 28: two-stage eight-wave tiles without a measured rate;
#   32 / 34: round 6's two-role eight-wave tiles (gemm_v5_kernel: 256 x 256 and 160 x 320).  Isolated they are the fastest kernels of the library
#   (8192^3: 1397 TF/s against 1297 for tile 18; the 64x40-level convolutions 5-20 % ahead of tile 6), inside the step they only TIE with
#   the two-per-CU four-wave tiles (in-situ sweep: +-3 % per problem, one problem -10 %; cost-model selection +0.45 ms / step):
#   profiles/r6e_tune_dump_reworked.txt, DESIGN.md 6.4.  A 160 x 320 two-role workgroup IS two 160 x 160 workgroups side by side.
#   36: round 6's 144 x 160 SIX-wave two-stage tile whose row tiles are 140 apart (two workgroups per CU): 35840 = 256 x 140 and 8960 = 64 x 140,
#   so N = 320 at the 64x40 level / N = 1280 at the 32x20 level launch exactly 512 workgroups -- every slot of the chip -- where the
#   160-row tile of variant 6 fills 448.  The entry's first number is the row STEP (what tile counts follow); it computes 144 rows.
#   Isolated it is 3-12 % ahead of tile 6 on the N = 320 problems of the 64x40 level (profiles/r6m_ring_time_tile36.txt); selected for every
#   (1, 6) choice whose grid it fills in one round it moved the step by -0.07 ms (profiles/r6n_ab_tile36.txt: 48.01 against 48.08 ms) -- inside the
#   step those launches wait for their cold operands, not for workgroup slots.  A tuner candidate only.
STAGED_TILES = {27: (128, 128, 2, 8), 28: (128, 160, 2, 8), 32: (256, 256, 2, 8), 34: (160, 320, 2, 8), 36: (140, 160, 2, 6)}
# TFLOP/s one CU sustains on a variant's K-loop when the CU is full (8192^3 runs of tools/ring_check.py divided by 256 CUs, trimmed by
# the in-situ sweeps of bench.py --tune): the two-stage four-wave tiles need two workgroups per CU for it
_TILE_RATE = {6: 4.05, 7: 3.5, 8: 3.5, 16: 4.4, 17: 4.0, 18: 3.6, 20: 2.75, 21: 2.5, 22: 3.8, 23: 3.8, 24: 2.4, 25: 2.05}
_ALONE, _FILL_STEPS, _EPI_US, _FIN_US, _FIN_BYTES_PER_US = 0.5, 1.5, 3.0, 12.0, 6.0e6


def _xcd_block_tiles(Tm: int, Tn: int, a_bytes: float, b_bytes: float) -> int:
    """Tiles owned by the fullest XCD under launch_gemm_v4's arrangement of the 8 XCDs over the tile grid (csrc/gemm.hip).  (With 2 / 4 / 8
    K slices the launcher may give every slice XCDs of its own -- `z_xcd`, round 4 -- which fills no worse by construction; the model
    keeps the slice-agnostic count its rates were fitted with.)"""
    best_eff, best = 0.0, None
    for xn in (1, 2, 4, 8):
        sm, sn = -(-Tm // (8 // xn)), -(-Tn // xn)
        best_eff = max(best_eff, Tm * Tn / (8.0 * sm * sn))
    for xn in (1, 2, 4, 8):
        sm, sn = -(-Tm // (8 // xn)), -(-Tn // xn)
        eff, cost = Tm * Tn / (8.0 * sm * sn), a_bytes * xn + b_bytes * (8 // xn)
        if eff >= 0.9 * best_eff and (best is None or cost < best[0]):
            best = (cost, min(sm, Tm) * min(sn, Tn))
    return best[1]


def estimate_gemm_us(M: int, N: int, Kd: int, split: int, variant: int, cin: int = 0) -> float:
    """Cost model behind `choose_cfg`: rounds of workgroups on the fullest XCD x (K-steps + pipeline fill) x time per K-step of the
    tile, + the epilogue, + the float-slab round trip of a split reduction.  Fitted to two in-situ sweeps of round 3 (108 problems of
    the 14 x 512 x 320 step: its picks cost 0.6 % more than the measured best of every problem)."""
    bm, bn, stages, waves = TILE_OF_VARIANT[variant]
    Tm, Tn = -(-M // bm), -(-N // bn)
    two_stage = stages == 2 and waves == 4
    per_xcd = 64 if two_stage else 32                       # resident workgroups of one XCD's 32 CUs
    block = _xcd_block_tiles(Tm, Tn, 2.0 * M * (2 * cin if cin else Kd), 2.0 * N * Kd) * split
    rounds = -(-block // per_xcd)
    rate = _TILE_RATE[variant]
    if two_stage:
        rate = rate / 2 if block > 32 else rate * _ALONE    # shares its CU | alone on it (a drained K-step is exposed latency)
    ksteps = -(-(Kd // 64) // split)
    t = rounds * ((ksteps + _FILL_STEPS) * (2.0 * bm * bn * 64) / (rate * 1e6) + _EPI_US * bm * bn / (128 * 128))
    if split > 1:
        t += _FIN_US + split * M * N * 8.0 / _FIN_BYTES_PER_US
    return t


def _nt_candidates(M: int, N: int, Kd: int, splittable: bool, fused_epilogue: bool = False, staged: bool = False):
    kt = (Kd + 63) // 64
    out = []
    for v, (bm, bn, _stages, waves) in list(TILE_OF_VARIANT.items()) + (list(STAGED_TILES.items()) if staged else []):
        if bn == 160 and (N % 160 or fused_epilogue):       # the GEGLU-forward epilogue pairs 64 value with 64 gate columns: 128-wide tiles
            continue
        if bn == 128 and N % 160 == 0 and N % 128 and N > 160:
            continue
        if bn == 256 and N % 256:
            continue
        if bn == 320 and N % 320:
            continue
        if waves == 8 and M < 2 * bm:
            continue
        tiles = -(-M // bm) * -(-N // bn)
        for s in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if s > 1 and (not splittable or tiles >= 512 or tiles * s > 1024 or kt // s < 4 or -(-kt // s) * (s - 1) >= kt):
                continue
            out.append((s, v))
    return out


def _dual_candidates(M: int, N: int, Kd: int):
    """The second-operand loop (LoRA) exists for the two-stage four-wave tiles only."""
    return [(1, v) for v in ((7, 6, 8) if N % 160 == 0 and N % 128 == 0 else ((7, 6) if N % 160 == 0 else (8,)))]


def choose_cfg(rt: "Runtime", M: int, N: int, Kd: int, ldc: int, cin: int = 0, dual: bool = False):
    """(split, variant) without a measurement: the candidate with the smallest `estimate_gemm_us`.  dual: the launch carries a
    second operand pair (LoRA), which only the unsplit two-stage four-wave tiles implement.  Memoised on the problem: the search
    walks ~100 (split, tile) candidates (0.16 ms of Python), and an eager step asks ~600 times."""
    if rt.gemm_variant != 4:
        return (1 if dual else choose_split(rt, M, N, Kd, ldc)), rt.gemm_variant
    return _choose_cfg_v4(bool(rt.split_k), M, N, Kd, ldc, cin, bool(dual))


@functools.lru_cache(maxsize=None)
def _choose_cfg_v4(split_k: bool, M: int, N: int, Kd: int, ldc: int, cin: int, dual: bool):
    splittable = split_k and N % 4 == 0 and ldc % 4 == 0
    if not dual and M >= 30000 and N % 128 == 0 and N % 160:
        # the conditioners' convolutions (VAE widths 128 / 256 / 512 over 38400 - 2.46 M pixel rows; no UNet width is a multiple of 128 and
        # not of 160): the cost model's rates were fitted to the UNet's shapes and picked tiles 3-17 % behind the fastest here.  In-situ
        # sweep of round 6 (tools/cond_tune.py, profiles/r6k_cond_tune.txt): 256 x 256 tiles where two column tiles and >= 100 k rows
        # exist, the two-stage eight-wave 128 x 128 tile elsewhere; VAE encode of 15 frames 16.08 -> 15.47 ms.
        return 1, (18 if (M >= 100000 and N % 256 == 0) else 27)
    cands = _dual_candidates(M, N, Kd) if dual else _nt_candidates(M, N, Kd, splittable)
    if not cands:
        return choose_split(SimpleNamespace(split_k=split_k), M, N, Kd, ldc), 4
    return min(cands, key=lambda c: estimate_gemm_us(M, N, Kd, c[0], c[1], cin))


GEGLU_TWO_PER_CU = 26      # 192 x 128, eight waves, two stages: 80 KB of LDS, two workgroups per CU (csrc/gemm.hip); used under GEGLU epilogues only


def geglu_candidates(M: int, N: int, Kd: int, fwd: bool = True):
    """Tile variants the in-situ tuner tries for a GEMM with a fused GEGLU epilogue (no split-K there)."""
    vs = [v for _, v in _nt_candidates(M, N, Kd, False, fused_epilogue=fwd, staged=True)]
    return vs + ([GEGLU_TWO_PER_CU] if fwd or N % 128 == 0 else [])


def choose_geglu_variant(M: int, N: int, Kd: int, fwd: bool = True) -> int:
    return _choose_geglu_variant(M, N, Kd, fwd)


@functools.lru_cache(maxsize=None)
def _choose_geglu_variant(M: int, N: int, Kd: int, fwd: bool) -> int:
    """Tile variant of a GEMM with a fused GEGLU epilogue (no split-K there) without a measurement.  Under these epilogues a
    workgroup runs its main loop, the GELU polynomial and its 200-400 KB of stores one after the other, so two workgroups per CU
    matter more than the main loop: the two-stage eight-wave 192 x 128 tile is the default (isolated, us: forward M = 35840
    111.4 -> 100.2, M = 2240 73.8 -> 70.1; backward 126.4 -> 105.3 / 68.6 -> 57.8 / 53.8 -> 44.3 at M = 35840 / 8960 / 2240);
    the 256 x 256 tile keeps the forward at the 32x20 level (80.3 against 80.9), ring tiles the 8x5 level (M = 560).  (Rounds 3-6 A/B-ed
    the alternatives in the step -- the in-situ sweep's one-per-CU winners, the two-role tiles 32 / 34: +0.1 to +1.7 ms, profiles/r5_ab_c2.txt,
    r6c_ab_two_role_first_cut.txt.)"""
    if not fwd and N % 128:
        return 4                                             # the backward epilogue takes whole column tiles: 160-wide ones here (N % 160 == 0)
    if M < 1024:
        return (17 if M >= 512 else 4) if fwd else 21
    if fwd and 4096 <= M < 16384 and N % 256 == 0:
        return 18
    return GEGLU_TWO_PER_CU


class GemmTuner:
    """In-situ choice of tile shape / split-K per distinct GEMM problem.

    While `active`, every tuned call site asks `pick(key, candidates)`; all calls of one problem use the same candidate
    during one step and are bracketed by events on the launch stream, so each candidate is timed inside the real step
    (real cache state, real neighbours -- isolated back-to-back timing of one GEMM is MALL-warm and picked configurations
    that were slower in the step).  `end_step()` folds the timings in and moves every problem to its next candidate; after
    `rounds` sweeps the fastest candidate per problem is frozen into `table`."""

    def __init__(self, rounds: int = 2):
        self.rounds = rounds
        self.active = True
        self.step = 0
        self.cands = {}
        self.stats = {}
        self.pending = []
        self.table = {}

    def pick(self, key, make_cands):
        if key not in self.cands:
            self.cands[key] = list(make_cands())
            self.stats[key] = [[0.0, 0] for _ in self.cands[key]]
        idx = self.step % len(self.cands[key])
        return self.cands[key][idx], idx

    def record(self, key, idx, e0, e1) -> None:
        self.pending.append((key, idx, e0, e1))

    def end_step(self) -> bool:
        """-> True once every problem has been swept `rounds` times and the table is frozen."""
        if self.pending:
            self.pending[-1][3].synchronize()
        for key, idx, e0, e1 in self.pending:
            st = self.stats[key][idx]
            st[0] += e0.elapsed_time(e1)
            st[1] += 1
        self.pending = []
        self.step += 1
        if self.cands and self.step >= self.rounds * max(len(c) for c in self.cands.values()):
            self.freeze()
        return not self.active

    def freeze(self) -> None:
        """Adopt the fastest measured candidate of every problem (unmeasured ones never win) and stop exploring."""
        for key, cands in self.cands.items():
            avg = [(st[0] / st[1]) if st[1] else float("inf") for st in self.stats[key]]
            i = min(range(len(cands)), key=avg.__getitem__)
            if avg[i] < float("inf"):
                self.table[key] = cands[i]
        self.active = False


def _tuning(rt: "Runtime") -> bool:
    t = rt.tuner
    return t is not None and t.active and rt.dev.type == "cuda" and not torch.cuda.is_current_stream_capturing()


def tuned_call(rt: "Runtime", key, make_cands, fallback, run) -> None:
    """Run `run(cfg)` with the frozen choice for `key`, the tuner's candidate of this step (timed), or `fallback()`."""
    if _tuning(rt):
        cfg, idx = rt.tuner.pick(key, make_cands)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(cfg)
        e1.record()
        rt.tuner.record(key, idx, e0, e1)
        return
    cfg = rt.tuner.table.get(key) if rt.tuner is not None else None
    run(cfg if cfg is not None else fallback())


def gn_tile_ok(variant: int, N: int, rows: int, cg: int) -> bool:
    """Can the tile `variant` take GroupNorm statistics in its store loop (svdx_gemm_gn)?  Whole column tiles, and a tile that touches
    at most 8 samples and 36 groups (csrc/gemm.hip: GN_MAX_S / GN_MAX_G)."""
    t = TILE_OF_VARIANT.get(variant) or STAGED_TILES.get(variant)
    if t is None:
        return False
    bm, bn = (144 if variant == 36 else t[0]), t[1]          # (variant 36 steps 140 rows and computes 144: the launcher checks the height)
    return N % bn == 0 and (bm - 1) // rows + 2 <= 8 and (bn - 1) // cg + 2 <= 36


def gemm_act(rt: "Runtime", A, B, out, M, N, Kd, lda, ldb, ldc, bias=None, rowvec=None, rv_ld=0, rv_rpg=0, rv_mod=0,
             res=None, ldres=0, gather=None, dual=None, alpha: float = 1.0, gn=None) -> bool:
    """Activation-dtype GEMM.  Tile shape and split-K factor come from the GemmTuner table when the model was tuned
    (Trainer.tune_gemms), else from a formula: the 10x16 / 5x8 latent levels (M = 2240 / 560 rows against K up to 23040)
    cannot fill 256 CUs with output tiles alone, so the reduction is split across blocks -- partial sums go to float slabs
    that a small epilogue kernel reduces, applying bias/row-vector/residual.
    dual = (A2, B2, K2, lda2, ldb2): out also gets A2 B2^T (the LoRA term) -- inside the same launch when the reduction is not
    split, by a second accumulate launch when it is.
    gn = (stats, rows, cg): the GroupNorm that consumes `out` wants its statistics (zeroed opaque buffer of svdx_gn_stats; sample =
    `rows` consecutive rows, group = `cg` consecutive channels).  Returns True when the launch left them there (unsplit launch of a
    tile whose store loop can take them); False: the caller runs svdx_gn_stats as before."""
    k = rt.k
    splittable = rt.split_k and N % 4 == 0 and ldc % 4 == 0
    key = ("nt", M, N, Kd, lda, ldc, 0 if gather is None else (gather.mode, gather.stride, gather.ups, gather.cin),
           bias is not None, rowvec is not None, res is not None, gn is not None) + (() if dual is None else (("dual", dual[2]),))
    # svdx_gemm_gn serves operands below 2 GiB only (buffer descriptors; csrc/gemm.hip returns -2 beyond that): larger operands keep the
    # separate statistics pass instead of turning a working forward into an error
    gn_fits = gn is not None and max(A.numel() * A.element_size(), B.numel() * B.element_size()) < (1 << 31)
    assert alpha == 1.0 or dual is None

    done = [False]

    def run(cfg):
        split, variant = cfg
        fused = dual if (split == 1 and rt.fuse_dual) else None
        if split == 1:
            take = gn if (gn_fits and dual is None and rt.fuse_gn_stats and ldc % 8 == 0 and gn_tile_ok(_tile_launched(variant, M, N), N, gn[1], gn[2])) else None
            done[0] = take is not None
            k.gemm(A, B, out, M, N, Kd, lda, ldb, ldc, bias=bias, rowvec=rowvec, rv_ld=rv_ld, rv_rpg=rv_rpg, rv_mod=rv_mod,
                   res=res, ldres=ldres, gather=gather, variant=variant, dual=fused, alpha=alpha, gn=take)
        else:
            acc = rt.f32(split, M, N)
            k.gemm(A, B, acc, M, N, Kd, lda, ldb, N, gather=gather, out_mode=K.OUT_F32_SLAB, split_k=split, variant=variant, alpha=alpha)
            # the reducing launch writes the tensor: the statistics ride there (a block's 1024 consecutive elements: <= 2 samples)
            # (dual is None: an unfused adapter term below would change `out` after its statistics were taken)
            take = gn if (gn is not None and dual is None and rt.fuse_gn_stats and N * gn[1] >= 1024 and N // gn[2] <= 64) else None
            done[0] = take is not None
            k.gemm_finalize(acc, split, M * N, out, M, N, ldc, bias=bias, rowvec=rowvec, rv_ld=rv_ld, rv_rpg=rv_rpg,
                            rv_mod=rv_mod, res=res, ldres=ldres, gn=take)
        if dual is not None and fused is None:
            A2, B2, K2, lda2, ldb2 = dual[:5]
            seg = dual[5] if len(dual) > 5 and dual[5] else N
            for j in range(N // seg):            # the adapter term as its own accumulate launch (per q/k/v segment)
                oj = out[:, j * seg:]
                k.gemm(A2[:, j * K2:], B2[j * seg:], oj, M, seg, K2, lda2, ldb2, ldc, res=oj, ldres=ldc, variant=rt.gemm_variant)

    tuned_call(rt, key, lambda: _nt_candidates(M, N, Kd, splittable and dual is None, staged=True) if dual is None else _dual_candidates(M, N, Kd),
               lambda: choose_cfg(rt, M, N, Kd, ldc, 0 if gather is None else gather.cin, dual is not None and rt.fuse_dual), run)
    return done[0]


def _tile_launched(variant: int, M: int, N: int) -> int:
    """The tile svdx_gemm resolves `variant` to (csrc/gemm.hip: variant 4 is a rule among 6 / 7 / 8; a 160-wide request on an N that 160
    does not divide takes the 128-wide sibling, 256-wide tiles need N % 256 == 0)."""
    if variant == 4:
        if N % 160:
            return 8
        t128, t160 = -(-M // 128) * (N // 160), -(-M // 160) * (N // 160)
        return 6 if (-(-t128 // 512) * 4 > -(-t160 // 512) * 5 and t160 >= 384) else 7
    sib = {16: 17, 23: 22, 25: 24, 20: 21, 28: 27, 7: 8, 6: 8, 36: 27}
    if variant in sib and N % 160:
        return sib[variant]
    if variant == 18 and N % 256:
        return 17
    return variant



# --------------------------------------------------------------------------------------------------
# Linear
# --------------------------------------------------------------------------------------------------
class LinearOp:
    """y = x W^T (+ b).  `weights` may be several nn.Parameters concatenated along the output dim (fused QKV).

    Replaces nn.Linear inside diffusers Attention / FeedForward / proj_in / proj_out.
    Packed copies (activation dtype): w [N,K] for fwd, wt [K,N] for the data-grad.  Weight-grad goes to the
    float .grad of the master parameter(s) through an NT GEMM on transposed operands."""

    def __init__(self, weights: Sequence[nn.Parameter], biases: Optional[Sequence[Optional[nn.Parameter]]] = None):
        self.weights = list(weights)
        self.biases = list(biases) if biases is not None else None
        self.N = sum(w.shape[0] for w in self.weights)
        self.Kdim = self.weights[0].shape[1]
        self.trainable = any(w.requires_grad for w in self.weights)
        self.w = self.wt = self.b = None
        self.w_grad = self.b_grad = None

    # ---- packing ----
    def pack(self, rt: Runtime, need_dx: bool = True) -> None:
        k = rt.k
        master = self._flat_view([w.data for w in self.weights])
        if master is None:
            master = torch.cat([w.data.reshape(w.shape[0], -1) for w in self.weights], 0).contiguous()
        twin = rt.act_view(master) if self.trainable else None
        self.w_is_view = twin is not None
        if twin is not None:
            self.w = twin.view(self.N, self.Kdim)          # kept current by AdamW / refresh_trainable (no per-op cast)
        else:
            self.w = rt.empty(self.N, self.Kdim)
            k.cast_from_f32(master, self.w, self.N * self.Kdim)
        self.wt_managed = False
        if need_dx:
            n_el = self.Kdim * self.N
            if twin is not None and rt.wt16_flat is not None and rt.wt_pos + n_el <= rt.wt16_flat.numel():
                off = rt.wt_pos
                rt.wt_pos += rup(n_el, 64)
                self.wt = rt.wt16_flat[off:off + n_el].view(self.Kdim, self.N)
                n0 = 0
                for w in self.weights:               # fused q/k/v: each weight owns a column block of the [K, 3C] twin
                    rt.wt_map[id(w)] = (off + n0, self.N)
                    n0 += w.shape[0]
                self.wt_managed = True
            else:
                self.wt = rt.empty(self.Kdim, self.N)
            k.cast_transpose_from_f32(master, self.wt, self.N, self.Kdim)
        if self.biases is not None and self.biases[0] is not None:
            b = self._flat_view([b.data for b in self.biases])
            self.b = b if b is not None else torch.cat([b.data for b in self.biases]).contiguous()
        if self.trainable:
            self.w_grad = self._flat_view([w.grad for w in self.weights])
            if self.w_grad is None:
                raise RuntimeError("trainable fused weights must have contiguous .grad views (use Trainer)")
            rt.write_once.update(id(w) for w in self.weights)     # bwd_dw is their only gradient producer
            if self.b is not None:
                self.b_grad = self._flat_view([b.grad for b in self.biases])

    @staticmethod
    def _flat_view(ts: List[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
        """Return one contiguous tensor covering `ts` when they are adjacent in memory, else None."""
        if any(t is None for t in ts):
            return None
        if len(ts) == 1:
            return ts[0] if ts[0].is_contiguous() else None
        ptr = ts[0].data_ptr()
        store = ts[0].untyped_storage().data_ptr()
        total = 0
        for t in ts:
            # adjacent addresses are not enough: the caching allocator often places separate tensors back to back
            if (not t.is_contiguous() or t.data_ptr() != ptr + total * t.element_size() or t.dtype != ts[0].dtype
                    or t.untyped_storage().data_ptr() != store):
                return None
            total += t.numel()
        base = ts[0]
        return torch.as_strided(base, (total,), (1,), base.storage_offset())

    def refresh(self, rt: Runtime, need_dx: bool = True) -> None:
        """Re-cast after an optimizer step (trainable weights only)."""
        master = self._flat_view([w.data for w in self.weights])
        if not getattr(self, "w_is_view", False):
            rt.k.cast_from_f32(master, self.w, self.N * self.Kdim)
        if self.wt is not None and not (rt.adam_writes_wt and getattr(self, "wt_managed", False)):
            if getattr(self, "w_is_view", False):       # the 16-bit twin is current: 2+2 bytes per weight instead of 4+2
                rt.k.transpose(self.w, self.Kdim, self.wt, self.N, self.N, self.Kdim)
            else:
                rt.k.cast_transpose_from_f32(master, self.wt, self.N, self.Kdim)

    # ---- compute ----
    def fwd(self, rt: Runtime, x: torch.Tensor, M: int, res: Optional[torch.Tensor] = None,
            rowvec: Optional[torch.Tensor] = None, rv_ld: int = 0, rv_rpg: int = 0, rv_mod: int = 0,
            out: Optional[torch.Tensor] = None, dual=None, gn=None):
        """gn: see gemm_act; with it the return value is (y, statistics taken)."""
        y = out if out is not None else rt.empty(M, self.N)
        took = gemm_act(rt, x, self.w, y, M, self.N, self.Kdim, self.Kdim, self.Kdim, self.N, bias=self.b,
                        rowvec=rowvec, rv_ld=rv_ld, rv_rpg=rv_rpg, rv_mod=rv_mod, res=res,
                        ldres=self.N if res is not None else 0, dual=dual, gn=gn)
        return y if gn is None else (y, took)

    def bwd_dx(self, rt: Runtime, dy: torch.Tensor, M: int, out: Optional[torch.Tensor] = None, dual=None) -> torch.Tensor:
        dx = out if out is not None else rt.empty(M, self.Kdim)
        gemm_act(rt, dy, self.wt, dx, M, self.Kdim, self.N, self.N, self.N, self.Kdim, dual=dual)
        return dx

    def bwd_dw(self, rt: Runtime, dy: torch.Tensor, x: torch.Tensor, M: int, colsum_to: Optional[torch.Tensor] = None) -> None:
        """w.grad += dy^T x ; b.grad += colsum(dy).  TN GEMM straight from the row-major dy [M,N] and x [M,K].
        colsum_to (float [N], zeroed): receives colsum(dy) instead of b.grad -- the caller wants the column sums themselves and adds
        them to b.grad on its own (TemporalBasicTransformerBlock.bwd: they are also d(cross-attention vector))."""
        if not self.trainable:
            return
        gemm_tn_acc(rt, dy, x, self.w_grad, M, self.N, self.Kdim, self.N, self.Kdim,
                    a_colsum=self.b_grad if colsum_to is None else colsum_to, write_once=True)


# Round 4 timed eight-wave 128 x 256 / 128 x 384 / 256 x 128 weight-gradient tiles (two and three stages) inside the step: the four-wave
# 128 x 128 tile with two workgroups per CU won every problem by 5-40 % (profiles/r4_dropped_experiments.txt, item 2); they were removed again.
# Round 5: with the staging hidden from the compiler's wait pass (csrc/gemm.hip hidden_dma) loads and MFMAs overlap for the first time, so the
# deeper rings of the 128 x 128 tile (stages 3 / 4: one workgroup per CU, 2 / 3 tiles in flight) are offered to the in-situ tuner again.
STAGED_TN_TILES = {3: (128, 128), 4: (128, 128)}


@functools.lru_cache(maxsize=None)
def _tn_formula(M: int, N: int, Kd: int):
    """(row slices, `stages` code) of a weight-gradient GEMM without a measurement."""
    rtiles = (M + 63) // 64

    def tiles_of(tm, tk):
        return -(-N // tm) * -(-Kd // tk)
    # 256 x 256 eight-wave tiles when (with a few row slices) they cover the output in one round of <= 256 workgroups: the
    # 16x10 / 32x20-level feed-forward gradients (10240 x 1280 over 2240 rows: 98 against 107 us in the step; 5120 x 640 over 8960)
    if N >= 1024 and Kd >= 512 and rtiles >= 16:
        t18 = tiles_of(256, 256)
        sk = _tn_slices(max(1, min(256 // t18, rtiles // 32)))
        if 180 <= t18 * sk <= 256:
            return sk, 18
    tiles = tiles_of(128, 128)
    sk = 1
    if tiles < 256 and rtiles >= 16:
        sk = max(1, min(512 // tiles, rtiles // 4, 128 if tiles <= 4 else 32))
        sk = _tn_slices(sk)
        while sk > 1 and (rtiles + sk - 1) // sk * (sk - 1) >= rtiles:
            sk = _tn_slices(sk - 1)
    return sk, 2


def _tn_slices(sk: int) -> int:
    """Row-slice counts the TN kernels place well on the 8 XCDs (csrc/gemm.hip tn_who): 1, 2, 4 (a slice owns 8 / sk XCDs) or a multiple of 8
    (an XCD owns whole slices); the largest such count <= sk."""
    return sk // 8 * 8 if sk >= 8 else (4 if sk >= 4 else 2 if sk >= 2 else 1)


def gemm_tn_acc(rt: Runtime, dy: torch.Tensor, x: torch.Tensor, dst: torch.Tensor, M: int, N: int, Kd: int, lda: int, ldb: int,
                a_colsum: Optional[torch.Tensor] = None, write_once: bool = False) -> None:
    """dst[N, Kd] (float, contiguous) += dy[:, :N]^T x[:, :Kd] over M rows (row pitches lda / ldb); a_colsum += colsum(dy).
    The reduction over rows is split across blocks when the [N, Kd] tile grid cannot fill the chip (float slabs + finalize):
    weight-grad outputs are small (down to 320 x 64 for a LoRA factor) while M is 35840."""
    k = rt.k
    rtiles = (M + 63) // 64

    store = rt.grad_overwrite and write_once       # first micro-batch of a step: dst = ..., later ones: dst += ...

    def run(cfg):
        sk, stages = cfg if isinstance(cfg, tuple) else (cfg, 0)
        # the bias gradient (column sums of dy) rides on the same launch
        # GradScaler's inf check where the gradient is written (Runtime.found_inf: the trainer's opt_state[3]; None = no folding)
        found = rt.found_inf if (write_once and rt.fold_finite) else None
        if sk == 1:
            k.gemm_tn(dy, x, dst, M, N, Kd, lda, ldb, Kd, out_mode=K.OUT_F32 if store else K.OUT_F32_ADD, a_colsum=a_colsum, stages=stages,
                      found_inf=found)
        else:
            slabs = rt.f32(sk, N, Kd)
            cs = rt.f32(sk, N) if a_colsum is not None else None       # per-slice column sums, added in slice order by the finalize
            k.gemm_tn(dy, x, slabs, M, N, Kd, lda, ldb, Kd, out_mode=K.OUT_F32_SLAB, split_k=sk, a_colsum=cs, stages=stages)
            if rt.batch_small and rt.defer_grad_finalize and (N * Kd) % 4 == 0 and not _tuning(rt):
                # nothing reads a weight gradient before the optimizer (or the block's gradient bucket): the reducing launch waits for
                # the sweep's one table-driven launch (Runtime.flush_deferred); the slabs stay alive in the queue until then
                rt.defer_grad_finalize_job((slabs, sk, N * Kd, dst, N * Kd, cs, a_colsum, store, found))
            else:
                k.gemm_finalize(slabs, sk, N * Kd, dst, N, Kd, Kd, accumulate_f32=2 if store else 1, dtype=rt.dt, colsum_slabs=cs,
                                colsum_out=a_colsum)
                if found is not None:
                    rt.unchecked_grads = True          # the single-launch reduction carries no flag: this step takes the full pass instead

    TN_TILES = {2: (128, 128), 18: (256, 256)}     # svdx_gemm_tn `stages`: output tile (rows of dst, columns)
    ALL_TN_TILES = {**TN_TILES, **STAGED_TN_TILES}

    def tiles_of(v):
        tm, tk = ALL_TN_TILES[v]
        return -(-N // tm) * -(-Kd // tk)

    def cands():
        return [(s, v) for v in ALL_TN_TILES if v == 2 or (N >= ALL_TN_TILES[v][0] and Kd > ALL_TN_TILES[v][1] - 128)
                for s in (1, 2, 4, 8, 16, 24, 32, 40, 48, 64, 96)
                if s == 1 or (tiles_of(v) * s <= (2048 if v == 2 else 768) and rtiles // s >= 2 and -(-rtiles // s) * (s - 1) < rtiles)]

    tuned_call(rt, ("tn", M, N, Kd, lda, ldb), cands, lambda: _tn_formula(M, N, Kd), run)


_ONES = {}


def _ones(rt: Runtime) -> torch.Tensor:
    key = str(rt.dev)
    if key not in _ONES:
        _ONES[key] = torch.ones(1, 1, dtype=torch.float32, device=rt.dev)
    return _ONES[key]


# --------------------------------------------------------------------------------------------------
# skinny linear (float activations): embedding MLPs and the KV-length-1 cross-attention
# --------------------------------------------------------------------------------------------------
class SmallLinearOp:
    def __init__(self, weight: nn.Parameter, bias: Optional[nn.Parameter]):
        self.weight, self.bias = weight, bias
        self.N, self.Kdim = weight.shape
        self.trainable = weight.requires_grad
        self.w = None

    def pack(self, rt: Runtime) -> None:
        twin = rt.act_view(self.weight.data) if self.trainable else None
        self.w_is_view = twin is not None
        if twin is not None:
            self.w = twin.view(self.N, self.Kdim)
            return
        self.w = rt.empty(self.N, self.Kdim)
        rt.k.cast_from_f32(self.weight.data.contiguous(), self.w, self.N * self.Kdim)

    def refresh(self, rt: Runtime) -> None:
        if not self.w_is_view:
            rt.k.cast_from_f32(self.weight.data.contiguous(), self.w, self.N * self.Kdim)

    def fwd(self, rt: Runtime, x: torch.Tensor, M: int, silu_in: bool = False, out: Optional[torch.Tensor] = None,
            accumulate: bool = False) -> torch.Tensor:
        y = out if out is not None else rt.f32(M, self.N)
        rt.k.small_linear(x, self.w, None if self.bias is None else self.bias.data, y, M, self.N, self.Kdim,
                          self.Kdim, 0, int(silu_in), int(accumulate))
        return y

    def bwd(self, rt: Runtime, dy: torch.Tensor, x: torch.Tensor, M: int, need_dx: bool) -> Optional[torch.Tensor]:
        k = rt.k
        if self.trainable:
            k.outer_acc(dy, x, self.weight.grad, M, self.N, self.Kdim, 1.0)
            if self.bias is not None:
                k.outer_acc(dy, _ones(rt).expand(M, 1).contiguous(), self.bias.grad.view(self.N, 1), M, self.N, 1, 1.0)
        if not need_dx:
            return None
        dx = rt.f32(M, self.Kdim)
        k.small_linear(dy, self.w, None, dx, M, self.N, self.Kdim, self.Kdim, 1, 0, 0)
        return dx


# --------------------------------------------------------------------------------------------------
# LoRA branches (reference config 5: train_svd_lora.py:659-674; peft.tuners.lora.Linear restated in lora.py)
# --------------------------------------------------------------------------------------------------
class LoraOp:
    """Adapter branch of a (possibly fused) LinearOp: y[:, seg_j] += s * (x A_j^T) B_j^T for the J wrapped projections that
    share the input x (J = 3 for fused q/k/v, 1 for to_out).

    Forward: one skinny NT GEMM xs = s * x [A_1; ..; A_J]^T, then the adapter term rides on the base projection as its second
    operand pair (xs, B_bd) -- B_bd [sum N_j, J*rp] is block-structured, segment j's rows hold B_j in columns j*rp.. -- so y is
    written once.  Backward: one skinny NT GEMM per segment for d(xA^T) = s * dy_j B_j, TN GEMMs for dB_j and dA_j, and the term
    d(xA^T) [A_1; ..; A_J] rides on the base data-grad GEMM the same way.  The rank is zero-padded to a multiple of 64 (the GEMM K
    granule); the packed 16-bit copies of the small matrices are refreshed after every optimizer step."""

    def __init__(self, mods):
        self.mods = list(mods)
        self.J = len(self.mods)
        self.r = self.mods[0].r
        self.rp = rup(self.r, 64)
        self.s = self.mods[0].scaling
        self.in_f = self.mods[0].in_features
        self.outs = [m.out_features for m in self.mods]
        assert all(m.r == self.r and m.in_features == self.in_f and m.scaling == self.s for m in self.mods)
        self.trainable = any(m.A.requires_grad or m.B.requires_grad for m in self.mods)
        self.A3 = self.A3T = self.Bbd = self.Bst = None
        self.fast = self.seg_ok = False
        self.Bp: List[torch.Tensor] = []
        self.BTp: List[torch.Tensor] = []

    def pack(self, rt: Runtime) -> None:
        J, rp = self.J, self.rp
        n_sum = sum(self.outs)
        if self.r == rp:
            rt.write_once.update(id(q) for m in self.mods for q in (m.A, m.B) if q.requires_grad)
        # Fast layout: the flat buffer holds A_1..A_J and B_1..B_J as two contiguous blocks (ops._layout_order), so their 16-bit
        # twins ARE the stacked operands, and the transposed copies live in the arena the tiled AdamW writes: nothing to re-pack.
        a_tw = rt.act_view(LinearOp._flat_view([m.A.data for m in self.mods])) if self.r == rp else None
        b_tw = rt.act_view(LinearOp._flat_view([m.B.data for m in self.mods])) if self.r == rp else None
        n_wt = self.in_f * J * rp + sum(rp * n for n in self.outs)
        self.fast = (a_tw is not None and b_tw is not None and rt.wt16_flat is not None
                     and rt.wt_pos + n_wt + 64 * (J + 1) <= rt.wt16_flat.numel())
        n0 = self.outs[0]
        # one B2 [sum N, r] with per-segment A2 columns needs segments that are whole tiles of the dual kernel (include/svdx.h)
        self.seg_ok = J == 1 or (all(n == n0 for n in self.outs) and (n_sum % 160 != 0 or n0 % 160 == 0)
                                 and (n0 % 128 == 0 or (n_sum % 160 == 0 and n_sum % 128 != 0)))
        if self.fast:
            self.A3 = a_tw.view(J * rp, self.in_f)
            self.Bst = b_tw.view(n_sum, rp)

            def arena(n_el):
                off = rt.wt_pos
                rt.wt_pos += rup(n_el, 64)
                return off
            off = arena(self.in_f * J * rp)
            self.A3T = rt.wt16_flat[off:off + self.in_f * J * rp].view(self.in_f, J * rp)
            for j, m in enumerate(self.mods):
                rt.wt_map[id(m.A)] = (off + j * rp, J * rp)         # A_j^T is the column block j of A3T
            self.BTp = []
            for m, n in zip(self.mods, self.outs):
                off = arena(rp * n)
                self.BTp.append(rt.wt16_flat[off:off + rp * n].view(rp, n))
                rt.wt_map[id(m.B)] = (off, n)
        else:
            self.A3 = torch.zeros(J * rp, self.in_f, dtype=rt.dt, device=rt.dev)
            self.A3T = rt.empty(self.in_f, J * rp)
            self.BTp = [rt.empty(rp, n) for n in self.outs]
            self.Bst = None
        if not (self.fast and self.seg_ok):
            self.Bbd = torch.zeros(n_sum, J * rp, dtype=rt.dt, device=rt.dev)
            self.Bp, off = [], 0
            for j, n in enumerate(self.outs):
                self.Bp.append(self.Bbd[off:off + n, j * rp:(j + 1) * rp])          # view, row pitch J*rp
                off += n
        self.refresh(rt, force=True)

    def refresh(self, rt: Runtime, force: bool = False) -> None:
        k, r, rp, J = rt.k, self.r, self.rp, self.J
        if self.fast:
            if self.Bbd is not None:                 # segments that are not whole tiles: block-structured B for the dual launch
                off = 0
                for j, n in enumerate(self.outs):
                    self.Bp[j].copy_(self.Bst[off:off + n])
                    off += n
            if rt.adam_writes_wt and not force:
                return                               # AdamW wrote the twins and their transposes
            off = 0
            for j, n in enumerate(self.outs):
                k.transpose(self.Bst[off:off + n], rp, self.BTp[j], n, n, rp)
                off += n
            k.transpose(self.A3, self.in_f, self.A3T, J * rp, J * rp, self.in_f)
            return
        for j, m in enumerate(self.mods):
            if r == rp:
                k.cast_from_f32(m.A.data, self.A3[j * rp:(j + 1) * rp], r * self.in_f)
            else:                                   # padded rank: strided re-layout of a tiny matrix
                self.A3[j * rp:j * rp + r].copy_(m.A.data)
            if r == rp and J == 1:
                k.cast_from_f32(m.B.data, self.Bbd, self.outs[j] * r)
            else:
                self.Bp[j][:, :r].copy_(m.B.data)       # strided block of B_bd
            k.transpose(self.Bp[j], J * rp, self.BTp[j], self.outs[j], self.outs[j], rp)
        k.transpose(self.A3, self.in_f, self.A3T, J * rp, J * rp, self.in_f)

    def fwd_xs(self, rt: Runtime, x: torch.Tensor, M: int) -> torch.Tensor:
        """xs = s * x A^T [M, J*rp]; hand `self.fwd_dual(xs)` to the base projection's `fwd`."""
        xs = rt.empty(M, self.J * self.rp)
        rt.k.gemm(x, self.A3, xs, M, self.J * self.rp, self.in_f, self.in_f, self.in_f, self.J * self.rp, alpha=self.s,
                  variant=rt.gemm_variant)
        return xs

    def fwd_dual(self, xs: torch.Tensor):
        w = self.J * self.rp
        if self.fast and self.seg_ok:                # B2 = the stacked B twins [sum N, r]; segment j reads xs[:, j*r:(j+1)*r]
            return (xs, self.Bst, self.rp, w, self.rp, self.outs[0] if self.J > 1 else 0)
        return (xs, self.Bbd, w, w, w)

    def can_colsum(self) -> bool:
        """the dB GEMM of a single-projection adapter can hand out colsum(dy) (`bwd(colsum_to=...)`)"""
        return self.J == 1 and self.r == self.rp and self.mods[0].B.requires_grad

    def bwd(self, rt: Runtime, dy: torch.Tensor, lddy: int, x: torch.Tensor, xs: torch.Tensor, M: int,
            colsum_to: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dy [M, sum N_j] (row pitch lddy); accumulates A_j.grad / B_j.grad and returns dxa = s * dy B [M, J*rp]; hand
        `self.bwd_dual(dxa)` to the base projection's `bwd_dx` for the dx term.  colsum_to (float [N], zeroed; needs `can_colsum()`):
        += colsum(dy), computed by the dB GEMM on the side (the transformer blocks' d(cross-attention vector) at one clip per rank)."""
        k, r, rp, J = rt.k, self.r, self.rp, self.J
        assert colsum_to is None or self.can_colsum()
        dxa = rt.empty(M, J * rp)
        off = 0
        for j, n in enumerate(self.outs):
            dyj = dy[:, off:off + n]
            k.gemm(dyj, self.BTp[j], dxa[:, j * rp:], M, rp, n, lddy, n, J * rp, alpha=self.s, variant=rt.gemm_variant)
            m = self.mods[j]
            if m.B.requires_grad:
                if r == rp:
                    gemm_tn_acc(rt, dyj, xs[:, j * rp:], m.B.grad, M, n, rp, lddy, J * rp, a_colsum=colsum_to, write_once=True)
                else:
                    tmp = rt.zeros_f32(n, rp)
                    gemm_tn_acc(rt, dyj, xs[:, j * rp:], tmp, M, n, rp, lddy, J * rp)
                    m.B.grad.add_(tmp[:, :r])
            off += n
        # dA_j = d(xA_j^T)^T x: the J factors' gradients are adjacent in the flat buffer (ops._layout_order), so they are ONE
        # [J*r, in] TN GEMM on the stacked d(xA^T) -- a third of the launches (and of their slab finalizes), and output tiles that
        # are not three-quarters padding
        a_grads = [m.A.grad for m in self.mods]
        stacked = (LinearOp._flat_view(a_grads) if J > 1 and r == rp and rt.lora_stack_da and all(m.A.requires_grad for m in self.mods)
                   and all(g is not None for g in a_grads) else None)
        if stacked is not None:
            gemm_tn_acc(rt, dxa, x, stacked.view(J * rp, self.in_f), M, J * rp, self.in_f, J * rp, self.in_f, write_once=True)
            return dxa
        for j, m in enumerate(self.mods):
            if m.A.requires_grad:
                if r == rp:
                    gemm_tn_acc(rt, dxa[:, j * rp:], x, m.A.grad, M, rp, self.in_f, J * rp, self.in_f, write_once=True)
                else:
                    tmp = rt.zeros_f32(rp, self.in_f)
                    gemm_tn_acc(rt, dxa[:, j * rp:], x, tmp, M, rp, self.in_f, J * rp, self.in_f)
                    m.A.grad.add_(tmp[:r])
        return dxa

    def bwd_dual(self, dxa: torch.Tensor):
        w = self.J * self.rp
        return (dxa, self.A3T, w, w, w)


class SmallLoraOp:
    """Adapter branch of a SmallLinearOp (float activations, a handful of rows: the KV-length-1 cross-attention's to_v / to_out
    act on one context vector per clip)."""

    def __init__(self, mod):
        self.mod = mod
        self.s = mod.scaling
        self.a = SmallLinearOp(mod.A, None)
        self.b = SmallLinearOp(mod.B, None)
        self.trainable = mod.A.requires_grad or mod.B.requires_grad

    def pack(self, rt: Runtime) -> None:
        self.a.pack(rt)
        self.b.pack(rt)

    def refresh(self, rt: Runtime) -> None:
        self.a.refresh(rt)
        self.b.refresh(rt)

    def fwd(self, rt: Runtime, x: torch.Tensor, y: torch.Tensor, M: int) -> torch.Tensor:
        xs = self.a.fwd(rt, x, M)
        if self.s != 1.0:
            xs.mul_(self.s)                          # [M, r] floats; the reference always uses lora_alpha == r (s = 1)
        self.b.fwd(rt, xs, M, out=y, accumulate=True)
        return xs

    def bwd(self, rt: Runtime, dy: torch.Tensor, x: torch.Tensor, xs: torch.Tensor, M: int, need_dx: bool):
        dxa = self.b.bwd(rt, dy, xs, M, need_dx=True)
        if self.s != 1.0:
            dxa.mul_(self.s)
        return self.a.bwd(rt, dxa, x, M, need_dx=need_dx)


# --------------------------------------------------------------------------------------------------
# convolutions as implicit GEMM
# --------------------------------------------------------------------------------------------------
class ConvOp:
    """3x3 conv2d (stride 1/2, optional nearest-x2 source), 1x1 conv2d, or Conv3d (3,1,1), all as NT GEMMs
    whose A operand is gathered on the fly (K1-K4 of SURVEY.md 2.3).  Weights are frozen on this path
    (train_svd.py:761-766 trains only temporal transformer blocks), so only fwd + data-grad exist."""

    def __init__(self, weight: nn.Parameter, bias: Optional[nn.Parameter], kind: str, stride: int = 1,
                 ups: bool = False, cin_pad: Optional[int] = None, cout_pad: Optional[int] = None, pad0: bool = False):
        """pad0: the VAE encoder's Downsample2D(padding=0) -- stride 2 over F.pad(x, (0, 1, 0, 1)); forward only."""
        assert kind in ("3x3", "1x1", "t3")
        assert not pad0 or (kind == "3x3" and stride == 2 and not ups)
        self.weight, self.bias, self.kind, self.stride, self.ups, self.pad0 = weight, bias, kind, stride, ups, pad0
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.cin_p = cin_pad or self.cin
        self.cout_p = cout_pad or self.cout     # padded channel count of the *incoming gradient* rows
        self.taps = {"3x3": 9, "1x1": 1, "t3": 3}[kind]
        self.w = self.wd = None
        if weight.requires_grad:
            raise NotImplementedError("conv weight-grad is outside this round's trainable set")

    def pack(self, rt: Runtime, need_dx: bool = True, out_scale: float = 1.0) -> None:
        """out_scale: constant folded into the packed weights, data-grad weights and bias (a frozen AlphaBlender after the conv:
        unet.SpatioTemporalResBlock)."""
        W = self.weight.data * out_scale if out_scale != 1.0 else self.weight.data
        co, ci, tp = self.cout, self.cin, self.taps
        w4 = W.reshape(co, ci, tp)                                   # taps flattened dy*3+dx / dt
        wp = torch.zeros(co, tp, self.cin_p, dtype=torch.float32, device=W.device)
        wp[:, :, :ci] = w4.permute(0, 2, 1)
        self.w = rt.empty(co, tp * self.cin_p)
        rt.k.cast_from_f32(wp.reshape(-1), self.w, wp.numel())
        if need_dx:
            flip = self.kind in ("3x3", "t3") and self.stride == 1
            src = w4.flip(2) if flip else w4
            wd = torch.zeros(ci, tp, self.cout_p, dtype=torch.float32, device=W.device)
            wd[:, :, :co] = src.permute(1, 2, 0)
            self.wd = rt.empty(ci, tp * self.cout_p)
            rt.k.cast_from_f32(wd.reshape(-1), self.wd, wd.numel())
        self.b = None if self.bias is None else (self.bias.data * out_scale if out_scale != 1.0 else self.bias.data)

    def _gather(self, n_img, hi, wi, ho, wo, cin, lda, T=0, hw=0, dgrad=False) -> Optional[K.Gather]:
        if self.kind == "1x1":
            return None
        if self.kind == "t3":
            return K.Gather(K.GATHER_TEMPORAL3, n_img=n_img, cin=cin, t=T, hw=hw, lda=lda)
        if self.pad0:
            if dgrad:
                raise NotImplementedError("the pad-0 downsample exists on the frozen VAE encoder only (no data-grad)")
            return K.Gather(K.GATHER_CONV3X3_PAD0, n_img=n_img, hi=hi, wi=wi, ho=ho, wo=wo, cin=cin, stride=2, lda=lda)
        if dgrad and self.stride == 2:
            return K.Gather(K.GATHER_CONV3X3_DGRAD2, n_img=n_img, hi=hi, wi=wi, ho=ho, wo=wo, cin=cin, lda=lda)
        return K.Gather(K.GATHER_CONV3X3, n_img=n_img, hi=hi, wi=wi, ho=ho, wo=wo, cin=cin,
                        stride=1 if dgrad else self.stride, ups=int(self.ups and not dgrad), lda=lda)

    def out_hw(self, h: int, w: int):
        if self.ups:
            return 2 * h, 2 * w
        if self.pad0:
            return (h - 2) // 2 + 1, (w - 2) // 2 + 1
        if self.stride == 2:
            return (h - 1) // 2 + 1, (w - 1) // 2 + 1
        return h, w

    def fwd(self, rt: Runtime, x: torch.Tensor, n_img: int, h: int, w: int, T: int = 0,
            res: Optional[torch.Tensor] = None, rowvec=None, rv_ld=0, rv_rpg=0, ldc: Optional[int] = None,
            out: Optional[torch.Tensor] = None, gn=None):
        """x: [n_img*h*w, cin_p] (t3: n_img = B, rows = B*T*h*w).  Returns ([M, cout], ho, wo); `out` [M, ldc]: write the cout
        columns into the caller's (wider) rows instead of a fresh tensor."""
        ho, wo = self.out_hw(h, w)
        if self.kind == "t3":
            M = n_img * T * h * w
            g = self._gather(n_img, 0, 0, 0, 0, self.cin_p, self.cin_p, T=T, hw=h * w)
        else:
            M = n_img * ho * wo
            g = self._gather(n_img, ho if self.ups else h, wo if self.ups else w, ho, wo, self.cin_p, self.cin_p)
        ldc = ldc or self.cout
        y = out if out is not None else rt.empty(M, ldc)
        Kd = self.taps * self.cin_p
        self.took_gn = gemm_act(rt, x, self.w, y, M, self.cout, Kd, self.cin_p, Kd, ldc, bias=self.b, rowvec=rowvec, rv_ld=rv_ld,
                                rv_rpg=rv_rpg, res=res, ldres=self.cout if res is not None else 0, gather=g, gn=gn)
        return y, ho, wo

    def bwd_dx(self, rt: Runtime, dy: torch.Tensor, n_img: int, h: int, w: int, T: int = 0) -> torch.Tensor:
        """dy: [M_out, cout_p] -> dx [n_img*h*w, cin] where (h, w) are the conv INPUT dims (pre-upsample)."""
        k = rt.k
        ho, wo = self.out_hw(h, w)
        Kd = self.taps * self.cout_p
        if self.kind == "t3":
            M = n_img * T * h * w
            g = self._gather(n_img, 0, 0, 0, 0, self.cout_p, self.cout_p, T=T, hw=h * w, dgrad=True)
            dx = rt.empty(M, self.cin)
            gemm_act(rt, dy, self.wd, dx, M, self.cin, Kd, self.cout_p, Kd, self.cin, gather=g)
            return dx
        if self.ups:
            Mh = n_img * ho * wo
            g = self._gather(n_img, ho, wo, ho, wo, self.cout_p, self.cout_p, dgrad=True)
            dxh = rt.empty(Mh, self.cin)
            gemm_act(rt, dy, self.wd, dxh, Mh, self.cin, Kd, self.cout_p, Kd, self.cin, gather=g)
            dx = rt.empty(n_img * h * w, self.cin)
            k.sum2x2(dxh, dx, n_img, h, w, self.cin)
            return dx
        M = n_img * h * w
        g = self._gather(n_img, ho, wo, h, w, self.cout_p, self.cout_p, dgrad=True)
        dx = rt.empty(M, self.cin)
        gemm_act(rt, dy, self.wd, dx, M, self.cin, Kd, self.cout_p, Kd, self.cin, gather=g)
        return dx


# --------------------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------------------
class GroupNormOp:
    """GroupNorm(32, C) (+ fused SiLU).  n_s samples of `rows` rows: 2-D norm -> sample = frame,
    3-D (TemporalResnetBlock) -> sample = clip with rows = T*HW (the group spans all frames)."""

    def __init__(self, mod: nn.GroupNorm, silu: bool):
        self.mod, self.silu = mod, silu
        self.C, self.eps = mod.num_channels, mod.eps
        if mod.weight.requires_grad:
            raise NotImplementedError("GroupNorm affine grads are outside this round's trainable set")

    def want(self, rt: Runtime, n_s: int, rows: int):
        """-> (stats, rows, cg): what the GEMM that writes this norm's input gets as `gn` (ops.gemm_act) so that the statistics are
        there when `fwd(..., pre=)` runs.  The buffer is a zeroed slice of the sweep's arena; None when the fusion is switched off."""
        if not rt.fuse_gn_stats:
            return None
        stats, pz = rt.take_zeroed(K.GN_REPLICAS * n_s * GN_GROUPS * K.GN_STAT_FLOATS)
        if not pz:
            rt.k.zero(stats)
        return stats, rows, self.C // GN_GROUPS

    def fwd(self, rt: Runtime, x: torch.Tensor, n_s: int, rows: int, pre=None):
        """pre = (stats buffer from `want`, filled): filled -- the GEMM that wrote x took the statistics; not filled -- the (zeroed)
        buffer is used for the pass over x here.  None: a buffer of this norm's own."""
        y = rt.empty(n_s * rows, self.C)
        if pre is not None:
            stats, filled = pre
            if not filled:
                rt.k.gn_stats(x, stats, n_s, rows, self.C, GN_GROUPS, prezeroed=1)
            rt.k.gn_apply(x, stats, self.mod.weight.data, self.mod.bias.data, y, n_s, rows, self.C, GN_GROUPS, self.eps, self.silu)
            return y, stats
        stats, pz = rt.take_zeroed(K.GN_REPLICAS * n_s * GN_GROUPS * K.GN_STAT_FLOATS)
        rt.k.gn_stats(x, stats, n_s, rows, self.C, GN_GROUPS, prezeroed=pz)
        rt.k.gn_apply(x, stats, self.mod.weight.data, self.mod.bias.data, y, n_s, rows, self.C, GN_GROUPS,
                      self.eps, self.silu)
        return y, stats

    def bwd(self, rt: Runtime, dy, x, stats, n_s: int, rows: int, add: Optional[torch.Tensor] = None):
        """The backward statistics stay a pass of their own over (dy, x): round 4 built them into the store loop of the data-gradient GEMM that
        writes dy (the norm's input x read beside it) and measured the step 0.12 ms SLOWER (profiles/r4_ab_in_step.txt) -- the pass
        reads dy and x at 3 TB/s beside nothing else, the store loop read x through a workgroup that had just finished its K-loop."""
        dx = rt.empty(n_s * rows, self.C)
        g, b = self.mod.weight.data, self.mod.bias.data
        bst, pz = rt.take_zeroed(K.GN_REPLICAS * n_s * GN_GROUPS * K.GN_STAT_FLOATS)
        rt.k.gn_bwd_stats(dy, x, stats, g, b, bst, n_s, rows, self.C, GN_GROUPS, self.eps, self.silu, prezeroed=pz)
        rt.k.gn_bwd_apply(dy, x, stats, bst, g, b, add, dx, n_s, rows, self.C, GN_GROUPS, self.eps, self.silu)
        return dx


class LayerNormOp:
    def __init__(self, mod: nn.LayerNorm):
        self.mod = mod
        self.C, self.eps = mod.normalized_shape[0], mod.eps
        self.trainable = mod.weight.requires_grad

    def fwd(self, rt: Runtime, x: torch.Tensor, M: int):
        y = rt.empty(M, self.C)
        stats = rt.f32(M, 2)
        rt.k.ln_fwd(x, self.mod.weight.data, self.mod.bias.data, y, stats, M, self.C, self.eps)
        return y, stats

    def bwd(self, rt: Runtime, dy, x, stats, M: int, add: Optional[torch.Tensor] = None, add2: Optional[torch.Tensor] = None,
            add2_scale: float = 1.0):
        """dx = LN'(dy) + add + add2_scale * add2 (gradient fan-in folded into the one pass that writes dx)."""
        dx = rt.empty(M, self.C)
        dg = self.mod.weight.grad if self.trainable else None
        db = self.mod.bias.grad if self.trainable else None
        if self.trainable and rt.batch_small:        # the partial rows wait for the sweep's one reducing launch (Runtime.flush_deferred)
            nblk = K.ln_bwd_blocks(M, self.C)
            scratch = rt.f32(nblk * 2 * self.C)
            rt.k.ln_bwd(dy, x, stats, self.mod.weight.data, add, dx, dg, db, M, self.C, scratch=scratch, add2=add2,
                        add2_scale=add2_scale, defer_reduce=True)
            rt.defer_ln_reduce((scratch, dg, db, nblk, self.C))
            return dx
        scratch = rt.f32(K.LN_PARTIAL_ROWS * 2 * self.C) if self.trainable else None
        rt.k.ln_bwd(dy, x, stats, self.mod.weight.data, add, dx, dg, db, M, self.C, scratch=scratch, add2=add2,
                    add2_scale=add2_scale)
        return dx

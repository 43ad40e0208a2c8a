"""Where the waves of the step's main kernels spend their cycles: a few representative launches, standalone, for two rocprofv3 counter passes.

    python tools/stall_pmc.py run                       # the launches (12 per case, rotating over buffer sets); run it under
        rocprofv3 --pmc <counters of one pass> --kernel-trace --output-format csv -d <dir> -- python tools/stall_pmc.py run
    python tools/stall_pmc.py report <dir1> <dir2> ...  # joins the counter CSVs of the passes: per case, averages over launches 3..12

Passes (8 SQ slots each, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
  A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
  B: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall)
+ ACTIVE_INST_ANY ~ WAVE_CYCLES.  Cases are told apart by their grid size in the trace (each case has a distinct one)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases(K):
    """name -> (callable(be, i), flops)"""
    import torch
    dev, dt = torch.device("cuda"), torch.float16
    out = {}

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dt)

    def nt(name, M, N, Kd, variant, gather=None, a_rows=None, lda=None, epi=0, res=False):
        As = [rnd(a_rows or M, lda or Kd) for _ in range(4)]
        B = rnd(N, Kd, scale=Kd ** -0.5)
        Cs = [torch.zeros(M, N, device=dev, dtype=dt) for _ in range(4)]
        aux = [torch.zeros(M, N // 2, device=dev, dtype=dt) for _ in range(4)] if epi == 1 else None
        Rs = [rnd(M, N) for _ in range(4)] if res else None
        def run(be, i):
            kw = dict(gather=gather, variant=variant)
            if res:
                kw.update(res=Rs[(i + 1) % 4], ldres=N)
            if epi == 1:
                kw.update(epilogue=K.EPI_GEGLU_FWD, aux_out=aux[i % 4], aux_dim=N // 2)
            be.gemm(As[i % 4], B, Cs[i % 4], M, N, Kd, lda or Kd, Kd, N, **kw)
        out[name] = (run, 2.0 * M * N * Kd)

    def geglu_bwd(name, M, F, Kd, variant):
        dys = [rnd(M, Kd) for _ in range(3)]
        wt = rnd(F, Kd, scale=Kd ** -0.5)
        pres = [rnd(M, 2 * F) for _ in range(3)]
        dpre = [torch.zeros(M, 2 * F, device=dev, dtype=dt) for _ in range(3)]
        def run(be, i):
            be.gemm(dys[i % 3], wt, dpre[i % 3], M, F, Kd, Kd, Kd, 2 * F, variant=variant, epilogue=K.EPI_GEGLU_BWD, aux_in=pres[i % 3], aux_dim=F)
        out[name] = (run, 2.0 * M * F * Kd)

    def tn(name, R, N, Kd, sk, stages):
        As = [rnd(R, N) for _ in range(3)]
        Bs = [rnd(R, Kd) for _ in range(3)]
        C = torch.zeros(max(sk, 1), N, Kd, device=dev)
        def run(be, i):
            be.gemm_tn(As[i % 3], Bs[i % 3], C, R, N, Kd, N, Kd, Kd, out_mode=K.OUT_F32_SLAB if sk > 1 else K.OUT_F32, split_k=sk, stages=stages)
        out[name] = (run, 2.0 * R * N * Kd)

    nt("nt conv3x3 L0 35840x320x2880 v6", 35840, 320, 2880, 6, K.Gather(K.GATHER_CONV3X3, n_img=14, hi=40, wi=64, ho=40, wo=64, cin=320, stride=1, lda=320), 35840, 320)
    nt("nt conv3x3 L1 8960x640x5760 v22", 8960, 640, 5760, 22, K.Gather(K.GATHER_CONV3X3, n_img=14, hi=20, wi=32, ho=20, wo=32, cin=640, stride=1, lda=640), 8960, 640)
    nt("nt geglu fwd L0 35840x2560x320 v26", 35840, 2560, 320, 26, epi=1)
    geglu_bwd("nt geglu bwd L0 35840x1280x320 v26", 35840, 1280, 320, 26)
    nt("nt linear L0 35840x320x1280 v6", 35840, 320, 1280, 6)
    nt("nt linear L2 2240x1280x1280 v24", 2240, 1280, 1280, 24)
    nt("nt linear+res L0 35840x320x1280 v6", 35840, 320, 1280, 6, res=True)
    nt("nt linear+res L1 8960x640x2560 v22", 8960, 640, 2560, 22, res=True)
    nt("nt linear+res L2 2240x1280x5120 v24", 2240, 1280, 5120, 24, res=True)
    nt("nt linear L1 8960x640x5120 v22", 8960, 640, 5120, 22)
    tn("tn L0 2560x320 over 35840 sk8 128x128", 35840, 2560, 320, 8, 0)
    tn("tn L0 320x1280 over 35840 sk16 128x128", 35840, 320, 1280, 16, 0)
    tn("tn L2 10240x1280 over 2240 256x256", 2240, 10240, 1280, 1, 18)
    tn("tn L2 1280x5120 over 2240 128x128", 2240, 1280, 5120, 1, 0)
    tn("tn L1 5120x640 over 8960 sk4 128x128", 8960, 5120, 640, 4, 0)
    return out


def run():
    import torch
    from svd_xtend_amd import kernels as K
    be = K.backend()
    cs = cases(K)
    marks = []
    for name, (fn, fl) in cs.items():
        n0 = be.n_calls
        for i in range(12):
            fn(be, i)
        torch.cuda.synchronize()
        marks.append((name, fl))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(marks, open(os.path.join(ROOT, "gpurun_out", "stall_cases.json"), "w"))


def timeit():
    """python tools/stall_pmc.py time: us per launch of every case (HIP events around 40 launches, buffers rotating), no profiler"""
    import torch
    from svd_xtend_amd import kernels as K
    be = K.backend()
    for name, (fn, fl) in cases(K).items():
        for i in range(6):
            fn(be, i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            fn(be, i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print(f"{name:48s} {us:8.1f} us  {fl / us / 1e6:7.0f} TFLOP/s", flush=True)


def report(dirs):
    marks = json.load(open(os.path.join(ROOT, "gpurun_out", "stall_cases.json")))
    per_case = [defaultdict(float) for _ in marks]
    dur = [0.0] * len(marks)
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        rows = defaultdict(dict)
        for f in files:
            for r in csv.DictReader(open(f)):
                i = int(r["Dispatch_Id"])
                rows[i]["name"] = r["Kernel_Name"]
                rows[i][r["Counter_Name"]] = float(r["Counter_Value"])
                if "Start_Timestamp" in r and r.get("End_Timestamp"):
                    rows[i]["dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                i = int(r["Dispatch_Id"])
                if i in rows:
                    rows[i]["dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        # the launches of a case are 12 consecutive dispatches of one kernel family (gemm_v4 / gemm_tn / gemm_tn8); zero-fills and
        # torch's own kernels in between are skipped by name
        ids = [i for i in sorted(rows) if any(t in rows[i]["name"] for t in ("gemm_v4_kernel", "gemm_tn_kernel", "gemm_tn8_kernel"))]
        if len(ids) != 12 * len(marks):
            print(f"{d}: {len(ids)} GEMM dispatches, expected {12 * len(marks)}", file=sys.stderr)
        for c in range(len(marks)):
            sel = ids[c * 12 + 2:c * 12 + 12]
            for i in sel:
                for k, v in rows[i].items():
                    if k not in ("name", "dur"):
                        per_case[c][k] += v / len(sel)
                dur[c] = max(dur[c], sum(rows[i].get("dur", 0.0) for i in sel) / max(len(sel), 1))
    for (name, fl), cnt, us in zip(marks, per_case, dur):
        wc = cnt.get("SQ_WAVE_CYCLES", 0.0)
        print(f"\n{name}   ({fl / 1e9:.1f} GFLOP" + (f", {us:.1f} us under the profiler = {fl / us / 1e6:.0f} TFLOP/s)" if us else ")"))
        if wc:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if k in cnt:
                    print(f"    {k:24s} {cnt[k] / wc:6.3f} of SQ_WAVE_CYCLES")
        mf = cnt.get("SQ_INSTS_MFMA", 0.0)
        if mf:
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
                if k in cnt:
                    print(f"    {k:24s} {cnt[k] / mf:6.3f} per MFMA")
            if "SQ_LDS_IDX_ACTIVE" in cnt:
                print(f"    LDS bank-conflict cycles {cnt.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(cnt['SQ_LDS_IDX_ACTIVE'], 1.0):6.3f} of SQ_LDS_IDX_ACTIVE"
                      f"   ({cnt['SQ_LDS_IDX_ACTIVE'] / mf:.2f} LDS-array cycles per MFMA)")
            if "GRBM_GUI_ACTIVE" in cnt and "SQ_VALU_MFMA_BUSY_CYCLES" in cnt:
                print(f"    MFMA busy               {cnt['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cnt['GRBM_GUI_ACTIVE'] / 8):6.3f} of (1024 SIMDs x active cycles)")
        print("    raw: " + ", ".join(f"{k}={v:.3g}" for k, v in sorted(cnt.items())))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "time":
        timeit()
    else:
        report(sys.argv[2:])

"""MFMA-pipe utilisation per kernel symbol over the real training step, from one rocprofv3 counter pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- \
        python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
    python tools/pmc_mfma.py <dir>/.../*_counter_collection.csv [out.json]
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (16 cycles per v_mfma_f32_16x16x32_f16/bf16 wave instruction);
GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (one GRBM each: a 52 us GEMM reports ~1.07 M), so the kernel's active cycles
are GRBM_GUI_ACTIVE / 8 and utilisation = busy / (1024 x active / 8).  Cross-check: the GEMM family's figure equals bench.py's
flop-derived `roofline.frac`.  Only the LAST optimizer step of the trace is used (split at the adamw kernel).
    python tools/pmc_mfma.py --from-json <out.json>     re-prints a stored summary"""
import csv
import json
import re
import sys
from collections import defaultdict

SIMDS = 1024
XCDS = 8


def sym(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"_GLOBAL__N_1\d+(\w+?_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    m = re.search(r"N\d+_GLOBAL__N_1(\d+)(\w+?_kernel)(I\w+?E)?Ev", name)
    return (m.group(2) + (m.group(3) or "")) if m else name[:60]


def report(agg, out_path=None):
    tot_busy = sum(a[1] for a in agg.values())
    tot_act = sum(a[2] for a in agg.values())
    out = {}
    print(f"{'kernel':44s} {'launches':>8s} {'active Mcyc (per XCD)':>22s} {'MFMA busy / (1024 SIMDs x active)':>34s}")
    for k, (n, busy, act) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        util = busy / (SIMDS * act / XCDS) if act else 0.0
        out[k] = dict(launches=n, grbm_gui_active_sum=act, mfma_busy_cycles=busy, mfma_util=util)
        if act / max(tot_act, 1) > 0.002:
            print(f"{k[:44]:44s} {n:8d} {act/XCDS/1e6:22.2f} {util:34.3f}")
    print(f"{'whole step':44s} {sum(a[0] for a in agg.values()):8d} {tot_act/XCDS/1e6:22.2f} {tot_busy/(SIMDS*tot_act/XCDS):34.3f}")
    out["_step"] = dict(grbm_gui_active_sum=tot_act, mfma_busy_cycles=tot_busy, mfma_util=tot_busy / (SIMDS * tot_act / XCDS))
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


def main():
    if sys.argv[1] == "--from-json":
        d = json.load(open(sys.argv[2]))
        report({k: [v["launches"], v["mfma_busy_cycles"], v.get("grbm_gui_active_sum", v.get("active_cycles"))] for k, v in d.items()
                if k != "_step"})
        return
    rows = defaultdict(dict)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    ends = [i for i in ids if "adamw" in rows[i]["name"]]
    if len(ends) >= 2:
        ids = [i for i in ids if ends[-2] < i <= ends[-1]]
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for i in ids:
        r = rows[i]
        a = agg[sym(r["name"])]
        a[0] += 1
        a[1] += r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += r.get("GRBM_GUI_ACTIVE", 0.0)
    report(agg, sys.argv[2] if len(sys.argv) > 2 else None)


if __name__ == "__main__":
    main()

"""Aggregate tools/step_trace.py's JSON into time per op family and latent level (which part of the step the kernel time sits in)."""
import collections
import json
import sys


def main():
    d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_trace.json"))
    cat = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in d["calls"]:
        a, e = r["args"], r["entry"]
        if e in ("svdx_gemm", "svdx_gemm_dual", "svdx_gemm_gn"):
            M, N, K = a[3], a[4], a[5]
            lvl = {35840: "L0", 8960: "L1", 2240: "L2", 560: "L3"}.get(M, str(M))
            epi = a[22] if e == "svdx_gemm" else 0
            kind = "conv" if r["extra"] else ("geglu_fwd" if epi == 1 else "geglu_bwd" if epi == 2 else ("linear K<=1280" if K <= 1280 else "linear K>1280"))
            k, fl = (lvl, kind), 2.0 * M * N * K
        elif e == "svdx_gemm_tn":
            k, fl = ("", "weight-grad (tn)"), 2.0 * a[3] * a[4] * a[5]
        else:
            k, fl = ("", e[5:]), 0.0
        c = cat[k]
        c[0] += 1
        c[1] += r["us"]
        c[2] += fl
    tot = sum(c[1] for c in cat.values())
    print(f"# joined kernel time {tot / 1e3:.2f} ms of {d['kernel_ms_per_step']:.2f} ms per step")
    for k, c in sorted(cat.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[0]:3s} {k[1]:20s} n={c[0]:4d} {c[1] / 1e3:7.2f} ms" + (f"  {c[2] / c[1] / 1e6:7.1f} TF/s" if c[2] else ""))


if __name__ == "__main__":
    main()

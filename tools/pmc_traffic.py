"""HBM traffic per kernel family from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950).
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads, which is what every kernel here issues, so it is doubled.  WRITE_SIZE is uncalibrated (taken as is).
Only the LAST optimizer step of the trace is used (split at adamw_kernel)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel)(I\w+?E)?E?v.*? (grid=\d+)$", name)
    return f"{m.group(1).split(chr(95)+chr(78)+chr(95))[-1]}{m.group(2) or str()} {m.group(3)}" if m else name[-60:]


def family(name: str) -> str:
    for key, fam in (("gemm_v4", "gemm"), ("gemm_tn", "gemm"), ("gemm_kernel", "gemm"), ("gemm_finalize", "gemm_finalize"),
                     ("attn", "attention"), ("gn_", "groupnorm"), ("ln_", "layernorm"), ("adamw", "adamw")):
        if key in name:
            return fam
    return "other"


def load(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"] + " grid=" + r["Grid_Size"], float(r["Counter_Value"])))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw" in r[1]]
    if len(ends) >= 2:
        rows = rows[ends[-2] + 1:ends[-1] + 1]
    return rows


def by_shape(fetch, write, log):
    """Fabric-side bytes per GEMM problem: the k-th dispatch of a kernel family in the step belongs to the k-th logged call of the
    entries that launch it (tools/step_trace.py's join).  Algorithmic bytes = each operand once + the result once (float slabs: 4 B x
    slices), activations of a convolution once (not once per tap) -- what bench.py's `algorithmic_bytes_per_launch` counts."""
    fams = [(("gemm_v4", "gemm_kernel"), ("svdx_gemm", "svdx_gemm_dual", "svdx_gemm_gn")), (("gemm_tn",), ("svdx_gemm_tn",))]
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for keys, entries in fams:
        calls = [c for c in log if c[0] in entries]
        fr = [r for r in fetch if any(k in r[1] for k in keys) and "finalize" not in r[1]]
        wr = [r for r in write if any(k in r[1] for k in keys) and "finalize" not in r[1]]
        if len(fr) != len(calls) or len(wr) != len(calls):
            print(f"# by-shape: {keys}: {len(fr)} / {len(wr)} dispatches vs {len(calls)} logged calls -- skipped")
            continue
        for c, f, w in zip(calls, fr, wr):
            a = c[1]
            if c[0] == "svdx_gemm_tn":
                R, N, K, sk = a[3], a[4], a[5], a[12]
                key = ("tn", N, K, R, 0, sk)
                alg = 2.0 * R * N + 2.0 * R * K + 4.0 * N * K * max(1, sk)
            else:
                M, N, K = a[3], a[4], a[5]
                dual = c[0] == "svdx_gemm_dual"
                g = tuple(c[2]) if len(c) > 2 and c[2] else 0
                sk, om, epi = (a[20], a[18], a[22]) if c[0] == "svdx_gemm" else (1, a[18], 0) if dual else (1, 0, 0)
                taps = {0: 1, 1: 9, 2: 9, 3: 3, 4: 9}.get(g[0] if g else 0, 1)
                osz = 4 * sk if om != 0 else 2
                alg = 2.0 * M * K / taps + 2.0 * N * K + osz * M * N + (2.0 * M * N if a[14] is not None else 0.0)
                if c[0] == "svdx_gemm_gn" and a[23] is not None:
                    alg += 2.0 * M * N              # the backward-statistics form also reads the norm's input x [M, N]
                if epi == 1:
                    alg += 2.0 * M * N / 2          # GEGLU forward also writes h [M, F]
                elif epi == 2:
                    alg += 2.0 * M * N * 3          # GEGLU backward reads pre [M, 2F] and writes d(pre) [M, 2F] instead of [M, F]
                key = ("nt", M, N, K, g, sk, epi)
            e = agg[key]
            e[0] += 1
            e[1] += 2.0 * f[2] * 1024
            e[2] += w[2] * 1024
            e[3] += alg
    print(f"# by shape: fabric bytes against algorithmic bytes per GEMM problem, largest excess first")
    print(f"{'problem':64s} {'n':>4s} {'read MB':>9s} {'write MB':>9s} {'algo MB':>9s} {'ratio':>6s} {'excess GB/step':>14s}")
    tm = ta = 0.0
    for key, (n, rd, wr, alg) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2] - kv[1][3])):
        tm += rd + wr
        ta += alg
        print(f"{str(key):64s} {n:4d} {rd / n / 1e6:9.1f} {wr / n / 1e6:9.1f} {alg / n / 1e6:9.1f} {(rd + wr) / alg:6.2f} {(rd + wr - alg) / 1e9:14.2f}")
    print(f"# GEMM family total {tm / 1e9:.2f} GB against {ta / 1e9:.2f} GB algorithmic = {tm / max(ta, 1):.2f}x")


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for _, name, v in fetch:
        a = agg[family(name)]
        a[0] += 1
        a[1] += 2.0 * v * 1024
    for _, name, v in write:
        agg[family(name)][2] += v * 1024
    out = {}
    for fam, (n, rd, wr) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        out[fam] = dict(launches=n, read_bytes=rd, write_bytes=wr, bytes_per_launch=(rd + wr) / max(n, 1))
        print(f"{fam:16s} launches {n:5d}  read {rd/1e9:8.3f} GB  write {wr/1e9:8.3f} GB  per launch {(rd+wr)/max(n,1)/1e6:8.2f} MB")
    if "--by-grid" in sys.argv:
        g = defaultdict(lambda: [0, 0.0, 0.0])
        for _, name, v in fetch:
            a = g[short(name)]
            a[0] += 1
            a[1] += 2.0 * v * 1024
        for _, name, v in write:
            g[short(name)][2] += v * 1024
        for k, (n, rd, wr) in sorted(g.items(), key=lambda kv: -kv[1][1])[:70]:
            print(f"  {k:72s} n={n:4d} read/launch {rd/n/1e6:8.2f} MB  write/launch {wr/n/1e6:8.2f} MB  total read {rd/1e9:6.2f} GB")
    if "--by-shape" in sys.argv:
        by_shape(fetch, write, json.load(open(sys.argv[sys.argv.index("--by-shape") + 1])))
    tot_r = sum(v["read_bytes"] for v in out.values())
    tot_w = sum(v["write_bytes"] for v in out.values())
    print(f"step total: read {tot_r/1e9:.2f} GB, write {tot_w/1e9:.2f} GB")
    out["_step_total"] = dict(read_bytes=tot_r, write_bytes=tot_w)
    if len(sys.argv) > 3 and not sys.argv[3].startswith("--"):
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()

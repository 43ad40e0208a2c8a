"""Join a rocprofv3 kernel trace of graph-replayed steps with the ordered libsvdx call list of one step.

    rocprofv3 --kernel-trace -d gpurun_out/tr -o tr -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline \
        --launch-log gpurun_out/launch_log.json
    python tools/step_trace.py gpurun_out/tr/<...>_results.db gpurun_out/launch_log.json [out.json]

Every libsvdx entry launches a fixed list of kernels, so the k-th dispatch of a kernel family inside one step of the trace belongs to
the k-th logged call of the entries that launch that family.  Steps are cut at the optimizer kernel.  Durations are averaged over the
complete steps found in the trace (the first one is dropped)."""
import json
import re
import sqlite3
import sys

FAMILIES = [   # (kernel-name regex, entries whose calls launch exactly one such kernel each, in call order)
    (r"gemm_v4_kernel|[^_]gemm_kernel|gemm_v5", ["svdx_gemm", "svdx_gemm_dual", "svdx_gemm_gn"]),
    (r"gemm_tn", ["svdx_gemm_tn"]),
    (r"gemm_finalize_kernel", ["svdx_gemm_finalize", "svdx_gemm_finalize_gn"]),
    (r"attn_fwd_kernel", ["svdx_attn_fwd"]), (r"attn_bwd_dkv_kernel", ["svdx_attn_bwd_dkv"]), (r"attn_bwd_dq_kernel", ["svdx_attn_bwd_dq"]),
    (r"attn_bwd_prep_kernel", ["svdx_attn_bwd_prep"]),
    (r"tattn_fwd_kernel", ["svdx_tattn_fwd"]), (r"tattn_bwd_kernel", ["svdx_tattn_bwd"]), (r"tsa_fwd_kernel", ["svdx_tsa_fwd"]),
    (r"tsa_bwd_kernel", ["svdx_tsa_bwd"]),
    (r"gn_reduce_kernel", ["svdx_gn_stats", "svdx_gn_bwd_stats"]), (r"gn_apply_kernel", ["svdx_gn_apply", "svdx_gn_bwd_apply"]),
    (r"ln_fwd(16)?_kernel", ["svdx_ln_fwd"]), (r"ln_bwd(16)?_kernel", ["svdx_ln_bwd"]),
    (r"binary_kernel", ["svdx_add", "svdx_blend", "svdx_blend_bwd"]), (r"add_rowvec_kernel", ["svdx_add_rowvec"]),
    (r"colsum_kernel", ["svdx_colsum"]), (r"concat2_kernel", ["svdx_concat2"]), (r"split2_kernel", ["svdx_split2"]),
    (r"sum2x2_kernel", ["svdx_sum2x2"]), (r"small_linear_n", ["svdx_small_linear"]), (r"outer_acc_kernel", ["svdx_outer_acc"]),
    (r"adamw_tiled_kernel", ["svdx_adamw_tiled"]), (r"check_finite_kernel", ["svdx_check_finite"]), (r"edm_loss_kernel", ["svdx_edm_loss"]),
]
STEP_END = r"adamw_tiled_kernel|adamw_kernel"


def main():
    db = sqlite3.connect(sys.argv[1])
    log = json.load(open(sys.argv[2]))
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ix = {c: i for i, c in enumerate(cols)}
    rows = cur.execute("select * from kernels").fetchall()
    recs = sorted((r[ix["start"]], r[ix["end"]], r[ix["name"]], r[ix["grid_x"]]) for r in rows)
    ends = [i for i, r in enumerate(recs) if re.search(STEP_END, r[2])]
    steps = [recs[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])]
    # keep the steps that have the modal dispatch count (graph-replayed ones), drop the first of them
    from collections import Counter
    modal = Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == modal][1:]
    print(f"# {len(recs)} dispatches, {len(steps)} complete steps of {modal} dispatches used", file=sys.stderr)
    out = []
    covered = 0
    for rx, entries in FAMILIES:
        calls = [(i, c) for i, c in enumerate(log) if c[0] in entries]
        per_step = [[r for r in s if re.search(rx, r[2])] for s in steps]
        n = len(per_step[0]) if per_step else 0
        if n != len(calls) or any(len(p) != n for p in per_step):
            if n or calls:
                print(f"# family {rx}: {n} dispatches per step vs {len(calls)} logged calls -- skipped", file=sys.stderr)
            continue
        for k, (i, c) in enumerate(calls):
            d = [(p[k][1] - p[k][0]) / 1e3 for p in per_step]
            out.append({"seq": i, "entry": c[0], "args": c[1], "extra": c[2] if len(c) > 2 else None, "kernel": re.sub(r"\(.*", "", per_step[0][k][2])[:80],
                        "grid": per_step[0][k][3], "us": sum(d) / len(d), "us_min": min(d)})
            covered += sum(d) / len(d)
    tot = sum((r[1] - r[0]) / 1e3 for s in steps for r in s) / max(1, len(steps))
    span = sum((s[-1][1] - s[0][0]) / 1e3 for s in steps) / max(1, len(steps))
    print(f"# kernel time per step {tot / 1e3:.3f} ms (span {span / 1e3:.3f} ms), joined {covered / 1e3:.3f} ms", file=sys.stderr)
    out.sort(key=lambda r: r["seq"])
    json.dump({"kernel_ms_per_step": tot / 1e3, "span_ms_per_step": span / 1e3, "calls": out}, open(sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/step_trace.json", "w"))
    # GEMM table
    agg = {}
    for r in out:
        a = r["args"]
        if r["entry"] in ("svdx_gemm", "svdx_gemm_dual", "svdx_gemm_gn"):
            M, N, K = a[3], a[4], a[5]
            dual = r["entry"] == "svdx_gemm_dual"
            gn = r["entry"] == "svdx_gemm_gn"            # ... gather, zero_page, alpha, variant, gn_stats, rows, cg, gn_bwd: unsplit, activation output
            key = ("nt", M, N, K, tuple(r["extra"]) if r["extra"] else 0, 1 if (dual or gn) else a[20], a[19] if gn else (a[21] if not dual else a[20]),
                   (3 if a[23] is not None else 4) if gn else (a[22] if not dual else 0),      # epi column: 4 = + GroupNorm statistics, 3 = + backward statistics
                   a[9] is not None, a[10] is not None, a[14] is not None, 0 if gn else a[18])
            fl = 2.0 * M * N * K
        elif r["entry"] == "svdx_gemm_tn":
            R, N, K = a[3], a[4], a[5]
            key = ("tn", N, K, R, 0, a[12], 0, 0, a[9] is not None, False, False, a[11])
            fl = 2.0 * R * N * K
        elif r["entry"] == "svdx_gemm_finalize":
            key = ("fin", a[5], a[6], a[1], 0, 0, 0, 0, a[8] is not None, a[9] is not None, a[13] is not None, a[4])
            fl = 0.0
        elif r["entry"] == "svdx_gemm_finalize_gn":
            key = ("fin", a[4], a[5], a[1], 0, 0, 0, 4, a[7] is not None, a[8] is not None, a[12] is not None, 0)
            fl = 0.0
        else:
            continue
        e = agg.setdefault(key, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += r["us"]
        e[2] += fl
    print(f"{'kind':4s} {'M':>6s} {'N':>6s} {'K':>6s} {'gather':>18s} {'spl':>3s} {'var':>3s} {'epi':>3s} b r s {'om':>2s} {'n':>4s} {'tot_us':>9s} {'avg_us':>8s} {'TF/s':>7s}")
    for key, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        kind, M, N, K, g, sp, var, epi, b, rv, rs, om = key
        print(f"{kind:4s} {M:6d} {N:6d} {K:6d} {str(g):>18s} {sp:3d} {var:3d} {epi:3d} {int(b)} {int(rv)} {int(rs)} {om:2d} {n:4d} {us:9.1f} {us / n:8.2f} {fl / us / 1e6 if us else 0:7.1f}")
    # everything else by entry
    other = {}
    for r in out:
        if r["entry"] in ("svdx_gemm", "svdx_gemm_dual", "svdx_gemm_gn", "svdx_gemm_tn", "svdx_gemm_finalize", "svdx_gemm_finalize_gn"):
            continue
        ints = tuple(x for x in r["args"] if isinstance(x, int) and 0 < x < (1 << 24))[:6]
        e = other.setdefault((r["entry"], r["kernel"][-40:], ints), [0, 0.0])
        e[0] += 1
        e[1] += r["us"]
    print()
    for key, (n, us) in sorted(other.items(), key=lambda kv: -kv[1][1])[:80]:
        print(f"{key[0]:22s} {key[1]:42s} {str(key[2]):44s} {n:4d} {us:9.1f} {us / n:8.2f}")


if __name__ == "__main__":
    main()

"""Summarise a rocprofv3 rocpd sqlite database into per-kernel statistics (like --stats CSV).
usage: python tools/prof_summary.py <results.db> [--skip-first-fraction F] [--by-grid]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<.*>)?\(", name)
    if m:
        t = m.group(2) or ""
        return (m.group(1) + t)[:90]
    return name[:90]


def main():
    path = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ix else "kernel_name"
    recs = []
    for r in rows:
        recs.append((r[ix["start"]], r[ix["end"]], r[ix[name_c]], r[ix.get("grid_x", ix.get("grid_size_x", 0))] if ("grid_x" in ix or "grid_size_x" in ix) else 0))
    recs.sort()
    # keep the LAST step only: the bench runs warmup + steps identical iterations; take the final 1/n of dispatches
    frac = 0.0
    for a in sys.argv:
        if a.startswith("--last-fraction="):
            frac = float(a.split("=")[1])
    if frac:
        recs = recs[int(len(recs) * (1 - frac)):]
    agg = {}
    for s, e, n, gx in recs:
        key = short(n) + (f" grid={gx}" if by_grid else "")
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    span = (recs[-1][1] - recs[0][0]) / 1e3
    print(f"# {len(recs)} dispatches, kernel time {tot/1e3:.3f} ms, wall span {span/1e3:.3f} ms")
    print(f"{'kernel':92s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:92s} {c:7d} {t:12.1f} {t/c:10.2f} {100*t/tot:6.2f}")


if __name__ == "__main__":
    main()

"""Idle-time analysis of a rocprofv3 kernel trace (rocpd sqlite): splits the trace into optimizer steps (at adamw_kernel),
then reports kernel time, wall span and the gaps between consecutive dispatches, attributed to the kernel that FOLLOWS the gap.
usage: python tools/prof_gaps.py <results.db>"""
import sqlite3
import sys

from prof_summary import short


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ix else "kernel_name"
    recs = sorted((r[ix["start"]], r[ix["end"]], short(r[ix[name_c]])) for r in cur.execute("select * from kernels"))
    steps, cur_step = [], []
    for r in recs:
        cur_step.append(r)
        if "adamw" in r[2]:
            steps.append(cur_step)
            cur_step = []
    if cur_step:
        steps.append(cur_step)
    for i, st in enumerate(steps):
        kt = sum(e - s for s, e, _ in st) / 1e6
        span = (st[-1][1] - st[0][0]) / 1e6
        gaps = [max(0, st[j + 1][0] - st[j][1]) for j in range(len(st) - 1)]
        print(f"step {i}: {len(st)} dispatches, kernel {kt:.2f} ms, span {span:.2f} ms, idle {sum(gaps)/1e6:.2f} ms, "
              f"median gap {sorted(gaps)[len(gaps)//2]/1e3 if gaps else 0:.2f} us")
    st = steps[-1] if "adamw" in steps[-1][-1][2] else steps[-2]
    gaps = [(max(0, st[j + 1][0] - st[j][1]), st[j + 1][2], st[j][2]) for j in range(len(st) - 1)]
    edges = [0, 1e3, 2e3, 4e3, 8e3, 16e3, 32e3, 64e3, 1e12]
    print("gap histogram (last full step):")
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = [g for g, _, _ in gaps if lo <= g < hi]
        print(f"  {lo/1e3:5.0f}-{hi/1e3 if hi < 1e11 else float('inf'):5.0f} us: {len(sel):5d} gaps, {sum(sel)/1e6:7.3f} ms")
    for which, col in (("following", 1), ("preceding", 2)):
        agg = {}
        for g in gaps:
            a = agg.setdefault(g[col], [0, 0])
            a[0] += 1
            a[1] += g[0]
        print(f"idle attributed to the {which} kernel:")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"  {k[:80]:80s} {c:5d} {t/1e6:8.3f} ms {t/c/1e3:7.2f} us/gap")


if __name__ == "__main__":
    main()

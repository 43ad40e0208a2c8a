import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]; ix = {c: i for i, c in enumerate(cols)}
rows = sorted((r[ix["start"]], r[ix["end"]], r[ix["name"]], r[ix["grid_x"]]) for r in cur.execute("select * from kernels").fetchall())
ends = [i for i, r in enumerate(rows) if "adamw_tiled" in r[2]]
steps = [rows[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])]
print("steps:", [len(s) for s in steps])
last4 = steps[-4:]
for i, s in enumerate(last4):
    print(i, "n", len(s), "kernel ms", sum(r[1] - r[0] for r in s) / 1e6, "span ms", (s[-1][1] - s[0][0]) / 1e6)
a, b = last4[1], last4[2]          # replay before / after the copy
ca = collections.Counter(re.sub(r"\(.*", "", r[2])[:70] for r in a); cb = collections.Counter(re.sub(r"\(.*", "", r[2])[:70] for r in b)
for k in sorted(set(ca) | set(cb)):
    if ca[k] != cb[k]: print("COUNT DIFF", k, ca[k], cb[k])
da = collections.defaultdict(float); dbb = collections.defaultdict(float)
for r in a: da[re.sub(r"\(.*", "", r[2])[:70]] += (r[1] - r[0]) / 1e3
for r in b: dbb[re.sub(r"\(.*", "", r[2])[:70]] += (r[1] - r[0]) / 1e3
for k in sorted(da, key=lambda k: -abs(da[k] - dbb.get(k, 0)))[:12]:
    print(f"{k:72s} before {da[k]:9.1f} us  after {dbb.get(k, 0):9.1f} us")
# per-dispatch comparison when the lists align
if len(a) == len(b):
    big = sorted(((abs((y[1]-y[0]) - (x[1]-x[0])) / 1e3, i, x[2][:60], (x[1]-x[0])/1e3, (y[1]-y[0])/1e3) for i, (x, y) in enumerate(zip(a, b))), reverse=True)[:15]
    for d in big: print("dispatch", d[1], d[2], f"{d[3]:.1f} -> {d[4]:.1f} us")

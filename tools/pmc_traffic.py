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
    tot_r = sum(v["read_bytes"] for v in out.values())
    tot_w = sum(v["write_bytes"] for v in out.values())
    print(f"step total: read {tot_r/1e9:.2f} GB, write {tot_w/1e9:.2f} GB")
    out["_step_total"] = dict(read_bytes=tot_r, write_bytes=tot_w)
    if len(sys.argv) > 3 and not sys.argv[3].startswith("--"):
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()

"""Which GEMMs of the step leave CUs idle?  For every NT problem of `bench.py --gemm-table` (gpurun_out/gemm_table.json) print the
cost model's (split, tile) choice, the workgroups it launches, how many rounds they take on the chip's resident slots (256, or 512 for
the two-stage four-wave tiles that fit twice on a CU), the share of the slots the last round fills and the share of a tile that is
padding -- sorted by the time that quantisation costs.  This is the table the 192-row and 96-row tiles of round 3 came from.

    python bench.py --gemm-table --steps 20 --no-cpu-baseline      (on the GPU box; writes gpurun_out/gemm_table.json)
    python tools/fill_table.py [gpurun_out/gemm_table.json]
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svd_xtend_amd import ops  # noqa: E402


class _RT:
    gemm_variant, split_k = 4, True


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gemm_table.json")
    if not os.path.exists(path):
        # no fresh table on this machine: the tracked per-call join of the latest profiled step (tools/step_trace.py) holds the same columns
        import glob
        import re
        traces = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_trace.txt")))
        if not traces:
            raise SystemExit(f"{path} not found and no profiles/r*_step_trace.txt: run `python bench.py --gemm-table --steps 20 --no-cpu-baseline` on the GPU box")
        path = traces[-1]
        print(f"# reading {os.path.relpath(path, ROOT)} (no gpurun_out/gemm_table.json here)")
        table = []
        for line in open(path):
            m = re.match(r"(nt|tn)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\(.*?\)|0)\s+(\d+)\s+\d+\s+\d \d \d \d\s+\d+\s+(\d+)\s+([\d.]+)\s+[\d.]+\s+([\d.]+)", line)
            if m:
                table.append([m[1], int(m[2]), int(m[3]), int(m[4]), m[5], int(m[6]), int(m[7]), float(m[8]) / 1e3, float(m[9])])
    else:
        table = json.load(open(path))
    rows = []
    for kind, M, N, Kd, _gather, _spl, n, ms, tf in table:
        if kind != "nt":
            continue
        s, v = ops.choose_cfg(_RT(), M, N, Kd, N, 0)
        bm, bn, stages, waves = ops.TILE_OF_VARIANT[v]
        tiles = math.ceil(M / bm) * math.ceil(N / bn)
        slots = 512 if (stages == 2 and waves == 4) else 256
        wgs = tiles * s
        rounds = math.ceil(wgs / slots)
        fill = wgs / (rounds * slots)
        useful = M * N / (tiles * bm * bn)
        rows.append((ms * (1 - fill * useful), ms, M, N, Kd, n, f"{bm}x{bn}" + (f" s{s}" if s > 1 else ""), wgs, rounds, fill, useful, tf))
    rows.sort(reverse=True)
    print(f"{'lost ms':>8} {'ms/step':>8} {'M':>6} {'N':>6} {'K':>6} {'calls':>5} {'tile':>12} {'wgs':>5} {'rounds':>6} {'fill':>5} {'useful':>6} {'TF/s':>6}")
    for lost, ms, M, N, Kd, n, tile, wgs, rounds, fill, useful, tf in rows[:40]:
        print(f"{lost:8.2f} {ms:8.2f} {M:6d} {N:6d} {Kd:6d} {n:5d} {tile:>12} {wgs:5d} {rounds:6d} {fill:5.2f} {useful:6.2f} {tf:6.0f}")
    print("(GEGLU launches appear with the plain cost-model tile: their tile comes from ops.choose_geglu_variant)")


if __name__ == "__main__":
    main()

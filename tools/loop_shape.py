"""Print the instruction-class sequence of the hottest basic block (most MFMAs) of a kernel in a hipcc -S listing.
usage: python tools/loop_shape.py file.s <mangled-kernel-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):", s, re.M)
i = m.start()
j = s.index(".Lfunc_end", i)
body = s[i:j].split("\n")
labels = [k for k, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)]
best = None
for a, b in zip(labels, labels[1:] + [len(body)]):
    n = sum("v_mfma" in l for l in body[a:b])
    if best is None or n > best[0]:
        best = (n, a, b)
n, a, b = best
seq = []
for l in body[a:b]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    seq.append("mfma" if "mfma" in op else "dsr" if op.startswith("ds_read") else "dma" if "buffer_load" in op else op)
out, prev, cnt = [], None, 0
for op in seq + [None]:
    if op == prev:
        cnt += 1
        continue
    if prev:
        out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
    prev, cnt = op, 1
print(f"{m.group(1)}: hottest block has {n} MFMAs, {len(seq)} instructions")
print(" ".join(out))

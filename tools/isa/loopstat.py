#!/usr/bin/env python
"""Counts instructions of the innermost loop that contains a marker instruction in a hipcc -S listing.
usage: loopstat.py file.s kernel_substring marker   (design tool)"""
import re, sys
path, kern, marker = sys.argv[1:4]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and kern in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
mi = next(i for i, l in enumerate(body) if marker in l)
# innermost loop = nearest preceding label that is the target of a later backward branch
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
best = None
for i in range(mi, len(body)):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", body[i])
    if m and m.group(1) in labels and labels[m.group(1)] <= mi:
        best = (labels[m.group(1)], i)
        break
lo, hi = best
cnt = {}
for l in body[lo:hi + 1]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    cls = ("valu_mov" if op.startswith("v_mov") else "trans" if op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32")
           else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_"))
           else "wait" if op.startswith("s_waitcnt") else "salu")
    cnt[cls] = cnt.get(cls, 0) + 1
print(f"loop lines {lo}-{hi}:", dict(sorted(cnt.items())), "total", sum(cnt.values()))

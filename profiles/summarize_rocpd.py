#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max (us),
short kernel names.  usage: summarize_rocpd.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]{0,60}>)?)", name)
    s = m.group(1) if m else name[:80]
    if s.startswith("rocprim::trampoline_kernel") or len(name) > 300:
        inner = re.search(r"rocprim::(radix_sort_\w+|merge_sort_\w+|transform_impl|partition\w*|onesweep\w*|histogram\w*|scan\w*)", name)
        tag = re.search(r"lambda\(auto:1\)#(\d)", name)
        s = "rocprim::" + (inner.group(1) if inner else "kernel") + (("#" + tag.group(1)) if tag else "")
    return s[:90]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, duration from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dur / 1e3
        a[2] = min(a[2], dur / 1e3)
        a[3] = max(a[3], dur / 1e3)
    tot = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Static VALU / memory instruction counts of one kernel per source line (from a -gline-tables-only -save-temps .s):
  python3 tools/isa_lines.py FILE.s MANGLED_KERNEL_SUBSTRING [source-file-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
want_file = sys.argv[3] if len(sys.argv) > 3 else ""
m = re.search(r"^(_Z\S*%s\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel" % re.escape(name), s, re.S | re.M)
body = m.group(2)
files = dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s))
files.update(dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"([^"]+)"\s*$', s, re.M)))
cur = (0, 0)
valu = collections.Counter()
mem = collections.Counter()
for ln in body.splitlines():
    t = ln.strip()
    mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if mm:
        cur = (int(mm.group(1)), int(mm.group(2)))
        continue
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        valu[cur] += 1
    elif op.startswith(("ds_", "global_", "scratch_", "flat_", "buffer_")):
        mem[cur] += 1
tot = sum(valu.values())
print("kernel %s: %d VALU, %d memory instructions (static)" % (m.group(1)[:60], tot, sum(mem.values())))
rows = sorted(set(valu) | set(mem), key=lambda k: (files.get(k[0], ""), k[1]))
for k in rows:
    f = files.get(k[0], "?")
    if want_file in f:
        print("%-28s %5d  valu %4d  mem %3d" % (f.split("/")[-1], k[1], valu[k], mem[k]))

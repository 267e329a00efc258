#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing (tools/isa_mix.py k.s <mangled-name-prefix>)."""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    name = sys.argv[2]
    m = re.search(r'^(%s[^:\n]*):[^\n]*\n(.*?)^\s*s_endpgm' % re.escape(name), txt, re.S | re.M)
    if not m:
        sys.exit("kernel not found")
    ops = collections.Counter()
    for line in m.group(2).splitlines():
        line = line.split(';')[0].strip()
        if not line or line.startswith('.') or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    tot = sum(ops.values())
    valu = sum(v for k, v in ops.items() if k.startswith('v_'))
    salu = sum(v for k, v in ops.items() if k.startswith('s_'))
    print(m.group(1)[:60], "total", tot, "valu", valu, "salu", salu, "other", tot - valu - salu)

    def n(prefix):
        return sum(v for k, v in ops.items() if k.startswith(prefix))
    for k in ('v_div_scale', 'v_div_fmas', 'v_div_fixup', 'v_rcp_f32', 'v_fma_f32', 'v_fmac_f32', 'v_pk_', 'v_mul_f32', 'v_add_f32', 'v_sub_f32',
              'v_cndmask', 'v_mad_u64_u32', 'v_mul_hi_u32', 'v_mul_lo_u32', 'v_sqrt_f32', 'v_readlane', 'v_writelane', 'v_readfirstlane',
              's_waitcnt', 'v_cmp', 's_load', 'global_load', 'scratch_', 'ds_', 's_cbranch', 's_and', 's_or', 's_mov', 'v_mov'):
        print("   %-18s %d" % (k, n(k)))
    print(ops.most_common(12))


main()

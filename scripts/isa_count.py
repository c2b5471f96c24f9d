#!/usr/bin/env python3
"""Instruction census of the column kernels in an ISA dump (hipcc -S --cuda-device-only ... -o x.s): scripts/isa_count.py x.s [substr ...]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2:]
cur = None
res = {}
for ln in lines:
    m = re.match(r'(_ZN4pow2\S*kernel\S*):', ln)
    if m:
        cur = m.group(1)
        res[cur] = collections.Counter()
        continue
    if cur and ln.startswith('.Lfunc_end'):
        cur = None
        continue
    if cur:
        m = re.match(r'\s+([a-z_0-9]+)\s', ln)
        if m:
            res[cur][m.group(1)] += 1
for name, ops in res.items():
    if want and not any(k in name for k in want):
        continue
    valu = sum(c for o, c in ops.items() if o.startswith('v_'))
    v64 = sum(c for o, c in ops.items() if o.startswith('v_') and 'f64' in o)
    print(name[9:60], 'total', sum(ops.values()), 'valu', valu, 'f64', v64, 'rcp', ops['v_rcp_f64_e32'], 'rsq',
          ops['v_rsq_f64_e32'], 'div_scale', ops['v_div_scale_f64'], 'ds_w', ops['ds_write_b64'], 'scratch',
          sum(c for o, c in ops.items() if 'scratch' in o), 'gload', sum(c for o, c in ops.items() if o.startswith('global_load')))

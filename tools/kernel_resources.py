#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in an AMDGPU assembly file (hipcc -S --cuda-device-only)."""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    name, body = m.group(1), m.group(2)
    def g(key):
        r = re.search(key + r':\s+(\d+)', body)
        return r.group(1) if r else '?'
    print("%-72s vgpr=%4s spill=%3s lds=%6s scratch=%s" % (name[:72], g('vgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))

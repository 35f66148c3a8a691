#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in a hipcc -S device listing: python scripts/asm_all.py build/asm/kernels.s"""
import re, sys
name = None
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*\.amdhsa_kernel (\S+)", l)
    if m:
        name, cur = m.group(1), {}
        continue
    if name:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size) (\d+)", l)
        if m:
            cur[m.group(1)] = int(m.group(2))
        if ".end_amdhsa_kernel" in l:
            rows.append((name, cur))
            name = None
for n, c in rows:
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)   # fcp_ragged_kernelILi4ELb0EEEv9FcpLaunch: <V, SHARDED>
    print(f"{c.get('next_free_vgpr', 0):4d} vgpr  {c.get('private_segment_fixed_size', 0):4d} scratch  {c.get('group_segment_fixed_size', 0):6d} lds  {short[:90]}")

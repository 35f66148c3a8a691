#!/usr/bin/env python3
"""Extract one kernel from a hipcc -S device listing and summarise it (static instruction mix, registers).
  python scripts/asm_kernel.py build/asm/kernels.s 'fcp_ragged_kernelILi4ELb0' [--dump out.s]"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
body = lines[start:end]
if dump:
    open(dump, "w").write("\n".join(body))
mix = collections.Counter()
for l in body:
    m = re.match(r"^\t([a-z_0-9]+)", l)
    if not m: continue
    op = m.group(1)
    if op.startswith("v_"): k = "VALU"
    elif op.startswith("s_waitcnt"): k = "s_waitcnt"
    elif op.startswith("s_load") or op.startswith("s_buffer"): k = "SMEM"
    elif op.startswith("s_"): k = "SALU"
    elif op.startswith("ds_"): k = "LDS"
    elif op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load"): k = "VMEM_RD"
    elif op.startswith("global_store") or op.startswith("flat_store"): k = "VMEM_WR"
    elif op.startswith("global_atomic"): k = "ATOMIC"
    else: k = "other:" + op
    mix[k] += 1
print(dict(mix))
for l in lines[end - 80:end + 5]:
    if re.search(r"next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed|accum_offset", l):
        print(l.strip())

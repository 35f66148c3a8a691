#!/usr/bin/env python3
"""Rewrite profiles/traffic.json from the PMC summaries of one refresh pass (scripts/pmc.sh / pmc_py.sh
output: per kernel a block of "COUNTER n= N mean= X" lines; the first block is the path's own kernel).
Usage: traffic_from_pmc.py DIR [PREFIX]   (DIR holds <PREFIX>_s2_pmc_fcp_bench.txt, <PREFIX>_ragged_pmc.txt, <PREFIX>_ae_model_e_pmc.txt; PREFIX defaults to r03)
The record carries the sha of fcp_kernels.hip: bench.py reports `roofline.traffic` only while the kernels
are the ones these passes measured."""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIX = sys.argv[2] if len(sys.argv) > 2 else "r03"
FILES = {"s2": (f"{PREFIX}_s2_pmc_fcp_bench.txt", "fcp_dense_kernel<4,4,false>"),
         "ragged": (f"{PREFIX}_ragged_pmc.txt", "fcp_ragged_kernel<4,false>"),
         "ragged_as_delivered": (f"{PREFIX}_ragged_as_delivered_pmc.txt", "fcp_ragged_kernel<4,false>"),
         "e": (f"{PREFIX}_ae_model_e_pmc.txt", "fcp_hybrid_kernel<4,4,false>")}


def kernel_block(path, kernel):
    """Counters of the block headed by `kernel` (a prefix of the header line); the request's other kernels (the
    segment-offset pre-pass of RAGGED with SparseTensor indices) are added: traffic is per request."""
    blocks, cur = {}, None
    for line in open(path):
        m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", line)
        if m and cur is not None:
            blocks[cur].setdefault(m.group(1), (int(m.group(2)), float(m.group(3))))
        elif not m and line.strip():
            cur = line.strip()
            blocks.setdefault(cur, {})
    main = [k for k in blocks if k.startswith(kernel.split("<")[0])]
    if not main:
        raise SystemExit(f"{path}: no block for {kernel}: {list(blocks)}")
    n_main = blocks[main[0]]["FETCH_SIZE"][0]
    vals = {}
    for name, b in blocks.items():
        if not name.startswith("fcp_") or "probe" in name or "FETCH_SIZE" not in b:
            continue
        if name != main[0] and b["FETCH_SIZE"][0] != n_main:
            continue                                  # not launched once per request
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[c] = vals.get(c, 0.0) + b[c][1]
    return vals


def main(d):
    with open(os.path.join(ROOT, "recom_amd", "csrc", "fcp_kernels.hip"), "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    rec = {"_comment": "HBM traffic per launch from rocprofv3 PMC passes (scripts/pmc.sh over the torch-free fcp_bench for S2, "
                       "scripts/pmc_py.sh for the others; separate --pmc runs, counters restricted to fcp_* kernels). FETCH_SIZE / "
                       "WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 tallies 128-byte requests "
                       "at 64 bytes, so the read side is doubled (an upper bound here: requests for 32/64-byte rows are not wide).",
           "kernels_sha16": sha}
    for key, (name, kernel) in FILES.items():
        if not os.path.exists(os.path.join(d, name)):
            continue
        v = kernel_block(os.path.join(d, name), kernel)
        fetch, write = v["FETCH_SIZE"], v["WRITE_SIZE"]
        rec[key] = {"source": f"profiles/{name}", "kernel": kernel, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                    "traffic_bytes": int(round((2 * fetch + write) * 1024))}
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(rec, f, indent=2)
        f.write("\n")
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])

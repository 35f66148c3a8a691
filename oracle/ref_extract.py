#!/usr/bin/env python3
"""Builds oracle/_ref/*.so from the REFERENCE's own source, where it lies.

Almost nothing of the reference's hot path can be compiled in this image: its device code exists only as C++ string
fragments that need TensorFlow 2.6.2, SymEngine, nvcc and CUB.  The fragments that are
self-contained C++ inside their literals are compiled here, unmodified, behind small wrappers of ours:

  libref_bucketize.so (g++, host)         `Bucketize` (tensorflow_addons/graph_optimizers/cuda_emitter.cc:233-247) and the
                                           arena alignment helper `alignmem` of the generated host code (:967-969)
  libref_device.so (hipcc, gfx950 device)  `GatherRowsToGlbMem` (:250-293), `GatherScatterRows` (:296-345), `AlignedVector`
                                           (:664-765), `experiment::ComputeSegmentOffsets` / `SparseSegmentReduce` (:768-962);
                                           run on the GPU by tests/test_gpu_reference_kernels.py
  libref_device_scan.so (hipcc, gfx950)    `SparseSegmentSum` / `SparseSegmentMean` for dim <= 20 (:348-661) — against hipCUB,
                                           the image's port of the CUB interface, NOT CUB 1.8: the templates' own logic runs
                                           as written, the order inside the 64-item scan is rocPRIM's (the oracle restates
                                           both orders behind one selector and equals this library bit for bit in rocPRIM's)

This recipe
  1. reads that file under /root/reference (never copied into the repository),
  2. takes the adjacent string literals of each `headers += "..." "..." ...;` statement and un-escapes them into a
     TEMPORARY directory,
  3. compiles oracle/ref_bucketize_wrap.cc / oracle/ref_device_wrap.hip (ours) against them into oracle/_ref/,
  4. deletes the temporary directory: no reference text is left in the tree, only the built libraries travel to the GPU
     box (oracle/_ref/ is git-ignored).  oracle/_ref/BUILD_STAMP holds a sha256 over literals + wrappers so that a
     second call does nothing.

Used by tests only; a no-op with exit code 0 when /root/reference is absent (the GPU box uses the prebuilt libraries).
CUB 1.8 itself is absent: the order inside the reference's 64-item scan for dim <= 20 stays restated (ORC_SCAN_CUB18)."""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tensorflow_addons/graph_optimizers/cuda_emitter.cc"
OUT = os.path.join(HERE, "_ref")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LIBS = ("libref_bucketize.so", "libref_device.so", "libref_device_scan.so")


def _statement(text: str, first_literal: str, occurrence: int = 0) -> str:
    """The un-escaped text of the `headers += "..." "..." ... ;` statement whose first literal is `first_literal`."""
    start = -1
    for _ in range(occurrence + 1):
        start = text.index(first_literal, start + 1)
    end = text.index('";\n', start) + 1               # the statement ends after a closing quote
    lits = re.findall(r'"((?:[^"\\]|\\.)*)"', text[start:end])
    return "".join(lits).encode().decode("unicode_escape")


def extract_all() -> dict:
    text = open(SRC).read()
    bucketize = _statement(text, '"template <int NUM_BOUNDARIES, typename T>\\n"')
    alignmem = _statement(text, '"static __inline__ int alignmem(int x) {\\n"')
    gather_first = '"template <bool FULL_BLOCK, int EmbedDim, int BLOCK_THREADS, typename "'
    gather = _statement(text, gather_first, 0) + "\n" + _statement(text, gather_first, 1)
    experiment = _statement(text, '"// Ported from TensorFlow 2.6\\n"')
    scan = (_statement(text, '"template <int ScanDim, typename Tparam> struct ScanVecPair {\\n"') + "\n" +
            _statement(text, '"template <int ScanDim, typename Tparam> struct ScanVecCntTuple {\\n"'))
    checks = (("Bucketize(" in bucketize and bucketize.count("{") == bucketize.count("}")),
              ("alignmem(int x)" in alignmem and alignmem.count("{") == 1),
              ("GatherRowsToGlbMem(" in gather and "GatherScatterRows(" in gather and gather.count("{") == gather.count("}")),
              ("class alignas(alignof(T) * N) AlignedVector" in experiment and "namespace experiment" in experiment
               and "ComputeSegmentOffsets" in experiment and experiment.rstrip().endswith("// namespace experiment")),
              ("SparseSegmentSum(" in scan and "SparseSegmentMean(" in scan and scan.count("cub::BlockScan") == 4
               and scan.count("{") == scan.count("}")))
    if not all(checks):
        raise SystemExit(f"ref_extract: the reference literals do not look as expected {checks}")
    return {"bucketize_ref.inc": bucketize, "alignmem_ref.inc": alignmem, "ref_gather.inc": gather, "ref_experiment.inc": experiment,
            "ref_segment_scan.inc": scan}


def _stamp(files: dict) -> str:
    h = hashlib.sha256()
    for name in sorted(files):
        h.update(name.encode() + b"\0" + files[name].encode() + b"\0")
    for w in ("ref_bucketize_wrap.cc", "ref_device_wrap.hip", "ref_device_scan_wrap.hip"):
        h.update(open(os.path.join(HERE, w), "rb").read())
    return h.hexdigest()


def build(force: bool = False) -> bool:
    """True when the libraries are there (built now or earlier)."""
    have = all(os.path.exists(os.path.join(OUT, lib)) for lib in LIBS)
    if not os.path.exists(SRC):
        return have or os.path.exists(os.path.join(OUT, LIBS[0]))
    files = extract_all()
    stamp, stamp_path = _stamp(files), os.path.join(OUT, "BUILD_STAMP")
    if have and not force and os.path.exists(stamp_path) and open(stamp_path).read().strip() == stamp:
        return True
    os.makedirs(OUT, exist_ok=True)
    for stale in os.listdir(OUT):                      # earlier rounds left the un-escaped text here
        if stale.endswith(".inc"):
            os.remove(os.path.join(OUT, stale))
    tmp = tempfile.mkdtemp(prefix="fcp_ref_")
    try:
        for name, body in files.items():
            with open(os.path.join(tmp, name), "w") as f:
                f.write(body)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fno-fast-math", "-I", tmp,
                               os.path.join(HERE, "ref_bucketize_wrap.cc"), "-o", os.path.join(OUT, LIBS[0])])
        if os.path.exists(HIPCC):
            # -ffp-contract=off: the reference's sums contain no multiply, but nothing may be fused either way
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-Wno-unused-value", "-I", tmp, os.path.join(HERE, "ref_device_wrap.hip"), "-o",
                                   os.path.join(OUT, LIBS[1])])
            # the dim <= 20 templates against hipCUB (the image's port of the CUB interface; CUB 1.8 itself is absent: the
            # scan's fp32 association is rocPRIM's, see the header of ref_device_scan_wrap.hip)
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-I", tmp, os.path.join(HERE, "ref_device_scan_wrap.hip"), "-o", os.path.join(OUT, LIBS[2])])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(stamp_path, "w") as f:
        f.write(stamp + "\n")
    return True


def device_lib_path(which: int = 1):
    """oracle/_ref/libref_device.so (which = 1) or libref_device_scan.so (2), or None when it has not been built (no
    /root/reference and nothing prebuilt)."""
    build()
    p = os.path.join(OUT, LIBS[which])
    return p if os.path.exists(p) else None


if __name__ == "__main__":
    ok = build("--force" in sys.argv)
    print("oracle/_ref: " + ("built from " + SRC if os.path.exists(SRC) else ("present" if ok else "absent (no reference here)")))

#!/usr/bin/env python3
"""Builds oracle/_ref/libref_bucketize.so from the REFERENCE's own source, where it lies.

Almost nothing of the reference's hot path can be compiled in this image: its device code exists only as C++ string
fragments that need TensorFlow 2.6.2, SymEngine, nvcc and CUB.  Two functions are plain C++ inside
their literals: `Bucketize` (tensorflow_addons/graph_optimizers/cuda_emitter.cc:233-247) and the arena alignment helper
`alignmem` of the generated host code (:967-969, used by :2151-2179).  This recipe

  1. reads that file under /root/reference (never copied into the repository),
  2. takes the adjacent string literals of the `Bucketize` template and of `alignmem` and un-escapes them into
     oracle/_ref/bucketize_ref.inc / alignmem_ref.inc (generated files: oracle/_ref/ is git-ignored, it only travels to
     the GPU box next to the built library),
  3. compiles oracle/ref_bucketize_wrap.cc (ours: it defines the two CUDA qualifiers away and instantiates the
     template for every boundary count the tests use) with g++ into oracle/_ref/libref_bucketize.so.

Used by tests only (the oracle's a5 restatement and the HIP kernels' three Bucketize tiers are compared with it); a
no-op with exit code 0 when /root/reference is absent (the GPU box uses the prebuilt library)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tensorflow_addons/graph_optimizers/cuda_emitter.cc"
OUT = os.path.join(HERE, "_ref")


def extract() -> str:
    text = open(SRC).read()
    start = text.index('"template <int NUM_BOUNDARIES, typename T>\\n"')
    end = text.index('";\n', start) + 1               # the statement `headers += "..." "..." ... ;` ends after a closing quote
    lits = re.findall(r'"((?:[^"\\]|\\.)*)"', text[start:end])
    body = "".join(lits).encode().decode("unicode_escape")
    if "Bucketize(" not in body or body.count("{") != body.count("}"):
        raise SystemExit("ref_extract: the Bucketize literal does not look as expected")
    return body


def extract_alignmem() -> str:
    text = open(SRC).read()
    start = text.index('"static __inline__ int alignmem(int x) {\\n"')
    end = text.index('";\n', start) + 1
    body = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', text[start:end])).encode().decode("unicode_escape")
    if "alignmem(int x)" not in body or body.count("{") != 1 or body.count("}") != 1:
        raise SystemExit("ref_extract: the alignmem literal does not look as expected")
    return body


def build(force: bool = False) -> bool:
    if not os.path.exists(SRC):
        return os.path.exists(os.path.join(OUT, "libref_bucketize.so"))
    os.makedirs(OUT, exist_ok=True)
    inc, lib = os.path.join(OUT, "bucketize_ref.inc"), os.path.join(OUT, "libref_bucketize.so")
    for path, body in ((inc, extract()), (os.path.join(OUT, "alignmem_ref.inc"), extract_alignmem())):
        if force or not os.path.exists(path) or open(path).read() != body:
            open(path, "w").write(body)
            force = True
    wrap = os.path.join(HERE, "ref_bucketize_wrap.cc")
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(wrap):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fno-fast-math", "-I", OUT, wrap, "-o", lib])
    return True


if __name__ == "__main__":
    ok = build("--force" in sys.argv)
    print("oracle/_ref/libref_bucketize.so " + ("built from " + SRC if os.path.exists(SRC) else ("present" if ok else "absent (no reference here)")))

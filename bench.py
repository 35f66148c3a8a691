#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s2|dlrm|ragged|shard|shard-col|e|f]
                  [--seg indices|csr|rowids32]   (ragged: how row membership arrives; default SparseTensor indices)
                  [--max-len N] [--ids uniform|zipf]   (ragged: bag lengths U{0..N}; id distribution)
                  [--staged]         (ragged: headline = the staged form; default: as delivered, the staged form beside it)
                  [--as-delivered]   (e / f: int64 ids + SparseTensor indices resident on the device, pre-pass there;
                                      default: as the staged Addons>ConcatInputs leaves the request, host cost stated)

A *step* is one request: one pass of the fused feature-column path (ids resident
in HBM -> [batch, sum(dim)] concat output resident in HBM) over one batch of
synthetic input.  Default workload = BASELINE.json configs[1] "S2": 1000 columns,
dims 8/16/32/64, vocab 1M (120 GB of tables), batch 512, on 1 MI355X.

N > 1 (one rank per GPU: launched by torch.distributed.run, or by bench.py itself when no launcher did): the path shards by
requests — every rank serves its own requests on its own replica of the tables,
no data-path collective ("weak" scaling).  Only `--workload shard` / `shard-col`
(tables larger than one GPU's HBM) shard the tables — by rows (partial sums) or by
columns (final column blocks) — and exchange with one RCCL all-to-all.

Whether tables are replicated or sharded is decided by the placement gate
(recom_amd/placement.py) from the workload's table bytes and the GPU's HBM.

The timed loop is native (recom_amd/csrc/fcp_harness.hip); rank 0 prints ONE JSON
line: `value` = exactly --steps requests from one serve worker; `overlapped_serving` =
the reference's multi-worker protocol (best of 2 / 3 / 4 workers, >= 400 requests each).  `roofline.achieved` = algorithmic bytes per request (SURVEY.md §8d formula,
DESIGN.md §5) / average device time per request from HIP events recorded on the
launch stream around the timed region.  `cpu_baseline` times the CPU oracle
(oracle/, a port of the reference's TF-CPU semantics; TensorFlow is absent) on
this box's host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Serve workers issue on their own streams; the HIP runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4), and two workers sharing a queue do not overlap.  A serving process gives every worker its own queue
# (measured, overlapped pass, 2 / 3 / 4 workers: 23.3 / 25.6 / 24.0 us with 4 queues, 23.4 / 22.5-23.8 / 22.7-23.3 us with 8;
# the single-stream figure does not move).  Must be set before the runtime initialises; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREWARM_S = 0.25       # untimed pre-warm in front of the timed region, seconds of the same requests (whatever --warmup is)
PREWARM_CHUNK = 256    # ... issued in calls of this many requests
REPEATS = 5            # the K timed steps are repeated this many times (9 times when K < 200); the median repeat is the one reported
METRIC = "inference QPS + p50 latency, 1000-col synth model, batch 512, 1\u00d7MI355X"  # == BASELINE.json "metric"
try:
    with open(os.path.join(ROOT, "BASELINE.json")) as _f:
        METRIC = json.load(_f).get("metric", METRIC)
except (OSError, ValueError):
    pass


def kernels_sha16():
    import hashlib
    with open(os.path.join(ROOT, "recom_amd", "csrc", "fcp_kernels.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def measured_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json:
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md).
    PMC counters cannot be collected from inside this process, so the figure is a RECORD of a separate
    pass (scripts/pmc.sh); it is reported only while the kernels are the ones that pass measured
    (traffic.json carries the sha of fcp_kernels.hip), otherwise null.  Returns (bytes or None, source)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f)
        entry = rec[workload]
    except Exception:
        return None, "no PMC record for this workload in profiles/traffic.json"
    sha = rec.get("kernels_sha16")
    if sha is not None and sha != kernels_sha16():
        return None, f"profiles/traffic.json was collected on kernels {sha}, this build is {kernels_sha16()}: stale, not reported"
    return entry["traffic_bytes"], "profiles/traffic.json: separate rocprofv3 --pmc passes (scripts/pmc.sh), per launch"


def live_traffic(arena_ring: int):
    """HBM bytes per launch of the S2 kernel measured NOW, on this box: two separate `rocprofv3 --pmc` passes (FETCH_SIZE,
    WRITE_SIZE; counters restricted to fcp_* kernels, `--kernel-trace` the only trace domain) over the torch-free native
    binary of the same workload (`recom_amd/fcp_bench`, the program itself right after `--`), as MI355X_MICROARCH.md's HBM /
    rocprofv3 section prescribes: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — FETCH_SIZE tallies gfx950's 128-byte requests at 64.
    Returns (bytes, source) or (None, why): the caller then falls back to the committed record (profiles/traffic.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(os.environ.get("FCP_LIB_DIR", os.path.join(ROOT, "recom_amd")), "fcp_bench")
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or not os.path.exists(prof):
        return None, "rocprofv3 or fcp_bench not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="fcp_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--kernel-include-regex", "fcp_", "--output-format", "csv", "-d", out, "--",
                   exe, "--steps", "40", "--warmup", "10", "--verify", "0", "--ring", str(arena_ring)]
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"})
            got = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "fcp_dense_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None, f"rocprofv3 --pmc {counter}: no fcp_dense_kernel rows (rc {res.returncode}): {res.stderr[-200:]}"
            vals[counter] = sum(got) / len(got)
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return int(round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)), (
        f"measured in this run: two rocprofv3 --pmc passes (FETCH_SIZE {vals['FETCH_SIZE']:.1f} KiB, WRITE_SIZE {vals['WRITE_SIZE']:.1f} KiB per "
        f"launch) over recom_amd/fcp_bench --ring {arena_ring}, (2 x FETCH_SIZE + WRITE_SIZE) x 1024")


def pcie_inclusive():
    """The same workload with the request on the HOST when the clock starts: host id tensors -> fcp_stager (pinned ring,
    pack threads, int64 ids shipped as int32) -> one H2D -> kernel, from the torch-free native binary in its own process
    (never part of `value`: the bench contract times inputs resident in HBM).  Pipelined rate and the
    latency of a lone request, host clock."""
    import subprocess
    exe = os.path.join(os.environ.get("FCP_LIB_DIR", os.path.join(ROOT, "recom_amd")), "fcp_bench")
    out = {"what": "host int64 id tensors -> fcp_stager_stage_narrow (pinned ring, ids packed as int32) -> fused kernel, host clock, "
                   "`recom_amd/fcp_bench --h2d 1 --narrow 1 [--copy-kernel 0 | --zero-copy 1]`; h2d_copy: the stager's default since "
                   "round 5 - a copy KERNEL on the stager's stream reads the pinned ring (no SDMA engine: hipMemcpyAsync's submission "
                   "stalls for 6-14 ms a few times per thousand calls, profiles/r05_pcie_staging_stalls.txt), a lone request packed and "
                   "shipped in 4 groups; h2d_copy_sdma: hipMemcpyAsync as in rounds 2-4; zero_copy: the fused kernel reads the pinned ring "
                   "over PCIe itself (FCP_STAGER_ZERO_COPY); copy_calls_over_1ms / max_copy_call_us: stalls of the copy CALL on the host"}
    for key, extra in (("h2d_copy", []), ("h2d_copy_sdma", ["--copy-kernel", "0"]), ("zero_copy", ["--zero-copy", "1"])):
        cmd = [exe, "--h2d", "1", "--narrow", "1", "--steps", "300", "--warmup", "50", "--verify", "0", "--pack-threads", "16"] + extra
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("{") and "pcie_inclusive" in ln][-1]
            r = json.loads(line)
            out[key] = {k: r[k] for k in ("pack_threads", "blob_MB", "us_per_request_pipelined", "us_latency_single", "inferences_per_s",
                                          "host_us_stage_call", "host_us_process_call", "h2d_copy_alone_us", "h2d_GBs", "copy_calls",
                                          "copy_calls_over_1ms", "max_copy_call_us", "zero_copy_fallback_switches") if k in r}
        except Exception as e:  # the bench line must not depend on this extra
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def host_staging_cost(raw_model, n_requests: int = 8):
    """What the staged request form costs the HOST, stated beside the device figure: Addons>ConcatInputs packing the
    graph's tensors with the plan's stage section (fcp_concat_inputs_ex: int64 ids -> int32, sorted row ids /
    SparseTensor indices -> row offsets; one thread, like the TF op) next to the reference's byte-for-byte pack of the same
    tensors (fcp_concat_inputs = concat_inputs_ops.cc:42-77).  The C calls alone are timed (tensor views prepared
    beforehand).  Host time only; never part of `value`."""
    import ctypes as C
    from recom_amd import lib as _lib
    from recom_amd.ops import _host_tensors
    L = _lib.load()
    spec, stage = raw_model.spec.staged_for_concat_inputs()
    reqs = [raw_model.make_request(777 + i) for i in range(n_requests)]
    raws = [list(r.inputs) + ([r.symbols] if stage.symbols_input >= 0 else []) for r in reqs]
    n = len(raws[0])
    modes = np.asarray(stage.modes, np.uint8)
    prepared = []
    for x in raws:
        arrs, keep, tens = _host_tensors(x)
        args = np.asarray(stage.mode_args(arrs), np.int64)
        prepared.append((arrs, keep, tens, args))
    cap = sum(a.nbytes for a in raws[0]) * 2 + 4096
    blob, offsets, shapes = np.empty(cap, np.int8), np.empty(n, np.int32), np.empty(4 * n, np.int32)

    def timed(call):
        best = float("inf")
        for _ in range(4):
            t0 = time.perf_counter()
            for pr in prepared:
                _lib.check(call(pr), "ConcatInputs")
            best = min(best, (time.perf_counter() - t0) / len(prepared))
        return best * 1e6
    plain_us = timed(lambda pr: L.fcp_concat_inputs(pr[2], n, blob.ctypes.data, blob.nbytes, offsets.ctypes.data, shapes.ctypes.data))
    staged_us = timed(lambda pr: L.fcp_concat_inputs_ex(pr[2], n, modes.ctypes.data, pr[3].ctypes.data, blob.ctypes.data, blob.nbytes,
                                                        offsets.ctypes.data, shapes.ctypes.data))
    pooled = {}
    if hasattr(L, "fcp_pack_pool_create"):                           # the same pack on a worker pool (FCP_CONCAT_INPUTS_THREADS in the shim)
        for nt in (4, 8, 16):
            pool = C.c_void_p()
            _lib.check(L.fcp_pack_pool_create(nt, C.byref(pool)), "fcp_pack_pool_create")
            pooled[str(nt)] = timed(lambda pr: L.fcp_concat_inputs_ex_pool(pool, pr[2], n, modes.ctypes.data, pr[3].ctypes.data,
                                                                           blob.ctypes.data, blob.nbytes, offsets.ctypes.data,
                                                                           shapes.ctypes.data))
            L.fcp_pack_pool_destroy(pool)
    nb = C.c_int64(0)
    _lib.check(L.fcp_concat_inputs_ex_sizes(prepared[0][2], n, modes.ctypes.data, prepared[0][3].ctypes.data, C.byref(nb), None), "sizes")
    return {"what": "host cost of the request form timed above: Addons>ConcatInputs with the plan file's stage section "
                    "(fcp_concat_inputs_ex, one thread, like the TF op; fcp_concat_inputs_ex_pool on 4 / 8 / 16 threads) vs the "
                    "reference's byte copy of the same tensors (fcp_concat_inputs = concat_inputs_ops.cc:42-77); the C calls "
                    "alone; never part of `value`",
            "concat_inputs_staged_us": staged_us, "concat_inputs_staged_pool_us": pooled, "concat_inputs_byte_copy_us": plain_us,
            "blob_bytes_staged": int(nb.value), "blob_bytes_byte_copy": int(sum(a.nbytes for a in raws[0])),
            "inputs": n}


def side_error(e: BaseException) -> str:
    """What a failed SIDE record says in its place (the line's contract fields never depend on a side record)."""
    return f"{type(e).__name__}: {e}"[:400]


def cpu_baseline(model, budget_s: float = 20.0, sample_columns: int = 0):
    """The CPU oracle (oracle/: a C port of the TF-CPU semantics of the reference's path; TensorFlow is absent) on this
    box's host cores: the WHOLE workload when its tables fit host memory (S2: 120 GB; `sampled: false`), else the first k
    columns at full batch, scaled to the whole model (`sampled: true`).  ONE sweep decides everything that is reported:
    every worker count serves for the same duration (orc_serve_for: independent single-threaded workers, the reference
    harness' serve_workers on TF-CPU).  `value` is the 32-worker entry — the reference's TF-CPU budget is 32 cores
    (AE/build_and_run.py:57) — with the best of the sweep and the all-core entry beside it; they are entries of the same
    sweep and cannot disagree.  The intra-request mode (one request at a time, OpenMP over its columns) is measured beside
    it and reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fcp_oracle
    from recom_amd.ops import concat_inputs
    from recom_amd.plan import PlanSpec

    cores = len(os.sched_getaffinity(0))
    # The CPUs this process can actually USE: the affinity mask says 256 on the boxes of this pool, the cgroup's CPU quota
    # (cpu.max: "1600000 100000") says 16 — beyond the quota every runnable thread is throttled, which is why every sweep of
    # rounds 2-5 peaked at 16 workers and fell to a third at 256 (so does a plain OpenMP triad: 290 GB/s at 32 threads, 42 at
    # 256; profiles/r06_cpu_baseline_collapse.txt).  Worker counts beyond the quota are reported, never headlined.
    quota = host_cpu_quota()
    usable = max(1, min(cores, int(quota + 0.999))) if quota else cores
    spec = model.spec
    # host RAM bounds the sample: the tables (S2: 120 GB) + 8 GB of requests / outputs within 80 % of what is available
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 64 << 30
    if os.environ.get("FCP_BENCH_CPU_RAM_BYTES"):    # testing aid: pretend the host is this small
        avail = int(os.environ["FCP_BENCH_CPU_RAM_BYTES"])
    k = min(sample_columns or spec.n_columns, spec.n_columns)
    per_col = [t.vocab * t.dim * 4 for t in model.tables]

    def table_bytes(k_):
        return sum(per_col[:1 + max([c.table_input for c in spec.columns[:k_] if c.table_input >= 0] or [-1])])
    while k > 8 and table_bytes(k) + (8 << 30) > 0.8 * avail:
        k //= 2
    cols = spec.columns[:k]
    n_host = 1 + max(max(c.ids_input, c.seg_input) for c in cols)
    n_tab = 1 + max(c.table_input for c in cols)
    sub = spec if k == spec.n_columns else PlanSpec(cols, spec.host_input_ranks[:n_host], spec.host_input_elem_sizes[:n_host], n_tab,
                                                    n_groups=1, n_symbols=spec.n_symbols)
    # 64 distinct requests, rotated, so that the touched rows are not cache-resident
    reqs = [model.make_request(12345 + i) for i in range(64)]
    packed = [concat_inputs(r.inputs[:n_host]) for r in reqs]
    req = reqs[0]
    blob, offsets, shapes = packed[0]
    # table VALUES do not affect CPU time; every page is written (an untouched page would read from the shared zero page)
    # and it is written from all cores at once, so that on a two-socket host a table's pages are spread over both sockets'
    # memory instead of landing next to this thread
    orc = fcp_oracle.COracle()
    t_fill = time.perf_counter()
    tables = []
    for t in model.tables[:n_tab]:
        a = np.empty((t.vocab, t.dim), np.float32)
        orc.fill_parallel(a, 0.5)
        tables.append(a)
    t_fill = time.perf_counter() - t_fill
    plan = sub.to_dict()
    rows = sub.group_rows(0, shapes, req.symbols)
    scale = spec.n_columns / k
    # serving sweep: the same duration for every worker count.  `value` stays what rounds 2-5 reported: every column written
    # straight into the concat matrix (the checker's own, FUSED form: the CPU's best foot forward, ~8x faster per request at
    # 1000 columns than TensorFlow-CPU's dataflow for the unrewritten graph — one [rows, dim] tensor per column op, then
    # ConcatV2 copying 512 x 1000 row pieces — which is swept beside it at three worker counts:
    # orc_process_feature_columns_unfused).  Both get SLOWER beyond 16-32 workers on the boxes of this pool, and so does a plain
    # OpenMP triad (290 GB/s at 32 threads, 42 GB/s at 256: profiles/r06_cpu_baseline_collapse.txt): the host, not the port.
    serve_cands = sorted({t for t in (1, 8, 16, 32, 64, 128, cores) if t <= max(cores, 1)})
    per = max(0.5, 0.6 * budget_s / len(serve_cands))
    sweep, detail = {}, {}
    for t in serve_cands:
        done, sec = orc.serve_for(plan, packed, tables, req.symbols, t, per, 0)
        sweep[t] = rows * done / sec / scale
        detail[t] = {"requests": done, "seconds": sec}
    tf_cands = sorted({t for t in (16, 32, cores) if t <= max(cores, 1)})
    tf_flow = {}
    for t in tf_cands:
        done, sec = orc.serve_for(plan, packed, tables, req.symbols, t, max(0.5, 0.2 * budget_s / len(tf_cands)), 1)
        tf_flow[t] = rows * done / sec / scale
    best_t = max(sweep, key=sweep.get)
    # the reference's TF-CPU budget is 32 cores (AE/build_and_run.py:57): that entry where the box grants 32 CPUs, else the
    # entry at the CPUs it does grant
    head_t = max([t for t in sweep if t <= min(32, usable)] or [min(sweep)])
    # intra-request mode, the same duration in total
    out = [np.zeros((sub.group_rows(g, shapes, req.symbols), sub.group_width(g)), np.float32) for g in range(sub.n_groups)]
    intra = {}
    intra_cands = sorted({t for t in (8, 32, cores) if t <= max(cores, 1)})
    for t in intra_cands:
        orc.process_feature_columns(plan, blob, offsets, shapes, tables, req.symbols, t, out)  # warm
        n, t0 = 0, time.perf_counter()
        while True:
            orc.process_feature_columns(plan, *packed[n % len(packed)], tables, req.symbols, t, out)
            n += 1
            el = time.perf_counter() - t0
            if el > 0.2 * budget_s / len(intra_cands):
                break
        intra[t] = rows * n / el / scale
    return {
        "value": sweep[head_t], "unit": "inferences/s", "cores": head_t, "kind": "port", "sampled": k != spec.n_columns,
        "best_of_sweep": {"cores": best_t, "inferences_per_s": sweep[best_t]},
        "cores_32_inferences_per_s": sweep.get(32), "all_cores": cores, "all_cores_inferences_per_s": sweep.get(cores),
        "cpu_quota_cores": quota, "usable_cores": usable,
        "serve_workers_sweep": {str(t): v for t, v in sorted(sweep.items())},
        "dataflow": "fused: every column written straight into the concat matrix (orc_process_feature_columns, the checker's form)",
        "tf_cpu_dataflow_serve_workers_sweep": {str(t): v for t, v in sorted(tf_flow.items())},
        "tf_cpu_dataflow": "the same values through TensorFlow-CPU's dataflow for the unrewritten graph: one [rows, dim] tensor per "
                           "column op, then ConcatV2 row by row (orc_process_feature_columns_unfused)",
        "intra_request_openmp_inferences_per_s": {str(t): v for t, v in sorted(intra.items())},
        "host_table_fill_s": t_fill,
        "sample": (f"all {spec.n_columns} columns" if k == spec.n_columns else f"first {k} of {spec.n_columns} columns, scaled x{scale:.1f} to the whole model,")
                  + f" at batch {rows} ({sum(a.nbytes for a in tables) / 1e9:.1f} GB of host tables), "
                  f"{len(packed)} distinct requests rotated; one sweep of independent "
                  f"single-threaded workers ({serve_cands}), {per:.1f} s each (requests completed: "
                  f"{ {t: d['requests'] for t, d in detail.items()} }); value = the {head_t}-worker entry of that sweep (the reference's "
                  f"TF-CPU budget is 32 cores, AE/build_and_run.py:57; this box grants {usable}: {cores} CPUs in the affinity mask, "
                  f"cgroup CPU quota {quota if quota else 'none'}), best_of_sweep beside it; entries beyond the quota are throttled; "
                  f"tables first-touched from all cores (pages spread over the sockets); C port of TF-CPU semantics (TensorFlow absent)",
    }


def host_cpu_quota():
    """CPUs the container may use at once according to its cgroup (v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`),
    or None when there is no quota (or no cgroup file system to ask)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def access_mix_floor(model, h, bytes_alg, measured_us):
    """What the memory system itself needs for this request mix, from probes run now on this GPU: random
    row gathers at each row size (the memory system fetches 128 bytes whatever the row size; tables of
    <= 1 MiB, or all tables of a model that fits the Infinity Cache, are taken as cache-resident), the output at the best store rate of the kernel's store
    shape, ids / offsets as a sequential read; reads and writes summed (they hardly overlap on HBM).
    `frac` = floor / measured time — close to 1 means memory-system bound, small means latency bound."""
    from recom_amd.harness import gather_probe, read_probe, write_probe
    rates = {rb: gather_probe(rb) for rb in (32, 64, 128, 256)}
    per_class = {rb: 0.0 for rb in rates}
    packed, reqs = h.packed, h.requests
    all_tables = model.table_bytes()
    for (blob, offsets, shapes), r in zip(packed, reqs):
        so = model.spec.shape_offsets()
        for c in model.spec.columns:
            if c.form not in (1, 2, 3):
                continue
            nnz = 1
            for j in range(model.spec.host_input_ranks[c.ids_input]):
                nnz *= int(shapes[so[c.ids_input] + j])
            rb = c.dim * 4
            if all_tables <= (128 << 20) or c.vocab * rb <= (1 << 20):
                continue  # cache-resident (whole model within the Infinity Cache, or an L2-sized hot table)
            cls = 32 if rb <= 32 else 64 if rb <= 64 else 128 if rb <= 128 else 256
            per_class[cls] += nnz * rb / len(packed)
    w_rate, r_rate = write_probe(), read_probe()
    gather_us = sum(per_class[rb] / rates[rb] for rb in rates) * 1e6
    other_read = bytes_alg["ids"] + bytes_alg["segments"] + bytes_alg["boundaries"]
    floor_us = gather_us + bytes_alg["out"] / w_rate * 1e6 + max(other_read, 0.0) / r_rate * 1e6
    return {"gather_GBs_by_row_bytes": {str(k): v / 1e9 for k, v in rates.items()}, "store_GBs": w_rate / 1e9,
            "sequential_read_GBs": r_rate / 1e9, "gather_us": gather_us, "floor_us": floor_us,
            "frac": floor_us / measured_us,
            "note": "floor = row gathers at the probed rate of their row size + output at the probed store rate + "
                    "ids/offsets at the sequential read rate; frac = floor / measured device time per request"}


def launch_ranks(n: int, argv) -> int:
    """One process per GPU through torch.distributed.run (what the driver does for N > 1), rendezvous on 127.0.0.1 and
    a free port.  FCP_BENCH_DRY_LAUNCH=1 prints the command instead of running it (CPU test of the launcher)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if os.environ.get("FCP_BENCH_DRY_LAUNCH"):
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="s2", choices=["s2", "dlrm", "ragged", "shard", "shard-row", "shard-col", "e", "f"],
                    help="shard = BASELINE configs[4] (4000 columns, 480 GB) placed by the gate's mixed preference: whole tables "
                         "wherever a table fits one GPU, rows only for those that do not (configs[4]: every table fits -> "
                         "column blocks, 8x fewer bytes on the wire); shard-row / shard-col force one kind")
    ap.add_argument("--seg", default="indices", choices=["indices", "csr", "rowids32"],
                    help="ragged: how row membership arrives - SparseTensor indices [nnz, 2] int64 (what BASELINE configs[3] "
                         "names and TF graphs deliver; default), CSR offsets or int32 row ids")
    ap.add_argument("--ids", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--max-len", type=int, default=0,
                    help="ragged: ids per row drawn from U{0..max-len} (default 10); hundreds = multi-hot history features")
    ap.add_argument("--threads", type=int, default=1, help="serve_workers per GPU (reference harness flag)")
    ap.add_argument("--columns", type=int, default=0, help="override the column count (debug only)")
    ap.add_argument("--staged", action="store_true",
                    help="requests resident in HBM in the form the staging step (Addons>ConcatInputs with the plan's stage section / "
                         "fcp_stager_stage_ex, PlanSpec.staged()) leaves them: ids int32, sorted row ids / SparseTensor indices "
                         "already CSR offsets (no pre-pass, no search).  The default for --workload ragged: the reference delivers "
                         "SparseTensor indices on the HOST (ConcatInputs is a CPU op), so the conversion happens while packing")
    ap.add_argument("--as-delivered", action="store_true",
                    help="ragged: time the request as the graph's tensors are (int64 ids, SparseTensor indices resident on the "
                         "device, segment-offset pre-pass on the device) instead of as the staged ConcatInputs leaves it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the closed-form check of the resident requests before the warm-up")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive pass (host tensors -> pinned ring -> H2D -> kernel)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="skip the extra overlapped-serving pass (3 streams): keeps a kernel trace of this run single-stream")
    ap.add_argument("--vocab", type=int, default=0, help="override the vocabulary size (debug only)")
    ap.add_argument("--batch", type=int, default=0, help="override the batch size (debug only)")
    ap.add_argument("--requests", type=int, default=0,
                    help="distinct resident requests cycled through (default: 16; 64 for the dynamic-shape workloads ragged / e / f, "
                         "more than the plan's 32 descriptor slots: every request brings shapes that are NOT resident, as real "
                         "dynamic-shape traffic does)")
    ap.add_argument("--arena-ring", type=int, default=1,
                    help="output arenas the timed loop rotates through.  1 (default) = the reference's allocation pattern: the op takes "
                         "its arena from allocate_output(2) once per Compute (feature_column_process_op_gpu.cu.cc:107-111) and a one-thread "
                         "serving loop (benchmark_multi_thread, recom_examples.patch:193-216) gets back the block the previous request "
                         "freed; 6 = what rounds 1-5 timed (every output line evicted between two writes).  `arena_reuse` reports 1 / 2 / 6 "
                         "side by side whatever this is")
    ap.add_argument("--sla-ms", type=float, default=0.0,
                    help="instead of the latency protocol: the reference's THROUGHPUT benchmark (benchmark_throughput, "
                         "recom_examples.patch:264-465) for this path - grow the batch until the average request latency reaches this "
                         "many milliseconds (--threads = its serve_workers, --steps = its num_iterations); s2 / dlrm / ragged; one JSON line")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) normally; gloo only to exercise the N>1 control flow on a 1-GPU box")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, torch.distributed.run
    # on 127.0.0.1) and leave with their exit code.  Nothing has touched the GPU yet — torch is not even imported — so
    # this process never initialises HIP; rank 0 of the children prints the JSON line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # dmabuf IPC: RCCL across processes needs it on this driver (the pool exports it; a launcher's environment may not)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from recom_amd import synth
    from recom_amd.harness import ServingHarness, copy_probe

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FCP_BENCH_DEVICE"):  # testing aid: several ranks on one GPU (with --dist-backend gloo)
        local_rank = int(os.environ["FCP_BENCH_DEVICE"])
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    if args.sla_ms > 0:
        # the reference's other harness: the pressure test (recom_amd.harness.sla_throughput_search); N = 1 only
        from recom_amd.harness import sla_throughput_search
        if world != 1 or args.workload not in ("s2", "dlrm", "ragged"):
            print("bench.py --sla-ms: one GPU; --workload s2 | dlrm | ragged", file=sys.stderr)
            sys.exit(2)
        make = {"s2": lambda b: synth.model_s2(columns=args.columns or 1000, batch=b, **({'vocab': args.vocab} if args.vocab else {})),
                "dlrm": lambda b: synth.model_dlrm(batch=b),
                "ragged": lambda b: synth.model_ragged(columns=args.columns or 512, seg=args.seg, batch=b,
                                                       **({'vocab': args.vocab} if args.vocab else {}))}[args.workload]
        free_b, _ = torch.cuda.mem_get_info(local_rank)
        probe = make(16)
        res = sla_throughput_search(make, args.sla_ms, serve_workers=args.threads, num_iterations=min(args.steps, 100), device=local_rank,
                                    arena_budget_bytes=int(0.8 * max(free_b - probe.table_bytes(), 1 << 30)),
                                    log=lambda m: print(m, file=sys.stderr))
        print(json.dumps({"metric": f"max throughput under a {args.sla_ms} ms SLA of the embedding stage (reference protocol: benchmark_throughput, "
                                    f"recom_examples.patch:264-465), {probe.name}", "value": res["max_throughput"], "unit": "inferences/s",
                          "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic", "vs_baseline": None,
                          "config": {"workload": f"{probe.name}: {probe.description} (batch grown by the search)", "serve_workers": args.threads}, **res}))
        return
    if args.workload in ("shard", "shard-row", "shard-col"):
        # BASELINE.json config 5: 4000 S2-shaped columns (480 GB); the flag names the model and the preferred sharding
        model = synth.model_shard(columns=args.columns or 4000, **({'vocab': args.vocab} if args.vocab else {}),
                                  **({'batch': args.batch} if args.batch else {}))
    elif args.workload == "s2":
        model = synth.model_s2(columns=args.columns or 1000, dist=args.ids, **({'batch': args.batch} if args.batch else {}),
                               **({'vocab': args.vocab} if args.vocab else {}))
    elif args.workload == "dlrm":
        model = synth.model_dlrm()
    elif args.workload in ("e", "f"):
        model = synth.model_ae(args.workload, **({'batch': args.batch} if args.batch else {}))
    else:
        model = synth.model_ragged(columns=args.columns or 512, seg=args.seg, dist=args.ids,
                                   **({'batch': args.batch} if args.batch else {}),
                                   **({'vocab': args.vocab} if args.vocab else {}),
                                   **({'max_len': args.max_len} if args.max_len else {}))

    if not args.requests:
        args.requests = 64 if args.workload in ("ragged", "e", "f") else 16
    raw_model = model
    # Workloads with SparseTensor features (RAGGED; the reference's models E / F) are timed in the form the rewritten
    # graph's Addons>ConcatInputs leaves in HBM (plan-file stage section: ids int32, row offsets) unless --as-delivered
    # RAGGED (BASELINE configs[3]) is timed AS THE CONFIG NAMES IT — SparseTensor indices resident on the device, int64 ids, the
    # segment-offset pre-pass in the timed region — and carries the staged form (what Addons>ConcatInputs leaves in HBM when it
    # converts while it packs) beside it as `staged`, with the host cost of that conversion; `--staged` swaps the two.
    # The reference's models E / F keep the staged form as their headline (their ids are hashed strings: the host is in the
    # path either way).
    if args.workload in ("e", "f") and not args.as_delivered:
        args.staged = True
    if args.staged:
        model = synth.staged_model(model)

    # The placement gate (a13): tables that fit this GPU's HBM are served from replicas — every rank its own
    # requests, no collective; only beyond that are they sharded over the node's GPUs with one RCCL exchange per
    # request.  The branch below is the gate's decision, not the flag's.  Tables that fit no placement on
    # `world` GPUs are refused (FcpError names the smallest world that would do).
    from recom_amd.lib import FcpError
    from recom_amd.placement import REPLICATE, decide_placement, device_hbm_bytes
    try:
        # FCP_BENCH_HBM_BYTES: testing aid — pretend the GPU is this small, so that a toy model takes the sharded branch
        hbm_override = os.environ.get("FCP_BENCH_HBM_BYTES")
        placement = decide_placement(model.spec, world, hbm_bytes=int(hbm_override) if hbm_override else device_hbm_bytes(local_rank),
                                     prefer={"shard-col": "column", "shard-row": "row"}.get(args.workload, "mixed" if args.workload == "shard" else "row"),
                                     **({"reserve_bytes": 0} if hbm_override else {}))
    except FcpError as e:                            # the tables fit no placement on this many GPUs
        if rank == 0:
            print(f"bench.py: {model.name} ({model.table_bytes() / 1e9:.0f} GB of tables) cannot be placed on {world} GPU(s): {e}",
                  file=sys.stderr)
        if dist:
            dist.destroy_process_group()
        sys.exit(2)
    if placement.mode != REPLICATE:
        from recom_amd.shard import bench_sharded
        rec = bench_sharded(args, model, placement, rank, world, local_rank, dist)
        if rank == 0:
            print(json.dumps(rec))
        if dist:
            dist.destroy_process_group()
        return

    h = ServingHarness(model, device=local_rank, n_requests=args.requests, arena_ring=args.arena_ring, n_threads=args.threads,
                       seed0=1000 * rank)
    bytes_alg = h.algorithmic_bytes()
    # a wrong kernel is not timed: every resident request is served once and compared with the closed-form tables (no
    # oracle: recom_amd.harness.verify_resident); this also reads every request blob once before the warm-up
    verified = h.verify_resident() if not args.no_verify else {"checked": 0, "note": "--no-verify"}

    h.run(max(args.warmup, 1))                       # W untimed warm-up steps
    # ... and, whatever --warmup says, an untimed pre-warm of >= PREWARM_S of the same requests: the reference's protocol
    # discards >= 10 iterations (recom_examples.patch:110, 186-207) so that clocks, queues and TLBs are those of a serving
    # process; a 5-request warm-up in front of a 0.6-ms region measured the clock ramp (r05: 31.2 us by the driver's
    # --steps 20 against 28.6 over 2000 steps)
    extra_warmup, t_end = 0, time.perf_counter() + PREWARM_S
    while time.perf_counter() < t_end:
        h.run(PREWARM_CHUNK)
        extra_warmup += PREWARM_CHUNK
    # exactly K timed steps, REPEATS times, each repeat bracketed by barrier + synchronize on both sides and reduced to the
    # MAX over ranks; `ms_per_step` / `value` = the MEDIAN repeat, min / max beside it
    repeats = []
    # (short regions are noisy — one 20-request region in five came out at 32 us on a fresh box: more of them)
    for _ in range(REPEATS if args.steps >= 200 else 2 * REPEATS - 1):
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wall_ms, dev_ms, _ = h.run(args.steps)       # exactly K timed steps (native loop, ends with a stream sync)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        repeats.append((elapsed, dev_ms, wall_ms))
    repeats_in_order = list(repeats)
    repeats.sort()
    elapsed, _, wall_ms = repeats[len(repeats) // 2]           # ms_per_step / value: the repeat with the median wall time
    dev_ms = sorted(r[1] for r in repeats)[len(repeats) // 2]   # roofline: the median HIP-event time of the repeats

    # latency percentiles: separate pass with one HIP event pair per request, at least 200 samples
    # whatever --steps is
    _, _, it = h.run(min(max(args.steps, 200), 500), per_request=True)
    # What the output costs by reuse distance of the arena: the same loop over rings of 1 / 2 / 6 arenas,
    # the library choosing its store policy per request (plain stores into an arena one of the plan's last two requests wrote,
    # nt / sc1 nt otherwise: store_policy_for, fcp_process.hip).  Extra field only; `value` is the ring --arena-ring names.
    arena_reuse = None
    if args.threads == 1 and not dist and not args.no_overlap:
        arena_reuse = {"what": "stream order, one serve worker, HIP-event us per request by the number of output arenas the loop rotates "
                               "through; 1 = TF's allocate_output handing a one-thread serving loop the block it just freed; the library "
                               "picks the store policy from the arena's reuse (profiles/r06_arena_reuse_store_policy.txt)",
                       "value_used_ring": args.arena_ring}
        ar_steps = max(args.steps, 800)
        try:                                         # (side records never cost the line: a failure is reported in their place)
            for ring in (1, 2, 6):
                ha = ServingHarness(model, device=local_rank, n_requests=args.requests, arena_ring=ring, n_threads=1, tables=h.tables,
                                    seed0=1000 * rank)
                ha.run(max(args.warmup, 200))
                _, a_dev, _ = ha.run(ar_steps)
                arena_reuse[f"ring_{ring}_us"] = a_dev * 1e3 / ar_steps
                ha.close()
        except Exception as e:
            arena_reuse["error"] = side_error(e)
    # overlapped serving (the reference harness' serve_workers): independent requests on
    # 3 streams hide each launch's ramp / tail behind its neighbours.  Extra field only.
    overlap = None
    if args.threads == 1 and not dist and not args.no_overlap:
        # its own floor whatever --steps / --warmup are: the workers need some tens of requests each to reach
        # steady overlap (round 1: the driver's --steps 20 gave 6 requests per worker and no overlap).  The
        # reference sweeps its serve workers too (AE/build_and_run.py:73-80: 2 / 4 / 8); how many streams overlap
        # best depends on how the process' streams fall on the GPU's hardware queues, so 2, 3 and 4 are measured
        # and all three reported.
        sweep = {}
        try:
            for workers in (2, 3, 4):
                hw = ServingHarness(model, device=local_rank, n_requests=args.requests, arena_ring=args.arena_ring, n_threads=workers,
                                    tables=h.tables, seed0=1000 * rank)
                ov_warm, ov_steps = max(args.warmup // workers, 50), max(args.steps // workers, 400)
                hw.run(ov_warm)
                w_ms, _, _ = hw.run(ov_steps)
                sweep[workers] = {"requests_per_worker": ov_steps, "warmup_per_worker": ov_warm,
                                  "us_per_request": w_ms * 1e3 / (workers * ov_steps)}
                hw.close()
            best = min(sweep, key=lambda k: sweep[k]["us_per_request"])
            overlap = {"serve_workers": best, **sweep[best],
                       "sweep_us_per_request": {str(k): v["us_per_request"] for k, v in sweep.items()},
                       "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}}
        except Exception as e:
            overlap = {"error": side_error(e)}
    # The same overlap behind ONE caller stream and ONE host thread — what the TensorFlow op has: the plan's private
    # streams (fcp_plan_set_private_streams), the consumer of each request (fcp_result_wait + a reader kernel on the
    # caller's stream, standing for Addons>ConcatOutputs) enqueued `lanes - 1` requests behind it.  Extra field only.
    single_caller = None
    if args.threads == 1 and not dist and not args.no_overlap:
        try:
            sweep = {}
            sc_warm, sc_steps = max(args.warmup, 100), max(args.steps, 1200)
            # the figure every private-stream number has to beat: the SAME loop — same consumer kernel behind every request —
            # with the requests in stream order on the caller's stream (private streams off)
            hs = ServingHarness(model, device=local_rank, n_requests=args.requests, arena_ring=max(args.arena_ring, 3), n_threads=1, tables=h.tables,
                                seed0=1000 * rank)
            hs.run_private(sc_warm, 3)
            so_ms, so_dev = hs.run_private(sc_steps, 3)
            stream_order_consumer = {"us_per_request": so_ms * 1e3 / sc_steps, "device_us_per_request": so_dev * 1e3 / sc_steps}
            hs.close()
            for lanes in (2, 3):                         # the library creates at most three (more were slower than one)
                hp = ServingHarness(model, device=local_rank, n_requests=args.requests, arena_ring=max(args.arena_ring, 3), n_threads=1,
                                    tables=h.tables, seed0=1000 * rank)
                hp.plan.set_private_streams(lanes)
                hp.run(1)
                t_v = time.perf_counter()
                hp.plan.verify_private_streams(hp.caller_stream(), 400)   # ... and the verification, as the shim's first Compute runs it
                verify_ms = (time.perf_counter() - t_v) * 1e3
                hp.run_private(sc_warm, lanes)
                w_ms, d_ms = hp.run_private(sc_steps, lanes)
                sweep[lanes] = {"requests": sc_steps, "warmup": sc_warm, "us_per_request": w_ms * 1e3 / sc_steps,
                                "device_us_per_request": d_ms * 1e3 / sc_steps, "verify_at_warmup_ms": verify_ms,
                                # 1: the library found its streams to overlap behind this caller stream and used them; 0: it did not
                                # and kept the requests on the caller's stream; -1: never asked (requests below the work threshold)
                                "verified_overlap": hp.plan.private_streams_verdict(hp.caller_stream()),
                                "supervisor": {k: v for k, v in hp.plan.private_streams_stats().items() if k != "supervised_stream"}}
                hp.close()
            # The reference's real protocol: `serve_workers` host threads share ONE Session, hence one compute stream
            # (recom_examples.patch:193-216): T threads issue on the one caller stream over the 3 private streams, every thread
            # enqueues the consumer of its request `depth - 1` of its own requests later.  depth 1 = FeatureColumnProcess and
            # Addons>ConcatOutputs back to back inside one Session::Run (what the rewritten graph does); depth 3 = a graph that
            # runs other work of the same thread between the two.  Every figure next to the same threads in stream order.
            threads_sweep = {}
            for T in (1, 2, 4):
                for depth in (1, 3):
                    ht = ServingHarness(model, device=local_rank, n_requests=args.requests, arena_ring=max(args.arena_ring, 3), n_threads=T, tables=h.tables,
                                        seed0=1000 * rank)
                    per = max(sc_steps // T, 300)
                    ht.run_private(max(sc_warm // T, 50), depth, T)      # private streams off: stream order + the same consumers
                    b_ms, _ = ht.run_private(per, depth, T)
                    ht.plan.set_private_streams(3)
                    ht.run(1)
                    ht.plan.verify_private_streams(ht.caller_stream(), 400)
                    ht.run_private(max(1400 // T, 50), depth, T)         # (long enough for three evaluations of the supervisor)
                    w_ms, d_ms = ht.run_private(per, depth, T)
                    st = ht.plan.private_streams_stats()
                    threads_sweep[f"threads_{T}_depth_{depth}"] = {
                        "private_streams_us": w_ms * 1e3 / (per * T), "stream_order_same_consumer_us": b_ms * 1e3 / (per * T),
                        "verdict_at_the_end": ht.plan.private_streams_verdict(ht.caller_stream()), "supervisor_demoted": st["demoted"],
                        "supervisor_last_ratio": st["last_ratio"], "supervisor_evaluations": st["evaluations"], "requests_per_thread": per}
                    ht.close()
            # and the cheap variant for callers that own their buffers (this harness does): FCP_ORDER_INPUTS_READY — the same K
            # requests back to back on the one stream, the fused kernel launched without the queue's barrier bit
            hr = ServingHarness(model, device=local_rank, n_requests=args.requests, arena_ring=args.arena_ring, n_threads=1, tables=h.tables,
                                seed0=1000 * rank)
            hr.plan.set_inputs_ready(True)
            hr.run(max(args.warmup, 100))
            r_ms, r_dev, _ = hr.run(max(args.steps, 1200))
            inputs_ready = {"us_per_request": r_ms * 1e3 / max(args.steps, 1200), "device_us_per_request": r_dev * 1e3 / max(args.steps, 1200),
                            "what": "fcp_plan_set_request_order(FCP_ORDER_INPUTS_READY): requests back to back on ONE stream, no events, no "
                                    "extra streams; the fused kernel may begin under the previous request's tail (any-order launch)"}
            hr.close()
            best = min(sweep, key=lambda k: sweep[k]["us_per_request"])
            single_caller = {"what": "one host thread, one caller stream (the TF op's situation): requests run on the plan's private "
                                     "streams, the consumer of request k (fcp_result_wait + a reader kernel on the caller's stream) is "
                                     "enqueued `private_streams - 1` requests later; host wall clock over the loop incl. the final sync.  The "
                                     "library verified on the first request that its private streams overlap behind this caller stream (or searched a "
                                     "hardware-queue mapping that does; profiles/r04_private_streams_queue_mapping.txt)",
                             "private_streams": best, **sweep[best],
                             "stream_order_same_consumer": {**stream_order_consumer,
                                                            "what": "the same loop with private streams OFF: requests in stream order on the caller's "
                                                                    "stream, the same consumer kernel behind each — the figure the private-stream number "
                                                                    "has to beat (the plain `value` loop has no consumer)"},
                             "host_threads_sweep": {"what": "T host threads issuing on the ONE caller stream (the reference's serve_workers share one "
                                                            "Session = one compute stream, recom_examples.patch:193-216), 3 private streams; depth 1 = the "
                                                            "consumer right behind its request (FeatureColumnProcess -> Addons>ConcatOutputs inside one "
                                                            "Session::Run), depth 3 = two more requests of the thread in between; us per request over all "
                                                            "threads, next to the same threads in stream order.  The supervisor A/Bs the two modes while "
                                                            "serving (48 requests each, at request 1, ~360, ~970, ...): a caller for whom the private streams lose "
                                                            "(ratio > 0.97 twice) is demoted and finishes in stream order",
                                                    **threads_sweep},
                             "sweep_us_per_request": {str(k): v["us_per_request"] for k, v in sweep.items()},
                             "sweep_verified_overlap": {str(k): v["verified_overlap"] for k, v in sweep.items()},
                             "inputs_ready_back_to_back": inputs_ready}
        except Exception as e:
            single_caller = {"error": side_error(e)}
    batch = model.batch
    steps_total = args.steps * args.threads
    ms_per_step = elapsed * 1e3 / steps_total
    value = world * batch * steps_total / elapsed
    dev_ms_per_req = dev_ms / args.steps             # worker 0's stream, HIP events over the timed region

    rec = {
        "metric": METRIC if args.workload == "s2" else
                  f"inference QPS + p50 latency, {model.name} config{' (staged request form: as Addons>ConcatInputs leaves it in HBM)' if args.staged and raw_model is not model else ''}, batch {batch}, 1xMI355X",
        "value": value, "unit": "inferences/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "extra_warmup_requests": extra_warmup, "timed_region_s": elapsed,
        # what the bracket itself costs per region (first launch out of an idle queue, the closing torch.cuda.synchronize() =
        # hipDeviceSynchronize): host wall time of the region minus the HIP-event time of its K requests; ~30 us, i.e. 5 % of
        # a 20-request region (0.55 ms) and 0.05 % of the default 2000-request one
        "bracket_overhead_us": (elapsed - dev_ms * 1e-3) * 1e6,
        "repeats": {"n": len(repeats), "reported": "median", "ms_per_step_min": repeats[0][0] * 1e3 / steps_total,
                    "ms_per_step_max": repeats[-1][0] * 1e3 / steps_total,
                    "ms_per_step_all": [r[0] * 1e3 / steps_total for r in repeats],
                    "kernel_avg_us_all": [r[1] * 1e3 / args.steps for r in repeats],
                    "kernel_avg_us_in_time_order": [r[1] * 1e3 / args.steps for r in repeats_in_order],
                    "what": f"exactly --steps requests timed {len(repeats)} times (barrier + synchronize on both sides of each, max over "
                            f"ranks) behind >= {PREWARM_S} s of untimed requests; ms_per_step / value / roofline are the median repeat"},
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{model.name}: {model.description}", "batch": batch,
                   "columns": model.spec.n_columns, "table_bytes": model.table_bytes(),
                   "parallelism": f"{world} replica(s), requests sharded across ranks, no collective (placement gate: "
                                  f"{model.table_bytes() / 1e9:.0f} GB of tables fit one GPU)",
                   "serve_workers": args.threads, "arena_ring": args.arena_ring},
        "requests_per_s": value / batch,
        "verified": verified,
        "p50_latency_ms": float(np.percentile(it, 50)), "p95_latency_ms": float(np.percentile(it, 95)),
        "latency_scope": f"device latency of one request (HIP event pair around it on the launch stream, {len(it)} samples), "
                         "inputs resident in HBM; host packing + H2D are reported separately (profiles/HISTORY.md, PCIe-inclusive rate; DESIGN.md section 5)",
    }
    if rank == 0:
        achieved = bytes_alg["total"] / (dev_ms_per_req * 1e-3) / 1e9
        # (the PMC record of RAGGED is for the default segment encoding: SparseTensor indices, pre-pass included)
        # (PMC records of RAGGED: the staged form = the default, and the request as delivered with its pre-pass)
        traffic, traffic_source = measured_traffic(args.workload if args.workload != "ragged" else
                                                   "ragged" if args.seg == "indices" and args.staged else
                                                   "ragged_as_delivered" if args.seg == "indices" else f"ragged_{args.seg}")
        rec["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "fcp_dense_kernel" if all(c.form in (1, 4) for c in model.spec.columns) else "fcp_ragged_kernel",
            "kernel_avg_us": dev_ms_per_req * 1e3,
            "algorithmic_bytes_per_request": bytes_alg,
            "read_only_frac": bytes_alg["read"] / (dev_ms_per_req * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        try:
            rec["roofline"]["measured_copy_peak_GBs"] = copy_probe() / 1e9
            rec["roofline"]["access_mix"] = access_mix_floor(model, h, bytes_alg, dev_ms_per_req * 1e3)
        except Exception as e:
            rec["roofline"]["access_mix"] = {"error": side_error(e)}
        if arena_reuse:
            rec["arena_reuse"] = arena_reuse
        if overlap and "error" in overlap:
            rec["overlapped_serving"] = overlap
        elif overlap:
            overlap["inferences_per_s"] = batch / (overlap["us_per_request"] * 1e-6)
            overlap["aggregate_frac_of_peak"] = bytes_alg["total"] / (overlap["us_per_request"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            rec["overlapped_serving"] = overlap
            rec["qps_definition"] = ("`value` = one serve worker issuing exactly --steps requests back to back (the bench contract's timed "
                                     "region; roofline.frac is this single-stream kernel); the serving QPS under the reference's own "
                                     "protocol - benchmark_multi_thread: serve_workers threads x num_iterations requests, "
                                     "examples/cc/recom_examples.patch:193-225 - is overlapped_serving.inferences_per_s "
                                     f"({overlap['inferences_per_s'] / 1e6:.1f} M with {overlap['serve_workers']} workers), next to the single-request "
                                     "p50 above")
        if single_caller and "error" in single_caller:
            rec["single_caller_stream"] = single_caller
        elif single_caller:
            single_caller["inferences_per_s"] = batch / (single_caller["us_per_request"] * 1e-6)
            single_caller["frac_of_peak"] = bytes_alg["total"] / (single_caller["us_per_request"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            single_caller["inputs_ready_back_to_back"]["frac_of_peak"] = (
                bytes_alg["total"] / (single_caller["inputs_ready_back_to_back"]["us_per_request"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
            rec["single_caller_stream"] = single_caller
        if args.staged and raw_model is not model:
            # the same requests resident AS DELIVERED (int64 ids, SparseTensor indices; the segment-offset pre-pass runs on the
            # device): the like-for-like figure of rounds 1-2, in the same record
            hd = ServingHarness(raw_model, device=local_rank, n_requests=args.requests, arena_ring=args.arena_ring, n_threads=1, tables=h.tables,
                                seed0=1000 * rank)
            hd.run(max(args.warmup, 1))
            _, d_ms, _ = hd.run(args.steps)
            b_ad = hd.algorithmic_bytes()
            rec["as_delivered"] = {"us_per_request": d_ms * 1e3 / args.steps, "algorithmic_bytes_per_request": b_ad["total"],
                                   "frac": b_ad["total"] / (d_ms * 1e-3 / args.steps) / 1e9 / HBM_PEAK_GBS,
                                   "what": "the same requests resident as the graph's tensors are (int64 ids, SparseTensor indices), "
                                           "segment-offset pre-pass on the device; `bench.py --as-delivered` makes this the headline"}
            hd.close()
        if args.workload == "ragged" and args.seg == "indices" and not args.staged:
            # the same requests in the form the rewritten graph's Addons>ConcatInputs leaves in HBM (plan-file stage section: ids
            # int32, SparseTensor indices -> row offsets while packing): no pre-pass, no search — and what that conversion costs
            # the HOST per request, stated right beside it (one thread, like the TF op; pack pool of 4 / 8 / 16 threads)
            smodel = synth.staged_model(raw_model)
            hs2 = ServingHarness(smodel, device=local_rank, n_requests=args.requests, arena_ring=args.arena_ring, n_threads=1, tables=h.tables,
                                 seed0=1000 * rank)
            hs2.run(max(args.warmup, 1))
            _, s_ms, _ = hs2.run(args.steps)
            b_st = hs2.algorithmic_bytes()
            rec["staged"] = {"us_per_request": s_ms * 1e3 / args.steps, "algorithmic_bytes_per_request": b_st["total"],
                             "frac": b_st["total"] / (s_ms * 1e-3 / args.steps) / 1e9 / HBM_PEAK_GBS,
                             "what": "the same requests as the staged Addons>ConcatInputs leaves them in HBM (ids int32, row offsets instead "
                                     "of SparseTensor indices): the device side of the deployment the rewritten graph (--staged) produces; "
                                     "`bench.py --workload ragged --staged` makes this the headline.  The conversion is HOST work per request: "
                                     "see `staging` (never inside a timed region)"}
            hs2.close()
            if "FCP_LIB_DIR" not in os.environ:
                rec["staged"]["staging"] = host_staging_cost(raw_model)
        if args.staged and "FCP_LIB_DIR" not in os.environ:
            rec["staging"] = host_staging_cost(raw_model)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            try:
                rec["cpu_baseline"] = cpu_baseline(model)
            except Exception as e:                   # (e.g. a host that cannot hold the tables and cannot even sample them)
                rec["cpu_baseline"] = {"value": None, "unit": "inferences/s", "cores": 0, "kind": "port", "sample": "failed",
                                       "error": side_error(e)}
    h.close()
    if world > 1 and args.workload == "s2" and not os.environ.get("FCP_BENCH_NO_SHARDED_RECORD"):
        # The default multi-GPU line is replicas (S2's 120 GB fit one GPU: no collective); so that a scaling run on a multi-GPU
        # node still measures the one exchange this path has, BASELINE configs[4]'s row-sharded step runs on the same ranks for
        # <= 10 s afterwards and rides along as `sharded` (every rank takes part; rank 0 reports)
        from recom_amd.shard import bench_row_sharded_record
        del h
        torch.cuda.empty_cache()
        # The headline line must not depend on this extra.  An exception is caught (all ranks fail alike); a rank that never
        # comes back from the exchange (RCCL with N > 1 ranks has not run on this pool: every box has one GPU) would take the
        # whole line with it, so every rank arms a watchdog first: when it fires, rank 0 prints the finished headline with the
        # reason under `sharded` and every rank leaves with status 0 (ctypes and torch release the GIL inside their calls)
        import threading
        dist.barrier()                               # rank 0 arrives later (its side records): every rank's clock starts here
        limit_s = float(os.environ.get("FCP_BENCH_SHARDED_WATCHDOG_S", "240"))
        line_lock, leg_done = threading.Lock(), []

        def abandon():
            with line_lock:
                if leg_done:
                    return
                if rank == 0:
                    rec["sharded"] = {"error": f"abandoned: no result within {limit_s:g} s (the replicated headline above is complete)"}
                    print(json.dumps(rec), flush=True)
                os._exit(0)
        # (rank 0 gives up first: a peer that left earlier would only turn its wait into a "connection closed" error)
        watchdog = threading.Timer(limit_s + (0.0 if rank == 0 else 3.0), abandon)
        watchdog.daemon = True
        watchdog.start()
        leg_failed = False
        try:
            sharded = bench_row_sharded_record(args, rank, world, local_rank, dist,
                                               int(hbm_override) if hbm_override else device_hbm_bytes(local_rank))
        except Exception as e:
            sharded, leg_failed = {"error": f"{type(e).__name__}: {e}"[:400]}, True
        with line_lock:                              # from here the line is this thread's to print
            leg_done.append(True)
            watchdog.cancel()
        if rank == 0:
            rec["sharded"] = sharded
        if leg_failed:
            # the process group is suspect (a peer may be gone or stuck in the exchange): no further collective, no
            # destroy_process_group — the line (rank 0) and out
            if rank == 0:
                print(json.dumps(rec), flush=True)
            os._exit(0)
        h = None
    if rank == 0:
        if args.workload == "s2" and world == 1 and args.ids == "uniform" and not args.no_pcie and not args.no_cpu_baseline and not args.batch and not args.columns:
            del h
            torch.cuda.empty_cache()
            rec["pcie_inclusive"] = pcie_inclusive()
            # HBM traffic of the dominant kernel measured on THIS box in THIS run (the committed PMC record is the fallback
            # and stays in the line as `traffic_record` for comparison)
            live, why = live_traffic(args.arena_ring)
            rec["roofline"]["traffic_record"] = {"bytes": rec["roofline"]["traffic"], "source": rec["roofline"]["traffic_source"]}
            if live is not None:
                rec["roofline"]["traffic"], rec["roofline"]["traffic_source"] = live, why
            else:
                rec["roofline"]["traffic_live_error"] = why
        print(json.dumps(rec))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Plan-owned private streams (fcp_plan_set_private_streams / fcp_result_wait): ONE host thread and ONE caller stream —
what the TensorFlow op has (feature_column_process_op_gpu.cu.cc:65-131; the reference harness' serve workers share one
Session, recom_examples.patch:193-216) — with several requests in flight on the plan's own streams.  The consumer on
the caller's stream (what Addons>ConcatOutputs and the layers behind it are in the rewritten graph) must always see a
complete arena; results stay bit-exact with the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from recom_amd import lib
    lib.load()  # fail loudly if the HIP extension is missing
    return torch


def _busy(torch, stream, ms=2.0):
    """Keep `stream` busy for about `ms` milliseconds so that whatever is enqueued behind it runs LATE: a reader that did
    not wait for the private stream's kernel would then race with it instead of trivially finding it finished."""
    with torch.cuda.stream(stream):
        if hasattr(torch.cuda, "_sleep"):
            torch.cuda._sleep(int(ms * 2.0e6))
        else:  # pragma: no cover
            a = torch.empty((4096, 4096), device="cuda")
            for _ in range(8):
                a = a @ a


def _poison_allocator(torch, nbytes, n=8):
    """The arenas torch hands out next are blocks that held NaNs: an unfinished arena cannot look right by accident."""
    junk = [torch.full((nbytes // 4 + 64,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    del junk


@pytest.mark.parametrize("no_caller_wait", [False, True])
def test_three_requests_in_flight_behind_one_caller_stream(torch_cuda, oracle, no_caller_wait):
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    m = synth.model_mixed(batch=192, vocab=4999, n_groups=1)          # every column form, dynamic shapes
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    op.plan.set_private_streams(3, no_caller_wait=no_caller_wait, always=True)   # (small requests: below the work threshold)
    caller = torch.cuda.Stream()
    reqs = [m.make_request(50 + k, B=150 + 7 * k) for k in range(14)]  # new shapes on every request
    packed = [concat_inputs(r.inputs) for r in reqs]
    blobs = [torch.from_numpy(p[0]).cuda() for p in packed]
    torch.cuda.synchronize()
    _poison_allocator(torch, op.plan.arena_bytes(packed[-1][2], reqs[-1].symbols))
    depth, in_flight, snaps, keep = 3, [], [], []
    with torch.cuda.stream(caller):
        for k, (r, p, b) in enumerate(zip(reqs, packed, blobs)):
            if k % 4 == 0:
                _busy(torch, caller)                                   # the caller's stream runs behind the host
            out = op(b, p[1], p[2], tabs, r.symbols, defer_wait=True)  # enqueued on a private stream; caller does not wait
            in_flight.append(out)
            keep.append(out)                                           # arenas stay allocated until everything is checked
            if len(in_flight) >= depth:
                done = in_flight.pop(0)
                done.wait()                                            # Addons>ConcatOutputs: fcp_result_wait on ITS stream
                snaps.append(done.groups[0].clone())                   # the consumer kernel, on the caller's stream
        while in_flight:
            done = in_flight.pop(0)
            done.wait()
            snaps.append(done.groups[0].clone())
    caller.synchronize()
    for r, p, snap in zip(reqs, packed, snaps):
        want, _ = oracle.process_feature_columns(m.spec.to_dict(), p[0], p[1], p[2], tabs_np, r.symbols)
        assert np.array_equal(snap.cpu().numpy(), want[0])
    # a host reader: fcp_result_synchronize instead of a stream wait
    out = op(blobs[0], packed[0][1], packed[0][2], tabs, reqs[0].symbols, stream=caller.cuda_stream, defer_wait=True)
    from recom_amd import lib
    lib.check(lib.load().fcp_result_synchronize(out.buffer.data_ptr()), "fcp_result_synchronize")
    want, _ = oracle.process_feature_columns(m.spec.to_dict(), *packed[0], tabs_np, reqs[0].symbols)
    got = torch.empty_like(out.groups[0], device="cpu").pin_memory()
    side = torch.cuda.Stream()                                         # neither the caller's nor a private stream
    with torch.cuda.stream(side):
        got.copy_(out.groups[0], non_blocking=True)
    side.synchronize()
    assert np.array_equal(got.numpy(), want[0])
    torch.cuda.synchronize()
    op.plan.set_private_streams(0)                                     # off again: the request runs on the caller's stream
    with torch.cuda.stream(caller):
        out = op(blobs[1], packed[1][1], packed[1][2], tabs, reqs[1].symbols)
    caller.synchronize()
    want, _ = oracle.process_feature_columns(m.spec.to_dict(), *packed[1], tabs_np, reqs[1].symbols)
    assert np.array_equal(out.groups[0].cpu().numpy(), want[0])


def test_default_call_orders_the_callers_stream_itself(torch_cuda, oracle):
    """Without defer_wait the Python op enqueues the wait at once: torch code on the caller's stream reads results as
    before, the bad-id counter covers private lanes, and a stream capture keeps the request on the captured stream."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    from recom_amd.plan import FLAG_COUNT_BAD_IDS
    import dataclasses
    torch = torch_cuda
    m = synth.model_mixed(batch=96, vocab=997, n_groups=1)
    spec = dataclasses.replace(m.spec, flags=m.spec.flags | FLAG_COUNT_BAD_IDS)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(spec, 0)
    op.plan.set_private_streams(2, always=True)
    s = torch.cuda.Stream()
    total_bad = 0
    for k in range(6):
        r = m.make_request(300 + k, B=80 + k)
        ids0 = r.inputs[spec.columns[0].ids_input]
        ids0.reshape(-1)[:3] = 10 ** 6                                  # three ids outside the vocabulary
        blob, offsets, shapes = concat_inputs(r.inputs)
        with torch.cuda.stream(s):
            _busy(torch, s, 0.5)
            out = op(torch.from_numpy(blob).cuda(), offsets, shapes, tabs, r.symbols)
            got = out.groups[0].clone()
        s.synchronize()
        want, bad = oracle.process_feature_columns(spec.to_dict(), blob, offsets, shapes, tabs_np, r.symbols)
        total_bad += bad
        assert np.array_equal(got.cpu().numpy(), want[0])
    assert total_bad > 0 and op.plan.read_bad_ids(s.cuda_stream) == total_bad
    # capture: the request is recorded on the captured stream (no cross-stream events inside a capture)
    r = m.make_request(1, B=64)
    blob, offsets, shapes = concat_inputs(r.inputs)
    d_blob = torch.from_numpy(blob).cuda()
    op.plan.set_private_streams(0)
    with torch.cuda.stream(s):
        op(d_blob, offsets, shapes, tabs, r.symbols)                    # descriptors resident for stream s
    torch.cuda.synchronize()
    op.plan.set_private_streams(2, always=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = op(d_blob, offsets, shapes, tabs, r.symbols)
    g.replay()
    torch.cuda.synchronize()
    want, _ = oracle.process_feature_columns(spec.to_dict(), blob, offsets, shapes, tabs_np, r.symbols)
    assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
    from recom_amd import lib
    lib.check(lib.load().fcp_plan_release_captures(op.plan.handle), "release")


def test_request_on_the_callers_stream_forgets_the_older_private_stream_result_of_its_arena(torch_cuda, oracle, monkeypatch):
    """A plan with private streams keeps some requests on the caller's stream (below the work threshold; a stream that is
    being captured).  Such a request writes its arena in stream order: the registry entry an OLDER private-stream request
    left for the same memory is dropped, so the reader's fcp_result_wait finds nothing — inside a capture it would
    otherwise wait for an event of another stream that the capture knows nothing about.  One arena address for every
    request; first request on the caller's stream (nothing installed yet), second on a private stream, third captured."""
    from recom_amd import lib, synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    monkeypatch.setenv("FCP_PRIVATE_MIN_WORK_BYTES", "1")              # read by fcp_plan_set_private_streams
    m = synth.model_mixed(batch=96, vocab=997, n_groups=1)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    fixed = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")

    def allocators():
        state = {"arena": None, "temps": []}

        def _alloc(_ctx, nbytes):
            assert nbytes <= fixed.numel()
            state["arena"] = fixed[:int(nbytes)]
            return fixed.data_ptr()

        def _alloc_temp(_ctx, nbytes):
            t = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
            state["temps"].append(t)
            return t.data_ptr()
        return state, lib.ALLOC_FN(_alloc), lib.ALLOC_FN(_alloc_temp)
    op._allocators = allocators
    op.plan.set_private_streams(2)
    r = m.make_request(7, B=64)
    blob, offsets, shapes = concat_inputs(r.inputs)
    d_blob = torch.from_numpy(blob).cuda()
    want, _ = oracle.process_feature_columns(m.spec.to_dict(), blob, offsets, shapes, tabs_np, r.symbols)
    s = torch.cuda.Stream()
    L = lib.load()
    with torch.cuda.stream(s):
        out = op(d_blob, offsets, shapes, tabs, r.symbols)             # 1: caller's stream (no work figure yet), installs the shapes for s
        first = out.groups[0].clone()
        _busy(torch, s, 0.3)
        out = op(d_blob, offsets, shapes, tabs, r.symbols, defer_wait=True)   # 2: private stream; the registry now names the arena
        out.wait()
        second = out.groups[0].clone()
    torch.cuda.synchronize()
    assert np.array_equal(first.cpu().numpy(), want[0]) and np.array_equal(second.cpu().numpy(), want[0])
    fixed.zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = op(d_blob, offsets, shapes, tabs, r.symbols)             # 3: captured: stays on s, and so does the reader's wait
        lib.check(L.fcp_result_wait(fixed.data_ptr(), s.cuda_stream), "fcp_result_wait inside the capture")
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
    lib.check(L.fcp_plan_release_captures(op.plan.handle), "release")


def test_private_streams_are_verified_per_caller_stream_and_results_do_not_depend_on_the_verdict(torch_cuda, oracle, capfd, monkeypatch):
    """Whether event-linked streams overlap depends on the hardware queues the runtime mapped them to (N streams created
    before the lanes change it, profiles/r04_private_streams_queue_mapping.txt), so the first request of a caller stream
    probes it and the library may re-create its lanes or leave that caller's requests on its own stream.  Here: dummy
    streams shift the mapping; every caller stream gets its verdict exactly once (FCP_DIAG=private_verify_verbose names it);
    results are bit-exact with the oracle whatever the verdict; FCP_PRIVATE_NO_VERIFY probes nothing; the diagnostic entry
    point reports both times of the synthetic pattern."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    monkeypatch.setenv("FCP_DIAG", "private_verify_verbose")
    m = synth.model_mixed(batch=96, vocab=997, n_groups=1)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    reqs = [m.make_request(900 + k, B=70 + k) for k in range(6)]
    packed = [concat_inputs(r.inputs) for r in reqs]
    blobs = [torch.from_numpy(p[0]).cuda() for p in packed]
    want = [oracle.process_feature_columns(m.spec.to_dict(), p[0], p[1], p[2], tabs_np, r.symbols)[0][0] for r, p in zip(reqs, packed)]
    dummies = []
    for n_dummy in (0, 2, 3):
        while len(dummies) < n_dummy:                                  # streams that have run something hold hardware queues
            d = torch.cuda.Stream()
            with torch.cuda.stream(d):
                torch.zeros(8, device="cuda").add_(1)
            d.synchronize()
            dummies.append(d)
        for verify in (True, False):
            op = FeatureColumnProcess(m.spec, 0)
            op.plan.set_private_streams(3, always=True, verify=verify)
            callers = [torch.cuda.Stream(), torch.cuda.Stream()]
            capfd.readouterr()
            for rounds in range(2):
                for s in callers:
                    outs = []
                    with torch.cuda.stream(s):
                        for k in range(len(reqs)):
                            outs.append(op(blobs[k], packed[k][1], packed[k][2], tabs, reqs[k].symbols, defer_wait=True))
                        for o in outs:
                            o.wait()
                        got = [o.groups[0].clone() for o in outs]
                    s.synchronize()
                    for k in range(len(reqs)):
                        assert np.array_equal(got[k].cpu().numpy(), want[k]), (n_dummy, verify, rounds, k)
            err = capfd.readouterr().err
            verdicts = [ln for ln in err.splitlines() if "lanes as created" in ln]
            assert len(verdicts) == (2 if verify else 0), err          # once per caller stream, never again
            for s in callers:                                          # ... and the plan says what it decided
                assert op.plan.private_streams_verdict(s.cuda_stream) in ((0, 1) if verify else (-1,))
            assert op.plan.private_streams_verdict(torch.cuda.Stream().cuda_stream) == -1
            a, b = op.plan.probe_private_streams(callers[0].cuda_stream, requests=12, spin_us=20, grid_blocks=1)
            assert a > 12 * 20 and b > 0
            del op


class _RawBlob:
    """A device address handed to the op wrapper in place of a torch tensor (the stager's blob)."""

    def __init__(self, ptr, nbytes):
        self._p, self._n = ptr, nbytes

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n

    def element_size(self):
        return 1


@pytest.mark.parametrize("zero_copy", [False, True])
def test_stager_does_not_recycle_a_blob_that_a_private_stream_still_reads(torch_cuda, oracle, zero_copy):
    """The request stager recycles its ring behind events recorded on the CALLER's stream — which does not wait for
    private-stream kernels.  The library therefore files a private-stream request's input blob next to its arena, and the
    stager waits for that reader before it overwrites the slot.  Provoked here: ONE private stream (unverified) kept busy
    by a dozen slow requests, a ring of ONE slot, small requests staged back to back: each of them sits in the lane's queue while
    the next one is staged into the same slot.  Every result must be that of ITS request."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    torch = torch_cuda
    m = synth.model_s2(columns=120, vocab=3000, batch=256)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    op.plan.set_private_streams(1, always=True, verify=False)
    st = RequestStager(4 << 20, m.spec.n_host_inputs, sum(m.spec.host_input_ranks), depth=1, n_threads=2, zero_copy=zero_copy)
    big = m.make_request(1, B=50000)                                   # ~1 GB of output: the lane is busy for a while
    big_blob, big_off, big_shp = concat_inputs(big.inputs)
    d_big = torch.from_numpy(big_blob).cuda()
    small = [m.make_request(10 + k, B=48 + k) for k in range(5)]
    want = []
    for r in small:
        blob, offsets, shapes = concat_inputs(r.inputs)
        want.append(oracle.process_feature_columns(m.spec.to_dict(), blob, offsets, shapes, tabs_np, r.symbols)[0][0])
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    for attempt in range(3):
        with torch.cuda.stream(s):
            slow = [op(d_big, big_off, big_shp, tabs, big.symbols, defer_wait=True) for _ in range(12)]   # several ms of lane time
            outs = []
            for r in small:
                ptr, nbytes, offsets, shapes = st.stage(r.inputs, stream=s.cuda_stream)
                outs.append(op(_RawBlob(ptr, nbytes), offsets, shapes, tabs, r.symbols, defer_wait=True))
            got = []
            for o in outs:
                o.wait()
                got.append(o.groups[0].clone())
            for o in slow:
                o.wait()
        s.synchronize()
        for k in range(len(small)):
            assert np.array_equal(got[k].cpu().numpy(), want[k]), (zero_copy, attempt, k)
        del slow, outs
    st.close()


def test_native_single_caller_loop_runs_s2_shape_and_is_faster_than_serial(torch_cuda):
    """The native loop bench.py times (fcp_harness_run_private): S2's shape at a small vocabulary, one caller stream,
    depth 3 over 3 private streams against the same requests back to back on that stream.  Asserts that it runs, that the
    arenas it leaves are the closed-form results, and reports both times (no timing assertion: boxes differ)."""
    from recom_amd import synth
    from recom_amd.harness import ServingHarness
    torch = torch_cuda
    model = synth.model_s2(columns=1000, vocab=2000)
    h = ServingHarness(model, device=0, n_requests=8, arena_ring=6, n_threads=1)
    assert h.verify_resident()["checked"] > 0
    h.run(50)
    serial_ms, _, _ = h.run(300)
    h.plan.set_private_streams(3)
    h.run_private(50, 3)
    priv_ms, priv_dev = h.run_private(300, 3)
    torch.cuda.synchronize()
    assert h.verify_resident()["checked"] > 0                            # the Python op over the same plan, private streams on
    print(f"S2 shape, vocab 2000: {serial_ms / 300 * 1e3:.2f} us per request on one stream, "
          f"{priv_ms / 300 * 1e3:.2f} us with 3 private streams behind one caller stream")
    assert priv_ms > 0 and priv_dev > 0
    h.close()


def test_inputs_ready_requests_overlap_on_one_stream_and_consumers_still_wait(torch_cuda, oracle):
    """fcp_plan_set_request_order(FCP_ORDER_INPUTS_READY): the fused kernel is launched without the queue's barrier bit, so
    bursts of requests on ONE stream overlap; every ordinary command queued behind them (the consumer) still waits for
    them.  Bursts of 1..5 requests with new shapes each, then one clone per result on the same stream: bit-exact with the
    oracle.  Plans that queue a pre-pass of their own (SparseTensor indices as delivered) keep stream order: same check."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    for m in (synth.model_mixed(batch=150, vocab=4999, n_groups=1),                       # segment-id columns: pre-pass or in-block search
              synth.staged_model(synth.model_ragged(columns=64, vocab=5000, batch=128, seg="indices")),  # CSR: one kernel per request
              synth.model_s2(columns=200, vocab=3000, batch=256)):
        tabs_np = m.numpy_tables()
        tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
        op = FeatureColumnProcess(m.spec, 0)
        op.plan.set_inputs_ready(True)
        s = torch.cuda.Stream()
        k = 0
        for burst in (1, 2, 3, 5, 4):
            reqs = [m.make_request(900 + k + i, B=m.batch - 3 * i - burst) for i in range(burst)]
            k += burst
            packed = [concat_inputs(r.inputs) for r in reqs]
            blobs = [torch.from_numpy(p[0]).cuda() for p in packed]
            torch.cuda.synchronize()                                # the promise: blobs complete, arenas fresh
            _poison_allocator(torch, op.plan.arena_bytes(packed[0][2], reqs[0].symbols), n=burst + 1)
            with torch.cuda.stream(s):
                _busy(torch, s, 0.3)                               # the stream runs behind the host: the burst piles up
                outs = [op(b, p[1], p[2], tabs, r.symbols) for r, p, b in zip(reqs, packed, blobs)]
                snaps = [o.groups[0].clone() for o in outs]        # ordinary commands: ordered behind every request before them
            s.synchronize()
            for r, p, snap in zip(reqs, packed, snaps):
                want, _ = oracle.process_feature_columns(m.spec.to_dict(), p[0], p[1], p[2], tabs_np, r.symbols)
                assert np.array_equal(snap.cpu().numpy(), want[0]), (m.name, burst)


def test_host_threads_share_a_plan_with_private_streams(torch_cuda, oracle):
    """The reference harness' serve workers share one Session — several host threads call the op concurrently, here each
    with its own caller stream (and, second pass, ALL on one stream: TensorFlow's single compute stream) over ONE plan with
    three private streams: lane rotation, per-lane event pairs and the result registry under concurrency; every result
    bit-exact with the oracle."""
    import threading
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    m = synth.model_mixed(batch=120, vocab=2999, n_groups=1)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    op.plan.set_private_streams(3, always=True)
    n_threads, per_thread = 4, 10
    reqs = [[m.make_request(2000 + 100 * t + k, B=90 + 3 * t + k) for k in range(per_thread)] for t in range(n_threads)]
    packed = [[concat_inputs(r.inputs) for r in rs] for rs in reqs]
    blobs = [[torch.from_numpy(p[0]).cuda() for p in ps] for ps in packed]
    want = [[oracle.process_feature_columns(m.spec.to_dict(), p[0], p[1], p[2], tabs_np, r.symbols)[0][0] for r, p in zip(rs, ps)]
            for rs, ps in zip(reqs, packed)]
    torch.cuda.synchronize()
    for shared_stream in (False, True):
        one = torch.cuda.Stream()
        errors, results = [], [[None] * per_thread for _ in range(n_threads)]

        def worker(t):
            try:
                s = one if shared_stream else torch.cuda.Stream()
                pending = []
                with torch.cuda.stream(s):
                    for k in range(per_thread):
                        out = op(blobs[t][k], packed[t][k][1], packed[t][k][2], tabs, reqs[t][k].symbols, defer_wait=True)
                        pending.append((k, out))
                        if len(pending) >= 2:
                            j, o = pending.pop(0)
                            o.wait()
                            results[t][j] = (o, o.groups[0].clone())
                    for j, o in pending:
                        o.wait()
                        results[t][j] = (o, o.groups[0].clone())
                s.synchronize()
            except BaseException as e:  # noqa: BLE001
                errors.append((t, repr(e)))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        torch.cuda.synchronize()
        for t in range(n_threads):
            for k in range(per_thread):
                assert np.array_equal(results[t][k][1].cpu().numpy(), want[t][k]), (shared_stream, t, k)


def test_two_plans_share_the_devices_private_streams(torch_cuda, oracle):
    """The command processor overlaps at most four event-linked queues, so the private streams belong to the DEVICE, not to
    the plan: two models served by one process (two plans, two host threads, ONE caller stream — TensorFlow's situation)
    rotate over the same three streams.  Every result bit-exact; a plan that switches the mode off and on again, or is
    destroyed, leaves the other's results intact."""
    import threading
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    models = [synth.model_mixed(batch=110, vocab=1999, n_groups=1), synth.model_s2(columns=60, vocab=900, batch=128)]
    ops, tabs, tabs_np, reqs, packed, blobs, want = [], [], [], [], [], [], []
    for i, m in enumerate(models):
        tn = m.numpy_tables()
        tabs_np.append(tn)
        tabs.append([torch.from_numpy(t).cuda() for t in tn])
        op = FeatureColumnProcess(m.spec, 0)
        op.plan.set_private_streams(3, always=True)
        ops.append(op)
        rs = [m.make_request(700 + 50 * i + k, B=64 + 5 * k + i) for k in range(8)]
        ps = [concat_inputs(r.inputs) for r in rs]
        reqs.append(rs)
        packed.append(ps)
        blobs.append([torch.from_numpy(p[0]).cuda() for p in ps])
        want.append([oracle.process_feature_columns(m.spec.to_dict(), p[0], p[1], p[2], tn, r.symbols)[0][0] for r, p in zip(rs, ps)])
    torch.cuda.synchronize()
    one = torch.cuda.Stream()
    for rounds in range(3):
        errors, results = [], [[None] * 8 for _ in models]

        def worker(i):
            try:
                pending = []
                with torch.cuda.stream(one):
                    for k in range(8):
                        o = ops[i](blobs[i][k], packed[i][k][1], packed[i][k][2], tabs[i], reqs[i][k].symbols, defer_wait=True)
                        pending.append((k, o))
                        if len(pending) >= 3:
                            j, oo = pending.pop(0)
                            oo.wait()
                            results[i][j] = oo.groups[0].clone()
                    for j, oo in pending:
                        oo.wait()
                        results[i][j] = oo.groups[0].clone()
            except BaseException as e:  # noqa: BLE001
                errors.append((i, repr(e)))

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(models))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        one.synchronize()
        assert not errors, errors
        for i in range(len(models)):
            for k in range(8):
                assert np.array_equal(results[i][k].cpu().numpy(), want[i][k]), (rounds, i, k)
        if rounds == 0:
            ops[1].plan.set_private_streams(0)                         # off ...
            ops[1].plan.set_private_streams(2, always=True)            # ... and on again, with fewer streams
        elif rounds == 1:
            ops[1].plan.close()                                        # the other plan keeps working on the shared streams
            ops[1] = FeatureColumnProcess(models[1].spec, 0)
            ops[1].plan.set_private_streams(3, always=True, verify=False)


def test_per_column_layout_concat_outputs_waits_for_the_private_stream(torch_cuda, oracle):
    """FCP_LAYOUT_PER_COLUMN (the reference's arena: one buffer per column) with private streams: fcp_concat_outputs — the
    reference's second pass, here on the caller's stream — orders itself behind the lookup kernels it reads from."""
    import dataclasses
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs, concat_outputs
    from recom_amd.plan import LAYOUT_PER_COLUMN
    torch = torch_cuda
    m = synth.model_mixed(batch=140, vocab=1999, n_groups=1)
    spec = dataclasses.replace(m.spec, layout=LAYOUT_PER_COLUMN)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(spec, 0)
    op.plan.set_private_streams(2, always=True)
    s = torch.cuda.Stream()
    order = [k for _, k in sorted((c.concat_slot, k) for k, c in enumerate(spec.columns))]
    for seed in range(4):
        r = m.make_request(70 + seed, B=120 + seed)
        blob, offsets, shapes = concat_inputs(r.inputs)
        d_blob = torch.from_numpy(blob).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            _busy(torch, s, 1.0)
            out = op(d_blob, offsets, shapes, tabs, r.symbols, defer_wait=True)
            cat = concat_outputs([out.column(k) for k in order])       # fcp_concat_outputs waits for the arena itself
        s.synchronize()
        want, _ = oracle.process_feature_columns(m.spec.to_dict(), blob, offsets, shapes, tabs_np, r.symbols)
        assert np.array_equal(cat.cpu().numpy(), want[0])


def test_lane_waits_for_caller_stream_work_queued_up_to_the_allocation(torch_cuda, oracle, monkeypatch):
    """TF's allocator hands out memory in compute-stream order, and between a request's entry and its
    malloc_buff another Session::Run thread may queue a kernel K that still READS the memory the arena then gets.  Here the
    allocator itself plays that thread: inside malloc_buff it queues — on the caller's stream — a long sleep and then K
    (a copy of the block's old content), and returns the block.  The private stream must wait for K: the event it waits
    for is recorded on the caller's stream AFTER the allocation.  (Recorded before it, the lane writes the block while K is
    still waiting behind the sleep, and K copies the new result instead of the old content.)"""
    from recom_amd import lib, synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    monkeypatch.setenv("FCP_LANE_SUPERVISE", "0")                       # (its baseline would keep the first requests on the caller's stream)
    m = synth.model_mixed(batch=160, vocab=2999, n_groups=1)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    op.plan.set_private_streams(3, always=True, verify=False)           # the lanes as they are: every request takes one
    caller = torch.cuda.Stream()
    for seed in range(3):
        r = m.make_request(900 + seed, B=150 + seed)
        blob, offsets, shapes = concat_inputs(r.inputs)
        d_blob = torch.from_numpy(blob).cuda()
        nbytes = op.plan.arena_bytes(shapes, r.symbols)
        block = torch.full((max(nbytes, 128) + 256,), 0x5A, dtype=torch.uint8, device="cuda")     # "tensor X", still in use
        seen_by_k = torch.zeros_like(block)
        torch.cuda.synchronize()
        state = {"arena": None, "temps": [], "calls": 0}

        def _alloc(_ctx, n, state=state, block=block, seen_by_k=seen_by_k):
            state["calls"] += 1
            assert n <= block.numel()
            with torch.cuda.stream(caller):                             # the other thread's kernel K, queued just before X is "freed"
                _busy(torch, caller, 3.0)
                seen_by_k.copy_(block)
            state["arena"] = block                                       # ... and the allocator hands X out as the arena
            return block.data_ptr()

        def _alloc_temp(_ctx, n, state=state):
            t = torch.empty(int(n), dtype=torch.uint8, device="cuda")
            state["temps"].append(t)
            return t.data_ptr()

        monkeypatch.setattr(op, "_allocators", lambda: (state, lib.ALLOC_FN(_alloc), lib.ALLOC_FN(_alloc_temp)))
        with torch.cuda.stream(caller):
            out = op(d_blob, offsets, shapes, tabs, r.symbols, defer_wait=True)
            out.wait()
            got = out.groups[0].clone()
        caller.synchronize()
        assert state["calls"] == 1
        assert bool((seen_by_k == 0x5A).all()), "the private stream wrote the arena while a kernel queued before the allocation still read it"
        want, _ = oracle.process_feature_columns(m.spec.to_dict(), blob, offsets, shapes, tabs_np, r.symbols)
        assert np.array_equal(got.cpu().numpy(), want[0])
    torch.cuda.synchronize()


def test_supervisor_demotes_a_caller_whose_private_streams_stop_overlapping(torch_cuda, oracle, monkeypatch, capfd):
    """a verdict is learnt once, the hardware-queue mapping can go bad later — and a caller's traffic may
    never gain from the private streams at all.  The supervisor A/Bs the two while serving: 48 requests on the caller's
    stream, 48 on the private streams, time per byte of work compared; two consecutive evaluations that the streams lose
    demote the caller.  FCP_DIAG=lane_fault_us=N makes the lanes behave like a mapping that does not overlap (every lane request
    waits for the device's previous lane request and stalls 60 us: requests serialise at several times their stream-order
    cost); the lanes are used unverified (FCP_PRIVATE_NO_VERIFY) and driven by the native loop.  Demotion must come within
    1000 requests, be logged once, and every result before, at and after it stays bit-exact.  Then the fault goes away:
    a later evaluation finds the streams fine again — or not, on a box whose real mapping does not overlap — and either
    way the caller ends in the mode its own measurements name."""
    from recom_amd import synth
    from recom_amd.harness import ServingHarness
    from recom_amd.ops import FeatureColumnProcess
    torch = torch_cuda
    monkeypatch.delenv("FCP_LANE_SUPERVISE", raising=False)              # (the supervisor on, whatever the environment says)
    monkeypatch.setenv("FCP_DIAG", "lane_fault_us=60,lane_log")
    monkeypatch.setenv("FCP_LANE_SUPERVISE_PERIOD", "256")               # evaluations at request 1, 1 + 256 (+ their windows), ...
    model = synth.model_s2(columns=96, vocab=5000, batch=512)
    h = ServingHarness(model, device=0, n_requests=8, arena_ring=6, n_threads=1)
    tabs_np = model.numpy_tables()
    op = FeatureColumnProcess(model.spec, 0, plan=h.plan)
    caller = h.caller_stream()
    ext = torch.cuda.ExternalStream(caller)

    def check_one(k):
        blob, offsets, shapes = h.packed[k]
        with torch.cuda.stream(ext):
            out = op(h._keep[1 + 4 * k], offsets, shapes, h.tables, h.requests[k].symbols)
            got = out.groups[0].clone()
        ext.synchronize()
        want, _ = oracle.process_feature_columns(model.spec.to_dict(), blob, offsets, shapes, tabs_np, h.requests[k].symbols)
        assert np.array_equal(got.cpu().numpy(), want[0])

    h.plan.set_private_streams(3, always=True, verify=False)
    capfd.readouterr()
    issued, demoted_at = 0, None
    for chunk in range(10):
        h.run_private(100, 3)
        issued += 100
        check_one(chunk % 8)
        issued += 1
        st = h.plan.private_streams_stats()
        assert st["requests"] == issued
        if st["demoted"] and demoted_at is None:
            demoted_at = issued
    st = h.plan.private_streams_stats()
    assert st["demoted"] == 1 and demoted_at is not None and demoted_at <= 1000, st
    assert st["evaluations"] >= 2 and st["worst_ratio"] > 1.5 and st["stream_order_us_per_mib"] > 0, st
    assert 96 <= st["lane_requests"] < st["requests"], st               # two lane windows at least; not everything since
    assert h.plan.private_streams_verdict(caller) == 0
    err = capfd.readouterr().err
    assert err.count("DEMOTED") == 1 and "RE-ADMITTED" not in err, err
    # the fault goes away; evaluations go on (gap back to 256 after the switch): the caller ends where its measurements say
    monkeypatch.setenv("FCP_DIAG", "lane_log")
    h.plan.set_private_streams(3, always=True, verify=False)            # (re-reads the fault; the supervisor starts over)
    st0 = h.plan.private_streams_stats()
    assert (st0["demoted"], st0["requests"], st0["lane_requests"], st0["evaluations"]) == (0, 0, 0, 0), st0
    assert h.plan.private_streams_verdict(caller) == -1
    for chunk in range(8):
        h.run_private(100, 3)
        check_one(chunk % 8)
    st2 = h.plan.private_streams_stats()
    assert st2["evaluations"] >= 2 and st2["requests"] == 808, st2
    # consistency: demoted <=> the last decisions found the private streams losing
    assert h.plan.private_streams_verdict(caller) in ((0,) if st2["demoted"] else (-1, 1))   # (-1: unverified use, never switched)
    print("supervisor without the fault:", st2)
    torch.cuda.synchronize()
    h.close()


def test_supervisor_re_admits_a_demoted_caller_when_the_private_streams_win_again(torch_cuda, monkeypatch, capfd):
    """The other direction: a caller demoted while the lanes were faulty is re-admitted by later evaluations once they beat
    stream order — here made certain by a fault on the OTHER side: FCP_LANE_KEEP_RATIO far above 1 means "keep the private
    streams unless they are several times slower", so with the lane fault gone every evaluation votes for them."""
    from recom_amd import synth
    from recom_amd.harness import ServingHarness
    torch = torch_cuda
    monkeypatch.delenv("FCP_LANE_SUPERVISE", raising=False)
    monkeypatch.setenv("FCP_DIAG", "lane_fault_us=120,lane_log")
    monkeypatch.setenv("FCP_LANE_SUPERVISE_PERIOD", "192")
    monkeypatch.setenv("FCP_LANE_KEEP_RATIO", "3.0")
    model = synth.model_s2(columns=96, vocab=5000, batch=512)
    h = ServingHarness(model, device=0, n_requests=8, arena_ring=6, n_threads=1)
    caller = h.caller_stream()
    h.plan.set_private_streams(3, always=True, verify=False)
    capfd.readouterr()
    for _ in range(10):
        h.run_private(100, 3)
    st = h.plan.private_streams_stats()
    assert st["demoted"] == 1 and st["worst_ratio"] > 3.0, st             # 120 us of stall per request against ~10 us in stream order
    # the fault is a property of the device's lane pool, read when the mode is set: another plan on the device clears it
    monkeypatch.setenv("FCP_DIAG", "lane_log")
    other = ServingHarness(model, device=0, n_requests=2, arena_ring=2, n_threads=1, tables=h.tables)
    other.plan.set_private_streams(3, always=True, verify=False)
    for _ in range(12):
        h.run_private(100, 3)
    st = h.plan.private_streams_stats()
    assert st["demoted"] == 0 and h.plan.private_streams_verdict(caller) == 1, st
    err = capfd.readouterr().err
    assert err.count("DEMOTED") == 1 and err.count("RE-ADMITTED") == 1, err
    torch.cuda.synchronize()
    other.close()
    h.close()


def test_verification_at_warm_up_and_its_wall_time_budget(torch_cuda, monkeypatch):
    """fcp_plan_verify_private_streams runs the search at a time of the caller's choosing (the shim: its
    first Compute), bounded by wall time; a request after it only looks the verdict up.  Below the work threshold nothing is
    probed at all."""
    import time
    from recom_amd import synth
    from recom_amd.harness import ServingHarness
    torch = torch_cuda
    model = synth.model_s2(columns=96, vocab=5000, batch=512)
    h = ServingHarness(model, device=0, n_requests=4, arena_ring=6, n_threads=1)
    caller = h.caller_stream()
    h.run(4)                                                            # the warm-up request: the plan has seen its shapes
    h.plan.set_private_streams(3)                                       # default threshold 48 MiB: this model is far below
    t0 = time.perf_counter()
    assert h.plan.verify_private_streams(caller, 300) == -1             # nothing to verify: requests stay on the caller's stream
    assert time.perf_counter() - t0 < 0.25                              # (no search ran: far below its 300-ms budget; loose, a shared box)
    h.plan.set_private_streams(3, always=True)
    t0 = time.perf_counter()
    v = h.plan.verify_private_streams(caller, 60)                       # 60 ms of search at most (+ the probe in flight)
    took = time.perf_counter() - t0
    assert v in (0, 1) and took < 0.060 + 0.5, (v, took)                # the budget + the probe in flight + slack for a loaded box (an unbounded search takes seconds)
    assert h.plan.private_streams_verdict(caller) == v
    t0 = time.perf_counter()
    h.run_private(50, 3)                                                # requests after it: no probe, whatever the verdict
    assert h.plan.private_streams_verdict(caller) == v                  # (the verdict stands; timing only as a loose guard)
    assert time.perf_counter() - t0 < 0.5
    if v == 0:                                                          # a negative verdict is forgotten and searched again on request
        v2 = h.plan.verify_private_streams(caller, 400)
        assert v2 in (0, 1) and h.plan.private_streams_verdict(caller) == v2
    torch.cuda.synchronize()
    h.close()


def test_concat_outputs_waits_for_every_arena_its_inputs_come_from(torch_cuda, oracle, monkeypatch):
    """fcp_concat_outputs waited only for the arena of inputs[0], the scatter variants for none.  Here the
    columns of TWO FeatureColumnProcess ops (per-column layout, both on private streams, both kept busy behind a sleeping
    caller stream) are interleaved into one matrix by fcp_concat_outputs and by fcp_concat_outputs_scatter on the caller's
    stream, with NO explicit wait by the caller: both must order themselves behind both arenas.  A host reader through
    fcp_result_synchronize (which no longer holds the registry's lock while it waits) sees complete results too."""
    import dataclasses
    from recom_amd import lib, synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs, concat_outputs, concat_outputs_scatter
    from recom_amd.plan import LAYOUT_PER_COLUMN
    torch = torch_cuda
    monkeypatch.setenv("FCP_LANE_SUPERVISE", "0")
    models = [synth.model_mixed(batch=130, vocab=1999, n_groups=1), synth.model_mixed(batch=130, vocab=2999, n_groups=1, seed_dims=(8, 4, 16, 12, 20, 32, 64))]
    ops, tabs, tabs_np, specs = [], [], [], []
    for m in models:
        spec = dataclasses.replace(m.spec, layout=LAYOUT_PER_COLUMN)
        op = FeatureColumnProcess(spec, 0)
        op.plan.set_private_streams(3, always=True, verify=False)
        ops.append(op)
        specs.append(spec)
        tabs_np.append(m.numpy_tables())
        tabs.append([torch.from_numpy(t).cuda() for t in tabs_np[-1]])
    s = torch.cuda.Stream()
    for seed in range(3):
        reqs = [m.make_request(40 + seed, B=130) for m in models]
        packed = [concat_inputs(r.inputs) for r in reqs]
        blobs = [torch.from_numpy(p[0]).cuda() for p in packed]
        wants = [oracle.process_feature_columns(m.spec.to_dict(), *p, tn, r.symbols)[0][0] for m, p, tn, r in zip(models, packed, tabs_np, reqs)]
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            _busy(torch, s, 2.0)                                       # the lanes wait for this; the caller's stream is behind the host
            outs = [op(b, p[1], p[2], t, r.symbols, defer_wait=True) for op, b, p, t, r in zip(ops, blobs, packed, tabs, reqs)]
            cols, want_cols = [], []
            for which, (out, spec, want) in enumerate(zip(outs, specs, wants)):
                offs = spec.column_offsets()
                for k in sorted(range(len(spec.columns)), key=lambda k: spec.columns[k].concat_slot):
                    cols.append((which, out.column(k)))
                    want_cols.append(want[:, offs[k]:offs[k] + spec.columns[k].dim])
            order = list(range(len(cols)))
            order = order[1::2] + order[0::2]                          # inputs[0] belongs to one arena, most others to the other
            cat = concat_outputs([cols[i][1] for i in order])          # no fcp_result_wait by the caller
            width = int(cat.shape[1])
            sc = torch.full((130, width), float("nan"), device="cuda")
            col_offs = np.cumsum([0] + [int(cols[i][1].shape[1]) for i in order])[:-1]
            concat_outputs_scatter([cols[i][1] for i in order], col_offs, sc)
        s.synchronize()
        want_cat = np.concatenate([want_cols[i] for i in order], axis=1)
        assert np.array_equal(cat.cpu().numpy(), want_cat)
        assert np.array_equal(sc.cpu().numpy(), want_cat)
        # a host reader
        with torch.cuda.stream(s):
            _busy(torch, s, 1.0)
            out = ops[0](blobs[0], packed[0][1], packed[0][2], tabs[0], reqs[0].symbols, defer_wait=True)
        lib.check(lib.load().fcp_result_synchronize(out.buffer.data_ptr()), "fcp_result_synchronize")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            got = torch.cat([out.column(k).contiguous() for k in sorted(range(len(specs[0].columns)), key=lambda k: specs[0].columns[k].concat_slot)], dim=1).cpu()
        side.synchronize()
        assert np.array_equal(got.numpy(), wants[0])
    torch.cuda.synchronize()

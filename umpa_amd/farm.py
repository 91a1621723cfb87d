"""
Step-scan farm: many independent projections matched against resident reference stacks ("replicas only": no
collective).

This is the GPU shape of the reference's batch script ``UMPA/umpa_multi.py``: a pool of worker processes takes
projections from a queue, picks the nearest reference (``:133``), flat-corrects ``(proj - dark) / flat[refnum]``
(``:144``), builds ``UMPAModelDF(sam, ref, ...).match()`` (``:149-150``) and hands the result to a writer
(``:163-190``).  Here

  * ``StreamingMatcher`` is one GPU's pipeline: the reference stacks, flats and the dark frame stay in HBM; a
    projection is uploaded as the detector delivered it (uint16 / float32 / float64) and flat-corrected by a small
    kernel on the way into the model's sample buffer (``umpa_hip_stage_sample``); the upload of projection p+1 and the
    download of the maps of projection p-1 overlap the match of projection p (double-buffered sample stack, page-locked
    host buffers, three HIP streams); the reference stack is switched only when the nearest reference changes.
  * ``ProjectionFarm`` runs one ``StreamingMatcher`` per GPU in worker processes.  Projections and result maps travel
    through shared-memory slots that the workers page-lock once, so nothing is pickled and the PCIe transfers run at
    link rate; ``save_pattern`` also writes every result as an ``.npz`` file (``umpa_multi.py:185``).

    farm = ProjectionFarm(refs, window_size=5, max_shift=5, devices=[0, 1, 2, 3], flats=flats, dark=dark, ref_nums=ref_nums)
    for pid, res in farm.map((pid, raw_stack) for ...):      # res: dict of arrays (copied out of the result slot)
        ...
    farm.close()

The loading / unwarping steps of the reference script are the caller's business (outside the matching path).
"""
import multiprocessing as mp
import os
import queue as _queue
import time

import numpy as np

__all__ = ["StreamingMatcher", "ProjectionFarm", "bench_c5"]

RESULT_KEYS_DF = ("f", "T", "dx", "dy", "df", "err")


def nearest_reference(proj_num, ref_nums):
    """``umpa_multi.py:133``: index of the reference acquired closest to this projection."""
    return int(np.argmin(np.abs(float(proj_num) - np.asarray(ref_nums, dtype=np.float64))))


class StreamingMatcher:
    """One GPU's projection pipeline (see the module docstring).

    refs   [R, K, H, W] (or [K, H, W]) float64 reference stacks
    flats  [R, K, H, W] float64 or None;  dark [K, H, W] / [H, W] float64 or None;  ref_nums: acquisition numbers of
           the R references (``umpa_multi.py`` ``ref_nums``), default 0..R-1
    """

    def __init__(self, refs, window_size, max_shift=4, df=True, device=0, flats=None, dark=None, ref_nums=None,
                 model_cls=None, debug=False):
        import torch
        from . import model as _model
        refs = np.asarray(refs, dtype=np.float64)
        if refs.ndim == 3:
            refs = refs[None]
        self.refs = np.ascontiguousarray(refs)
        self.R, self.K, self.H, self.W = self.refs.shape
        self.ref_nums = list(range(self.R)) if ref_nums is None else list(ref_nums)
        self.device = int(device)
        self._torch = torch
        self._dev = torch.device("cuda", self.device)
        self._flat = None
        self._dark = None
        if flats is not None:
            flats = np.asarray(flats, dtype=np.float64)
            if flats.ndim == 3:
                flats = flats[None]
            self._flat = torch.from_numpy(np.ascontiguousarray(flats)).to(self._dev)          # resident: R x K frames
        if dark is not None:
            dark = np.asarray(dark, dtype=np.float64)
            dark = np.array(np.broadcast_to(dark, (self.K, self.H, self.W)))
            self._dark = torch.from_numpy(dark).to(self._dev)
        cls = model_cls or (_model.UMPAModelDF if df else _model.UMPAModelNoDF)
        self._refnum = 0
        # the model owns device copies; the sample stack is replaced by every staged projection
        self.model = cls(np.zeros_like(self.refs[0]), self.refs[0], window_size=window_size, max_shift=max_shift, device=self.device)
        self.model.debug = debug
        self.out_alloc = None              # optional hook: (shape, dtype, zero) -> page-locked array (ProjectionFarm's result slots)

    # -- page-locked input buffers for producers that can write straight into them
    def input_buffer(self, dtype=np.float64):
        from . import _lib
        return _lib.pinned_empty((self.K, self.H, self.W), dtype)

    def _stage(self, proj_num, raw):
        refnum = nearest_reference(proj_num, self.ref_nums)
        flat = list(self._flat[refnum]) if self._flat is not None else None
        dark = list(self._dark) if self._dark is not None else None
        self.model.stage_sample(list(raw), dark=dark, flat=flat)
        return refnum

    def _switch_reference(self, refnum):
        if refnum != self._refnum:                                  # umpa_multi.py:145: ref = spiral_ref_arr[refnum]
            self.model.wait()
            self.model.update_frames(ref_list=self.refs[refnum])
            self._refnum = refnum

    def run(self, items, **match_kw):
        """``items`` yields ``(proj_num, raw_stack)``; yields ``(proj_num, result)`` in the same order.  Two matches are
        in flight: while the maps of projection p travel to the host, projection p+1 is being matched (into the library's
        other device output set) and projection p+2 is on its way to the GPU."""
        from collections import deque
        match_kw.setdefault("quiet", True)
        it = iter(items)
        pending = deque()
        # The library keeps a FIFO of asynchronous matches and `wait` returns the OLDEST one.  If the consumer abandons
        # this generator (break, close), or staging / a match raises mid-series, the matches still in flight are waited for
        # here: a later run() on the same matcher must not pair its waits with this series' matches.
        try:
            cur = next(it, None)
            cur_ref = self._stage(*cur) if cur is not None else None
            while cur is not None or pending:
                if cur is not None:
                    if cur_ref != self._refnum:                     # a reference switch needs an idle model: hand out what is in flight
                        while pending:
                            pid, res = pending[0]
                            self.model.wait()
                            pending.popleft()
                            yield pid, res
                    self._switch_reference(cur_ref)
                    self.model._out_alloc = self.out_alloc
                    res = self.model.match_async(**match_kw)        # adopts the staged stack, enqueues kernels + downloads
                    pending.append((cur[0], res))
                    nxt = next(it, None)
                    nxt_ref = self._stage(*nxt) if nxt is not None else None   # upload the next one while this one is being matched
                    cur, cur_ref = nxt, nxt_ref
                if len(pending) == 2 or cur is None:
                    pid, res = pending[0]
                    self.model.wait()                               # the OLDEST match in flight
                    pending.popleft()
                    yield pid, res
        finally:
            while pending:                                          # nothing of this series stays in the library's queue
                pending.popleft()
                try:
                    self.model.wait()
                except Exception:
                    pass


# ------------------------------------------------------------------------------------------------
# worker processes
# ------------------------------------------------------------------------------------------------
class _Slots:
    """A ring of equally sized shared-memory slots."""

    def __init__(self, ctx, count, nbytes, name=None):
        from multiprocessing import shared_memory
        self.count, self.nbytes = count, int(nbytes)
        if name is None:
            self.shm = shared_memory.SharedMemory(create=True, size=max(1, self.count * self.nbytes))
            self.owner = True
        else:
            self.shm = shared_memory.SharedMemory(name=name)
            self.owner = False
        self.name = self.shm.name

    def view(self, q, shape, dtype, offset=0):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        start = q * self.nbytes + offset
        return np.frombuffer(self.shm.buf, dtype=dtype, count=int(np.prod(shape)), offset=start).reshape(shape), offset + ((n + 63) & ~63)

    def close(self):
        if self.owner:                                              # the name goes first: views of the mapping may still be alive
            try:
                self.shm.unlink()
            except Exception:
                pass
        try:
            self.shm.close()
        except Exception:                                           # BufferError while numpy views exist: the mapping dies with them
            pass


def _result_layout(N0, N1, df):
    """(key, shape, dtype) of the arrays in a result slot, in the order model.match allocates them."""
    nparam = 5 if df else 4
    return [("values", (nparam, N0, N1), np.float64), ("err", (N0, N1), np.int32)]


def _worker_main(body, device, cfg, tasks, results, in_name, out_name):
    """What every worker process runs: ``body`` (the GPU worker below, or a stand-in a caller injected with
    ``ProjectionFarm(worker=...)``) inside the farm's error protocol -- an exception becomes an "error" message, the process
    always says "exit"."""
    try:
        body(device, cfg, tasks, results, in_name, out_name)
    except BaseException as e:                                      # noqa: BLE001 -- the parent must hear about everything
        import traceback
        results.put(("error", None, None, None, repr(e) + "\n" + traceback.format_exc()))
    finally:
        results.put(("exit", device, None, None, None))


def _farm_worker(device, cfg, tasks, results, in_name, out_name):
    """The GPU worker: one StreamingMatcher on `device`, projections and maps through the shared-memory slots."""
    if device is not None:
        os.environ["UMPA_HIP_DEVICE"] = str(device)
    import ctypes
    from . import _lib
    lib = _lib.hip()
    sm = StreamingMatcher(cfg["refs"], cfg["window_size"], cfg["max_shift"], df=cfg["df"], device=device or 0,
                          flats=cfg["flats"], dark=cfg["dark"], ref_nums=cfg["ref_nums"])
    slots_in = _Slots(None, cfg["depth_in"], cfg["in_bytes"], name=in_name)
    slots_out = _Slots(None, cfg["depth_out"], cfg["out_bytes"], name=out_name)
    # page-lock both rings once: uploads and downloads then run as DMA at link rate, straight from / into the
    # memory the parent process sees
    base_in = ctypes.addressof(ctypes.c_char.from_buffer(slots_in.shm.buf))
    base_out = ctypes.addressof(ctypes.c_char.from_buffer(slots_out.shm.buf))
    lib.check(lib.host_register(base_in, slots_in.count * slots_in.nbytes), "host_register")
    lib.check(lib.host_register(base_out, slots_out.count * slots_out.nbytes), "host_register")
    shape, dtype = (sm.K, sm.H, sm.W), np.dtype(cfg["raw_dtype"])
    state = {"slot": 0, "off": 0}

    def out_alloc(shp, dt, zero=False):                         # model.match's result arrays live in the result slot
        a, state["off"] = slots_out.view(state["slot"], shp, dt, state["off"])
        if zero:
            a[...] = 0
        return a

    def stage(task):
        seq, pid, q_in, q_out, kw = task
        raw, _ = slots_in.view(q_in, shape, dtype)
        return sm._stage(pid, raw)

    def stage_or_report(task):
        """(reference number, None) of a staged task, or (None, message) when its upload failed"""
        try:
            return stage(task), None
        except Exception as e:                                  # this projection only: the worker goes on
            import traceback
            return None, repr(e) + "\n" + traceback.format_exc()

    cur = tasks.get()
    cur_ref, cur_err = stage_or_report(cur) if cur is not None else (None, None)
    while cur is not None:
        seq, pid, q_in, q_out, kw = cur
        matched = False
        if cur_err is None:
            try:
                sm._switch_reference(cur_ref)
                state["slot"], state["off"] = q_out, 0
                sm.model._out_alloc = out_alloc
                kw = dict(kw)
                kw.setdefault("quiet", True)
                sm.model.match_async(**kw)                      # adopts the staged stack; kernels + downloads enqueued
                matched = True
            except Exception as e:
                import traceback
                cur_err = repr(e) + "\n" + traceback.format_exc()
        nxt, nxt_ref, nxt_err, have_next = None, None, None, False
        try:
            nxt = tasks.get_nowait()                            # upload the next projection while this one is matched
            have_next = True
            if nxt is not None:
                nxt_ref, nxt_err = stage_or_report(nxt)
        except _queue.Empty:
            pass
        if matched:
            try:
                sm.model.wait()
            except Exception as e:
                import traceback
                cur_err = repr(e) + "\n" + traceback.format_exc()
        results.put(("done", seq, q_in, q_out, cur_err))
        if not have_next:
            nxt = tasks.get()
            nxt_ref, nxt_err = stage_or_report(nxt) if nxt is not None else (None, None)
        cur, cur_ref, cur_err = nxt, nxt_ref, nxt_err


class ProjectionFailed(RuntimeError):
    """One projection failed in its worker (the farm itself is intact)."""


class ProjectionFarm:
    def __init__(self, ref_stack, window_size, max_shift=4, df=True, devices=None, flats=None, dark=None, ref_nums=None,
                 raw_dtype=np.float64, depth=2, worker=None, save_pattern=None):
        """``devices``: HIP device indices, one worker process each (default: all visible devices).
        ``flats`` / ``dark`` / ``ref_nums``: the flat-field data of ``umpa_multi.py:133-145`` (optional).
        ``raw_dtype``: dtype of the projections as submitted (float64, float32 or uint16).
        ``depth``: shared-memory slots per worker for inputs and for results.
        ``worker``: a picklable callable ``(device, cfg, tasks, results, in_name, out_name)`` to run in the worker processes
        instead of the GPU worker (same queue / slot protocol).  The package itself has no other worker; the CPU tests
        inject one of their own (tests/farm_cpu_worker.py) to exercise the farm's plumbing on machines without a GPU.
        ``save_pattern``: e.g. ``"/data/out/umpa_%04d.npz"``: results are also written there (``umpa_multi.py:185``)."""
        if devices is None:
            from . import _lib
            devices = list(range(max(1, _lib.hip().device_count())))
        self._ctx = mp.get_context("spawn")                         # never fork a process that has touched the GPU
        refs = np.asarray(ref_stack, dtype=np.float64)
        r4 = refs if refs.ndim == 4 else refs[None]
        self.K, self.H, self.W = r4.shape[1:]
        P = int(window_size) + int(max_shift)
        self.N0, self.N1 = self.H - 2 * P, self.W - 2 * P
        self.df = bool(df)
        self.raw_dtype = np.dtype(raw_dtype)
        self.save_pattern = save_pattern
        in_bytes = (self.K * self.H * self.W * self.raw_dtype.itemsize + 4095) & ~4095
        out_bytes = sum(((int(np.prod(s)) * np.dtype(d).itemsize + 63) & ~63) for _, s, d in _result_layout(self.N0, self.N1, df))
        out_bytes = (out_bytes + 4095) & ~4095
        cfg = dict(refs=refs, window_size=int(window_size), max_shift=int(max_shift), df=self.df, flats=flats, dark=dark,
                   ref_nums=ref_nums, raw_dtype=self.raw_dtype.str, depth_in=depth, depth_out=depth, in_bytes=in_bytes,
                   out_bytes=out_bytes)
        self._workers = []
        self._results = self._ctx.Queue()
        for d in devices:
            w = dict(device=d, tasks=self._ctx.Queue(), slots_in=_Slots(self._ctx, depth, in_bytes),
                     slots_out=_Slots(self._ctx, depth, out_bytes), free_in=list(range(depth)), free_out=list(range(depth)),
                     inflight=0)
            w["proc"] = self._ctx.Process(target=_worker_main, daemon=True,
                                          args=(worker or _farm_worker, d, cfg, w["tasks"], self._results, w["slots_in"].name, w["slots_out"].name))
            w["proc"].start()
            self._workers.append(w)
        self._by_seq = {}                                           # in flight: internal sequence number -> (worker, caller's id)
        self._seq = 0
        self._alive = len(self._workers)

    # -- submission
    def _pick_worker(self):
        ready = [w for w in self._workers if w["free_in"] and w["free_out"] and w["proc"].is_alive()]
        return min(ready, key=lambda w: w["inflight"]) if ready else None

    def input_buffer(self):
        """A free input slot as ``(handle, array [K, H, W] of raw_dtype)``: fill the array, then ``submit(pid, handle)``.
        Returns None while every slot is in flight."""
        w = self._pick_worker()
        if w is None:
            return None
        q = w["free_in"].pop()
        arr, _ = w["slots_in"].view(q, (self.K, self.H, self.W), self.raw_dtype)
        return (w, q), arr

    def submit(self, pid, handle, **match_kw):
        """Queue the projection in input slot ``handle`` under the caller's id ``pid`` (ids may repeat).  The result slots
        hold full-extent maps: ``ROI`` / ``step`` are not accepted here (match a sub-region with a model of your own)."""
        bad = [k for k in match_kw if k in ("ROI", "step")]
        if bad:
            w, q_in = handle
            w["free_in"].append(q_in)
            raise ValueError("ProjectionFarm results are full-extent maps: %s is not supported" % ", ".join(bad))
        w, q_in = handle
        q_out = w["free_out"].pop()
        w["inflight"] += 1
        self._seq += 1
        self._by_seq[self._seq] = (w, pid)
        w["tasks"].put((self._seq, pid, q_in, q_out, match_kw))

    # -- collection
    def _collect(self, timeout):
        """One finished projection as ``(pid, result dict of views into the result slot, release())`` or None."""
        deadline = time.time() + timeout
        while True:
            try:
                kind, pid, q_in, q_out, err = self._results.get(timeout=1.0)
            except _queue.Empty:
                dead = [w for w in self._workers if w["inflight"] and not w["proc"].is_alive()]
                if dead:
                    raise RuntimeError("farm worker for device %r died with %d projections in flight (exit code %r)" % (
                        dead[0]["device"], dead[0]["inflight"], dead[0]["proc"].exitcode))
                if time.time() > deadline:
                    raise RuntimeError("no result from the farm workers for %.0f s" % timeout)
                continue
            if kind == "exit":
                self._alive -= 1
                if self._alive == 0 and any(w["inflight"] for w in self._workers):
                    raise RuntimeError("all farm workers exited with projections in flight")
                continue
            if kind == "error":
                raise RuntimeError("farm worker failed: %s" % err)
            w, pid = self._by_seq.pop(pid)                          # (the second field of a "done" message is the sequence number)
            w["free_in"].append(q_in)
            if err is not None:                                     # this projection failed in its worker; the farm goes on
                w["free_out"].append(q_out)
                w["inflight"] -= 1
                raise ProjectionFailed("projection %r failed in the farm worker for device %r: %s" % (pid, w["device"], err))
            off = 0
            res = {}
            for key, shape, dtype in _result_layout(self.N0, self.N1, self.df):
                res[key], off = w["slots_out"].view(q_out, shape, dtype, off)
            vals = res.pop("values")
            res["f"], res["T"], res["dx"], res["dy"] = vals[0], vals[1], vals[2], vals[3]
            if self.df:
                res["df"] = vals[4]

            def release(w=w, q_out=q_out):
                w["free_out"].append(q_out)
                w["inflight"] -= 1
            return pid, res, release

    def _raise_if_dead(self, timeout):
        """Called when no input slot can be had and nothing is in flight: drain the workers' messages (raising on 'error'
        and once every worker has exited); blocks for at most a second."""
        try:
            kind, _a, _b, _c, err = self._results.get(timeout=1.0)
        except _queue.Empty:
            if not any(w["proc"].is_alive() for w in self._workers):
                raise RuntimeError("all farm workers have exited")
            return
        if kind == "error":
            raise RuntimeError("farm worker failed: %s" % err)
        if kind == "exit":
            self._alive -= 1
            if self._alive <= 0:
                raise RuntimeError("all farm workers have exited")

    def map(self, projections, timeout=600.0, **match_kw):
        """Yield ``(id, result dict)`` as projections complete (not in submission order).  Convenience form: every
        projection is copied into an input slot and every result out of its slot; producers that can write into
        ``input_buffer()`` arrays and consumers that can work on the slot views (``_collect``) avoid both copies."""
        it = iter(projections)
        exhausted = False
        inflight = 0
        failed = None
        while not exhausted or inflight:
            while not exhausted:
                got = self.input_buffer()
                if got is None:
                    break
                try:
                    pid, sam = next(it)
                except StopIteration:
                    exhausted = True
                    w, q = got[0]
                    w["free_in"].append(q)
                    break
                handle, arr = got
                arr[...] = sam
                self.submit(pid, handle, **match_kw)
                inflight += 1
            if not inflight and not exhausted:
                # no free slot although nothing is in flight: no worker is alive (each posts 'error' / 'exit' when it
                # dies, e.g. in its constructor): surface that instead of spinning
                self._raise_if_dead(timeout)
                continue
            if inflight:
                try:
                    pid, res, release = self._collect(timeout)
                except ProjectionFailed as e:
                    # this projection is lost, the ones still in flight are not: nothing new is submitted, what is in
                    # flight is collected (and handed out), then the first failure is raised -- a later map() on this farm
                    # finds no stale results in the queue
                    failed = failed or e
                    inflight -= 1
                    exhausted = True
                    continue
                out = {k: np.array(v) for k, v in res.items()}
                if self.save_pattern:
                    np.savez(self.save_pattern % pid, **out)
                release()
                inflight -= 1
                yield pid, out
        if failed is not None:
            raise failed

    def close(self):
        for w in self._workers:
            try:
                w["tasks"].put(None)
            except Exception:
                pass
        for w in self._workers:
            w["proc"].join(timeout=30)
            if w["proc"].is_alive():
                w["proc"].terminate()
            w["slots_in"].close()
            w["slots_out"].close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ------------------------------------------------------------------------------------------------
# bench.py --config C5
# ------------------------------------------------------------------------------------------------
def bench_c5(device=0, steps=1, warmup=1, n_proj=32, raw_dtype=np.uint16, keep=None):
    """BASELINE config C5 on one GPU: 32 projections of 2048 x 2048 x 5 frames (Nw=5, max_shift=5, dark-field on) stream
    through one ``StreamingMatcher``: detector counts (uint16) in page-locked host memory -> upload + flat correction ->
    match -> result maps back in page-locked host memory.  A "step" is the whole series; the rate includes every
    transfer.  Two reference stacks, the nearest one per projection (``umpa_multi.py:133``).
    ``keep``: a projection number whose corrected sample stack, reference stack and result maps are handed back beside
    the JSON line (``(line, extras)``), for the caller's parity check of that projection."""
    from .synth import CONFIGS, make_stack
    cfg = CONFIGS["C5"]
    H, W, K, Nw, ms = cfg["H"], cfg["W"], cfg["K"], cfg["Nw"], cfg["max_shift"]
    t0 = time.time()
    # two references (acquired at projection 0 and 31), a dark frame and flats; raw = counts such that
    # (raw - dark) / flat = the synthetic sample stack
    sam0, ref0, _ = make_stack(H, W, K, ms, df=True, seed=0, order=1)
    sam1, ref1, _ = make_stack(H, W, K, ms, df=True, seed=100, order=1)
    refs = np.stack([ref0, ref1])
    rng = np.random.default_rng(5)
    dark = 100.0 + rng.uniform(0, 2, size=(K, H, W))
    flats = 20000.0 * (1.0 + 0.05 * rng.standard_normal((2, K, H, W)))
    ref_nums = [0, n_proj - 1]
    sm = StreamingMatcher(refs, Nw, ms, df=True, device=device, flats=flats, dark=dark, ref_nums=ref_nums)
    bufs = []
    for p in range(n_proj):
        refnum = nearest_reference(p, ref_nums)
        base = sam0 if refnum == 0 else sam1
        raw = np.rint(base * (1.0 - 0.004 * p) * flats[refnum] + dark)               # a different stack per projection
        b = sm.input_buffer(raw_dtype)
        b[...] = raw.astype(raw_dtype)
        bufs.append(b)
    t_gen = time.time() - t0
    N0, N1 = sm.model.extent
    ok = 0.0

    kept = {}

    def series(keep_now=False):
        nonlocal ok
        n, last = 0, None
        for pid, res in sm.run(((p, bufs[p]) for p in range(n_proj))):
            n += 1
            last = res                                              # the consumer of this bench only counts
            if keep_now and pid == keep:
                kept["res"] = {k: np.array(v) for k, v in res.items() if isinstance(v, np.ndarray)}
        ok = float(last["err"].mean())
        return n

    for _ in range(warmup):
        series()
    t0 = time.perf_counter()
    for _ in range(steps):
        done = series()
    dt = (time.perf_counter() - t0) / steps
    assert done == n_proj
    if keep is not None:
        series(keep_now=True)                                       # one more pass, untimed: the 181 MB copy of the kept maps stays out of the rate
    in_bytes = K * H * W * np.dtype(raw_dtype).itemsize
    out_bytes = N0 * N1 * (5 * 8 + 4)
    line = {
        "metric": "Mpixels/s (output map) at Nw=%d, max_shift=%d, %d frames" % (Nw, ms, K),
        "value": round(n_proj * N0 * N1 / dt / 1e6, 3), "unit": "Mpx/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C5: step-scan series of %d projections x %dx%d, %d frames, Nw=%d, max_shift=%d, dark-field on; "
                               "%s counts uploaded + flat-corrected + matched + maps downloaded, transfers included" % (
                                   n_proj, H, W, K, Nw, ms, np.dtype(raw_dtype).name),
                   "ms_per_projection": round(dt * 1e3 / n_proj, 3), "upload_bytes_per_projection": in_bytes,
                   "download_bytes_per_projection": out_bytes, "references": 2, "err_ok_fraction_last": round(ok, 5),
                   "pcie_floor_ms_per_projection": round(max(in_bytes, out_bytes) / 63e9 * 1e3, 3),
                   "input_generation_s": round(t_gen, 1)},
        "roofline": None, "cpu_baseline": None,
    }
    if keep is None:
        return line
    refnum = nearest_reference(keep, ref_nums)
    kept["sam"] = (bufs[keep].astype(np.float64) - dark) / flats[refnum]      # what umpa_multi.py:144 matches
    kept["ref"] = refs[refnum]
    kept["Nw"], kept["ms"] = Nw, ms
    return line, kept

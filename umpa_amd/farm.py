"""
Projection farm: many independent projections matched against one resident reference stack, one
worker process per GPU ("replicas only": no collective).

This is the GPU shape of the reference's batch script: a pool of worker processes, each building
``UMPAModelDF(sam, ref, ...).match()`` per projection from a task queue, with the reference frames
shared (``UMPA/umpa_multi.py:100-160, 261-270``).  Differences: the reference stack is uploaded to
each GPU once and stays in HBM (``update_frames`` swaps only the sample stack); the loading / unwarping
/ flat-field steps of that script are the caller's business (they are outside the matching path).

    farm = ProjectionFarm(ref, window_size=5, max_shift=5, devices=[0, 1, 2, 3])
    for pid, res in farm.map((pid, sam_stack) for pid, sam_stack in projections):
        np.savez(..., **res)
    farm.close()
"""
import multiprocessing as mp
import os

import numpy as np

__all__ = ["ProjectionFarm"]


def _worker(device, ref, kw, model_path, tasks, results):
    try:
        if device is not None:
            os.environ["UMPA_HIP_DEVICE"] = str(device)
        mod_name, cls_name = model_path
        import importlib
        ns = importlib.import_module(mod_name)
        for part in cls_name.split("."):
            ns = getattr(ns, part)
        cls, model = ns, None
        while True:
            item = tasks.get()
            if item is None:
                break
            pid, sam, match_kw = item
            try:
                if model is None:
                    model = cls(sam, ref, **kw)
                    model.debug = False
                elif hasattr(model, "update_frames") and model._lib.is_hip:
                    model.update_frames(sam_list=sam)
                    model.ROI = None
                else:
                    model = cls(sam, ref, **kw)
                    model.debug = False
                res = model.match(quiet=True, **match_kw)
                results.put((pid, res, None))
            except Exception as e:                      # one bad projection must not stop the farm
                results.put((pid, None, repr(e)))
    finally:
        results.put(("__exit__", device, None))


class ProjectionFarm:
    def __init__(self, ref_stack, window_size, max_shift=4, df=True, devices=None,
                 model=("umpa_amd.model", None), queue_depth=2):
        """``devices``: HIP device indices, one worker each (default: all visible devices).
        ``model``: (module, class path) of the model class -- the tests point it at the CPU checker."""
        if devices is None:
            from . import _lib
            devices = list(range(max(1, _lib.hip().device_count())))
        mod, cls = model
        cls = cls or ("UMPAModelDF" if df else "UMPAModelNoDF")
        self._ctx = mp.get_context("spawn")             # never fork a process that has touched the GPU
        self._tasks = self._ctx.Queue(maxsize=queue_depth * len(devices))
        self._results = self._ctx.Queue()
        ref = np.ascontiguousarray(ref_stack, dtype=np.float64)
        kw = dict(window_size=window_size, max_shift=max_shift)
        self._procs = [self._ctx.Process(target=_worker, args=(d, ref, kw, (mod, cls), self._tasks, self._results),
                                         daemon=True) for d in devices]
        for p in self._procs:
            p.start()
        self._alive = len(self._procs)

    def map(self, projections, **match_kw):
        """Yield ``(id, result_dict)`` as projections complete (not in submission order)."""
        pending = 0
        it = iter(projections)
        exhausted = False
        while not exhausted or pending:
            while not exhausted and pending < self._tasks._maxsize:
                try:
                    pid, sam = next(it)
                except StopIteration:
                    exhausted = True
                    break
                self._tasks.put((pid, np.ascontiguousarray(sam, dtype=np.float64), match_kw))
                pending += 1
            if pending:
                pid, res, err = self._results.get()
                if pid == "__exit__":
                    self._alive -= 1
                    if self._alive == 0:
                        raise RuntimeError("all farm workers exited")
                    continue
                pending -= 1
                if err is not None:
                    raise RuntimeError("projection %r failed: %s" % (pid, err))
                yield pid, res

    def close(self):
        for _ in self._procs:
            self._tasks.put(None)
        for p in self._procs:
            p.join(timeout=30)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

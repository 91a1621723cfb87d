"""A stand-in worker for umpa_amd.farm.ProjectionFarm(worker=...): the farm's queue / shared-memory protocol served by a CPU
model class (the checker under oracle/), so that the farm's plumbing -- slots, ids, failure modes, flat-field correction,
nearest reference -- is tested on machines without a GPU.  Test infrastructure: the package has no CPU worker."""
import numpy as np

from umpa_amd.farm import RESULT_KEYS_DF, _Slots, nearest_reference


def run(device, cfg, tasks, results, in_name, out_name, model=("oracle.cpu_model", "port.UMPAModelDF"), fail_pids=()):
    import importlib
    mod_name, cls_name = model
    ns = importlib.import_module(mod_name)
    for part in cls_name.split("."):
        ns = getattr(ns, part)
    slots_in = _Slots(None, cfg["depth_in"], cfg["in_bytes"], name=in_name)
    slots_out = _Slots(None, cfg["depth_out"], cfg["out_bytes"], name=out_name)
    refs = np.asarray(cfg["refs"], dtype=np.float64)
    if refs.ndim == 3:
        refs = refs[None]
    K, H, W = refs.shape[1:]
    while True:
        item = tasks.get()
        if item is None:
            return
        seq, pid, q_in, q_out, match_kw = item
        if pid in fail_pids:                                        # a projection that fails in its worker (the GPU worker reports
            results.put(("done", seq, q_in, q_out, "injected failure of projection %r" % (pid,)))   # an upload / match error this way)
            continue
        raw, _ = slots_in.view(q_in, (K, H, W), np.dtype(cfg["raw_dtype"]))
        refnum = nearest_reference(pid, cfg["ref_nums"]) if cfg["ref_nums"] is not None else 0
        sam = raw.astype(np.float64)
        if cfg["dark"] is not None:
            sam = sam - np.asarray(cfg["dark"], dtype=np.float64)
        if cfg["flats"] is not None:
            fl = np.asarray(cfg["flats"], dtype=np.float64)
            sam = sam / (fl[refnum] if fl.ndim == 4 else fl)
        m = ns(np.ascontiguousarray(sam), refs[refnum], window_size=cfg["window_size"], max_shift=cfg["max_shift"])
        m.debug = False
        res = m.match(quiet=True, **match_kw)
        N0, N1 = res["err"].shape
        off = 0
        keys = RESULT_KEYS_DF if cfg["df"] else tuple(k for k in RESULT_KEYS_DF if k != "df")
        vals, off = slots_out.view(q_out, (len(keys) - 1, N0, N1), np.float64, off)
        for n, k in enumerate(k for k in keys if k != "err"):
            vals[n] = res[k]
        e, off = slots_out.view(q_out, (N0, N1), np.int32, off)
        e[...] = res["err"]
        results.put(("done", seq, q_in, q_out, None))

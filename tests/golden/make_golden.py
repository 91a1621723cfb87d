#!/usr/bin/env python3
"""
Generator of the golden fixtures in tests/golden/*.npz  (run in the BUILD container only).

The reference's own tests pin no numbers, so parity is pinned by
importing the *unmodified* reference here and freezing its outputs.  This script

  1. makes a scratch copy of /root/reference under /tmp (the reference tree is read-only),
     cythonises UMPA/model.pyx with the package directory on the Cython include path (the one
     thing its setup.py lacks under Cython 3) and compiles it with the reference's own flags
     (setup.py:26);
  2. imports that build and runs the reference API on seeded inputs from this repo's own
     generator (umpa_amd/synth.py);
  3. stores inputs + outputs as small .npz files.

Nothing of the reference (source, generated C++, binaries) is written into this repository;
only data: input arrays and the reference's output arrays.

Usage:  python tests/golden/make_golden.py [--rebuild]
"""
import json
import os
import subprocess
import sys
import sysconfig

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SCRATCH = "/tmp/umpa_oracle"
sys.path.insert(0, REPO)


def build_reference(rebuild=False):
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = os.path.join(SCRATCH, "UMPA", "model" + ext)
    if os.path.exists(so) and not rebuild:
        return
    subprocess.run(["rm", "-rf", SCRATCH], check=True)
    subprocess.run(["cp", "-r", "/root/reference", SCRATCH], check=True)
    subprocess.run(["chmod", "-R", "u+w", SCRATCH], check=True)
    subprocess.run(["cython", "--cplus", "-3", "-I", "UMPA", "UMPA/model.pyx", "-o", "model_gen.cpp"],
                   cwd=SCRATCH, check=True)
    inc = sysconfig.get_paths()["include"]
    subprocess.run(["g++", "-std=c++17", "-O3", "-ffast-math", "-march=native", "-fopenmp", "-fPIC",
                    "-shared", "-I" + inc, "-I" + np.get_include(), "-IUMPA", "model_gen.cpp",
                    "-o", so, "-lm"], cwd=SCRATCH, check=True)


def import_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, SCRATCH)
    from UMPA import model as RM          # noqa: E402
    return RM


OUT_KEYS = ["f", "T", "dx", "dy", "df", "err", "debug_Ncalls"]


def run_variant(RM, data, v):
    """One reference run described by the dict ``v``; returns {key: array}."""
    cls = getattr(RM, v["model"])
    kw = dict(window_size=data["Nw"], max_shift=data["max_shift"])
    if data.get("mask") is not None and v.get("use_mask", True):
        kw["mask_list"] = data["mask"]
    if data.get("pos") is not None:
        kw["pos_list"] = [np.array(p) for p in data["pos"]]
    m = cls(data["sam"], data["ref"], **kw)
    m.assign_coordinates = v.get("assign", "sam")
    m.sub_pixel_mode = v.get("subpx", -1)
    if "Nw_set" in v:
        m.Nw = v["Nw_set"]
    mk = dict(num_threads=1, quiet=True)
    if "dxdy" in v:
        mk["dxdy"] = tuple(v["dxdy"])
    if "step" in v:
        mk["step"] = v["step"]
    if "ROI" in v:
        roi = v["ROI"]
        if v.get("ROI_kind") == "slice":
            roi = tuple(slice(*r) for r in roi)
        else:
            roi = tuple(tuple(r) for r in roi)
        mk["ROI"] = roi
    if v["model"] == "UMPAModelDFKernel":
        sh = m.sh if "ROI" not in v and "step" not in v else None
        if sh is None:
            s0, s1 = m._convert_ROI_slice(mk.get("ROI"), mk.get("step"))
            sh = (1 + (s0[1] - s0[0] - 1) // s0[2], 1 + (s1[1] - s1[0] - 1) // s1[2])
        abc = np.zeros(sh + (3,))
        abc[..., 0], abc[..., 1], abc[..., 2] = v["abc"]
        if v.get("abc_ramp"):
            abc[..., 0] += np.linspace(0, 0.2, sh[1])[None, :]
        mk["abc"] = abc
    r = m.match(**mk)
    out = {k: r[k] for k in OUT_KEYS if k in r}
    if v.get("keep_debug"):
        out["debug_d"] = r["debug_d"]
        e = r["err"] == 1
        out["debug_a"] = r["debug_a"] * e[..., None]     # only defined where the walk completed
    if v.get("coverage"):
        out["coverage"] = m.coverage()
    return out


def save_case(name, data, variants, RM, f32_inputs=False, generator=None):
    arrays = {}
    for key in ("sam", "ref", "mask"):
        if generator is not None:
            break                                        # inputs are regenerated from `generator` (umpa_amd.synth)
        if data.get(key) is not None:
            frames = data[key]
            if isinstance(frames, np.ndarray):
                arrays[key] = frames.astype(np.float32) if f32_inputs else frames
            else:                                        # unequal shapes: store per frame
                for k, fr in enumerate(frames):
                    arrays["%s_%d" % (key, k)] = fr
    meta = dict(Nw=data["Nw"], max_shift=data["max_shift"], pos=data.get("pos"),
                ragged=not isinstance(data["sam"], np.ndarray), f32_inputs=f32_inputs,
                variants=variants, generator=generator)
    for n, v in enumerate(variants):
        out = run_variant(RM, data, v)
        for k, a in out.items():
            arrays["v%d_%s" % (n, k)] = a
        print("  %-10s v%-2d %-60s err=%.3f Ncalls mean %.2f max %d" % (
            name, n, json.dumps({k: v[k] for k in v if k not in ("keep_debug",)})[:60],
            out["err"].mean(), out["debug_Ncalls"].mean(), out["debug_Ncalls"].max()))
    arrays["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.2f MB)" % (path, os.path.getsize(path) / 1e6))


def quantise32(a):
    return a.astype(np.float32).astype(np.float64)


def case_H_cap(RM):
    # ---- case H: the 500-call cap (Optim.cpp:14).  A valley along the shift diagonal keeps the walk zigzagging
    # (probe, probe, move, switch axis, ...) until MAX_CALLS stops it: the cap is tested at the loop header only
    # (Optim.cpp:267), so walks end with 500..509 calls depending on where in an iteration they were.
    from umpa_amd.synth import valley_stack
    gen = dict(func="valley_stack", n=408, D=150, q=32.0, K=1)
    sam, ref = valley_stack(gen["n"], gen["D"], gen["q"], gen["K"])
    save_case("H_cap", dict(sam=sam, ref=ref, Nw=1, max_shift=200),
              [dict(model="UMPAModelNoDF"), dict(model="UMPAModelDF"), dict(model="UMPAModelNoDF", assign="ref")],
              RM, generator=gen)


def main():
    build_reference("--rebuild" in sys.argv)
    RM = import_reference()
    from umpa_amd.synth import make_stack
    if "--only-H" in sys.argv:
        return case_H_cap(RM)

    # ---- F1: sub-pixel fit known-answer tests.  NB the reference's wrapper names are swapped:
    # model.spmq -> C++ spmin, model.spm -> C++ spmin_quad (model.pyx:31-80).
    rng = np.random.default_rng(4242)
    A = np.empty((256, 4, 4))
    for n in range(256):
        # a bowl with its minimum inside the central cell plus some roughness
        cx, cy = rng.uniform(0.0, 1.0, 2)
        g = np.arange(-1, 3)
        bowl = (rng.uniform(0.5, 2) * (g[:, None] - cx) ** 2 + rng.uniform(0.5, 2) * (g[None, :] - cy) ** 2
                + rng.uniform(-0.3, 0.3) * (g[:, None] - cx) * (g[None, :] - cy))
        A[n] = 1e-3 * (bowl + 0.05 * rng.standard_normal((4, 4)) + rng.uniform(0, 2))
    sp_pos = np.empty((256, 2)); sp_val = np.empty(256); sq_pos = np.empty((256, 2)); sq_val = np.empty(256)
    for n in range(256):
        p, c = RM.spmq(A[n].copy()); sp_pos[n], sp_val[n] = p, c
        p, c = RM.spm(A[n].copy()); sq_pos[n], sq_val[n] = p, c
    np.savez_compressed(os.path.join(HERE, "F1_subpixel.npz"), a=A, spmin_pos=sp_pos, spmin_val=sp_val,
                        quad_pos=sq_pos, quad_val=sq_val)
    print("wrote F1_subpixel.npz")

    # ---- case A: small, every option
    sam, ref, _ = make_stack(64, 72, 3, 4, df=True, seed=0, amplitude=2.5)
    dataA = dict(sam=sam, ref=ref, Nw=2, max_shift=4)
    variants = []
    for mdl in ("UMPAModelNoDF", "UMPAModelDF"):
        for assign in ("sam", "ref"):
            for subpx in (-1, 0, 1):
                variants.append(dict(model=mdl, assign=assign, subpx=subpx))
    variants[0]["keep_debug"] = True        # NoDF sam -1
    variants[9]["keep_debug"] = True        # DF ref -1
    variants += [
        dict(model="UMPAModelDF", dxdy=[2, -1]),
        dict(model="UMPAModelNoDF", dxdy=[-1.4, 2.6], assign="ref"),
        dict(model="UMPAModelDF", step=3),
        dict(model="UMPAModelDF", ROI=[[5, 40, 2], [3, 50, 3]]),
        dict(model="UMPAModelNoDF", ROI=[[4, 30, 1], [10, 50, 2]], ROI_kind="slice"),
        dict(model="UMPAModelDF", Nw_set=1),
        dict(model="UMPAModelDF", dxdy=[5, 0]),          # first call out of bounds: Ncalls == 1
    ]
    save_case("A_small", dataA, variants, RM)

    # ---- F2: single cost() evaluations incl. the bound cases |s| = ms-1 and ms
    rngc = np.random.default_rng(77)
    pts = []
    for n in range(200):
        i = int(rngc.integers(6, 64 - 6)); j = int(rngc.integers(6, 72 - 6))
        si = int(rngc.integers(-4, 5)); sj = int(rngc.integers(-4, 5))
        pts.append((i, j, si, sj))
    pts = np.array(pts, dtype=np.int32)
    costs = {}
    for mdl, nv in (("UMPAModelNoDF", 2), ("UMPAModelDF", 3)):
        for assign in ("sam", "ref"):
            m = getattr(RM, mdl)(sam, ref, window_size=2, max_shift=4)
            m.assign_coordinates = assign
            out = np.zeros((len(pts), nv))
            for n, (i, j, si, sj) in enumerate(pts):
                if abs(si) >= 4 or abs(sj) >= 4:
                    out[n] = np.nan                      # bound error: the reference returns uninitialised stack
                else:
                    out[n] = m.cost(int(i), int(j), float(si), float(sj))
            costs["%s_%s" % (mdl, assign)] = out
    np.savez_compressed(os.path.join(HERE, "F2_cost.npz"), pts=pts, **costs)
    print("wrote F2_cost.npz")

    # ---- case B: larger walks, restarts, bound exits
    sam, ref, _ = make_stack(96, 112, 4, 5, df=True, seed=10, amplitude=3.5)
    dataB = dict(sam=sam, ref=ref, Nw=3, max_shift=5)
    save_case("B_walks", dataB, [
        dict(model="UMPAModelDF"), dict(model="UMPAModelNoDF"),
        dict(model="UMPAModelDF", assign="ref"), dict(model="UMPAModelDF", subpx=0),
        dict(model="UMPAModelNoDF", assign="ref", subpx=1),
    ], RM)

    # ---- case C: masks (95 % random, and all ones)
    rngm = np.random.default_rng(15)
    sam, ref, _ = make_stack(64, 72, 3, 4, df=True, seed=20, amplitude=2.0)
    mask = (rngm.uniform(size=sam.shape) < 0.95).astype(np.float64)
    save_case("C_mask", dict(sam=sam, ref=ref, mask=mask, Nw=2, max_shift=4), [
        dict(model="UMPAModelDF", coverage=True), dict(model="UMPAModelNoDF"),
        dict(model="UMPAModelDF", assign="ref"), dict(model="UMPAModelNoDF", step=2),
    ], RM)
    save_case("C_mask_ones", dict(sam=sam, ref=ref, mask=np.ones_like(sam), Nw=2, max_shift=4), [
        dict(model="UMPAModelDF"), dict(model="UMPAModelNoDF"),
    ], RM)

    # ---- case D: sample stepping (pos_list) with unequal frame shapes
    shapes = [(64, 72), (64, 72), (60, 76), (64, 72)]
    pos = [[0, 0], [0, 12], [10, 0], [10, 12]]
    sams, refs = [], []
    for k, (h, w) in enumerate(shapes):
        s, r, _ = make_stack(h, w, 1, 4, df=True, seed=30 + k, amplitude=2.0)
        sams.append(np.ascontiguousarray(s[0])); refs.append(np.ascontiguousarray(r[0]))
    save_case("D_stepping", dict(sam=sams, ref=refs, pos=pos, Nw=2, max_shift=4), [
        dict(model="UMPAModelDF", coverage=True), dict(model="UMPAModelNoDF"),
        dict(model="UMPAModelDF", assign="ref", step=2),
    ], RM)

    # ---- case E: kernel dark-field model (tiny: 289 blur taps per window pixel)
    sam, ref, _ = make_stack(48, 52, 2, 4, df=True, seed=40, amplitude=1.5)
    save_case("E_dfkernel", dict(sam=sam, ref=ref, Nw=2, max_shift=4), [
        dict(model="UMPAModelDFKernel", abc=[0.1, 0.0, 0.1], step=2),
        dict(model="UMPAModelDFKernel", abc=[0.3, 0.05, 0.2], abc_ramp=True, step=3),
    ], RM)

    # ---- F7: BASELINE config C1 verbatim (max_shift=2: every pixel fails) and with max_shift=4
    sam, ref, _ = make_stack(256, 256, 3, 4, df=False, seed=50)
    sam, ref = quantise32(sam), quantise32(ref)
    save_case("F7_C1_ms2", dict(sam=sam, ref=ref, Nw=3, max_shift=2), [dict(model="UMPAModelNoDF")], RM, f32_inputs=True)
    save_case("F7_C1_ms4", dict(sam=sam, ref=ref, Nw=3, max_shift=4), [dict(model="UMPAModelNoDF")], RM, f32_inputs=True)

    # ---- F8: crops with the C2 (K=10, Nw=5, ms=5) and C3 (K=20, Nw=7, ms=8) parameters
    sam, ref, _ = make_stack(84, 148, 10, 5, df=True, seed=60, amplitude=2.5)
    save_case("F8_C2_crop", dict(sam=quantise32(sam), ref=quantise32(ref), Nw=5, max_shift=5),
              [dict(model="UMPAModelDF"), dict(model="UMPAModelNoDF")], RM, f32_inputs=True)
    sam, ref, _ = make_stack(62, 94, 20, 8, df=True, seed=70, amplitude=5.5)
    save_case("F8_C3_crop", dict(sam=quantise32(sam), ref=quantise32(ref), Nw=7, max_shift=8),
              [dict(model="UMPAModelDF")], RM, f32_inputs=True)
    case_H_cap(RM)


if __name__ == "__main__":
    main()

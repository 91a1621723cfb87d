#!/usr/bin/env python3
"""What does cutting one rank's slab of BASELINE config C4 (1042 x 8192) into row pieces cost?

bench.py's multi-GPU leg matches a slab in pieces so that the rows of piece c travel to rank 0 while piece c + 1 is matched
(`set_rows_callback`); the part of the gather that cannot overlap is the last piece's.  This tool times the match of the slab on
ONE GPU with a do-nothing callback for several piece counts.

    python tools/piece_rate.py [pieces ...]
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from umpa_amd import _lib, model
    from umpa_amd.synth import CONFIGS, make_stack
    cfg = CONFIGS["C4"]
    H, W, K, Nw, ms, df = 1042, cfg["W"], cfg["K"], cfg["Nw"], cfg["max_shift"], cfg["df"]
    sam, ref, _ = make_stack(H, W, K, ms, df=df, seed=0, order=1)
    m = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms, device=0)
    lib, h = m._lib, m._handle
    N0, N1 = m.extent
    dev = torch.device("cuda", 0)
    values = torch.zeros((N0, N1, 5), dtype=torch.float64, device=dev)
    err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    calls = []
    cb = _lib.ROWS_FN(lambda lo, hi, _u: calls.append((lo, hi)))

    def step(piece_rows):
        if piece_rows:
            lib.check(lib.set_rows_callback(h, ctypes.cast(cb, ctypes.c_void_p), None, piece_rows), "set_rows_callback")
        rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), 5, None, err.data_ptr(), None, 0.0, None, None, None,
                              _lib.F_DEVICE_IO, ctypes.c_void_p(stream))
        if piece_rows:
            lib.set_rows_callback(h, None, None, 0)
        lib.check(rc, "match_region")

    for pieces in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6, 8, 12]:
        piece_rows = 0 if pieces == 1 else ((-(-N0 // pieces)) + 31) // 32 * 32
        for _ in range(3):
            step(piece_rows)
        torch.cuda.synchronize()
        del calls[:]
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            step(piece_rows)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("pieces wanted %2d: rows per piece %4d, callbacks per match %2d, %.3f ms per slab (%.0f Mpx/s)" % (
            pieces, piece_rows or N0, len(calls) // n, dt * 1e3, N0 * N1 / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()

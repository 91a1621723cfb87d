# cython: language_level=3
# hip_backend.pyx -- the caller a maintainer of the reference would write on top of UMPA/HipModel.pxd (INTEGRATION.md B):
# what UMPAModelBase.__cinit__ / _create_c_model / _match / __dealloc__ do with `ModelBase[double]* c_model`
# (UMPA/model.pyx:121, 764-770, 476-492, 305-309), against the C ABI of include/umpa_hip.h.  Compiled and run by
# tests/test_cython_binding.py; not part of the library's own build (which binds the same symbols with ctypes).
import numpy as np
cimport numpy as cnp
from HipModel cimport (umpa_hip_model, umpa_hip_create, umpa_hip_destroy, umpa_hip_match_region, umpa_hip_set_subpx,
                       umpa_hip_last_error)

cnp.import_array()


cdef class HipModelDF:
    cdef umpa_hip_model* h_model
    cdef object keep                   # the frames stay alive like model.pyx:123-129 keeps sam_list / ref_list
    cdef public int Na, Nw, max_shift, padding

    def __cinit__(self, sam, ref, int Nw, int max_shift, int device=0):
        cdef cnp.ndarray[double, ndim=3, mode="c"] s = np.ascontiguousarray(sam, dtype=np.float64)
        cdef cnp.ndarray[double, ndim=3, mode="c"] r = np.ascontiguousarray(ref, dtype=np.float64)
        cdef int Na = s.shape[0], k, S = 2 * Nw + 1
        cdef cnp.ndarray[int, ndim=2, mode="c"] dims = np.empty((Na, 2), dtype=np.intc)
        cdef cnp.ndarray[int, ndim=2, mode="c"] pos = np.zeros((Na, 2), dtype=np.intc)
        cdef double* sp[64]
        cdef double* rp[64]
        if Na > 64:
            raise RuntimeError("too many frames for this sketch")
        for k in range(Na):
            dims[k, 0] = s.shape[1]; dims[k, 1] = s.shape[2]
            sp[k] = &s[k, 0, 0]; rp[k] = &r[k, 0, 0]
        w = np.outer(np.hamming(S), np.hamming(S))                      # model.pyx:691-696
        cdef cnp.ndarray[double, ndim=2, mode="c"] win = np.ascontiguousarray(w / w.sum())
        self.keep = (s, r)
        self.Na, self.Nw, self.max_shift, self.padding = Na, Nw, max_shift, Nw + max_shift
        self.h_model = umpa_hip_create(1, Na, &dims[0, 0], sp, rp, NULL, &pos[0, 0], Nw, &win[0, 0], max_shift,
                                       self.padding, device, 0)
        if self.h_model == NULL:
            raise RuntimeError(umpa_hip_last_error().decode())

    def __dealloc__(self):
        if self.h_model != NULL:
            umpa_hip_destroy(self.h_model)
            self.h_model = NULL

    def match(self):
        """The prange block of model.pyx:476-492 as ONE call."""
        s = self.keep[0]
        cdef int N0 = s.shape[1] - 2 * self.padding, N1 = s.shape[2] - 2 * self.padding, rc
        cdef cnp.ndarray[double, ndim=3, mode="c"] values = np.zeros((N0, N1, 5))
        cdef cnp.ndarray[double, ndim=3, mode="c"] uv = np.zeros((N0, N1, 2))
        cdef cnp.ndarray[int, ndim=2, mode="c"] err = np.zeros((N0, N1), dtype=np.intc)
        cdef umpa_hip_model* h = self.h_model
        with nogil:
            rc = umpa_hip_match_region(h, 0, 1, N0, 0, 1, N1, &values[0, 0, 0], 5, &uv[0, 0, 0], &err[0, 0],
                                       NULL, 0.0, NULL, NULL, NULL, 0, NULL)
        if rc < 0:
            raise RuntimeError(umpa_hip_last_error().decode())
        return {"f": values[:, :, 0].copy(), "T": values[:, :, 1].copy(), "dx": values[:, :, 2].copy(),
                "dy": values[:, :, 3].copy(), "df": values[:, :, 4].copy(), "err": err}


def last_error():
    return umpa_hip_last_error().decode()

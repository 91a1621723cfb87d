/*
 * umpa_hip.h -- C ABI of libumpa_hip.so, the MI355X (gfx950) implementation of the UMPA
 * per-pixel matching path.
 *
 * This is the drop-in boundary.  The reference has exactly one native interface under its
 * Python API: the C++ class models::ModelBase<double> and its three subclasses
 * (reference UMPA/lib/Model.h:78-189), bound by Cython in UMPA/Model.pxd:30-68 and driven by
 * the pixel loop of UMPA/model.pyx:476-492.  Every entry point below names the reference
 * interface it replaces.  Signatures use plain pointers and sizes only.
 *
 * Conventions (all taken from the reference):
 *  - frames are row-major float64, frame k has shape dims[2k] x dims[2k+1];
 *  - (i, j) are absolute frame coordinates (row, column); padding is NOT added by
 *    umpa_hip_cost/min/coverage (UMPA/model.pyx:853-854) but IS added by the *_region calls,
 *    exactly like the Cython loop adds `offset` (UMPA/model.pyx:482-483);
 *  - shifts: component 0 = rows, component 1 = columns (UMPA/model.pyx:461-465);
 *  - values = [f, T, dx(col), dy(row), df] (UMPA/lib/Model.cpp:934-938);
 *  - status word: bit0 ok, bit1 bound_error, bit2 dimension, bit3 positive
 *    (UMPA/lib/Optim.h:7-12).
 *
 * Every function that returns int returns a status word (>= 0) or a negative error code;
 * umpa_hip_last_error() then holds a message.  The library never falls back to a CPU path:
 * with no usable HIP device every call fails with UMPA_HIP_E_DEVICE.
 */
#ifndef UMPA_HIP_H
#define UMPA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMPA_HIP_KIND_NODF      0   /* models::ModelNoDF      Model.h:126-140 */
#define UMPA_HIP_KIND_DF        1   /* models::ModelDF        Model.h:145-164 */
#define UMPA_HIP_KIND_DFKERNEL  2   /* models::ModelDFKernel  Model.h:170-189 (direct kernel only; values[4..6] = a,b,c in) */

#define UMPA_HIP_ST_OK          1
#define UMPA_HIP_ST_BOUND       2
#define UMPA_HIP_ST_DIMENSION   4
#define UMPA_HIP_ST_POSITIVE    8

#define UMPA_HIP_E_ARG         (-1)
#define UMPA_HIP_E_DEVICE      (-2)
#define UMPA_HIP_E_NOMEM       (-3)
#define UMPA_HIP_E_UNSUPPORTED (-4)
#define UMPA_HIP_E_LAUNCH      (-5)

/* create flags */
#define UMPA_HIP_F_DEVICE_FRAMES  1  /* sam/ref/mask pointers are device pointers on `device`; borrowed, not copied */
/* match flags */
#define UMPA_HIP_F_DEVICE_IO      1  /* values/uv/err/covermap/debug pointers are device pointers; the call only enqueues work on `stream`
                                        and returns: the results are valid, and a kernel fault surfaces, when the caller
                                        synchronises that stream (hipStreamSynchronize / an event), as with any launch */
#define UMPA_HIP_F_FORCE_DIRECT   2  /* use the general direct kernel even where the tiled fast path applies */
#define UMPA_HIP_F_FORCE_TILED    4  /* fail with E_UNSUPPORTED instead of silently using the direct kernel */
#define UMPA_HIP_F_PLANAR         8  /* values is [nparam][N0*N1] (one plane per map) instead of the reference's [N0*N1][nparam] */
#define UMPA_HIP_F_REUSE_REF_MAPS 16 /* the reference stack is the one of this model's previous match (match_unbiased, a projection
                                        series against one reference): the tiled path keeps its reference-side maps and recomputes
                                        only the sample side.  For frames the model owns it also checks that itself
                                        (umpa_hip_update_frames / set_window invalidate); for borrowed device frames the caller vouches.
                                        Results are identical with and without. */

#define UMPA_HIP_F_USE_STAGED      32 /* the stack uploaded by umpa_hip_stage_sample becomes the sample stack of this match */
#define UMPA_HIP_F_FORCE_PLAIN_DIRECT 128 /* with F_FORCE_DIRECT: the plain direct kernel (windows through L1) even where the
                                             staged one (windows out of LDS) applies; the two give identical results */
#define UMPA_HIP_F_ASYNC           64 /* host-array call: enqueue the kernels and the downloads (into page-locked arrays) and
                                         return; the arrays are valid after umpa_hip_wait() */

typedef struct umpa_hip_model umpa_hip_model;

/* library-wide */
int         umpa_hip_device_count(void);
const char *umpa_hip_last_error(void);
const char *umpa_hip_version(void);

/*
 * Replaces the ModelNoDF/ModelDF/ModelDFKernel constructors (Model.cpp:193-216, :597-604,
 * :963-968; called from model.pyx:769, :835, :911).  Host frames are copied to the device
 * once here and stay resident for the model's lifetime (the reference keeps borrowed host
 * pointers instead, model.pyx:123-129).  mask may be NULL.  pos is [Na][2], >= 0.
 * padding >= Nw + max_shift, and >= Nw + max_shift + 8 for UMPA_HIP_KIND_DFKERNEL, whose on-the-fly blur
 * reads 8 pixels further (KERNEL_WINDOW_SIZE, Model.h:7; safe_crop, model.pyx:904); less is UMPA_HIP_E_ARG.
 * Returns NULL on error.
 */
umpa_hip_model *umpa_hip_create(int kind, int Na, const int *dims,
                                double *const *sam, double *const *ref, double *const *mask,
                                const int *pos, int Nw, const double *win,
                                int max_shift, int padding, int device, int flags);

/* Replace the contents of the resident sample and/or reference stacks (same shapes; NULL = keep).
 * No counterpart in the reference, whose models borrow host pointers (model.pyx:123-129): there a caller
 * such as umpa_multi.py:149 builds a new model per projection.  Here the reference stack can stay in HBM
 * while projections stream through. */
int umpa_hip_update_frames(umpa_hip_model *m, double *const *sam, double *const *ref);

/* The step-scan pipeline of UMPA/umpa_multi.py:133-150 on one GPU: upload the NEXT projection while the current
 * one is being matched.  raw[k] are host frames of the model's shapes (raw_dtype 0 float64, 1 float32, 2 uint16
 * detector counts; page-locked memory -- umpa_hip_host_alloc / umpa_hip_host_register -- makes the call return at
 * once), copied on the model's upload stream and written to the model's BACK sample buffer as
 * (raw - dark) / flat   (dark[k], flat[k]: device float64 frames, or NULL tables: plain conversion) -- the flat-field
 * correction of umpa_multi.py:144 fused into the upload.  The staged stack becomes the sample stack at the next
 * umpa_hip_match_region with UMPA_HIP_F_USE_STAGED, in stream order.  Only for models that own their frames. */
int umpa_hip_stage_sample(umpa_hip_model *m, const void *const *raw, int raw_dtype,
                          const double *const *dark, const double *const *flat);
/* Device-resident outputs (UMPA_HIP_F_DEVICE_IO) in row pieces: the region is matched in row chunks of `piece_rows`
 * rows (rounded up to a multiple of 32; the same boundaries on every rank of a sharded match whatever its slab size), and
 * fn(lo, hi, user) is called on the calling thread as soon as the kernels that complete output rows [lo, hi) are enqueued
 * on the match's stream -- the caller can queue work behind them there (a send of those rows to another GPU, a
 * download) that overlaps the matching of the next piece.  fn = NULL switches it off.  (bench.py --gpus N: the gather
 * of the result slabs on rank 0 travels piece by piece while the slab is still being matched.) */
typedef void (*umpa_hip_rows_fn)(int row_lo, int row_hi, void *user);
int umpa_hip_set_rows_callback(umpa_hip_model *m, umpa_hip_rows_fn fn, void *user, int piece_rows);

/* wait for the OLDEST UMPA_HIP_F_ASYNC match of this model that is still in flight (kernels and downloads); returns its
 * status.  Up to two may be in flight (the library keeps two sets of device output buffers: the maps of match p travel to
 * the host while match p + 1 is computed); a third UMPA_HIP_F_ASYNC call, or a synchronous one, before the wait is
 * UMPA_HIP_E_ARG.  With nothing in flight it waits for the model's streams. */
int umpa_hip_wait(umpa_hip_model *m);
/* page-lock / release memory the caller owns (a shared-memory ring that feeds umpa_hip_stage_sample) */
int umpa_hip_host_register(void *p, size_t bytes);
int umpa_hip_host_unregister(void *p);

/* ~ModelBase (Model.cpp:224) */
void umpa_hip_destroy(umpa_hip_model *m);

/* ModelBase::set_window (Model.cpp:239-246); negative Nw -> UMPA_HIP_E_ARG (the reference throws) */
int umpa_hip_set_window(umpa_hip_model *m, const double *win, int Nw);
/* ModelBase::subpx_func (Model.h:91; model.pyx:753-755): -1 spline, 0 none, 1 paraboloid */
int umpa_hip_set_subpx(umpa_hip_model *m, int mode);
/* ModelBase::reference_shift (Model.h:92; model.pyx:732-742): 0 'sam', 1 'ref' */
int umpa_hip_set_reference_shift(umpa_hip_model *m, int v);

/* ModelBase::coverage (Model.cpp:273-314), one pixel */
int umpa_hip_coverage(umpa_hip_model *m, double *out, int i, int j);
/* the serial double loop of model.pyx:524-528, as one launch; out is host [N0*N1] */
int umpa_hip_coverage_region(umpa_hip_model *m, int start0, int step0, int N0,
                             int start1, int step1, int N1, double *out);

/* Model*::cost_interface (Model.cpp:533-542, :887-897, :1181-1192): values[0]=cost, [1]=T, [2]=df;
 * DFKernel: values[2..4] = a, b, c on input */
int umpa_hip_cost(umpa_hip_model *m, int i, int j, int shift_i, int shift_j, double *values);

/* Model*::min (Model.cpp:562-578, :923-940) with the minimizer_debug fields (Optim.h:15-21)
 * returned separately; uv is in/out (start shift -> result).  dbg_* may be NULL. */
int umpa_hip_min(umpa_hip_model *m, int i, int j, double *values, double *uv,
                 double *dbg_d, double *dbg_a, int *ncalls);

/*
 * The pixel loop of model.pyx:476-492 as one call: for xi < N0, xj < N1 run min() at
 * (padding+start0+step0*xi, padding+start1+step1*xj) unless covermap[xi,xj] < cover_threshold.
 *   values [N0*N1*nparam] in/out, uv [N0*N1*2] in/out, err [N0*N1] out (status.ok),
 *   covermap [N0*N1] or NULL, dbg_d [N0*N1*25] / dbg_a [N0*N1*16] / dbg_ncalls [N0*N1] or NULL.
 * With UMPA_HIP_F_DEVICE_IO all of these are device pointers and the call only enqueues work
 * on `stream` (a hipStream_t, or NULL for the default stream).
 */
int umpa_hip_match_region(umpa_hip_model *m, int start0, int step0, int N0,
                          int start1, int step1, int N1,
                          double *values, int nparam, double *uv, int *err,
                          const double *covermap, double cover_threshold,
                          double *dbg_d, double *dbg_a, int *dbg_ncalls,
                          int flags, void *stream);

/* sub-pixel fits, exposed by the reference as model.spmq / model.spm (model.pyx:31-80):
 * spmin (Optim.cpp:41-130) and spmin_quad (Optim.cpp:155-185), evaluated on the device. */
int umpa_hip_spmin(int device, const double *a16, double *pos2, double *value);
int umpa_hip_spmin_quad(int device, const double *a16, double *pos2, double *value);

/* Bad-pixel repair of result maps, the epilogue of the reference's align.UMPA_normal / align.UMPA_nobias
 * (UMPA/align.py:51-52, 115-116, 661-732): pixels of `in` outside [lo, hi] are "bad"; each of `iterations`
 * passes replaces every bad pixel by the median of its 2*ndims neighbours as they were before the pass
 * (ndims = 2: along H and W; ndims = 1: along W only; edges reflect), everything else is copied.
 * `in`/`out` hold `nimg` images of H x W doubles (may be the same buffer); host pointers, or device
 * pointers with UMPA_HIP_F_DEVICE_IO.  The call returns when the result is in `out`. */
int umpa_hip_correct_bad_pixels(const double *in, double *out, long nimg, int H, int W, int ndims,
                                double lo, double hi, int iterations, int device, int flags, void *stream);

/* Page-locked host memory for the arrays of the host-array entry points (result maps above all: 181 MB per C2
 * match): into such memory the download is a DMA at PCIe rate that overlaps the matching of the next row chunk;
 * into pageable memory the runtime has to stage it.  Blocks handed back are kept pinned in a pool (pinning is the
 * slow part) until umpa_hip_host_trim().  The reference's arrays are ordinary numpy allocations (model.pyx:442-474). */
void *umpa_hip_host_alloc(size_t bytes);
void  umpa_hip_host_free(void *p);
void  umpa_hip_host_trim(void);

/* instrumentation used by bench.py for the roofline line: when enabled every kernel launch is
 * bracketed by HIP events on its launch stream; collect() waits for them and folds them into
 * per-kernel totals (returns the number of distinct kernels), read() returns one total. */
int umpa_hip_timing_enable(umpa_hip_model *m, int on);
int umpa_hip_timing_collect(umpa_hip_model *m);
int umpa_hip_timing_read(umpa_hip_model *m, int index, const char **name, double *total_ms, int *launches);
/* fp64 fused multiply-adds executed by the collected launches of kernel `index` (counted on the host from the launch
 * geometry; filled in for corr_volume, 0 for kernels that do not report it): the numerator of roofline.fp64_fma */
int umpa_hip_timing_fma(umpa_hip_model *m, int index, double *fma);
/* which path the last match_region took: 0 none, 1 direct (plain), 2 tiled, 3 direct (staged: windows out of LDS),
 * 4 sample stepping: tiled on the rectangle every frame contributes to + general kernels on the border strips */
int umpa_hip_last_path(umpa_hip_model *m);
/* the tiled path computes only the passes of its shift table that walks read (seed tiles predict, misses are repaired:
 * the maps are those of the exhaustive table): out4 = { (tile, pass) units computed by the last match, units of the
 * exhaustive table, pixels whose walk had to be run again, tiles whose prediction fell short }.  Waits for the device. */
int umpa_hip_last_stats(umpa_hip_model *m, double *out4);

#ifdef __cplusplus
}
#endif
#endif /* UMPA_HIP_H */

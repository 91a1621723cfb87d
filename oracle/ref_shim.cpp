/*
 * oracle/ref_shim.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A thin extern "C" face over the *unmodified* reference C++ core, compiled from
 * the reference sources where they lie (-I/root/reference/UMPA/lib, see
 * oracle/Makefile).  Nothing of the reference is copied into this repository:
 * this file only #includes the reference translation units exactly as the
 * reference's own Cython layer does (UMPA/Model.pxd:3 `cdef extern from
 * "lib/Model.cpp"`, UMPA/Optim.pxd:1, UMPA/CUtils.pxd:1) and forwards calls.
 *
 * The only logic restated here is the pixel loop of UMPA/model.pyx:476-492
 * (OpenMP team, one minimizer_debug per thread, dynamic schedule over rows,
 * coverage threshold test, err = status.ok), because that loop lives in Cython
 * and cannot be compiled without Python.
 *
 * Output: oracle/_ref/libumpa_ref.so (git-ignored; travels to the GPU box as a
 * prebuilt binary).  Used (a) to validate oracle/umpa_oracle.c, (b) as the
 * "reference" CPU baseline in bench.py.
 */
#include <vector>
#include <cstring>
#include <omp.h>

// The reference is a single translation unit built from these three files
// (UMPA/CUtils.pxd, UMPA/Optim.pxd, UMPA/Model.pxd).
#include "Utils.cpp"
#include "Optim.cpp"
#include "Model.cpp"

namespace {

struct Holder {
    std::vector<int>     dims;      // Na*2
    std::vector<int>     poss;      // Na*2
    std::vector<int*>    dim_ptrs;
    std::vector<int*>    pos_ptrs;
    std::vector<double*> sams, refs, masks;
    std::vector<double>  win;
    int kind;                       // 0 NoDF, 1 DF, 2 DFKernel
    models::ModelBase<double>* m;
};

inline int pack(error_status s)
{
    return (int)s.ok | ((int)s.bound_error << 1) | ((int)s.dimension << 2) | ((int)s.positive << 3);
}

} // namespace

extern "C" {

void* umparef_create(int kind, int Na, const int* dims, double* const* sam, double* const* ref,
                     double* const* mask, const int* pos, int Nw, const double* win,
                     int max_shift, int padding)
{
    Holder* h = new Holder;
    h->kind = kind;
    h->dims.assign(dims, dims + 2 * Na);
    h->poss.assign(pos, pos + 2 * Na);
    for (int k = 0; k < Na; k++) {
        h->dim_ptrs.push_back(&h->dims[2 * k]);
        h->pos_ptrs.push_back(&h->poss[2 * k]);
        h->sams.push_back(sam[k]);
        h->refs.push_back(ref[k]);
        if (mask) h->masks.push_back(mask[k]);
    }
    int S = 2 * Nw + 1;
    h->win.assign(win, win + S * S);
    // model.pyx:769, :835, :911
    if (kind == 0)
        h->m = new models::ModelNoDF<double>(Na, h->dim_ptrs, h->sams, h->refs, h->masks, h->pos_ptrs,
                                             Nw, h->win.data(), max_shift, padding);
    else if (kind == 1)
        h->m = new models::ModelDF<double>(Na, h->dim_ptrs, h->sams, h->refs, h->masks, h->pos_ptrs,
                                           Nw, h->win.data(), max_shift, padding);
    else
        h->m = new models::ModelDFKernel<double>(Na, h->dim_ptrs, h->sams, h->refs, h->masks, h->pos_ptrs,
                                                 Nw, h->win.data(), max_shift, padding);
    return h;
}

void umparef_destroy(void* p)
{
    Holder* h = (Holder*)p;
    delete h->m;
    delete h;
}

void umparef_set_window(void* p, const double* win, int Nw)   // model.pyx:702-704
{
    Holder* h = (Holder*)p;
    int S = 2 * Nw + 1;
    h->win.assign(win, win + S * S);
    h->m->set_window(h->win.data(), Nw);
}

void umparef_set_subpx(void* p, int mode) { ((Holder*)p)->m->subpx_func = mode; }            // model.pyx:753-755
void umparef_set_reference_shift(void* p, int v) { ((Holder*)p)->m->reference_shift = v; }   // model.pyx:732-742

int umparef_coverage(void* p, double* out, int i, int j)
{
    return pack(((Holder*)p)->m->coverage(out, i, j));
}

int umparef_cost(void* p, int i, int j, int si, int sj, double* values)
{
    return pack(((Holder*)p)->m->cost_interface(i, j, si, sj, values));
}

int umparef_min(void* p, int i, int j, double* values, double* uv, double* dbg_d, double* dbg_a, int* ncalls)
{
    minimizer_debug<double> db;
    std::memset(&db, 0, sizeof(db));
    error_status s = ((Holder*)p)->m->min(i, j, values, uv, &db);
    if (dbg_d) std::memcpy(dbg_d, db.d, sizeof(db.d));
    if (dbg_a) std::memcpy(dbg_a, db.a, sizeof(db.a));
    if (ncalls) *ncalls = db.Ncalls;
    return pack(s);
}

/* The pixel loop of UMPA/model.pyx:476-492. `offset` (= padding) is added here as there. */
void umparef_match_region(void* p, int start0, int step0, int N0, int start1, int step1, int N1,
                          double* values, int nparam, double* uv, int* err,
                          const double* covermap, double cover_threshold,
                          double* dbg_d, double* dbg_a, int* dbg_ncalls, int num_threads)
{
    Holder* h = (Holder*)p;
    models::ModelBase<double>* m = h->m;
    const int offset = m->padding;
#pragma omp parallel num_threads(num_threads)
    {
        minimizer_debug<double>* db = new minimizer_debug<double>();
        std::memset(db, 0, sizeof(*db));
#pragma omp for schedule(dynamic)
        for (int xi = 0; xi < N0; xi++) {
            for (int xj = 0; xj < N1; xj++) {
                size_t px = (size_t)xi * N1 + xj;
                if (covermap && covermap[px] < cover_threshold) continue;
                error_status s = m->min(offset + start0 + step0 * xi, offset + start1 + step1 * xj,
                                        &values[px * nparam], &uv[px * 2], db);
                err[px] = s.ok;
                if (dbg_d) std::memcpy(&dbg_d[px * 25], db->d, 25 * sizeof(double));
                if (dbg_a) std::memcpy(&dbg_a[px * 16], db->a, 16 * sizeof(double));
                if (dbg_ncalls) dbg_ncalls[px] = db->Ncalls;
            }
        }
        delete db;
    }
}

double umparef_spmin(double* a, double* pos) { return spmin<double>(a, pos); }
double umparef_spmin_quad(double* a, double* pos) { return spmin_quad<double>(a, pos); }
int umparef_max_threads(void) { return omp_get_max_threads(); }

} // extern "C"

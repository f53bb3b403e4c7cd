/*
 * oracle/csrc/oracle_kernels.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C twins of the heavy ReFeX loops, single thread, used (a) as the checker for large
 * parity cases where the numpy loops of oracle/refex.py would take minutes and (b) as the
 * "port" CPU baseline timed by bench.py.  Each function restates the reference lines cited;
 * tests/test_oracle_pinned.py checks them against oracle/refex.py and the golden vectors.
 *
 *   gcc -O2 -shared -fPIC -o oracle/liboracle.so oracle/csrc/oracle_kernels.c -lm
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum_DOUBLE) of the
 * rows X[col[k]], k in [b, b+cnt), for all f columns at once: the association order of the
 * Series.sum() the reference runs per column (features/extract.py:110-113).
 *   cnt < 8     sequential
 *   cnt <= 128  r[j] = x[j] + x[j+8] + ..., ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail
 *   cnt > 128   halves split at cnt/2 rounded down to a multiple of 8 */
static void pairwise_rows(const int32_t *col, const double *X, int f, int64_t b, int64_t cnt, double *out)
{
    if (cnt < 8) {
        for (int c = 0; c < f; ++c) out[c] = 0.0;
        for (int64_t k = b; k < b + cnt; ++k) {
            const double *x = X + (int64_t)col[k] * f;
            for (int c = 0; c < f; ++c) out[c] += x[c];
        }
    } else if (cnt <= 128) {
        const int64_t c8 = cnt - cnt % 8;
        double r[8][f];
        for (int j = 0; j < 8; ++j) {
            const double *x = X + (int64_t)col[b + j] * f;
            for (int c = 0; c < f; ++c) r[j][c] = x[c];
        }
        for (int64_t i = 8; i < c8; i += 8)
            for (int j = 0; j < 8; ++j) {
                const double *x = X + (int64_t)col[b + i + j] * f;
                for (int c = 0; c < f; ++c) r[j][c] += x[c];
            }
        for (int c = 0; c < f; ++c)
            out[c] = ((r[0][c] + r[1][c]) + (r[2][c] + r[3][c])) + ((r[4][c] + r[5][c]) + (r[6][c] + r[7][c]));
        for (int64_t i = c8; i < cnt; ++i) {
            const double *x = X + (int64_t)col[b + i] * f;
            for (int c = 0; c < f; ++c) out[c] += x[c];
        }
    } else {
        int64_t c2 = cnt / 2;
        c2 -= c2 % 8;
        double right[f];
        pairwise_rows(col, X, f, b, c2, out);
        pairwise_rows(col, X, f, b + c2, cnt - c2, right);
        for (int c = 0; c < f; ++c) out[c] += right[c];
    }
}

/* graphrole/features/extract.py:98-119 -- sum and mean over the neighbour rows; `col` lists every
 * row's neighbours in the order the reference visits them.  X, S, M row-major n x f. */
/* Threads (test speed only): every loop below that is parallel splits over ROWS (or column pairs) whose results do not
 * depend on each other, so the values are the same for any thread count.  bench.py's cpu_baseline leg sets 1. */
static int g_threads = 0;
void orc_set_threads(int t) { g_threads = t; }
static int orc_threads(void)
{
#ifdef _OPENMP
    if (g_threads > 0) return g_threads;
    int t = omp_get_max_threads();
    return t > 32 ? 32 : t;
#else
    return 1;
#endif
}

void orc_aggregate(int64_t n, const int64_t *row_ptr, const int32_t *col, int f,
                   const double *X, double *S, double *M)
{
    if (f <= 0) return;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(orc_threads())
    for (int64_t v = 0; v < n; ++v) {
        double *s = S + v * f, *m = M + v * f;
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        /* ndarray.sum() walks a column in chunks of 8192 elements (ufunc buffer size): every
         * chunk is summed pairwise and added to the running total, ((0 + p(c0)) + p(c1)) + ... */
        for (int c = 0; c < f; ++c) s[c] = 0.0;
        for (int64_t k = b; k < e; k += 8192) {
            double part[f];
            pairwise_rows(col, X, f, k, (e - k < 8192) ? e - k : 8192, part);
            for (int c = 0; c < f; ++c) s[c] += part[c];
        }
        const double cnt = (double)(e - b);
        for (int c = 0; c < f; ++c) m[c] = (e > b) ? s[c] / cnt : 0.0;
    }
}

/* graphrole/features/extract.py:98-119 with 'var' / 'std' among the aggregations: pandas' nanvar, ddof = 1:
 * avg = sum / count, var = sum((avg - x)^2) / (count - 1), both sums in ndarray.sum() order; std = sqrt(var);
 * fewer than two neighbours -> NaN -> 0. */
void orc_aggregate_var(int64_t n, const int64_t *row_ptr, const int32_t *col, int f,
                       const double *X, double *VAR, double *STD)
{
    if (f <= 0) return;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(orc_threads())
    for (int64_t v = 0; v < n; ++v) {
        double *var = VAR + v * f, *sd = STD + v * f;
        const int64_t b = row_ptr[v], e = row_ptr[v + 1], cnt = e - b;
        for (int c = 0; c < f; ++c) var[c] = sd[c] = 0.0;
        if (cnt < 2) continue;
        /* the transformed rows (avg - x)^2 in neighbour order, then the same summation routine */
        double avg[f];
        int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)cnt);
        double *T = (double *)malloc(sizeof(double) * (size_t)cnt * (size_t)f);
        for (int c = 0; c < f; ++c) avg[c] = 0.0;
        for (int64_t k = b; k < e; k += 8192) {
            double part[f];
            pairwise_rows(col, X, f, k, (e - k < 8192) ? e - k : 8192, part);
            for (int c = 0; c < f; ++c) avg[c] += part[c];
        }
        for (int c = 0; c < f; ++c) avg[c] /= (double)cnt;
        for (int64_t k = 0; k < cnt; ++k) {
            const double *x = X + (int64_t)col[b + k] * f;
            idx[k] = (int32_t)k;
            for (int c = 0; c < f; ++c) { const double t = avg[c] - x[c]; T[k * f + c] = t * t; }
        }
        for (int64_t k = 0; k < cnt; k += 8192) {
            double part[f];
            pairwise_rows(idx, T, f, k, (cnt - k < 8192) ? cnt - k : 8192, part);
            for (int c = 0; c < f; ++c) var[c] += part[c];
        }
        for (int c = 0; c < f; ++c) { var[c] /= (double)(cnt - 1); sd[c] = sqrt(var[c]); }
        free(idx);
        free(T);
    }
}

/* graphrole/features/extract.py:98-119 with 'prod' among the aggregations: np.multiply.reduce is a
 * plain left-to-right product (neighbours in the order given); the empty product is 1. */
void orc_aggregate_prod(int64_t n, const int64_t *row_ptr, const int32_t *col, int f,
                        const double *X, double *P)
{
#pragma omp parallel for schedule(dynamic, 1024) num_threads(orc_threads())
    for (int64_t v = 0; v < n; ++v) {
        double *p = P + v * f;
        for (int c = 0; c < f; ++c) p[c] = 1.0;
        for (int64_t k = row_ptr[v]; k < row_ptr[v + 1]; ++k) {
            const double *x = X + (int64_t)col[k] * f;
            for (int c = 0; c < f; ++c) p[c] *= x[c];
        }
    }
}

/* graphrole/features/extract.py:98-119 with 'min' / 'max' among the aggregations: column-wise
 * minimum / maximum over the neighbours' rows; no neighbours -> NaN -> fillna(0) (:113). */
void orc_aggregate_minmax(int64_t n, const int64_t *row_ptr, const int32_t *col, int f,
                          const double *X, double *LO, double *HI)
{
    for (int64_t v = 0; v < n; ++v) {
        double *lo = LO + v * f, *hi = HI + v * f;
        int64_t b = row_ptr[v], e = row_ptr[v + 1];
        if (e == b) {
            for (int c = 0; c < f; ++c) lo[c] = hi[c] = 0.0;
            continue;
        }
        const double *x0 = X + (int64_t)col[b] * f;
        for (int c = 0; c < f; ++c) lo[c] = hi[c] = x0[c];
        for (int64_t k = b + 1; k < e; ++k) {
            const double *x = X + (int64_t)col[k] * f;
            for (int c = 0; c < f; ++c) {
                if (x[c] < lo[c]) lo[c] = x[c];
                if (x[c] > hi[c]) hi[c] = x[c];
            }
        }
    }
}

/* graphrole/graph/interface/networkx.py:48-63 -- weighted row sums; the undirected degree adds
 * the self-loop weight once more (networkx counts a self-loop twice). */
void orc_rowsum(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *w,
                int add_self_loop, double *out)
{
    for (int64_t v = 0; v < n; ++v) {
        double s = 0.0;
        for (int64_t k = row_ptr[v]; k < row_ptr[v + 1]; ++k) {
            double x = w ? w[k] : 1.0;
            s += x;
            if (add_self_loop && col[k] == v) s += x;
        }
        out[v] = s;
    }
}

/* graphrole/graph/interface/networkx.py:71-83,115-123 -- ego-net internal / boundary weight.
 * ego(v) = {v} U row(v).  Rows of the ego members are walked in ascending member order and
 * CSR order inside a row. */
int orc_egonet(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *w,
               int directed, double *internal, double *external)
{
    int T = orc_threads();
    if (T > 16) T = 16;                              /* one n-entry stamp array per thread */
    if (n < 100000) T = 1;
    int64_t *marks = (int64_t *)calloc((size_t)n * (size_t)T, sizeof(int64_t));
    if (!marks) return -1;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(T)
    for (int64_t v = 0; v < n; ++v) {
#ifdef _OPENMP
        int64_t *mark = marks + (size_t)omp_get_thread_num() * (size_t)n;
#else
        int64_t *mark = marks;
#endif
        int64_t stamp = v + 1;
        mark[v] = stamp;
        for (int64_t k = row_ptr[v]; k < row_ptr[v + 1]; ++k) mark[col[k]] = stamp;
        double ins = 0.0, ext = 0.0;
        int v_done = 0;
        /* members in ascending order: merge v into the sorted row */
        int64_t k = row_ptr[v], e = row_ptr[v + 1];
        for (;;) {
            int64_t a;
            if (!v_done && (k >= e || v <= col[k])) {
                a = v; v_done = 1;
                if (k < e && col[k] == v) ++k;
            } else if (k < e) {
                a = col[k++];
            } else break;
            for (int64_t j = row_ptr[a]; j < row_ptr[a + 1]; ++j) {
                int64_t b = col[j];
                double x = w ? w[j] : 1.0;
                if (mark[b] == stamp) { if (directed || b >= a) ins += x; }
                else ext += x;
            }
        }
        internal[v] = ins;
        external[v] = ext;
    }
    free(marks);
    return 0;
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* graphrole/features/prune.py:13-56 -- vertical logarithmic binning.  Returns the number of
 * bins, or -1 on allocation failure / bad frac. */
int64_t orc_vertical_log_binning(int64_t n, const double *arr, double frac, int32_t *out)
{
    if (!(frac > 0.0 && frac < 1.0)) return -1;
    if (n == 0) return 0;
    double *s = (double *)malloc((size_t)n * sizeof(double));
    if (!s) return -1;
    memcpy(s, arr, (size_t)n * sizeof(double));
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    /* one threshold per bin; at most one bin per distinct value (the array is grown on demand) */
    int64_t cap = 64;
    double *thr = (double *)malloc((size_t)cap * sizeof(double));
    if (!thr) { free(s); return -1; }
    int64_t nb = 0, done = 0;
    while (done < n) {
        if (nb == cap) {
            cap *= 2;
            double *grown = (double *)realloc(thr, (size_t)cap * sizeof(double));
            if (!grown) { free(thr); free(s); return -1; }
            thr = grown;
        }
        int64_t size = (int64_t)(frac * (double)(n - done));
        if (size < 1) size = 1;
        int64_t pos = done + size - 1;               /* first unique with cumcount >= done+size */
        double hi = s[pos];
        while (pos + 1 < n && s[pos + 1] == hi) ++pos;   /* extend to the end of the tie run */
        thr[nb++] = hi;
        done = pos + 1;
    }
    free(s);
    for (int64_t i = 0; i < n; ++i) {
        double x = arr[i];
        int64_t lo = 0, hi = nb;                     /* first threshold >= x  == bin index */
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (thr[mid] < x) lo = mid + 1; else hi = mid; }
        out[i] = (int32_t)lo;
    }
    free(thr);
    return nb;
}

/* graphrole/features/prune.py:108 -- pdist(binned.T, 'chebychev').  B column-major F x n. */
void orc_chebyshev(int64_t n, int F, const int32_t *B, int64_t *D)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_threads())
    for (int p = 0; p < F; ++p) {
        D[p * F + p] = 0;
        for (int q = p + 1; q < F; ++q) {
            const int32_t *bp = B + (int64_t)p * n, *bq = B + (int64_t)q * n;
            int64_t mx = 0;
            for (int64_t i = 0; i < n; ++i) {
                int64_t d = (int64_t)bp[i] - (int64_t)bq[i];
                if (d < 0) d = -d;
                if (d > mx) mx = d;
            }
            D[p * F + q] = D[q * F + p] = mx;
        }
    }
}

// grx_host_linalg.hip -- HOST-side small dense algebra of the NNDSVDa initialisation.
//
// Between its device passes (two Gram matrices, one projection) the initialisation of the NMF
// works on k x F matrices with k, F <= a few dozen: eigen-decompositions of the Gram matrices, the
// randomised range finder of sklearn's randomized_svd (LU-normalised power iterations, QR, a small
// SVD; extmath.py:531-604, 349-351) and the NNDSVD column choices (_nmf.py:324-352).  Through
// numpy/scipy these ~25 LAPACK calls cost 0.6 ms of interpreter and wrapper overhead per fit --
// a quarter of the whole NMF phase on the bench graph -- so they are restated here in plain C++
// (cyclic Jacobi for the symmetric eigenproblem and the SVD: small, simple, and accurate to the
// last bits).  No device code in this file.
#include "grx_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

using Mat = std::vector<double>;                         // row-major

// C (m x n) = A (m x p) * B (p x n)
void matmul(int m, int p, int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < m; ++i) {
        double *c = C + (size_t)i * n;
        for (int j = 0; j < n; ++j) c[j] = 0.0;
        for (int l = 0; l < p; ++l) {
            const double a = A[(size_t)i * p + l];
            const double *b = B + (size_t)l * n;
            for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
    }
}

// C (m x n) = A^T (A is p x m) * B (p x n)
void matmul_tn(int m, int p, int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) C[(size_t)i * n + j] = 0.0;
    for (int l = 0; l < p; ++l)
        for (int i = 0; i < m; ++i) {
            const double a = A[(size_t)l * m + i];
            const double *b = B + (size_t)l * n;
            double *c = C + (size_t)i * n;
            for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
}

// Symmetric eigen-decomposition by cyclic Jacobi rotations.  A (n x n, row-major, symmetric) is
// destroyed; w ascending, V[:, j] the eigenvector of w[j] (row-major n x n).
void jacobi_eigh(int n, double *A, double *w, double *V)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
    // A pair is converged when |a_pq| <= eps sqrt(|a_pp a_qq|) (rotating it would not move either eigenvalue in
    // double precision; the relative criterion keeps the small eigenvalues of a graded Gram matrix accurate);
    // the iteration ends with the first sweep that rotates nothing.  A bound on the off-diagonal NORM far below
    // eps ||A|| is never met once n is a few dozen: rounding noise alone keeps it up and all 64 sweeps run.
    const double eps = 2.220446049250313e-16;
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                if (apq == 0.0 || std::fabs(apq) <= eps * std::sqrt(std::fabs(app * aqq))) continue;
                rotated = true;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {                       // columns p, q
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {                       // rows p, q
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
        }
        if (!rotated) break;
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::vector<double> diag(n);
    for (int i = 0; i < n; ++i) diag[i] = A[(size_t)i * n + i];
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return diag[a] < diag[b]; });
    Mat Vs((size_t)n * n);
    for (int j = 0; j < n; ++j) {
        w[j] = diag[order[j]];
        for (int i = 0; i < n; ++i) Vs[(size_t)i * n + j] = V[(size_t)i * n + order[j]];
    }
    std::memcpy(V, Vs.data(), sizeof(double) * n * n);
}

// Symmetric eigen-decomposition in O(n^3): Householder reduction to tridiagonal form with the
// transformations accumulated in place, then implicit-shift QL sweeps on the tridiagonal (the classical
// EISPACK pair).  Used above 32 columns, where the ~10 Jacobi sweeps (9 n^3 flops each) would cost tens of
// milliseconds; same interface as jacobi_eigh (A destroyed, w ascending, eigenvectors in the columns of V).
// Absolute accuracy eps * ||A|| like LAPACK's tridiagonal drivers; the second Gram pass of the
// initialisation re-orthogonalises, so small eigenvalues need no more than that.
bool tridiagonal_eigh(int n, double *A, double *w, double *V)
{
    std::vector<double> e(n, 0.0), gv(n, 0.0);
    double *a = A, *d = w;
    auto at = [&](int i, int j) -> double & { return a[(size_t)i * n + j]; };
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0;
        if (l > 0) {
            double scale = 0.0;
            for (int k = 0; k <= l; ++k) scale += std::fabs(at(i, k));
            if (scale == 0.0) {
                e[i] = at(i, l);
            } else {
                for (int k = 0; k <= l; ++k) { at(i, k) /= scale; h += at(i, k) * at(i, k); }
                double f = at(i, l);
                double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                at(i, l) = f - g;
                // p = A u over the stored lower triangle, rows only (contiguous): the k <= j terms row by row, then
                // row k adds its share to every p_j, j < k -- per p_j the additions run in the same order as the
                // textbook column walk
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    at(j, i) = at(i, j) / h;
                    double g = 0.0;
                    for (int k = 0; k <= j; ++k) g += at(j, k) * at(i, k);
                    e[j] = g;
                }
                for (int k = 1; k <= l; ++k) {
                    const double uk = at(i, k);
                    const double *rowk = &at(k, 0);
                    for (int j = 0; j < k; ++j) e[j] += rowk[j] * uk;
                }
                for (int j = 0; j <= l; ++j) {
                    e[j] /= h;
                    f += e[j] * at(i, j);
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = at(i, j);
                    const double g = e[j] = e[j] - hh * f;
                    for (int k = 0; k <= j; ++k) at(j, k) -= f * e[k] + g * at(i, k);
                }
            }
        } else {
            e[i] = at(i, l);
        }
        d[i] = h;
    }
    d[0] = 0.0;
    e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        if (d[i] != 0.0) {
            // g_j = sum_k a(i,k) a(k,j), then a(k,j) -= g_j a(k,i): rows of the block, k outermost
            std::fill(gv.begin(), gv.begin() + i, 0.0);
            for (int k = 0; k < i; ++k) {
                const double uk = at(i, k);
                const double *rowk = &at(k, 0);
                for (int j = 0; j < i; ++j) gv[j] += uk * rowk[j];
            }
            for (int k = 0; k < i; ++k) {
                const double vk = at(k, i);
                double *rowk = &at(k, 0);
                for (int j = 0; j < i; ++j) rowk[j] -= gv[j] * vk;
            }
        }
        d[i] = at(i, i);
        at(i, i) = 1.0;
        for (int j = 0; j < i; ++j) at(j, i) = at(i, j) = 0.0;
    }
    // QL with implicit shifts on (d, e); the rotations mix two eigenvector columns -- held as the ROWS of zt so
    // that the update streams through contiguous memory
    Mat zt((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) zt[(size_t)j * n + i] = at(i, j);
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    bool ok = true;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 80) { ok = false; break; }
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    e[i + 1] = r = std::hypot(f, g);
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    double *z0 = &zt[(size_t)i * n], *z1 = z0 + n;
                    for (int k = 0; k < n; ++k) {
                        const double hi = z1[k], lo = z0[k];
                        z1[k] = s * lo + c * hi;
                        z0[k] = c * lo - s * hi;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
        if (!ok) break;
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] < d[y]; });
    std::vector<double> ds(d, d + n);
    for (int j = 0; j < n; ++j) {
        w[j] = ds[order[j]];
        for (int i = 0; i < n; ++i) V[(size_t)i * n + j] = zt[(size_t)order[j] * n + i];
    }
    return ok;
}

constexpr int JACOBI_MAX_N = 32;

// eigen-decomposition used by the initialisation: Jacobi for small matrices, tridiagonal QL above
void sym_eigh(int n, double *A, double *w, double *V)
{
    if (n <= JACOBI_MAX_N) { jacobi_eigh(n, A, w, V); return; }
    Mat keep(A, A + (size_t)n * n);
    if (!tridiagonal_eigh(n, A, w, V)) jacobi_eigh(n, keep.data(), w, V);   // QL did not converge (never seen)
}

// scipy.linalg.lu(a, permute_l=True)[0]: the m x min(m, k) factor P L of a (m x k) = (P L) U,
// LU with partial pivoting (first maximum, like LAPACK's idamax).  a is destroyed.
void lu_permuted_l(int m, int k, double *a, double *PL)
{
    const int kk = std::min(m, k);
    std::vector<int> perm(m);
    std::iota(perm.begin(), perm.end(), 0);
    for (int j = 0; j < kk; ++j) {
        int piv = j;
        double best = std::fabs(a[(size_t)j * k + j]);
        for (int i = j + 1; i < m; ++i) {
            const double v = std::fabs(a[(size_t)i * k + j]);
            if (v > best) { best = v; piv = i; }
        }
        if (piv != j) {
            for (int c = 0; c < k; ++c) std::swap(a[(size_t)j * k + c], a[(size_t)piv * k + c]);
            std::swap(perm[j], perm[piv]);
        }
        const double d = a[(size_t)j * k + j];
        if (d != 0.0) {
            for (int i = j + 1; i < m; ++i) {
                const double f = a[(size_t)i * k + j] / d;
                a[(size_t)i * k + j] = f;
                for (int c = j + 1; c < k; ++c) a[(size_t)i * k + c] -= f * a[(size_t)j * k + c];
            }
        }
    }
    // row i of the pivoted L belongs to original row perm[i]
    for (int i = 0; i < m; ++i) {
        double *dst = PL + (size_t)perm[i] * kk;
        for (int j = 0; j < kk; ++j) dst[j] = (j < i) ? a[(size_t)i * k + j] : (j == i ? 1.0 : 0.0);
    }
}

// Economic QR by Householder reflections: Q (m x kk, kk = min(m, k)) of a (m x k); a is destroyed.
void qr_q(int m, int k, double *a, double *Q)
{
    const int kk = std::min(m, k);
    std::vector<double> tau(kk, 0.0);
    for (int j = 0; j < kk; ++j) {
        double nrm = 0.0;
        for (int i = j; i < m; ++i) nrm += a[(size_t)i * k + j] * a[(size_t)i * k + j];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = a[(size_t)j * k + j];
        const double beta = alpha >= 0.0 ? -nrm : nrm;
        tau[j] = (beta - alpha) / beta;
        const double scale = 1.0 / (alpha - beta);
        for (int i = j + 1; i < m; ++i) a[(size_t)i * k + j] *= scale;   // v (v_j = 1 implicit)
        a[(size_t)j * k + j] = beta;
        for (int c = j + 1; c < k; ++c) {
            double s = a[(size_t)j * k + c];
            for (int i = j + 1; i < m; ++i) s += a[(size_t)i * k + j] * a[(size_t)i * k + c];
            s *= tau[j];
            a[(size_t)j * k + c] -= s;
            for (int i = j + 1; i < m; ++i) a[(size_t)i * k + c] -= s * a[(size_t)i * k + j];
        }
    }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < kk; ++j) Q[(size_t)i * kk + j] = (i == j) ? 1.0 : 0.0;
    for (int j = kk - 1; j >= 0; --j) {
        if (tau[j] == 0.0) continue;
        for (int c = 0; c < kk; ++c) {
            double s = Q[(size_t)j * kk + c];
            for (int i = j + 1; i < m; ++i) s += a[(size_t)i * k + j] * Q[(size_t)i * kk + c];
            s *= tau[j];
            Q[(size_t)j * kk + c] -= s;
            for (int i = j + 1; i < m; ++i) Q[(size_t)i * kk + c] -= s * a[(size_t)i * k + j];
        }
    }
}

// Thin SVD of B (m x n, m <= n): B = U diag(s) Vt, s descending, by one-sided Jacobi on the
// columns of A = B^T.  U m x m, Vt m x n.
void jacobi_svd(int m, int n, const double *B, double *U, double *s, double *Vt)
{
    // the m vectors being orthogonalised are the ROWS of A (= B) and of Rt (= R^T): contiguous streams
    Mat A(B, B + (size_t)m * n), Rt((size_t)m * m);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) Rt[(size_t)i * m + j] = (i == j) ? 1.0 : 0.0;
    const double tol = std::sqrt((double)m) * 2.220446049250313e-16;
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < m; ++p) {
            for (int q = p + 1; q < m; ++q) {
                double *ap = &A[(size_t)p * n], *aq = &A[(size_t)q * n];
                double app = 0.0, aqq = 0.0, apq = 0.0;
                for (int k = 0; k < n; ++k) {
                    const double x = ap[k], y = aq[k];
                    app += x * x; aqq += y * y; apq += x * y;
                }
                // converged pair: |cos| below sqrt(m) eps (the dgesvj criterion).  A tighter bound never fires for
                // the noise-level columns of a rank-deficient B and burns all 64 sweeps.
                if (apq == 0.0 || std::fabs(apq) <= tol * std::sqrt(app * aqq)) continue;
                rotated = true;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {
                    const double x = ap[k], y = aq[k];
                    ap[k] = c * x - sn * y;
                    aq[k] = sn * x + c * y;
                }
                double *rp = &Rt[(size_t)p * m], *rq = &Rt[(size_t)q * m];
                for (int k = 0; k < m; ++k) {
                    const double x = rp[k], y = rq[k];
                    rp[k] = c * x - sn * y;
                    rq[k] = sn * x + c * y;
                }
            }
        }
        if (!rotated) break;
    }
    std::vector<double> nrm(m);
    for (int j = 0; j < m; ++j) {
        double v = 0.0;
        for (int k = 0; k < n; ++k) v += A[(size_t)j * n + k] * A[(size_t)j * n + k];
        nrm[j] = std::sqrt(v);
    }
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nrm[a] > nrm[b]; });
    for (int j = 0; j < m; ++j) {
        const int o = order[j];
        s[j] = nrm[o];
        for (int i = 0; i < m; ++i) U[(size_t)i * m + j] = Rt[(size_t)o * m + i];
        for (int k = 0; k < n; ++k) Vt[(size_t)j * n + k] = nrm[o] > 0.0 ? A[(size_t)o * n + k] / nrm[o] : 0.0;
    }
}

}  // namespace

extern "C" {

int grx_host_eigh(int n, const double *A_in, double *w, double *V)
{
    GRX_REQUIRE(n >= 1 && A_in && w && V, "grx_host_eigh: bad arguments");
    Mat A(A_in, A_in + (size_t)n * n);
    sym_eigh(n, A.data(), w, V);
    return GRX_OK;
}

// Whitening transform of the first Gram matrix G1 = X^T X (F x F):  eigen-pairs above the
// numerical floor, T1 (F x k, row-major with k columns) with (X T1)^T (X T1) ~ I.
// lam_keep [k], V_keep [F x k] with X^T X = V_keep diag(lam_keep) V_keep^T on the kept subspace (what
// grx_host_range_finder rebuilds M = Q^T X from).  Returns k (0: the feature matrix is numerically zero).
//
// equilibrate = false: eigh(G1) itself, floor = lam_max * F * 16 eps.  On a GRADED table (column norms decades
// apart: a degree column next to a mean of means) that floor is set by the largest column, and the directions of the
// small columns fall under it -- harmless while the r leading directions survive, and it also keeps the noise
// directions of exact dependencies out (config 5: 41 of 115 directions dropped, factors 1e-12 from the oracle).
// equilibrate = true (round 4): the scaling is removed first and EXACTLY -- D = diag(2^e_j), 2^e_j <= ||x_j|| < 2^(e_j+1),
// Gs = D^-1 G1 D^-1 formed without rounding, eigh(Gs) = Vs diag(lam) Vs^T, T1 = D^-1 Vs / sqrt(lam), V_keep = D Vs --
// so that the floor judges the column-equilibrated table.  It keeps every direction of a table whose conditioning is
// pure column scaling (tools/fuzz_rolx.py seed 403 case 103: r = F = 17, the seventeenth direction 1.9e-14 of the
// first), but also the rounding-noise directions of a wide table with dependent columns (config 5: 113 of 115 kept,
// factors wrong) -- so it is the FALLBACK: grx_host_whiten_for_rank uses it only when the plain decomposition keeps
// fewer directions than the factorisation asks for.
static int whiten_impl(int F, const double *G1, bool equilibrate, double *T1, double *lam_keep, double *V_keep, int *k_out)
{
    std::vector<double> scale(F, 1.0);
    for (int j = 0; j < F && equilibrate; ++j) {
        const double d = G1[(size_t)j * F + j];
        if (d > 0.0 && std::isfinite(d)) {
            int e = 0;
            (void)std::frexp(std::sqrt(d), &e);                  // sqrt(d) = m * 2^e, m in [0.5, 1)
            scale[j] = std::ldexp(1.0, e - 1);                   // 2^(e-1) <= sqrt(d) < 2^e
        }
    }
    Mat A((size_t)F * F), V((size_t)F * F);
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < F; ++j) A[(size_t)i * F + j] = G1[(size_t)i * F + j] / scale[i] / scale[j];   // exact: powers of two
    std::vector<double> w(F);
    sym_eigh(F, A.data(), w.data(), V.data());
    const double lam_max = std::max(w[F - 1], 0.0);
    const double floor = lam_max * F * 2.220446049250313e-16 * 16;
    int k = 0;
    for (int j = 0; j < F; ++j) if (w[j] > floor) ++k;
    *k_out = k;
    int col = 0;
    for (int j = 0; j < F; ++j) {
        if (!(w[j] > floor)) continue;
        lam_keep[col] = w[j];
        const double inv = 1.0 / std::sqrt(w[j]);
        for (int i = 0; i < F; ++i) {
            V_keep[(size_t)i * k + col] = V[(size_t)i * F + j] * scale[i];
            T1[(size_t)i * k + col] = V[(size_t)i * F + j] * inv / scale[i];
        }
        ++col;
    }
    return GRX_OK;
}

int grx_host_whiten(int F, const double *G1, double *T1, double *lam_keep, double *V_keep, int *k_out)
{
    GRX_REQUIRE(F >= 1 && G1 && T1 && lam_keep && V_keep && k_out, "grx_host_whiten: bad arguments");
    return whiten_impl(F, G1, false, T1, lam_keep, V_keep, k_out);
}

// The whitening for a factorisation of rank r: the plain decomposition; if it keeps fewer than min(r, F) directions
// (a graded table whose small columns fell under the floor), the column-equilibrated one.
int grx_host_whiten_for_rank(int F, const double *G1, int r, double *T1, double *lam_keep, double *V_keep, int *k_out)
{
    GRX_REQUIRE(F >= 1 && r >= 0 && G1 && T1 && lam_keep && V_keep && k_out, "grx_host_whiten_for_rank: bad arguments");
    int rc = whiten_impl(F, G1, false, T1, lam_keep, V_keep, k_out);
    if (rc != GRX_OK || *k_out >= (r < F ? r : F)) return rc;
    int k2 = 0;
    std::vector<double> T2((size_t)F * F), lam2(F), V2((size_t)F * F);
    rc = whiten_impl(F, G1, true, T2.data(), lam2.data(), V2.data(), &k2);
    if (rc != GRX_OK || k2 <= *k_out) return rc;                 // nothing gained: keep the plain result
    *k_out = k2;
    std::memcpy(T1, T2.data(), (size_t)F * k2 * 8);
    std::memcpy(V_keep, V2.data(), (size_t)F * k2 * 8);
    std::memcpy(lam_keep, lam2.data(), (size_t)k2 * 8);
    return GRX_OK;
}

// From the second Gram matrix G2 = (X T1)^T (X T1) (k x k): T = T1 V2 / sqrt(lam2) (X T = Q
// orthonormal), M = Q^T X = diag(sqrt lam2) V2^T diag(sqrt lam) V^T (k x F), then sklearn's
// randomized_svd of X = Q M carried out on M (extmath.py:531-604): Z = T Us (F x r), S [r],
// Vt [r x F] (before svd_flip).  omega: F x n_over Gaussian test matrix (row-major), n_iter
// LU-normalised power iterations.
int grx_host_range_finder(int F, int k, const double *T1, const double *lam_keep, const double *V_keep,
                          const double *G2, const double *omega, int n_over, int r, int n_iter, double *Z,
                          double *S, double *Vt_out)
{
    GRX_REQUIRE(F >= 1 && k >= 1 && k <= F && n_over >= 1 && r >= 1 && n_iter >= 0, "grx_host_range_finder: bad shape");
    GRX_REQUIRE(T1 && lam_keep && V_keep && G2 && omega && Z && S && Vt_out, "grx_host_range_finder: NULL pointer");
    Mat A(G2, G2 + (size_t)k * k), V2((size_t)k * k);
    std::vector<double> lam2(k);
    sym_eigh(k, A.data(), lam2.data(), V2.data());
    // T = (T1 V2) / sqrt(lam2)           F x k
    Mat T((size_t)F * k);
    matmul(F, k, k, T1, V2.data(), T.data());
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < k; ++j) T[(size_t)i * k + j] /= std::sqrt(lam2[j]);
    // M = (sqrt(lam2)[:, None] * V2^T) @ (sqrt(lam)[:, None] * V^T)         k x F
    Mat L((size_t)k * k), Rm((size_t)k * F), M((size_t)k * F);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) L[(size_t)i * k + j] = std::sqrt(lam2[i]) * V2[(size_t)j * k + i];
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < F; ++j) Rm[(size_t)i * F + j] = std::sqrt(lam_keep[i]) * V_keep[(size_t)j * k + i];
    matmul(k, k, F, L.data(), Rm.data(), M.data());
    // range finder on M
    int w = n_over;                                               // current number of columns of Qs
    Mat Qs(omega, omega + (size_t)F * n_over), tmp, PL;
    for (int it = 0; it < n_iter; ++it) {
        tmp.assign((size_t)k * w, 0.0);
        matmul(k, F, w, M.data(), Qs.data(), tmp.data());         // k x w
        const int w1 = std::min(k, w);
        PL.assign((size_t)k * w1, 0.0);
        lu_permuted_l(k, w, tmp.data(), PL.data());               // k x w1
        tmp.assign((size_t)F * w1, 0.0);
        matmul_tn(F, k, w1, M.data(), PL.data(), tmp.data());     // M^T (F x k) @ PL -> F x w1
        const int w2 = std::min(F, w1);
        Qs.assign((size_t)F * w2, 0.0);
        lu_permuted_l(F, w1, tmp.data(), Qs.data());              // F x w2
        w = w2;
    }
    tmp.assign((size_t)k * w, 0.0);
    matmul(k, F, w, M.data(), Qs.data(), tmp.data());             // k x w
    const int m = std::min(k, w);
    Mat Q((size_t)k * m);
    qr_q(k, w, tmp.data(), Q.data());                             // k x m
    Mat B((size_t)m * F);
    matmul_tn(m, k, F, Q.data(), M.data(), B.data());             // Q^T M: m x F
    Mat Uh((size_t)m * m), Vt((size_t)m * F);
    std::vector<double> s(m);
    jacobi_svd(m, F, B.data(), Uh.data(), s.data(), Vt.data());
    Mat Us((size_t)k * m);
    matmul(k, m, m, Q.data(), Uh.data(), Us.data());              // k x m
    // Z = T Us[:, :r]  (F x r), zero padded when rank < r
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < r; ++j) {
            double v = 0.0;
            if (j < m)
                for (int l = 0; l < k; ++l) v += T[(size_t)i * k + l] * Us[(size_t)l * m + j];
            Z[(size_t)i * r + j] = v;
        }
    for (int j = 0; j < r; ++j) {
        S[j] = j < m ? s[j] : 0.0;
        for (int c = 0; c < F; ++c) Vt_out[(size_t)j * F + c] = j < m ? Vt[(size_t)j * F + c] : 0.0;
    }
    return GRX_OK;
}

// Fewer nodes than features (n < F): sklearn's randomized_svd takes its transposed branch (extmath.py:565-569,
// 587-604: the range finder runs on M = X^T, the factors are swapped back) and every matrix is small -- X itself is
// at most GRX_MAX_NMF_FEATURES squared.  The whole SVD part of the initialisation on the host, from the library's own
// routines: U (n x r, before svd_flip), S [r], V (r x F).  omega: n x n_over Gaussian test matrix.
int grx_host_small_svd(int n, int F, const double *X, const double *omega, int n_over, int r, int n_iter, double *U_out,
                       double *S, double *V_out)
{
    GRX_REQUIRE(n >= 1 && F >= 1 && n_over >= 1 && r >= 1 && n_iter >= 0, "grx_host_small_svd: bad shape");
    GRX_REQUIRE(X && omega && U_out && S && V_out, "grx_host_small_svd: NULL pointer");
    // M = X^T: F x n
    Mat M((size_t)F * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < F; ++j) M[(size_t)j * n + i] = X[(size_t)i * F + j];
    int w = n_over;
    Mat Qs(omega, omega + (size_t)n * n_over), tmp, PL;
    for (int it = 0; it < n_iter; ++it) {
        tmp.assign((size_t)F * w, 0.0);
        matmul(F, n, w, M.data(), Qs.data(), tmp.data());          // M Qs: F x w
        const int w1 = std::min(F, w);
        PL.assign((size_t)F * w1, 0.0);
        lu_permuted_l(F, w, tmp.data(), PL.data());                // F x w1
        tmp.assign((size_t)n * w1, 0.0);
        matmul_tn(n, F, w1, M.data(), PL.data(), tmp.data());      // M^T PL: n x w1
        const int w2 = std::min(n, w1);
        Qs.assign((size_t)n * w2, 0.0);
        lu_permuted_l(n, w1, tmp.data(), Qs.data());               // n x w2
        w = w2;
    }
    tmp.assign((size_t)F * w, 0.0);
    matmul(F, n, w, M.data(), Qs.data(), tmp.data());              // F x w
    const int m = std::min(F, w);
    Mat Q((size_t)F * m);
    qr_q(F, w, tmp.data(), Q.data());                              // F x m
    Mat B((size_t)m * n);
    matmul_tn(m, F, n, Q.data(), M.data(), B.data());              // Q^T M: m x n
    // thin SVD of B; jacobi_svd wants no more rows than columns
    const int mm = std::min(m, n);
    Mat Uh((size_t)m * mm), Vt((size_t)mm * n);
    std::vector<double> s(mm);
    if (m <= n) {
        jacobi_svd(m, n, B.data(), Uh.data(), s.data(), Vt.data());
    } else {
        Mat Bt((size_t)n * m), U2((size_t)n * n), Vt2((size_t)n * m);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) Bt[(size_t)j * m + i] = B[(size_t)i * n + j];
        jacobi_svd(n, m, Bt.data(), U2.data(), s.data(), Vt2.data());   // B^T = U2 s Vt2  =>  B = Vt2^T s U2^T
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) Uh[(size_t)i * n + j] = Vt2[(size_t)j * m + i];
        for (int j = 0; j < n; ++j)
            for (int c = 0; c < n; ++c) Vt[(size_t)j * n + c] = U2[(size_t)c * n + j];
    }
    // M = X^T ~ (Q Uh) s Vt  =>  X ~ Vt^T s (Q Uh)^T:  U = Vt^T (n x r),  V = (Q Uh)^T (r x F)
    for (int j = 0; j < r; ++j) {
        S[j] = j < mm ? s[j] : 0.0;
        for (int i = 0; i < n; ++i) U_out[(size_t)i * r + j] = j < mm ? Vt[(size_t)j * n + i] : 0.0;
        for (int c = 0; c < F; ++c) {
            double v = 0.0;
            if (j < mm)
                for (int l = 0; l < m; ++l) v += Q[(size_t)c * m + l] * Uh[(size_t)l * mm + j];
            V_out[(size_t)j * F + c] = v;
        }
    }
    return GRX_OK;
}

// NNDSVD column choices (_nmf.py:324-352) from the statistics of the raw U = X Z columns:
// stats[j] = {signed max-|.| entry, its row, sum sq of the positive part, of the negative part}.
// Outputs sign[r], scale[r] for grx_nndsvd_apply and H (r x F) before thresholding.
int grx_host_nndsvd_plan(int r, int F, const double *S, const double *Vt, const double *stats, double *sign,
                         double *scale, double *H)
{
    GRX_REQUIRE(r >= 1 && F >= 1 && S && Vt && stats && sign && scale && H, "grx_host_nndsvd_plan: bad arguments");
    for (int j = 0; j < r; ++j) { sign[j] = 0.0; scale[j] = 0.0; }
    for (int i = 0; i < r * F; ++i) H[i] = 0.0;
    scale[0] = std::sqrt(S[0]);
    for (int c = 0; c < F; ++c) H[c] = std::sqrt(S[0]) * std::fabs(Vt[c]);
    std::vector<double> y(F);
    for (int j = 1; j < r; ++j) {
        double flip = stats[(size_t)j * 4] > 0.0 ? 1.0 : (stats[(size_t)j * 4] < 0.0 ? -1.0 : 1.0);
        double yp = 0.0, yn = 0.0;
        for (int c = 0; c < F; ++c) {
            y[c] = Vt[(size_t)j * F + c] * flip;
            if (y[c] > 0.0) yp += y[c] * y[c]; else yn += y[c] * y[c];
        }
        const double x_p = std::sqrt(flip > 0.0 ? stats[(size_t)j * 4 + 2] : stats[(size_t)j * 4 + 3]);
        const double x_n = std::sqrt(flip > 0.0 ? stats[(size_t)j * 4 + 3] : stats[(size_t)j * 4 + 2]);
        const double y_p = std::sqrt(yp), y_n = std::sqrt(yn);
        const double m_p = x_p * y_p, m_n = x_n * y_n;
        double x_nrm, y_nrm, sigma, part;
        if (m_p > m_n) { x_nrm = x_p; y_nrm = y_p; sigma = m_p; part = 1.0; }
        else { x_nrm = x_n; y_nrm = y_n; sigma = m_n; part = -1.0; }
        const double lbd = std::sqrt(S[j] * sigma);
        if (!std::isfinite(lbd) || x_nrm == 0.0 || y_nrm == 0.0 || !std::isfinite(lbd / y_nrm)) {
            sign[j] = 1.0; scale[j] = 0.0;                       // degenerate component -> all fill
            continue;
        }
        sign[j] = flip * part;
        scale[j] = lbd / x_nrm;
        for (int c = 0; c < F; ++c) {
            const double v = part > 0.0 ? std::max(y[c], 0.0) : std::fabs(std::min(y[c], 0.0));
            H[(size_t)j * F + c] = lbd * (v / y_nrm);
        }
    }
    return GRX_OK;
}

}  // extern "C"

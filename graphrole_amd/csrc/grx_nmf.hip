// grx_nmf.hip -- RolX NMF: NNDSVDa building blocks and the multiplicative-update loop.
//
// Reference call site: graphrole/roles/factor.py:19,24 -> sklearn NMF(solver='mu',
// init='nndsvda') (sklearn/decomposition/_nmf.py).  Shapes: X is N x F with N up to millions
// and F <= ~100, rank r <= 16: every pass below streams X (and W) once from HBM and is bound by
// HBM bandwidth (2-3 flop/byte, far under the fp64 ridge), so the arithmetic is plain fp64 FMA
// on LDS-staged row tiles; the reduction over the N axis is a fixed-order tree (per-workgroup
// partials, then a fixed-order sum) -- bitwise reproducible, no floating-point atomics.
//
//   gather_columns_kernel   column pointers -> contiguous F x ld
//   gram_kernel             G = (X T)^T (X T)                 (init: orthogonal factorisation)
//   project_kernel          U = X Z + per-column statistics   (init: singular vectors, svd_flip,
//                                                              NNDSVD +/- norms)
//   nndsvd_apply_kernel     _nmf.py:324-359 elementwise part
//   nmf_w_pass_mfma_kernel  W <- W*(XH^T)/(W HH^T) fused with A = W^T X, B = W^T W partials (fp64 MFMA)
//   nmf_h_update_kernel     H <- H*A/(B H)
//   nmf_residual_kernel     ||X - WH||_F^2
#include "grx_common.h"

#include <array>
#include <utility>

namespace {

constexpr double NMF_EPSILON = 1.1920928955078125e-07;   // np.finfo(np.float32).eps, _nmf.py:39
constexpr int MAX_R = GRX_MAX_ROLES;                      // 32
constexpr int MFMA_R = 16;                                // one MFMA tile of roles: what the fused kernels factor with
constexpr int MAX_F = 120;                                // single-launch Gram / register-resident W-pass
constexpr int MAX_F_WIDE = GRX_MAX_NMF_FEATURES;          // 480: r * F * 8 <= 60 KiB of LDS at r = 16

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_columns_kernel(int64_t n, int F, GrxPtrTable ptr_tab,
                                                             double *__restrict__ out, int64_t ld)
{
    const double *src = reinterpret_cast<const double *>(ptr_tab.p[blockIdx.y]);
    double *dst = out + (size_t)blockIdx.y * ld;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// out[p] = sum_b partial[p*nb + b]: one wavefront per output, per-lane sequential partial sums
// then a fixed butterfly (bitwise reproducible).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ partial,
                                                              int nb, int P, double *__restrict__ out)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const double *src = partial + (size_t)p * nb;
    double s = 0.0;
    // sixteen loads in flight, then the additions in the same order as a plain loop (a lane's
    // dependent load-add chain was most of this kernel's 9 us)
    for (int b0 = lane; b0 < nb; b0 += 64 * 16) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = b0 + 64 * j;
            v[j] = (b < nb) ? src[b] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
    }
    s = grx_group_sum<64>(s);
    if (lane == 0) out[p] = s;
}

// One entry of the H update, H[k][c] * (A[k][c] / sum_l B[k][l] H[l][c]) with a zero denominator replaced by
// EPSILON (_nmf.py:638-641).  The products are accumulated with explicit fused multiply-adds so that the
// stand-alone kernel and the copy in the W-pass prologue give the same bits whatever the compiler contracts.
__device__ __forceinline__ double h_update_value(const double *H, const double *sB, double a, int idx, int F, int r)
{
    const int k = idx / F, c = idx % F;
    double denom = 0.0;
    for (int l = 0; l < r; ++l) denom = __fma_rn(sB[k * r + l], H[l * F + c], denom);
    if (denom == 0.0) denom = NMF_EPSILON;
    return H[idx] * (a / denom);
}

// ---------------------------------------------------------------------------------------
// Gram of the (optionally transformed) rows
// ---------------------------------------------------------------------------------------
constexpr int GR_TR = 64;
constexpr int GR_LD = GR_TR + 1;
constexpr int GR_PSLOTS = (MAX_F * (MAX_F + 1) / 2 + 255) / 256;    // 29

// The k outputs of a launch are the columns [ja, ja + na) followed by [jb, jb + k - na) of Y = X T
// (T has ldt columns; HAS_T = false: Y = X): Gram matrices wider than MAX_F are assembled from
// launches over pairs of column groups (see grx_gram).
template <bool HAS_T, int YS>
__global__ __launch_bounds__(256) void gram_kernel(int64_t row_begin, int64_t row_end, int F, int k,
                                                   const double *__restrict__ X, int64_t ldx,
                                                   const double *__restrict__ T, int t_in_lds,
                                                   double *__restrict__ partial, int ldt, int ja, int na, int jb)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    double *sY = gsm;                                  // k * GR_LD
    double *sT = gsm + k * GR_LD;                      // F * k (only when t_in_lds)
    __shared__ double xred[4];
    const int t = threadIdx.x, i = t & 63, g = t >> 6;
    if (HAS_T && t_in_lds) {                           // only when ldt == k (single launch)
        for (int idx = t; idx < F * k; idx += 256) sT[idx] = T[idx];
        __syncthreads();
    }
    const double *Tsrc = (HAS_T && t_in_lds) ? sT : T;
    auto col_of = [&](int j) { return j < na ? ja + j : jb + (j - na); };   // local output -> column of Y
    const int npairs = k * (k + 1) / 2;
    int pq[GR_PSLOTS];
    double acc[GR_PSLOTS];
#pragma unroll
    for (int s = 0; s < GR_PSLOTS; ++s) {
        const int id = t + 256 * s;
        acc[s] = 0.0;
        pq[s] = -1;
        if (id < npairs) {
            // id = b(b+1)/2 + a with a <= b
            int b = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
            while (b * (b + 1) / 2 > id) --b;
            while ((b + 1) * (b + 2) / 2 <= id) ++b;
            pq[s] = ((id - b * (b + 1) / 2) << 8) | b;
        }
    }
    double xsum = 0.0;
    for (int64_t r0 = row_begin + (int64_t)blockIdx.x * GR_TR; r0 < row_end;
         r0 += (int64_t)gridDim.x * GR_TR) {
        const bool live = (r0 + i) < row_end;
        __syncthreads();
        if (HAS_T) {
            double y[YS];
#pragma unroll
            for (int s = 0; s < YS; ++s) y[s] = 0.0;
            for (int c = 0; c < F; ++c) {
                const double x = live ? X[(size_t)c * ldx + r0 + i] : 0.0;
                const double *Tc = Tsrc + (size_t)c * ldt;
#pragma unroll
                for (int s = 0; s < YS; ++s) {
                    const int j = g + 4 * s;
                    if (j < k) y[s] += x * Tc[col_of(j)];
                }
            }
#pragma unroll
            for (int s = 0; s < YS; ++s) {
                const int j = g + 4 * s;
                if (j < k) sY[j * GR_LD + i] = y[s];
            }
        } else {
            for (int c = g; c < k; c += 4) {
                const double x = live ? X[(size_t)col_of(c) * ldx + r0 + i] : 0.0;
                sY[c * GR_LD + i] = x;
                xsum += x;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GR_PSLOTS; ++s) {
            if (pq[s] >= 0) {
                const double *a = sY + (pq[s] >> 8) * GR_LD;
                const double *b = sY + (pq[s] & 0xFF) * GR_LD;
                double v = acc[s];
#pragma unroll 8
                for (int ii = 0; ii < GR_TR; ++ii) v += a[ii] * b[ii];
                acc[s] = v;
            }
        }
    }
    // partial layout [npairs + 1][gridDim.x]
#pragma unroll
    for (int s = 0; s < GR_PSLOTS; ++s) {
        const int id = t + 256 * s;
        if (id < npairs) partial[(size_t)id * gridDim.x + blockIdx.x] = acc[s];
    }
    xsum = grx_group_sum<64>(xsum);
    if (i == 0) xred[g] = xsum;
    __syncthreads();
    if (t == 0) partial[(size_t)npairs * gridDim.x + blockIdx.x] = ((xred[0] + xred[1]) + xred[2]) + xred[3];
}

// ---------------------------------------------------------------------------------------
// The same Gram matrix on the matrix cores (v_mfma_f64_16x16x4_f64) for F, k <= 48: one wavefront
// per 16-row sub-tile.  Y^T (k x 16) = T^T X^T is one MFMA chain per 16-column tile of k with the
// X operand straight from global memory; the sub-tile of Y is then written to the wave's LDS
// slice and re-read transposed as both operands of G[j][j'] += sum_i Y[i][j] Y[i][j']
// (upper-triangular tile pairs only).  NQ = ceil(F / 4), KT = ceil(k / 16).
typedef double gv4d __attribute__((ext_vector_type(4)));
constexpr int GM_LD = 17;
constexpr int GM_MAX_NQ = 12, GM_MAX_KT = 3;

static inline size_t gram_mfma_lds_doubles(int KT)
{
    const size_t tile = (size_t)16 * KT * GM_LD, red = (size_t)(KT * (KT + 1) / 2) * 256;
    return 4 * (tile > red ? tile : red) + 8;
}

template <int NQ, int KT, bool HAS_T>
__global__ __launch_bounds__(256) void gram_mfma_kernel(int64_t row_begin, int64_t row_end, int F, int k,
                                                        const double *__restrict__ X, int64_t ldx,
                                                        const double *__restrict__ T, double *__restrict__ partial)
{
    constexpr int NT = KT * (KT + 1) / 2;
    constexpr int TILE = 16 * KT * GM_LD, RED = NT * 256;
    constexpr int WAVE_LDS = TILE > RED ? TILE : RED;
    extern __shared__ __attribute__((aligned(16))) double gms[];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    double *yT = gms + wave * WAVE_LDS;                          // [j][i], row stride GM_LD
    double tA[HAS_T ? KT : 1][HAS_T ? NQ : 1];
    if (HAS_T) {
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = 4 * q + lq, j = 16 * jt + li;
                tA[HAS_T ? jt : 0][HAS_T ? q : 0] = (c < F && j < k) ? T[(size_t)c * k + j] : 0.0;
            }
    }
    gv4d acc[NT];
#pragma unroll
    for (int p = 0; p < NT; ++p) acc[p] = (gv4d){0.0, 0.0, 0.0, 0.0};
    double xsum = 0.0;
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    const int64_t sub_stride = (int64_t)gridDim.x * 4;
    double xb[NQ];
    auto issue_loads = [&](int64_t sidx) {
        const int64_t row = row_begin + sidx * 16 + li;
        const int64_t rowc = row < row_end ? row : row_end - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = 4 * q + lq;
            xb[q] = X[(size_t)(c < F ? c : F - 1) * ldx + rowc];
        }
    };
    int64_t sidx = (int64_t)blockIdx.x * 4 + wave;
    if (sidx < nsub) issue_loads(sidx);
    for (; sidx < nsub; sidx += sub_stride) {
        const bool valid = row_begin + sidx * 16 + li < row_end;
#pragma unroll
        for (int q = 0; q < NQ; ++q) xb[q] = (valid && 4 * q + lq < F) ? xb[q] : 0.0;
        if (HAS_T) {
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) {
                gv4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    y = __builtin_amdgcn_mfma_f64_16x16x4f64(tA[HAS_T ? jt : 0][HAS_T ? q : 0], xb[q], y, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) yT[(16 * jt + lq + 4 * g) * GM_LD + li] = y[g];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4 * KT; ++q) {
                const double v = (q < NQ) ? xb[q < NQ ? q : 0] : 0.0;
                yT[(4 * q + lq) * GM_LD + li] = v;
                xsum += v;
            }
        }
        if (sidx + sub_stride < nsub) issue_loads(sidx + sub_stride);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            double a[KT];
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) a[jt] = yT[(16 * jt + li) * GM_LD + 4 * st + lq];
            int p = 0;
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int ju = jt; ju < KT; ++ju, ++p)
                    acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt], a[ju], acc[p], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // fixed-order sum over the four waves, then the [pair][block] partial layout of gram_finalize
    __syncthreads();
    double *red = gms + wave * WAVE_LDS;
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(p * 4 + g) * 64 + lane] = acc[p][g];
    double *xred = gms + 4 * WAVE_LDS;
    xsum = grx_group_sum<64>(xsum);
    if (lane == 0) xred[wave] = xsum;
    __syncthreads();
    const int npairs = k * (k + 1) / 2;
    for (int idx = t; idx < NT * 256; idx += 256) {
        const int p = idx >> 8, g = (idx >> 6) & 3, ln = idx & 63;
        const double v = ((gms[0 * WAVE_LDS + idx] + gms[1 * WAVE_LDS + idx]) + gms[2 * WAVE_LDS + idx]) +
                         gms[3 * WAVE_LDS + idx];
        int jt = 0, ju = 0, pp = p;                              // p -> (jt <= ju)
        for (jt = 0; jt < KT; ++jt) {
            if (pp < KT - jt) { ju = jt + pp; break; }
            pp -= KT - jt;
        }
        const int j = 16 * jt + (ln >> 4) + 4 * g, j2 = 16 * ju + (ln & 15);
        if (j <= j2 && j2 < k) partial[(size_t)(j2 * (j2 + 1) / 2 + j) * gridDim.x + blockIdx.x] = v;
    }
    if (t == 0) partial[(size_t)npairs * gridDim.x + blockIdx.x] = ((xred[0] + xred[1]) + xred[2]) + xred[3];
}

using GramKernel = void (*)(int64_t, int64_t, int, int, const double *, int64_t, const double *, double *);
template <int KT, bool HAS_T, int... NQs>
constexpr std::array<GramKernel, sizeof...(NQs)> gram_table(std::integer_sequence<int, NQs...>)
{
    return {gram_mfma_kernel<NQs + 1, KT, HAS_T>...};
}
const auto GRAM_T1 = gram_table<1, true>(std::make_integer_sequence<int, GM_MAX_NQ>{});
const auto GRAM_T2 = gram_table<2, true>(std::make_integer_sequence<int, GM_MAX_NQ>{});
const auto GRAM_T3 = gram_table<3, true>(std::make_integer_sequence<int, GM_MAX_NQ>{});
const auto GRAM_I1 = gram_table<1, false>(std::make_integer_sequence<int, 4>{});      // k = F: NQ <= 4 KT
const auto GRAM_I2 = gram_table<2, false>(std::make_integer_sequence<int, 8>{});
const auto GRAM_I3 = gram_table<3, false>(std::make_integer_sequence<int, 12>{});

GramKernel gram_mfma_pick(int F, int k, bool has_t)
{
    const int nq = (F + 3) / 4, kt = (k + 15) / 16;
    if (nq > GM_MAX_NQ || kt > GM_MAX_KT) return nullptr;
    if (has_t) return (kt == 1 ? GRAM_T1 : kt == 2 ? GRAM_T2 : GRAM_T3)[nq - 1];
    return kt == 1 ? GRAM_I1[nq - 1] : kt == 2 ? GRAM_I2[nq - 1] : GRAM_I3[nq - 1];
}

// ---------------------------------------------------------------------------------------
// Gram matrices of MORE than 48 columns on the matrix cores, in two kernels:
//   gram_transform_mfma_kernel  Y = X T (feature-major [k][ldy]): per 16-row sub-tile the X operand is
//                               loaded once per K-step and feeds up to 8 output tiles (128 columns of
//                               Y per grid.y block); T comes from global memory (cache resident)
//   gram_pairs_mfma_kernel      G = Y^T Y over pairs of 48-column groups (grid.y = pair): both operands
//                               are loaded from global memory directly in the transposed MFMA layout
//                               (lane (j, i): Y[j][row0 + i], four consecutive rows per lane group, the
//                               four K-steps of a sub-tile cover whole 128-byte lines) -- no LDS.
// Every group pair writes its own entries of the [pair id][workgroup] partial table of gram_finalize.
constexpr int GP_GROUP = 48;                                     // columns per group (3 tiles)

__global__ __launch_bounds__(256) void gram_transform_mfma_kernel(int64_t row_begin, int64_t row_end, int F, int k,
                                                                  const double *__restrict__ X, int64_t ldx,
                                                                  const double *__restrict__ T,
                                                                  double *__restrict__ Y, int64_t ldy)
{
    constexpr int JT = 8;                                        // output tiles per workgroup column block
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int j_base = 16 * JT * blockIdx.y;
    const int nq = (F + 3) / 4;
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < nsub; sidx += (int64_t)gridDim.x * 4) {
        const int64_t row = row_begin + sidx * 16 + li;
        const bool valid = row < row_end;
        const int64_t rowc = valid ? row : row_end - 1;
        gv4d y[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) y[jt] = (gv4d){0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < nq; ++q) {
            const int c = 4 * q + lq;
            const double xv = X[(size_t)(c < F ? c : F - 1) * ldx + rowc];
            const double x = (valid && c < F) ? xv : 0.0;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const int j = j_base + 16 * jt + li;
                const double a = (c < F && j < k) ? T[(size_t)c * k + j] : 0.0;
                y[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x, y[jt], 0, 0, 0);
            }
        }
        if (valid) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int j = j_base + 16 * jt + lq + 4 * g;
                    if (j < k) Y[(size_t)j * ldy + row] = y[jt][g];
                }
        }
    }
}

__global__ __launch_bounds__(256) void gram_pairs_mfma_kernel(int64_t row_begin, int64_t row_end, int k,
                                                              const double *__restrict__ Y, int64_t ldy,
                                                              int ngroups, double *__restrict__ partial)
{
    __shared__ double red[4][9 * 256];
    // pair index -> (A <= B)
    int A = 0, B = 0;
    {
        int p = blockIdx.y;
        for (A = 0; A < ngroups; ++A) {
            if (p < ngroups - A) { B = A + p; break; }
            p -= ngroups - A;
        }
    }
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int ja = A * GP_GROUP, jb = B * GP_GROUP;
    gv4d acc[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) acc[p] = (gv4d){0.0, 0.0, 0.0, 0.0};
    const double *ap[3], *bp[3];
    bool av[3], bv[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int j = ja + 16 * m + li, j2 = jb + 16 * m + li;
        av[m] = j < k && 16 * m + li < GP_GROUP;
        bv[m] = j2 < k && 16 * m + li < GP_GROUP;
        ap[m] = Y + (size_t)(av[m] ? j : 0) * ldy;
        bp[m] = Y + (size_t)(bv[m] ? j2 : 0) * ldy;
    }
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < nsub; sidx += (int64_t)gridDim.x * 4) {
        const int64_t row0 = row_begin + sidx * 16;
        double a[4][3], b[4][3];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int64_t row = row0 + 4 * st + lq;
            const bool valid = row < row_end;
            const int64_t rowc = valid ? row : row_end - 1;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const double u = ap[m][rowc];
                a[st][m] = (valid && av[m]) ? u : 0.0;
                if (A != B) {
                    const double w = bp[m][rowc];
                    b[st][m] = (valid && bv[m]) ? w : 0.0;
                } else {
                    b[st][m] = a[st][m];
                }
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2)
                    if (A != B || m2 >= m)
                        acc[m * 3 + m2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st][m], b[st][m2], acc[m * 3 + m2], 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[wave][(p * 4 + g) * 64 + lane] = acc[p][g];
    __syncthreads();
    for (int idx = t; idx < 9 * 256; idx += 256) {
        const int p = idx >> 8, g = (idx >> 6) & 3, ln = idx & 63;
        const int m = p / 3, m2 = p % 3;
        if (A == B && m2 < m) continue;
        const double v = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
        const int r1 = 16 * m + (ln >> 4) + 4 * g, r2 = 16 * m2 + (ln & 15);    // within the groups
        if (r1 >= GP_GROUP || r2 >= GP_GROUP) continue;
        const int j = ja + r1, j2 = jb + r2;
        if (j >= k || j2 >= k || j > j2) continue;
        partial[(size_t)((size_t)j2 * (j2 + 1) / 2 + j) * gridDim.x + blockIdx.x] = v;
    }
}

// ---- full-width Gram for 48 < k <= 128 output columns: X is read ONCE ----------------------------------
// The group-pair kernel above reads every 48-column group once per pair (3.75 x the matrix at k = 115) in 32-byte
// pieces (MFMA A layout straight from global memory: 16 columns x 4 rows per load) -- 10 ms per Gram matrix at
// 5 M x 115.  Here a workgroup owns 16-row sub-tiles: the (transformed) tile Y^T [k][16] is built once in LDS --
// no transform: coalesced 128-byte loads of X; with T: Y^T = T^T X^T on the matrix cores, T as A operand from L2,
// X as B operand straight from global memory in 128-byte segments -- and the KT (KT + 1) / 2 output tiles are
// split over the four waves (<= 9 accumulator tiles each, kept in registers over all sub-tiles).  Two LDS tiles
// alternate so that one barrier per sub-tile suffices.
constexpr int GF_LD = 17;
constexpr int GF_MAX_KT = 8;

// the MFMAs of one K step for wave W: every index is a compile-time constant (accumulators stay in registers)
template <int KT, int W>
__device__ __forceinline__ void gram_full_step(const double (&op)[KT], gv4d (&acc)[(KT * (KT + 1) / 2 + 3) / 4])
{
    int m = 0, m2 = 0;
#pragma unroll
    for (int pp = 0; pp < KT * (KT + 1) / 2; ++pp) {
        if ((pp & 3) == W) acc[pp >> 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[m], op[m2], acc[pp >> 2], 0, 0, 0);
        if (++m2 == KT) { ++m; m2 = m; }
    }
}

template <int PW>
__device__ __forceinline__ gv4d gram_full_pick(const gv4d (&acc)[PW], int idx)
{
    gv4d out = acc[0];
#pragma unroll
    for (int p = 1; p < PW; ++p)
        if (idx == p) out = acc[p];
    return out;
}

__device__ __forceinline__ double gram_full_keep(double v, bool keep)
{
    return __longlong_as_double(__double_as_longlong(v) & (keep ? -1ll : 0ll));
}

template <int KT, bool HAS_T>
__global__ __launch_bounds__(256, (HAS_T && KT <= 5) ? 2 : 1) void gram_full_mfma_kernel(int64_t row_begin, int64_t row_end, int F, int k,
                                                             const double *__restrict__ X, int64_t ldx,
                                                             const double *__restrict__ T, double *__restrict__ partial,
                                                             double *__restrict__ sum_partial)
{
    constexpr int NP = KT * (KT + 1) / 2;                       // output tile pairs (m <= m2)
    constexpr int PW = (NP + 3) / 4;                            // pairs per wave
    constexpr int XT = 8;                                       // X tile: up to 128 feature columns (F <= 128)
    constexpr int XTILE = 16 * XT * GF_LD;
    extern __shared__ __attribute__((aligned(16))) double gfs[];    // xbuf[2][XTILE] (+ ypart[4][16 KT LD] with T)
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    gv4d acc[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) acc[p] = (gv4d){0.0, 0.0, 0.0, 0.0};
    const int nq = (F + 3) / 4;
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    // thread (column t >> 4 + 16 u, row t & 15): 16 lanes read 16 consecutive rows of one column (128 bytes).
    // Two sub-tiles of loads stay in flight (register sets pa / pb) while a third is multiplied from LDS.
    double pa[XT], pb[XT];
    // issue() only loads (clamped addresses); the padding mask is applied in stash(), the first use -- a mask or
    // select next to the load makes the compiler wait for it there
    auto issue = [&](double (&pre)[XT], int64_t sidx) {
        const int64_t row = row_begin + sidx * 16 + (t & 15);
        const int64_t rowc = row < row_end ? row : row_end - 1;
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int c = (t >> 4) + 16 * u;
            pre[u] = X[(size_t)(c < F ? c : F - 1) * ldx + rowc];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    double xsum = 0.0;                                          // this thread's share of sum(X) (padding is zero)
    auto stash = [&](const double (&pre)[XT], int64_t sidx, double *xb) {
        __builtin_amdgcn_sched_barrier(0);
        const bool row_ok = row_begin + sidx * 16 + (t & 15) < row_end;
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int c = (t >> 4) + 16 * u;
            const double v = gram_full_keep(pre[u], row_ok && c < F);
            xb[c * GF_LD + (t & 15)] = v;
            xsum += v;
        }
    };
    // With T: the transform Y^T = T^T X^T is split over the feature (K) dimension -- wave w owns feature rows
    // [4 q_lo, 4 (q_lo + q_n)) and keeps its slice of T in registers for the whole kernel, so the loop has no
    // loads but the X stream. The four partial Y^T tiles are summed (fixed order) when they are read back.
    constexpr int QW = 8;                                       // K steps per wave (nq <= 32)
    constexpr int YPLANE = 16 * KT * GF_LD;
    double *ypart = gfs + 2 * XTILE;                            // [4][YPLANE]
    const int qw = (nq + 3) / 4, q_lo = wave * qw;
    const int q_n = nq - q_lo < qw ? nq - q_lo : qw;            // may be <= 0: the wave contributes zeros
    double treg[HAS_T ? KT : 1][QW];
    if (HAS_T) {
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int qq = 0; qq < QW; ++qq) {
                const int c = 4 * (q_lo + qq) + lq, j = 16 * jt + li;
                treg[jt][qq] = (qq < q_n && c < F && j < k) ? T[(size_t)c * k + j] : 0.0;
            }
    }
    auto compute = [&](const double *xT) {
        if (HAS_T) {
            gv4d y[KT];
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) y[jt] = (gv4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int qq = 0; qq < QW; ++qq) {
                if (qq < q_n) {                                 // uniform over the wave
                    const double x = xT[(4 * (q_lo + qq) + lq) * GF_LD + li];   // tile rows c >= F are zero
#pragma unroll
                    for (int jt = 0; jt < KT; ++jt)
                        y[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(treg[jt][qq], x, y[jt], 0, 0, 0);
                }
            }
            double *yp = ypart + wave * YPLANE;
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g) yp[(16 * jt + lq + 4 * g) * GF_LD + li] = y[jt][g];
            __syncthreads();
            // the four partial tiles summed ONCE (fixed order) into the first: read four times per operand below, the
            // planes made this kernel LDS-bound -- 128 ds_reads per lane and sub-tile against 55 MFMAs per wave
            // (profiles/r06_gram.txt)
            for (int e = t; e < 16 * KT * 16; e += 256) {
                const int a = (e >> 4) * GF_LD + (e & 15);
                ypart[a] = (ypart[a] + ypart[YPLANE + a]) + (ypart[2 * YPLANE + a] + ypart[3 * YPLANE + a]);
            }
            __syncthreads();
        }
        // this wave's pairs: pair number pp (order (0,0),(0,1),...,(0,KT-1),(1,1),...) belongs to wave pp & 3
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int i = 4 * st + lq;
            double op[KT];
#pragma unroll
            for (int m = 0; m < KT; ++m) {
                const int a = (16 * m + li) * GF_LD + i;
                op[m] = HAS_T ? ypart[a] : xT[a];
            }
            switch (wave) {
            case 0: gram_full_step<KT, 0>(op, acc); break;
            case 1: gram_full_step<KT, 1>(op, acc); break;
            case 2: gram_full_step<KT, 2>(op, acc); break;
            default: gram_full_step<KT, 3>(op, acc); break;
            }
        }
    };
    const int64_t G = gridDim.x;
    int64_t sidx = blockIdx.x;
    if (nsub > 0) {
        issue(pa, sidx);
        stash(pa, sidx, gfs);
        issue(pa, sidx + G);
    }
    __syncthreads();
    // invariant at the top: buffer 0 holds sub-tile sidx, pa carries sidx + G (in flight), pb is free
    while (sidx < nsub) {
        issue(pb, sidx + 2 * G);
        compute(gfs);
        stash(pa, sidx + G, gfs + XTILE);
        __syncthreads();                                        // next X tile visible; ybuf free again
        sidx += G;
        if (sidx >= nsub) break;
        issue(pa, sidx + 2 * G);
        compute(gfs + XTILE);
        stash(pb, sidx + G, gfs);
        __syncthreads();
        sidx += G;
    }
    if (sum_partial) {
        __syncthreads();
        xsum = grx_group_sum<64>(xsum);
        if (lane == 0) gfs[wave] = xsum;
        __syncthreads();
        if (t == 0) sum_partial[blockIdx.x] = ((gfs[0] + gfs[1]) + gfs[2]) + gfs[3];
    }
    // per-workgroup partials: each pair tile belongs to exactly one wave
    {
        int m = 0, m2 = 0;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            if ((pp & 3) == wave) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int j = 16 * m + lq + 4 * g, j2 = 16 * m2 + li;
                    if (j < k && j2 < k && j <= j2)
                        partial[(size_t)((size_t)j2 * (j2 + 1) / 2 + j) * gridDim.x + blockIdx.x] = gram_full_pick<PW>(acc, pp >> 2)[g];
                }
            }
            if (++m2 == KT) { ++m; m2 = m; }
        }
    }
}

// sum of X[:, ja .. ja+na) over the row range: per-workgroup partials (fixed order)
__global__ __launch_bounds__(256) void column_sum_kernel(int64_t row_begin, int64_t row_end,
                                                         const double *__restrict__ X, int64_t ldx, int ja, int na,
                                                         double *__restrict__ partial)
{
    __shared__ double wred[4];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_end; i += stride)
        for (int c = 0; c < na; ++c) s += X[(size_t)(ja + c) * ldx + i];
    s = grx_group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((wred[0] + wred[1]) + wred[2]) + wred[3];
}

// partial [npairs+1][nb] -> out: symmetric block of the K x K matrix (local column j <-> global column
// j < na ? ja + j : jb + j - na), then optionally the X sum at out[K*K] (one wavefront per output)
__global__ __launch_bounds__(256) void gram_finalize_kernel(const double *__restrict__ partial,
                                                            int nb, int k, double *__restrict__ out, int K,
                                                            int ja, int na, int jb, int write_sum)
{
    const int npairs = k * (k + 1) / 2;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx > k * k) return;
    int id, dst;
    if (idx == k * k) {
        if (!write_sum) return;
        id = npairs;
        dst = K * K;
    } else {
        const int a = idx / k, b = idx % k;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        id = hi * (hi + 1) / 2 + lo;
        const int ga = a < na ? ja + a : jb + (a - na), gb = b < na ? ja + b : jb + (b - na);
        dst = ga * K + gb;
    }
    const int lane = threadIdx.x & 63;
    const double *src = partial + (size_t)id * nb;
    double s = 0.0;
    for (int b = lane; b < nb; b += 64) s += src[b];
    s = grx_group_sum<64>(s);
    if (lane == 0) out[dst] = s;
}

// ---------------------------------------------------------------------------------------
// U = X Z with per-column statistics
// ---------------------------------------------------------------------------------------
struct ColStat { double maxabs; double signed_val; double idx; double sq_pos; double sq_neg; };

// U = X Z on the matrix cores (r <= 16 output columns = one MFMA tile): per 16-row
// sub-tile U^T (16 x 16) = Z^T X^T with the X operand straight from global memory and Z (F x r, a few
// KB) from the cache; the D layout (row j = (lane>>4) + 4g, col i = lane&15) stores U coalesced.  The
// column statistics are kept per lane over its sub-tiles (rows ascending, so ties keep the first row)
// and combined over the 16 lanes of a column group, the four waves, and then by project_finalize.
__global__ __launch_bounds__(256) void project_mfma_kernel(int64_t row_begin, int64_t row_end, int F, int r,
                                                           const double *__restrict__ X, int64_t ldx,
                                                           const double *__restrict__ Z, int ldz,
                                                           double *__restrict__ U, int64_t ldu,
                                                           double *__restrict__ partial)
{
    __shared__ double sred[4][MFMA_R][5];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int nq = (F + 3) / 4;
    double mx[4], sv[4], ix[4], sp[4], sn[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { mx[g] = -1.0; sv[g] = 0.0; ix[g] = 0.0; sp[g] = 0.0; sn[g] = 0.0; }
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < nsub; sidx += (int64_t)gridDim.x * 4) {
        const int64_t row = row_begin + sidx * 16 + li;
        const bool valid = row < row_end;
        const int64_t rowc = valid ? row : row_end - 1;
        gv4d u = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < nq; ++q) {
            const int c = 4 * q + lq;
            const double xv = X[(size_t)(c < F ? c : F - 1) * ldx + rowc];
            const double zv = Z[(size_t)(c < F ? c : F - 1) * ldz + (li < r ? li : r - 1)];
            u = __builtin_amdgcn_mfma_f64_16x16x4f64((c < F && li < r) ? zv : 0.0, (valid && c < F) ? xv : 0.0, u, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j = lq + 4 * g;
            if (valid && j < r) {
                const double v = u[g];
                U[(size_t)j * ldu + row] = v;
                const double a = fabs(v);
                if (a > mx[g]) { mx[g] = a; sv[g] = v; ix[g] = (double)row; }
                if (v > 0.0) sp[g] += v * v; else sn[g] += v * v;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {                 // the 16 lanes that share column j
            const double om = __shfl_xor(mx[g], off, 64), os = __shfl_xor(sv[g], off, 64);
            const double oi = __shfl_xor(ix[g], off, 64);
            if (om > mx[g] || (om == mx[g] && oi < ix[g])) { mx[g] = om; sv[g] = os; ix[g] = oi; }
            sp[g] += __shfl_xor(sp[g], off, 64);
            sn[g] += __shfl_xor(sn[g], off, 64);
        }
        const int j = lq + 4 * g;
        if (li == 0 && j < r) {
            sred[wave][j][0] = mx[g]; sred[wave][j][1] = sv[g]; sred[wave][j][2] = ix[g];
            sred[wave][j][3] = sp[g]; sred[wave][j][4] = sn[g];
        }
    }
    __syncthreads();
    if (t < r) {
        const int j = t;
        double bm = -1.0, bs = 0.0, bi = 0.0, p = 0.0, q2 = 0.0;
        for (int w = 0; w < 4; ++w) {
            const double *o = sred[w][j];
            if (o[0] > bm || (o[0] == bm && o[2] < bi)) { bm = o[0]; bs = o[1]; bi = o[2]; }
            p += o[3]; q2 += o[4];
        }
        double *o = partial + ((size_t)blockIdx.x * r + j) * 5;
        o[0] = bm; o[1] = bs; o[2] = bi; o[3] = p; o[4] = q2;
    }
}

__global__ __launch_bounds__(64) void project_finalize_kernel(const double *__restrict__ partial,
                                                              int nblocks, int r,
                                                              double *__restrict__ stats)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    double bm = -1.0, bs = 0.0, bi = 0.0, p = 0.0, q = 0.0;
    for (int b = lane; b < nblocks; b += 64) {
        const double *o = partial + ((size_t)b * r + j) * 5;
        if (o[0] > bm || (o[0] == bm && o[2] < bi)) { bm = o[0]; bs = o[1]; bi = o[2]; }
        p += o[3]; q += o[4];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(bm, off, 64);
        const double os = __shfl_xor(bs, off, 64);
        const double oi = __shfl_xor(bi, off, 64);
        if (om > bm || (om == bm && oi < bi)) { bm = om; bs = os; bi = oi; }
    }
    p = grx_group_sum<64>(p);
    q = grx_group_sum<64>(q);
    if (lane == 0) {
        stats[j * 4 + 0] = bs; stats[j * 4 + 1] = bi; stats[j * 4 + 2] = p; stats[j * 4 + 3] = q;
    }
}

struct NndsvdArgs { double sign[MAX_R]; double scale[MAX_R]; };

__global__ __launch_bounds__(256) void nndsvd_apply_kernel(int64_t row_begin, int64_t row_end, int r,
                                                           double *__restrict__ U, int64_t ldu,
                                                           NndsvdArgs a, double eps, double fill)
{
    const int j = blockIdx.y;
    const double sg = a.sign[j], sc = a.scale[j];
    double *u = U + (size_t)j * ldu;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_end; i += stride) {
        const double x = u[i];
        double v = (sg == 0.0) ? fabs(x) : fmax(sg * x, 0.0);
        v *= sc;
        u[i] = (v < eps) ? fill : v;
    }
}

// ---------------------------------------------------------------------------------------
// multiplicative update, W side (fused with the H-side reductions)
// ---------------------------------------------------------------------------------------
constexpr int MU_MAX_GRID = GRX_NUM_CU * 8;      // workgroups of the W pass (LDS admits >= 2 per CU)
// ---------------------------------------------------------------------------------------
// W <- W * (X H^T) / (W H H^T) fused with the H-side reductions A = W'^T X (r x F) and
// B = W'^T W' (r x r): ONE pass over X and W per iteration, on the matrix cores
// (v_mfma_f64_16x16x4_f64).  One wavefront owns a 16-row sub-tile from load to accumulate;
// nothing but its own 16 x (F + r) slice of X and W crosses HBM, and the only LDS traffic is one
// write + one read of that slice (the two GEMM pairs need X and W' in transposed lane layouts).
// (A VALU formulation on LDS row tiles measured 0.108 ms per launch at F = 20, r = 6: 5 KB of LDS
// reads per row made it LDS-bandwidth bound; this form moves 0.77 KB per row and runs 0.067 ms.)
//   phase 1 (D = A.B, rows k, cols i):  numer[k][i] = sum_c H[k][c] X[i][c]      A = H  (registers, loaded once)
//                                        denom[k][i] = sum_l HH[k][l] W[i][l]     A = HH (registers)
//            B operands straight from global memory: lane (q = lane>>4, i = lane&15) reads
//            X[4s+q][row i] -- 4 x 128-byte segments per wave load; the D layout of the f64 MFMA
//            (row = (lane>>4) + 4*reg, col = lane&15) is the layout the W operand was loaded in,
//            so W' = W * numer / denom needs no shuffle and is stored coalesced.
//   phase 2 (rows k, cols c | l):       A[k][c] += sum_i W'[i][k] X[i][c],  B[k][l] += sum_i W'[i][k] W'[i][l]
//            operands re-read from the wave's LDS slice in the transposed layout.
// Accumulators stay in registers over all sub-tiles of the wave; the four waves of a workgroup are
// summed in fixed order at the end (bitwise reproducible).
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int MF_LD = 17;                       // padded LDS row (doubles): conflict-free transposed reads

static inline size_t mfma_lds_doubles(int FT) { return (size_t)4 * (16 * FT + 16) * MF_LD + 256; }

// NQ = ceil(F / 4) K-steps over the feature columns, R4 = K-steps over the roles (2: r <= 8, 4: r <= 16);
// both compile-time so that the loads and MFMAs of a sub-tile form one straight-line block.
template <int NQ, int R4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NQ <= 8 ? 3 : 1, 8))) void nmf_w_pass_mfma_kernel(int64_t row_begin, int64_t row_end, int F, int r,
                                                              const double *__restrict__ X, int64_t ldx,
                                                              double *__restrict__ W, int64_t ldw,
                                                              const double *__restrict__ H,
                                                              double *__restrict__ partial,
                                                              const double *__restrict__ AB_prev,
                                                              double *__restrict__ H_out)
{
    constexpr int FT = (NQ + 3) / 4;                             // 16-column tiles of the A accumulator
    constexpr int F4 = NQ;                                       // K steps of 4 columns over c
    constexpr int WAVE_LDS = (16 * FT + 16) * MF_LD;             // doubles per wave: xT [16 FT][17] + wT [16][17]
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    double *sHH = fsm + 4 * WAVE_LDS;                            // 16 x 16
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    double *sH = fsm;                                            // r * F <= 16 * 120 doubles, aliases the wave slices
    // H H^T (r x r, zero padded to 16 x 16) through LDS
    {
        for (int idx = t; idx < r * F; idx += 256) sH[idx] = H[idx];
        if (AB_prev != nullptr) {
            // The H update of the PREVIOUS iteration, H <- H * A / (B H) from its reduced sums, recomputed by
            // every workgroup (r F <= 1920 outputs, a microsecond) instead of a launch of its own between two
            // W passes; workgroup 0 also stores it.  Same arithmetic as nmf_h_update_kernel (h_update_value).
            double *sB = fsm + r * F;                            // r x r
            for (int idx = t; idx < r * r; idx += 256) sB[idx] = AB_prev[r * F + idx];
            __syncthreads();
            constexpr int PER = (MFMA_R * MAX_F + 255) / 256;    // 8 outputs per thread at most
            double hn[PER];
#pragma unroll
            for (int s_ = 0; s_ < PER; ++s_) {
                const int idx = t + 256 * s_;
                hn[s_] = idx < r * F ? h_update_value(sH, sB, AB_prev[idx], idx, F, r) : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int s_ = 0; s_ < PER; ++s_) {
                const int idx = t + 256 * s_;
                if (idx < r * F) {
                    sH[idx] = hn[s_];
                    if (blockIdx.x == 0 && H_out != nullptr) H_out[idx] = hn[s_];
                }
            }
        }
        __syncthreads();
        {
            const int k = t >> 4, l = t & 15;
            double sacc = 0.0;
            if (k < r && l < r)
                for (int c = 0; c < F; ++c) sacc += sH[k * F + c] * sH[l * F + c];
            sHH[t] = sacc;
        }
        __syncthreads();
    }
    // SMALL (r <= 8): v_mfma_f64_4x4x4 (four independent 4x4x4 blocks per instruction, 19 cycles against
    // 66 for the 16x16x4 form) with the roles padded to 8 instead of 16.  Lane = 16 k + 4 block + i: A[block][i][k],
    // B[block][k][j] and D[block][i][j] at lane 16 i + 4 block + j (probed: tools/microbench/
    // mfma_f64_4x4x4_probe.hip).  The four blocks of an instruction take the four 4-row groups of the sub-tile
    // (phase 1) / four 4-column groups of a 16-column feature tile (phase 2), the A operand -- four roles -- is
    // the same in every block; two instructions cover the eight roles.  The B operands and the result lanes are
    // those of the 16x16x4 form (X as loaded; role = 4 rho + (lane >> 4), row / column = lane & 15).
    constexpr bool SMALL = (R4 == 2);
    const int l3 = lane & 3;
    double hA[SMALL ? 1 : F4], hhA[SMALL ? 1 : R4];
    double hA2[SMALL ? F4 : 1][2], hhA2[SMALL ? R4 : 1][2];
    if constexpr (SMALL) {
#pragma unroll
        for (int q = 0; q < F4; ++q)
#pragma unroll
            for (int rho = 0; rho < 2; ++rho) {
                const int k = 4 * rho + l3, c = 4 * q + lq;
                hA2[q][rho] = (k < r && c < F) ? sH[k * F + c] : 0.0;
            }
#pragma unroll
        for (int q = 0; q < R4; ++q)
#pragma unroll
            for (int rho = 0; rho < 2; ++rho) hhA2[q][rho] = sHH[(4 * rho + l3) * 16 + 4 * q + lq];
    } else {
#pragma unroll
        for (int q = 0; q < F4; ++q) {
            const int c = 4 * q + lq;
            hA[q] = (li < r && c < F) ? sH[li * F + c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < R4; ++q) hhA[q] = sHH[li * 16 + 4 * q + lq];
    }
    __syncthreads();                                             // sH (aliased) is dead from here on
    double *xT = fsm + wave * WAVE_LDS;                          // [c][i], row stride MF_LD
    double *wT = xT + 16 * FT * MF_LD;                           // [k][i]
    v4d accA[FT], accB = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ct = 0; ct < FT; ++ct) accA[ct] = (v4d){0.0, 0.0, 0.0, 0.0};
    double acc2[2][FT], accB2[2] = {0.0, 0.0};                   // SMALL: [rho][feature tile]
#pragma unroll
    for (int ct = 0; ct < FT; ++ct) acc2[0][ct] = acc2[1][ct] = 0.0;
    // When the last 16-column tile of A has r spare columns (F = 20, r = 6: columns 4..9 of tile 1),
    // W' rides along as extra "feature" columns F..F+r-1 of the LDS tile and B = W'^T W' comes out of
    // the same MFMAs as A: four matrix instructions fewer per sub-tile (12 -> 8 in phase 2).
    const bool fuse_b = (F & 15) != 0 && (F & 15) + r <= 16;
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    const int64_t sub_stride = (int64_t)gridDim.x * 4;
    double xb[F4], wb[R4];
    // Unconditional loads from clamped addresses (masked afterwards) so that all loads of a
    // sub-tile issue back to back; the loads of sub-tile n+1 are issued right after sub-tile n
    // has been copied to LDS and stay in flight during its phase 2.
    auto issue_loads = [&](int64_t sidx) {
        const int64_t row = row_begin + sidx * 16 + li;
        const int64_t rowc = row < row_end ? row : row_end - 1;
#pragma unroll
        for (int q = 0; q < F4; ++q) {
            const int c = 4 * q + lq;
            xb[q] = X[(size_t)(c < F ? c : F - 1) * ldx + rowc];
        }
#pragma unroll
        for (int q = 0; q < R4; ++q) {
            const int k = 4 * q + lq;
            wb[q] = W[(size_t)(k < r ? k : r - 1) * ldw + rowc];
        }
    };
    int64_t sidx = (int64_t)blockIdx.x * 4 + wave;
    if (sidx < nsub) issue_loads(sidx);
    for (; sidx < nsub; sidx += sub_stride) {
        const int64_t row = row_begin + sidx * 16 + li;
        const bool valid = row < row_end;
        v4d num = {0.0, 0.0, 0.0, 0.0}, den = {0.0, 0.0, 0.0, 0.0};
        if constexpr (SMALL) {
            double n0 = 0.0, n1 = 0.0, d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int q = 0; q < F4; ++q) {
                xb[q] = (valid && 4 * q + lq < F) ? xb[q] : 0.0;
                n0 = __builtin_amdgcn_mfma_f64_4x4x4f64(hA2[q][0], xb[q], n0, 0, 0, 0);
                n1 = __builtin_amdgcn_mfma_f64_4x4x4f64(hA2[q][1], xb[q], n1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < R4; ++q) {
                wb[q] = (valid && 4 * q + lq < r) ? wb[q] : 0.0;
                d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(hhA2[q][0], wb[q], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(hhA2[q][1], wb[q], d1, 0, 0, 0);
            }
            num[0] = n0; num[1] = n1; den[0] = d0; den[1] = d1;
        } else {
#pragma unroll
            for (int q = 0; q < F4; ++q) {
                xb[q] = (valid && 4 * q + lq < F) ? xb[q] : 0.0;
                num = __builtin_amdgcn_mfma_f64_16x16x4f64(hA[q], xb[q], num, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < R4; ++q) {
                wb[q] = (valid && 4 * q + lq < r) ? wb[q] : 0.0;
                den = __builtin_amdgcn_mfma_f64_16x16x4f64(hhA[q], wb[q], den, 0, 0, 0);
            }
        }
        double wnew[R4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k = lq + 4 * g;
            double wn = 0.0;
            if (g < R4) {
                double d = den[g];
                if (d == 0.0) d = NMF_EPSILON;
                const bool live = valid && k < r;
                wn = live ? wb[g] * (num[g] / d) : 0.0;
                if (live) W[(size_t)k * ldw + row] = wn;
                wnew[g] = wn;
            }
            wT[k * MF_LD + li] = wn;
        }
#pragma unroll
        for (int q = 0; q < 4 * FT; ++q) xT[(4 * q + lq) * MF_LD + li] = (q < F4) ? xb[q] : 0.0;
        if (fuse_b) {                                            // after the zero fill of the same rows (in-order LDS)
#pragma unroll
            for (int g = 0; g < R4; ++g) {
                const int k = lq + 4 * g;
                if (k < r) xT[(F + k) * MF_LD + li] = wnew[g];
            }
        }
        if (sidx + sub_stride < nsub) issue_loads(sidx + sub_stride);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int i = 4 * st + lq;
            if constexpr (SMALL) {
                const double a0 = wT[l3 * MF_LD + i], a1 = wT[(4 + l3) * MF_LD + i];   // W'[row i][role rho * 4 + l3]
#pragma unroll
                for (int ct = 0; ct < FT; ++ct) {
                    const double bX = xT[(16 * ct + li) * MF_LD + i];
                    acc2[0][ct] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, bX, acc2[0][ct], 0, 0, 0);
                    acc2[1][ct] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, bX, acc2[1][ct], 0, 0, 0);
                }
                if (!fuse_b) {
                    const double bW = wT[li * MF_LD + i];                          // rows k >= r of wT hold zeros
                    accB2[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, bW, accB2[0], 0, 0, 0);
                    accB2[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, bW, accB2[1], 0, 0, 0);
                }
            } else {
                const double aW = wT[li * MF_LD + i];
#pragma unroll
                for (int ct = 0; ct < FT; ++ct) {
                    const double bX = xT[(16 * ct + li) * MF_LD + i];
                    accA[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aW, bX, accA[ct], 0, 0, 0);
                }
                if (!fuse_b) accB = __builtin_amdgcn_mfma_f64_16x16x4f64(aW, aW, accB, 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // fixed-order sum of the four waves' accumulators: red[wave][tile][g][lane]
    __syncthreads();
    double *red = fsm + wave * WAVE_LDS;                         // (FT + 1) * 256 doubles <= WAVE_LDS
    if constexpr (SMALL) {                                       // g = rho: role (lane >> 4) + 4 g as in the other form
#pragma unroll
        for (int ct = 0; ct < FT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) red[(ct * 4 + g) * 64 + lane] = g < 2 ? acc2[g][ct] : 0.0;
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(FT * 4 + g) * 64 + lane] = g < 2 ? accB2[g] : 0.0;
    } else {
#pragma unroll
        for (int ct = 0; ct < FT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) red[(ct * 4 + g) * 64 + lane] = accA[ct][g];
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(FT * 4 + g) * 64 + lane] = accB[g];
    }
    __syncthreads();
    for (int idx = t; idx < (FT + 1) * 256; idx += 256) {
        const int tile = idx >> 8, g = (idx >> 6) & 3, ln = idx & 63;
        const double v = ((fsm[0 * WAVE_LDS + idx] + fsm[1 * WAVE_LDS + idx]) + fsm[2 * WAVE_LDS + idx]) +
                         fsm[3 * WAVE_LDS + idx];
        const int k = (ln >> 4) + 4 * g, j = ln & 15;
        if (k >= r) continue;
        if (tile < FT) {
            const int c = 16 * tile + j;
            if (c < F) partial[(size_t)(k * F + c) * gridDim.x + blockIdx.x] = v;
            else if (fuse_b && c < F + r) partial[(size_t)(r * F + k * r + (c - F)) * gridDim.x + blockIdx.x] = v;
        } else if (!fuse_b && j < r) {
            partial[(size_t)(r * F + k * r + j) * gridDim.x + blockIdx.x] = v;
        }
    }
}

// H <- H * (A / (B H)), one workgroup; B (r x r) in LDS, A and H (r x F, a few KB, cache resident); every
// H[l][c] is read before the barrier that precedes the writes, so H_out may be H_in.
template <int RMAX>
__global__ __launch_bounds__(256) void nmf_h_update_kernel(int F, int r, const double *H_in, double *H_out,
                                                           const double *__restrict__ AB)
{
    __shared__ double sB[RMAX * RMAX];
    for (int idx = threadIdx.x; idx < r * r; idx += 256) sB[idx] = AB[r * F + idx];
    __syncthreads();
    constexpr int PER = (RMAX * MAX_F_WIDE + 255) / 256;           // 30 outputs per thread at most (60 above 16 roles)
    double hnew[PER];
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        const int idx = threadIdx.x + 256 * s;
        hnew[s] = idx < r * F ? h_update_value(H_in, sB, AB[idx], idx, F, r) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        const int idx = threadIdx.x + 256 * s;
        if (idx < r * F) H_out[idx] = hnew[s];
    }
}

template <int RMAX>
__global__ __launch_bounds__(256) void nmf_residual_kernel(int64_t row_begin, int64_t row_end, int F, int r,
                                                           const double *__restrict__ X, int64_t ldx,
                                                           const double *__restrict__ W, int64_t ldw,
                                                           const double *__restrict__ H,
                                                           double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) double rsm[];
    double *sH = rsm;                 // r*F
    __shared__ double wred[4];
    for (int idx = threadIdx.x; idx < r * F; idx += 256) sH[idx] = H[idx];
    __syncthreads();
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_end; i += stride) {
        double w[RMAX];
#pragma unroll
        for (int k = 0; k < RMAX; ++k) w[k] = (k < r) ? W[(size_t)k * ldw + i] : 0.0;
        // eight columns of X in flight per row (a load-use loop waits for every load in turn); the squares are
        // added in column order as before
        for (int c0 = 0; c0 < F; c0 += 8) {
            double x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = X[(size_t)(c0 + j < F ? c0 + j : F - 1) * ldx + i];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                if (c < F) {
                    double wh = 0.0;
#pragma unroll
                    for (int k = 0; k < RMAX; ++k)
                        if (k < r) wh += w[k] * sH[k * F + c];
                    const double d = x[j] - wh;
                    s += d * d;
                }
            }
        }
    }
    s = grx_group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((wred[0] + wred[1]) + wred[2]) + wred[3];
}

// Generalised KL divergence of X from W H with the zero entries of X masked out
// (graphrole/roles/description_length.py:44-61): sum_{x != 0} x log(x / v) - x + v, v = (W H)_ic.
template <int RMAX>
__global__ __launch_bounds__(256) void nmf_kl_cost_kernel(int64_t row_begin, int64_t row_end, int F, int r,
                                                          const double *__restrict__ X, int64_t ldx,
                                                          const double *__restrict__ W, int64_t ldw,
                                                          const double *__restrict__ H,
                                                          double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) double rsm[];
    double *sH = rsm;                 // r*F
    __shared__ double wred[4];
    for (int idx = threadIdx.x; idx < r * F; idx += 256) sH[idx] = H[idx];
    __syncthreads();
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_end; i += stride) {
        double w[RMAX];
#pragma unroll
        for (int k = 0; k < RMAX; ++k) w[k] = (k < r) ? W[(size_t)k * ldw + i] : 0.0;
        for (int c = 0; c < F; ++c) {
            const double x = X[(size_t)c * ldx + i];
            if (x != 0.0) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < RMAX; ++k)
                    if (k < r) v += w[k] * sH[k * F + c];
                s += x * log(x / v) - x + v;
            }
        }
    }
    s = grx_group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((wred[0] + wred[1]) + wred[2]) + wred[3];
}

// The same update for MORE than 120 features (the register-resident form above would need more
// than 256 VGPRs), as two kernels with small register footprints (high occupancy):
//   nmf_w_update_wide_kernel : numerators over all feature columns in chunks of 8 K-steps (H operands
//                              from global memory, a few KB, cache resident), denominators, W' stored
//   nmf_w_accum_wide_kernel  : grid.y = 32-column chunk of the features; A[:, chunk] += W'^T X[:, chunk]
//                              (and B = W'^T W' in chunk 0) with both operands loaded from global memory
//                              directly in the transposed MFMA layout (4 consecutive rows per lane
//                              group; the four K-steps of a sub-tile cover whole 128-byte lines)
// X is read twice per iteration (W' is only known after the last chunk).
template <int R4>
__global__ __launch_bounds__(256) void nmf_w_update_wide_kernel(int64_t row_begin, int64_t row_end, int F, int r,
                                                                const double *__restrict__ X, int64_t ldx,
                                                                double *__restrict__ W, int64_t ldw,
                                                                const double *__restrict__ H)
{
    constexpr int CH = 8;                                        // K-steps per chunk
    __shared__ double sHH[256];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    {
        const int k = t >> 4, l = t & 15;
        double sacc = 0.0;
        if (k < r && l < r)
            for (int c = 0; c < F; ++c) sacc += H[k * F + c] * H[l * F + c];
        sHH[t] = sacc;
    }
    __syncthreads();
    double hhA[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hhA[q] = sHH[li * 16 + 4 * q + lq];
    const int nchunks = (F + 31) / 32;
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < nsub; sidx += (int64_t)gridDim.x * 4) {
        const int64_t row = row_begin + sidx * 16 + li;
        const bool valid = row < row_end;
        const int64_t rowc = valid ? row : row_end - 1;
        double wb[R4];
#pragma unroll
        for (int q = 0; q < R4; ++q) {
            const int k = 4 * q + lq;
            const double v = W[(size_t)(k < r ? k : r - 1) * ldw + rowc];
            wb[q] = (valid && k < r) ? v : 0.0;
        }
        v4d num = {0.0, 0.0, 0.0, 0.0}, den = {0.0, 0.0, 0.0, 0.0};
        for (int ch = 0; ch < nchunks; ++ch) {
            double x[CH], h[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int c = 32 * ch + 4 * q + lq;
                x[q] = X[(size_t)(c < F ? c : F - 1) * ldx + rowc];
                h[q] = H[(li < r ? li : r - 1) * F + (c < F ? c : F - 1)];
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int c = 32 * ch + 4 * q + lq;
                num = __builtin_amdgcn_mfma_f64_16x16x4f64((li < r && c < F) ? h[q] : 0.0,
                                                           (valid && c < F) ? x[q] : 0.0, num, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < R4; ++q) den = __builtin_amdgcn_mfma_f64_16x16x4f64(hhA[q], wb[q], den, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < R4; ++g) {
            const int k = lq + 4 * g;
            double d = den[g];
            if (d == 0.0) d = NMF_EPSILON;
            if (valid && k < r) W[(size_t)k * ldw + row] = wb[g] * (num[g] / d);
        }
    }
}

// partial layout [P][gridDim.x] as in the single-kernel form; chunk = blockIdx.y
__global__ __launch_bounds__(256) void nmf_w_accum_wide_kernel(int64_t row_begin, int64_t row_end, int F, int r,
                                                               const double *__restrict__ X, int64_t ldx,
                                                               const double *__restrict__ W, int64_t ldw,
                                                               double *__restrict__ partial)
{
    __shared__ double red[4][3 * 256];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int c_base = 32 * blockIdx.y;
    const bool with_b = blockIdx.y == 0;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0}, accB = {0.0, 0.0, 0.0, 0.0};
    const int c0 = c_base + li, c1 = c_base + 16 + li;
    const double *x0p = X + (size_t)(c0 < F ? c0 : F - 1) * ldx;
    const double *x1p = X + (size_t)(c1 < F ? c1 : F - 1) * ldx;
    const double *wp = W + (size_t)(li < r ? li : r - 1) * ldw;
    const int64_t nsub = (row_end - row_begin + 15) / 16;
    for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < nsub; sidx += (int64_t)gridDim.x * 4) {
        const int64_t row0 = row_begin + sidx * 16;
        double aW[4], b0[4], b1[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int64_t row = row0 + 4 * st + lq;
            const bool valid = row < row_end;
            const int64_t rowc = valid ? row : row_end - 1;
            const double w = wp[rowc], u = x0p[rowc], v = x1p[rowc];
            aW[st] = (valid && li < r) ? w : 0.0;
            b0[st] = (valid && c0 < F) ? u : 0.0;
            b1[st] = (valid && c1 < F) ? v : 0.0;
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[st], b0[st], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[st], b1[st], acc1, 0, 0, 0);
            if (with_b) accB = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[st], aW[st], accB, 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        red[wave][(0 * 4 + g) * 64 + lane] = acc0[g];
        red[wave][(1 * 4 + g) * 64 + lane] = acc1[g];
        red[wave][(2 * 4 + g) * 64 + lane] = accB[g];
    }
    __syncthreads();
    for (int idx = t; idx < 3 * 256; idx += 256) {
        const int tile = idx >> 8, g = (idx >> 6) & 3, ln = idx & 63;
        const double v = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
        const int k = (ln >> 4) + 4 * g, j = ln & 15;
        if (k >= r) continue;
        if (tile < 2) {
            const int c = c_base + 16 * tile + j;
            if (c < F) partial[(size_t)(k * F + c) * gridDim.x + blockIdx.x] = v;
        } else if (with_b && j < r) {
            partial[(size_t)(r * F + k * r + j) * gridDim.x + blockIdx.x] = v;
        }
    }
}

// launch table over (NQ, R4): function pointers of the instantiations
using WPassKernel = void (*)(int64_t, int64_t, int, int, const double *, int64_t, double *, int64_t, const double *,
                             double *, const double *, double *);
template <int R4, int... NQs>
constexpr std::array<WPassKernel, sizeof...(NQs)> w_pass_table(std::integer_sequence<int, NQs...>)
{
    return {nmf_w_pass_mfma_kernel<NQs + 1, R4>...};
}
const auto W_PASS_R2 = w_pass_table<2>(std::make_integer_sequence<int, MAX_F / 4>{});
const auto W_PASS_R4 = w_pass_table<4>(std::make_integer_sequence<int, MAX_F / 4>{});

WPassKernel w_pass_kernel(int F, int r) { return (r <= 8 ? W_PASS_R2 : W_PASS_R4)[(F + 3) / 4 - 1]; }

// one resident generation of workgroups (cached per instantiation)
static inline void launch_h_update(int F, int r, const double *h_in, double *h_out, const double *ab, hipStream_t st)
{
    if (r <= MFMA_R) nmf_h_update_kernel<MFMA_R><<<1, 256, 0, st>>>(F, r, h_in, h_out, ab);
    else nmf_h_update_kernel<MAX_R><<<1, 256, 0, st>>>(F, r, h_in, h_out, ab);
}

// ---------------------------------------------------------------------------------------
// More than 16 roles (17 .. GRX_MAX_ROLES): the fused kernels above hold the roles in ONE MFMA tile.  Beyond that the
// multiplicative update is composed from parts that exist: a plain per-row kernel for W <- W * (X H^T) / (W H H^T)
// (the shape of nmf_residual_kernel: a row of W and its numerators in registers, H read with uniform addresses), then
// [W^T X | W^T W] as the Gram matrix of the stacked table [W' | X] (the full-width Gram kernels read it once).
// Several times the traffic of the fused pass -- the reference accepts any rank
// (graphrole/roles/extract.py:22-33), and a slow exact path beats a refusal.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nmf_hht_kernel(int F, int r, const double *__restrict__ H, double *__restrict__ HH)
{
    for (int idx = threadIdx.x; idx < r * r; idx += 256) {
        const int k = idx / r, l = idx % r;
        double acc = 0.0;
        for (int c = 0; c < F; ++c) acc += H[k * F + c] * H[l * F + c];
        HH[idx] = acc;
    }
}

// W' = W * (numer / denom) for rows [row_begin, row_end); W' also goes to the first r rows of the stacked table Y
__global__ __launch_bounds__(256) void nmf_w_update_generic_kernel(int64_t row_begin, int64_t row_end, int F, int r,
                                                                   const double *__restrict__ X, int64_t ldx,
                                                                   double *__restrict__ W, int64_t ldw,
                                                                   const double *__restrict__ H,
                                                                   const double *__restrict__ HH,
                                                                   double *__restrict__ Y, int64_t ldy)
{
    __shared__ double sHH[MAX_R * MAX_R];
    for (int idx = threadIdx.x; idx < r * r; idx += 256) sHH[idx] = HH[idx];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_end; i += stride) {
        double w[MAX_R], num[MAX_R];
#pragma unroll
        for (int k = 0; k < MAX_R; ++k) { w[k] = (k < r) ? W[(size_t)k * ldw + i] : 0.0; num[k] = 0.0; }
        for (int c0 = 0; c0 < F; c0 += 8) {
            double x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = X[(size_t)(c0 + j < F ? c0 + j : F - 1) * ldx + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                if (c < F) {
#pragma unroll
                    for (int k = 0; k < MAX_R; ++k)
                        if (k < r) num[k] += x[j] * H[k * F + c];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MAX_R; ++k) {
            if (k < r) {
                double den = 0.0;
#pragma unroll
                for (int l = 0; l < MAX_R; ++l)
                    if (l < r) den += w[l] * sHH[l * r + k];
                if (den == 0.0) den = NMF_EPSILON;
                const double wn = w[k] * (num[k] / den);
                W[(size_t)k * ldw + i] = wn;
                Y[(size_t)k * ldy + i] = wn;
            }
        }
    }
}

// [A | B] from the Gram matrix G (K x K, K = r + F) of the stacked table [W' | X]
__global__ __launch_bounds__(256) void nmf_ab_from_gram_kernel(int F, int r, const double *__restrict__ G,
                                                               double *__restrict__ AB)
{
    const int K = r + F;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < r * F + r * r; idx += gridDim.x * 256) {
        if (idx < r * F) {
            const int k = idx / F, c = idx % F;
            AB[idx] = G[(size_t)k * K + r + c];
        } else {
            const int j = idx - r * F, k = j / r, l = j % r;
            AB[idx] = G[(size_t)k * K + l];
        }
    }
}

int mfma_resident_grid(int F, int r, size_t lds)
{
    static int cache[2][MAX_F / 4] = {};
    int &slot = cache[r <= 8 ? 0 : 1][(F + 3) / 4 - 1];
    if (slot == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, w_pass_kernel(F, r), 256, lds) != hipSuccess || nb < 1)
            nb = 1;
        const int g = nb * GRX_NUM_CU;
        slot = g > MU_MAX_GRID ? MU_MAX_GRID : g;
    }
    return slot;
}

constexpr int RES_GRID = GRX_NUM_CU * 4;

int check_nmf_shape(const char *who, int F, int r)
{
    if (F < 1 || r < 1 || F > MAX_F_WIDE || r > MAX_R || (r > MFMA_R && r + F > MAX_F_WIDE)) {
        grx_set_error("%s: F=%d r=%d outside the compiled limits (F<=%d, r<=%d, and r+F<=%d when r>%d)", who, F, r,
                      MAX_F_WIDE, MAX_R, MAX_F_WIDE, MFMA_R);
        return GRX_ERR_UNSUPPORTED;
    }
    return GRX_OK;
}

}  // namespace

// the wide-rank instantiations keep r * F <= 32 * 448 doubles of H in dynamic LDS: beyond the 64 KB default
#define GRX_TRY_LDS(kernel)                                                                                     \
    do {                                                                                                        \
        static const hipError_t lds_rc__ = hipFuncSetAttribute(reinterpret_cast<const void *>(&kernel),         \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); \
        GRX_CHECK_HIP(lds_rc__);                                                                                \
    } while (0)

extern "C" {

int grx_gather_columns(int64_t n, int F, const double *const *h_col_ptrs, double *d_out, int64_t ld,
                       void *stream)
{
    GRX_REQUIRE(n >= 0 && F >= 0 && ld >= n, "grx_gather_columns: bad shape");
    if (n == 0 || F == 0) return GRX_OK;
    GRX_REQUIRE(h_col_ptrs && d_out, "grx_gather_columns: NULL pointer");
    const int64_t want = grx_ceil_div(n, 256 * 4);
    for (int c0 = 0; c0 < F; c0 += GRX_MAX_PTRS) {               // pointer table: GRX_MAX_PTRS per launch
        const int fc = (F - c0 < GRX_MAX_PTRS) ? F - c0 : GRX_MAX_PTRS;
        GrxPtrTable tab;
        for (int c = 0; c < fc; ++c) tab.p[c] = h_col_ptrs[c0 + c];
        const dim3 grid((unsigned)(want > 1024 ? 1024 : want), fc);
        {
            GRX_PROF(GRX_K_GATHER_COLUMNS, grx_stream(stream));
            gather_columns_kernel<<<grid, 256, 0, grx_stream(stream)>>>(n, fc, tab, d_out + (size_t)c0 * ld, ld);
        }
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

static int gram_grid(int64_t nrows)
{
    const int64_t tiles = grx_ceil_div(nrows, GR_TR);
    return (int)(tiles > GRX_NUM_CU * 4 ? GRX_NUM_CU * 4 : (tiles < 1 ? 1 : tiles));
}

// workgroups of the Gram kernels: the [pair][workgroup] partial table is bounded to ~128 MB
static int gram_pairs_grid(int64_t nrows, int k)
{
    const int64_t npairs = (int64_t)k * (k + 1) / 2 + 1;
    int64_t cap = ((int64_t)1 << 24) / npairs;
    if (cap > GRX_NUM_CU * 4) cap = GRX_NUM_CU * 4;
    if (cap < 64) cap = 64;
    const int64_t want = grx_ceil_div(grx_ceil_div(nrows > 0 ? nrows : 1, 16), 4);
    return (int)(want > cap ? cap : (want < 1 ? 1 : want));
}

size_t grx_gram_workspace_bytes(int64_t n, int k)
{
    if (k < 1) k = 1;
    const size_t npairs = (size_t)k * (k + 1) / 2 + 1;
    size_t bytes = grx_align_up((size_t)gram_pairs_grid(n, k) * npairs * 8, 256) +
                   grx_align_up((size_t)MAX_F_WIDE * (size_t)(k > MAX_F ? k : MAX_F) * 8, 256);
    if (k > GM_MAX_KT * 16) bytes += grx_align_up((size_t)k * (size_t)(n > 0 ? n : 1) * 8, 256);   // Y = X T
    return bytes;
}

int grx_gram(int64_t n, int F, const double *d_X, int64_t ldx, int64_t row_begin, int64_t row_end,
             const double *h_T, int k, double *d_out, void *d_workspace, size_t workspace_bytes,
             void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n && ldx >= n,
                "grx_gram: bad row range");
    if (F < 1 || F > MAX_F_WIDE || k < 1 || k > MAX_F_WIDE) {
        grx_set_error("grx_gram: F=%d k=%d outside [1,%d]", F, k, MAX_F_WIDE);
        return GRX_ERR_UNSUPPORTED;
    }
    GRX_REQUIRE(h_T != nullptr || k == F, "grx_gram: identity transform needs k == F");
    GRX_REQUIRE(d_X && d_out && d_workspace, "grx_gram: NULL pointer");
    if (workspace_bytes < grx_gram_workspace_bytes(n, k)) {
        grx_set_error("grx_gram: workspace %zu < %zu", workspace_bytes, grx_gram_workspace_bytes(n, k));
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    int grid = gram_grid(row_end - row_begin);
    const int kl_max = k < MAX_F ? k : MAX_F;                    // outputs of one launch
    const size_t npairs_max = (size_t)kl_max * (kl_max + 1) / 2;
    char *ws = reinterpret_cast<char *>(d_workspace);
    double *partial = reinterpret_cast<double *>(ws);
    double *dT = reinterpret_cast<double *>(ws + grx_align_up((size_t)gram_grid(n) * (npairs_max + 1) * 8, 256));
    if (h_T) GRX_CHECK_HIP(hipMemcpyAsync(dT, h_T, (size_t)F * k * 8, hipMemcpyHostToDevice, st));
    if (GramKernel mk = gram_mfma_pick(F, k, h_T != nullptr)) {
        // matrix-core path (F, k <= 48): one resident generation of workgroups
        const size_t lds = gram_mfma_lds_doubles((k + 15) / 16) * 8;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mk, 256, lds) != hipSuccess || nb < 1) nb = 1;
        const int64_t want = grx_ceil_div(grx_ceil_div(row_end - row_begin, 16), 4);
        int cap = nb * GRX_NUM_CU;
        if (cap > gram_grid(n)) cap = gram_grid(n) > 0 ? gram_grid(n) : 1;     // partial buffer is sized by gram_grid
        grid = (int)(want > cap ? cap : (want < 1 ? 1 : want));
        {
            GRX_PROF(GRX_K_GRAM, st);
            hipLaunchKernelGGL(mk, dim3(grid), dim3(256), lds, st, row_begin, row_end, F, k, d_X, ldx,
                               h_T ? dT : (const double *)nullptr, partial);
        }
        GRX_LAUNCH_CHECK();
        gram_finalize_kernel<<<(k * k + 1 + 3) / 4, 256, 0, st>>>(partial, grid, k, d_out, k, 0, k, 0, 1);
        GRX_LAUNCH_CHECK();
        return GRX_OK;
    }
    if (h_T && k <= GM_MAX_KT * 16 && k <= MAX_F) {
        // rare: a wide X whitened to few columns (rank-deficient features) -- the VALU kernel
        size_t lds = (size_t)k * GR_LD * 8;
        const int t_in_lds = lds + (size_t)F * k * 8 <= 60 * 1024;
        if (t_in_lds) lds += (size_t)F * k * 8;
        {
            GRX_PROF(GRX_K_GRAM, st);
            if (k <= 16) gram_kernel<true, 4><<<grid, 256, lds, st>>>(row_begin, row_end, F, k, d_X, ldx, dT, t_in_lds, partial, k, 0, k, 0);
            else if (k <= 32) gram_kernel<true, 8><<<grid, 256, lds, st>>>(row_begin, row_end, F, k, d_X, ldx, dT, t_in_lds, partial, k, 0, k, 0);
            else gram_kernel<true, 16><<<grid, 256, lds, st>>>(row_begin, row_end, F, k, d_X, ldx, dT, t_in_lds, partial, k, 0, k, 0);
        }
        GRX_LAUNCH_CHECK();
        gram_finalize_kernel<<<(k * k + 1 + 3) / 4, 256, 0, st>>>(partial, grid, k, d_out, k, 0, k, 0, 1);
        GRX_LAUNCH_CHECK();
        return GRX_OK;
    }
    // more than 48 columns: Y = X T materialised once (feature-major, workspace), then the Gram matrix
    // over pairs of 48-column groups with both operands straight from global memory
    const size_t npairs_all = (size_t)k * (k + 1) / 2 + 1;
    double *dTw = reinterpret_cast<double *>(ws + grx_align_up((size_t)gram_pairs_grid(n, k) * npairs_all * 8, 256));
    double *Yw = reinterpret_cast<double *>(reinterpret_cast<char *>(dTw) +
                                            grx_align_up((size_t)MAX_F_WIDE * (size_t)(k > MAX_F ? k : MAX_F) * 8, 256));
    if (h_T) GRX_CHECK_HIP(hipMemcpyAsync(dTw, h_T, (size_t)F * k * 8, hipMemcpyHostToDevice, st));
    const int pgrid = gram_pairs_grid(row_end - row_begin, k);
    if (k <= 16 * GF_MAX_KT && F <= 128 && (h_T != nullptr || F == k)) {
        // 49..128 output columns of at most 128 feature columns: the full-width kernel reads X once
        const int KT = (k + 15) / 16;
        const size_t lds = ((size_t)2 * 16 * 8 * GF_LD + (h_T ? (size_t)4 * 16 * KT * GF_LD : 0)) * 8;
        {
            GRX_PROF(GRX_K_GRAM, st);
            const dim3 g(pgrid), b(256);
#define GRX_GRAM_FULL(KTV)                                                                                              \
    if (h_T && lds > 64 * 1024)                                                                                         \
        GRX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_full_mfma_kernel<KTV, true>),            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
    if (h_T) gram_full_mfma_kernel<KTV, true><<<g, b, lds, st>>>(row_begin, row_end, F, k, d_X, ldx, dTw, partial, nullptr); \
    else gram_full_mfma_kernel<KTV, false><<<g, b, lds, st>>>(row_begin, row_end, F, k, d_X, ldx, nullptr, partial,       \
                                                              partial + (size_t)(npairs_all - 1) * pgrid)
            switch (KT) {
            case 4: GRX_GRAM_FULL(4); break;
            case 5: GRX_GRAM_FULL(5); break;
            case 6: GRX_GRAM_FULL(6); break;
            case 7: GRX_GRAM_FULL(7); break;
            default: GRX_GRAM_FULL(8); break;
            }
#undef GRX_GRAM_FULL
        }
        GRX_LAUNCH_CHECK();
        gram_finalize_kernel<<<(k * k + 1 + 3) / 4, 256, 0, st>>>(partial, pgrid, k, d_out, k, 0, k, 0, h_T ? 0 : 1);
        GRX_LAUNCH_CHECK();
        return GRX_OK;
    }
    const double *src = d_X;
    int64_t lds_src = ldx;
    {
        GRX_PROF(GRX_K_GRAM, st);
        if (h_T) {
            gram_transform_mfma_kernel<<<dim3(pgrid, (k + 127) / 128), 256, 0, st>>>(row_begin, row_end, F, k, d_X, ldx, dTw,
                                                                                    Yw, n);
            src = Yw;
            lds_src = n;
        }
        const int ngroups = (k + GP_GROUP - 1) / GP_GROUP;
        gram_pairs_mfma_kernel<<<dim3(pgrid, ngroups * (ngroups + 1) / 2), 256, 0, st>>>(row_begin, row_end, k, src,
                                                                                         lds_src, ngroups, partial);
        if (!h_T) {
            // sum(X): every workgroup's share into the last row of the partial table
            column_sum_kernel<<<pgrid, 256, 0, st>>>(row_begin, row_end, d_X, ldx, 0, k,
                                                     partial + (size_t)(npairs_all - 1) * pgrid);
        }
    }
    GRX_LAUNCH_CHECK();
    gram_finalize_kernel<<<(k * k + 1 + 3) / 4, 256, 0, st>>>(partial, pgrid, k, d_out, k, 0, k, 0, h_T ? 0 : 1);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

constexpr int PROJ_GRID = GRX_NUM_CU * 4;

size_t grx_project_workspace_bytes(int64_t n, int r)
{
    (void)n;
    if (r < 1) r = 1;
    return grx_align_up((size_t)PROJ_GRID * r * 5 * 8, 256) + grx_align_up((size_t)MAX_F_WIDE * MAX_R * 8, 256);
}

int grx_project(int64_t n, int F, const double *d_X, int64_t ldx, int64_t row_begin, int64_t row_end,
                const double *h_Z, int r, double *d_U, int64_t ldu, double *d_stats,
                void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n && ldx >= n && ldu >= n,
                "grx_project: bad row range");
    int rc = check_nmf_shape("grx_project", F, r);
    if (rc != GRX_OK) return rc;
    GRX_REQUIRE(d_X && h_Z && d_U && d_stats && d_workspace, "grx_project: NULL pointer");
    if (workspace_bytes < grx_project_workspace_bytes(n, r)) {
        grx_set_error("grx_project: workspace too small");
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    double *partial = reinterpret_cast<double *>(ws);
    double *dZ = reinterpret_cast<double *>(ws + grx_align_up((size_t)PROJ_GRID * r * 5 * 8, 256));
    GRX_CHECK_HIP(hipMemcpyAsync(dZ, h_Z, (size_t)F * r * 8, hipMemcpyHostToDevice, st));
    const int64_t want = grx_ceil_div(grx_ceil_div(row_end - row_begin, 16), 4);
    const int grid = (int)(want > PROJ_GRID ? PROJ_GRID : (want < 1 ? 1 : want));
    // one MFMA tile of output columns per launch (a second one for ranks 17 .. 32)
    for (int j0 = 0; j0 < r; j0 += MFMA_R) {
        const int rc_ = r - j0 < MFMA_R ? r - j0 : MFMA_R;
        double *part = partial + (size_t)PROJ_GRID * j0 * 5;
        { GRX_PROF(GRX_K_PROJECT, st);
        project_mfma_kernel<<<grid, 256, 0, st>>>(row_begin, row_end, F, rc_, d_X, ldx, dZ + j0, r, d_U + (size_t)j0 * ldu,
                                                  ldu, part);
        }
        GRX_LAUNCH_CHECK();
        project_finalize_kernel<<<rc_, 64, 0, st>>>(part, grid, rc_, d_stats + (size_t)j0 * 4);
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

int grx_nndsvd_apply(int64_t n, int r, double *d_U, int64_t ldu, int64_t row_begin, int64_t row_end,
                     const double *h_sign, const double *h_scale, double eps, double fill, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n && ldu >= n,
                "grx_nndsvd_apply: bad row range");
    GRX_REQUIRE(r >= 1 && r <= MAX_R, "grx_nndsvd_apply: r=%d outside [1,%d]", r, MAX_R);
    GRX_REQUIRE(d_U && h_sign && h_scale, "grx_nndsvd_apply: NULL pointer");
    if (row_end == row_begin) return GRX_OK;
    NndsvdArgs a;
    for (int j = 0; j < MAX_R; ++j) { a.sign[j] = j < r ? h_sign[j] : 0.0; a.scale[j] = j < r ? h_scale[j] : 0.0; }
    const int64_t want = grx_ceil_div(row_end - row_begin, 256 * 4);
    const dim3 grid((unsigned)(want > 1024 ? 1024 : want), r);
    { GRX_PROF(GRX_K_NNDSVD_APPLY, grx_stream(stream));
    nndsvd_apply_kernel<<<grid, 256, 0, grx_stream(stream)>>>(row_begin, row_end, r, d_U, ldu, a, eps, fill);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

// buffers of the update for more than 16 roles, behind the layout above: H H^T, the stacked table [W' | X]
// ((r + F) x ld), its Gram matrix and the Gram kernels' own workspace
struct WideRankLayout { size_t hh, y, g, gram_ws, gram_ws_bytes, total; int64_t ldy; };
static WideRankLayout wide_rank_layout(int64_t n, int F, int r)
{
    WideRankLayout L;
    const int K = r + F;
    L.ldy = (n > 0 ? n : 1);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += grx_align_up(bytes, 256); return at; };
    L.hh = take((size_t)r * r * 8);
    L.y = take((size_t)K * (size_t)L.ldy * 8);
    L.g = take(((size_t)K * K + 1) * 8);
    L.gram_ws_bytes = grx_gram_workspace_bytes(n, K);
    L.gram_ws = take(L.gram_ws_bytes);
    L.total = o;
    return L;
}

size_t grx_nmf_workspace_bytes(int64_t n, int F, int r)
{
    if (F < 1) F = 1;
    if (r < 1) r = 1;
    const size_t P = (size_t)r * F + (size_t)r * r;
    const size_t a = grx_align_up((size_t)MU_MAX_GRID * P * 8, 256);
    const size_t b = grx_align_up((size_t)RES_GRID * 8, 256);
    size_t bytes = a + b + 2 * grx_align_up((size_t)r * F * 8, 256);        // + the two H buffers of grx_nmf_iterate
    if (r > MFMA_R) bytes += wide_rank_layout(n, F, r).total;
    return bytes;
}

// the two r x F scratch copies of H at the end of the workspace (grx_nmf_iterate)
static double *nmf_h_scratch(void *d_workspace, int F, int r, int which)
{
    const size_t P = (size_t)r * F + (size_t)r * r;
    const size_t off = grx_align_up((size_t)MU_MAX_GRID * P * 8, 256) + grx_align_up((size_t)RES_GRID * 8, 256) +
                       (size_t)which * grx_align_up((size_t)r * F * 8, 256);
    return reinterpret_cast<double *>(reinterpret_cast<char *>(d_workspace) + off);
}

// One W pass for 17 .. 32 roles (see nmf_w_update_generic_kernel): W' in place, then d_AB = [W'^T X | W'^T W'] over the
// rows [row_begin, row_end) from the Gram matrix of the stacked table.
static int w_pass_wide_rank(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                            int64_t row_begin, int64_t row_end, const double *d_H, double *d_AB, void *d_workspace,
                            void *stream)
{
    hipStream_t st = grx_stream(stream);
    const size_t P = (size_t)r * F + (size_t)r * r;
    const size_t base = grx_align_up((size_t)MU_MAX_GRID * P * 8, 256) + grx_align_up((size_t)RES_GRID * 8, 256) +
                        2 * grx_align_up((size_t)r * F * 8, 256);
    const WideRankLayout L = wide_rank_layout(n, F, r);
    char *ws = reinterpret_cast<char *>(d_workspace) + base;
    double *dHH = reinterpret_cast<double *>(ws + L.hh);
    double *dY = reinterpret_cast<double *>(ws + L.y);
    double *dG = reinterpret_cast<double *>(ws + L.g);
    const int K = r + F;
    const int64_t rows = row_end - row_begin;
    if (rows > 0) {
        GRX_PROF(GRX_K_NMF_W_PASS, st);
        nmf_hht_kernel<<<1, 256, 0, st>>>(F, r, d_H, dHH);
        const int64_t want = grx_ceil_div(rows, 256);
        nmf_w_update_generic_kernel<<<(int)(want > RES_GRID ? RES_GRID : want), 256, 0, st>>>(
            row_begin, row_end, F, r, d_X, ldx, d_W, ldw, d_H, dHH, dY, L.ldy);
        // the X half of the stacked table (this rank's rows)
        GRX_CHECK_HIP(hipMemcpy2DAsync(dY + (size_t)r * L.ldy + row_begin, (size_t)L.ldy * 8, d_X + row_begin, (size_t)ldx * 8,
                                       (size_t)rows * 8, (size_t)F, hipMemcpyDeviceToDevice, st));
    }
    GRX_LAUNCH_CHECK();
    int rc = grx_gram(n, K, dY, L.ldy, row_begin, row_end, nullptr, K, dG, ws + L.gram_ws, L.gram_ws_bytes, stream);
    if (rc != GRX_OK) return rc;
    nmf_ab_from_gram_kernel<<<(int)grx_ceil_div((int64_t)P, 256), 256, 0, st>>>(F, r, dG, d_AB);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

// d_AB_prev / d_H_out (single-launch kernels only, F <= MAX_F): the kernel first applies the H update of the
// previous iteration to d_H (from d_AB_prev) and stores the result in d_H_out -- see nmf_w_pass_mfma_kernel
static int w_pass_impl(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                       int64_t row_begin, int64_t row_end, const double *d_H, double *d_AB,
                       const double *d_AB_prev, double *d_H_out,
                       void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n && ldx >= n && ldw >= n,
                "grx_nmf_w_pass: bad row range");
    int rc = check_nmf_shape("grx_nmf_w_pass", F, r);
    if (rc != GRX_OK) return rc;
    GRX_REQUIRE(d_X && d_W && d_H && d_AB && d_workspace, "grx_nmf_w_pass: NULL pointer");
    if (workspace_bytes < grx_nmf_workspace_bytes(n, F, r)) {
        grx_set_error("grx_nmf_w_pass: workspace too small");
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    if (r > MFMA_R) {
        GRX_REQUIRE(d_AB_prev == nullptr, "w_pass_impl: more than 16 roles have no fused H update");
        return w_pass_wide_rank(n, F, r, d_X, ldx, d_W, ldw, row_begin, row_end, d_H, d_AB, d_workspace, stream);
    }
    double *partial = reinterpret_cast<double *>(d_workspace);
    const int P = r * F + r * r;
    int grid;
    {
        const int FT = (F + 15) / 16;
        const int64_t nsub = grx_ceil_div(row_end - row_begin, 16);
        const int64_t want = grx_ceil_div(nsub, 4);
        GRX_PROF(GRX_K_NMF_W_PASS, st);
        if (F <= MAX_F) {
            const size_t lds = mfma_lds_doubles(FT) * 8;
            // exactly one resident generation of workgroups: every wave keeps its accumulators over
            // all of its sub-tiles and there is no partially filled last wave of workgroups
            const int cap = mfma_resident_grid(F, r, lds);
            grid = (int)(want > cap ? cap : (want < 1 ? 1 : want));
            hipLaunchKernelGGL(w_pass_kernel(F, r), dim3(grid), dim3(256), lds, st, row_begin, row_end, F, r, d_X,
                               ldx, d_W, ldw, d_H, partial, d_AB_prev, d_H_out);
        } else {
            (void)FT;
            GRX_REQUIRE(d_AB_prev == nullptr, "w_pass_impl: the chunked kernels have no fused H update");
            const int cap = GRX_NUM_CU * 4;
            grid = (int)(want > cap ? cap : (want < 1 ? 1 : want));
            if (r <= 8) nmf_w_update_wide_kernel<2><<<grid, 256, 0, st>>>(row_begin, row_end, F, r, d_X, ldx, d_W, ldw, d_H);
            else nmf_w_update_wide_kernel<4><<<grid, 256, 0, st>>>(row_begin, row_end, F, r, d_X, ldx, d_W, ldw, d_H);
            nmf_w_accum_wide_kernel<<<dim3(grid, (F + 31) / 32), 256, 0, st>>>(row_begin, row_end, F, r, d_X, ldx, d_W,
                                                                               ldw, partial);
        }
    }
    GRX_LAUNCH_CHECK();
    { GRX_PROF(GRX_K_REDUCE_PARTIALS, st);
    reduce_partials_kernel<<<(P + 3) / 4, 256, 0, st>>>(partial, grid, P, d_AB);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_nmf_w_pass(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                   int64_t row_begin, int64_t row_end, const double *d_H, double *d_AB,
                   void *d_workspace, size_t workspace_bytes, void *stream)
{
    return w_pass_impl(n, F, r, d_X, ldx, d_W, ldw, row_begin, row_end, d_H, d_AB, nullptr, nullptr, d_workspace,
                       workspace_bytes, stream);
}

int grx_nmf_w_pass_next(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                        int64_t row_begin, int64_t row_end, const double *d_H_prev, const double *d_AB_prev,
                        double *d_H_out, double *d_AB, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(d_H_prev && d_AB_prev && d_H_out && d_H_prev != d_H_out, "grx_nmf_w_pass_next: needs distinct H buffers");
    if (F > MAX_F || r > MFMA_R) {                              // chunked / wide-rank kernels: the update as a launch of its own
        GRX_PROF(GRX_K_NMF_H_UPDATE, grx_stream(stream));
        launch_h_update(F, r, d_H_prev, d_H_out, d_AB_prev, grx_stream(stream));
        return w_pass_impl(n, F, r, d_X, ldx, d_W, ldw, row_begin, row_end, d_H_out, d_AB, nullptr, nullptr, d_workspace,
                           workspace_bytes, stream);
    }
    return w_pass_impl(n, F, r, d_X, ldx, d_W, ldw, row_begin, row_end, d_H_prev, d_AB, d_AB_prev, d_H_out, d_workspace,
                       workspace_bytes, stream);
}

int grx_nmf_h_update(int F, int r, double *d_H, const double *d_AB, void *stream)
{
    int rc = check_nmf_shape("grx_nmf_h_update", F, r);
    if (rc != GRX_OK) return rc;
    GRX_REQUIRE(d_H && d_AB, "grx_nmf_h_update: NULL pointer");
    { GRX_PROF(GRX_K_NMF_H_UPDATE, grx_stream(stream));
    launch_h_update(F, r, d_H, d_H, d_AB, grx_stream(stream));
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_nmf_residual(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *d_W,
                     int64_t ldw, int64_t row_begin, int64_t row_end, const double *d_H, double *d_out,
                     void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n && ldx >= n && ldw >= n,
                "grx_nmf_residual: bad row range");
    int rc = check_nmf_shape("grx_nmf_residual", F, r);
    if (rc != GRX_OK) return rc;
    GRX_REQUIRE(d_X && d_W && d_H && d_out && d_workspace, "grx_nmf_residual: NULL pointer");
    if (workspace_bytes < grx_nmf_workspace_bytes(n, F, r)) {
        grx_set_error("grx_nmf_residual: workspace too small");
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    const size_t P = (size_t)r * F + (size_t)r * r;
    double *partial = reinterpret_cast<double *>(reinterpret_cast<char *>(d_workspace) +
                                                 grx_align_up((size_t)MU_MAX_GRID * P * 8, 256));
    const int64_t want = grx_ceil_div(row_end - row_begin, 256);
    const int grid = (int)(want > RES_GRID ? RES_GRID : (want < 1 ? 1 : want));
    { GRX_PROF(GRX_K_NMF_RESIDUAL, st);
    if (r <= MFMA_R) {
        nmf_residual_kernel<MFMA_R><<<grid, 256, (size_t)r * F * 8, st>>>(row_begin, row_end, F, r, d_X, ldx, d_W, ldw,
                                                                         d_H, partial);
    } else {
        GRX_TRY_LDS(nmf_residual_kernel<MAX_R>);
        nmf_residual_kernel<MAX_R><<<grid, 256, (size_t)r * F * 8, st>>>(row_begin, row_end, F, r, d_X, ldx, d_W, ldw,
                                                                        d_H, partial);
    }
    }
    GRX_LAUNCH_CHECK();
    { GRX_PROF(GRX_K_REDUCE_PARTIALS, st);
    reduce_partials_kernel<<<1, 256, 0, st>>>(partial, grid, 1, d_out);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_nmf_kl_cost(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *d_W, int64_t ldw,
                    int64_t row_begin, int64_t row_end, const double *d_H, double *d_out, void *d_workspace,
                    size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n && ldx >= n && ldw >= n,
                "grx_nmf_kl_cost: bad row range");
    int rc = check_nmf_shape("grx_nmf_kl_cost", F, r);
    if (rc != GRX_OK) return rc;
    GRX_REQUIRE(d_X && d_W && d_H && d_out && d_workspace, "grx_nmf_kl_cost: NULL pointer");
    if (workspace_bytes < grx_nmf_workspace_bytes(n, F, r)) {
        grx_set_error("grx_nmf_kl_cost: workspace too small");
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    const size_t P = (size_t)r * F + (size_t)r * r;
    double *partial = reinterpret_cast<double *>(reinterpret_cast<char *>(d_workspace) +
                                                 grx_align_up((size_t)MU_MAX_GRID * P * 8, 256));
    const int64_t want = grx_ceil_div(row_end - row_begin, 256);
    const int grid = (int)(want > RES_GRID ? RES_GRID : (want < 1 ? 1 : want));
    { GRX_PROF(GRX_K_NMF_RESIDUAL, st);
    if (r <= MFMA_R) {
        nmf_kl_cost_kernel<MFMA_R><<<grid, 256, (size_t)r * F * 8, st>>>(row_begin, row_end, F, r, d_X, ldx, d_W, ldw, d_H,
                                                                        partial);
    } else {
        GRX_TRY_LDS(nmf_kl_cost_kernel<MAX_R>);
        nmf_kl_cost_kernel<MAX_R><<<grid, 256, (size_t)r * F * 8, st>>>(row_begin, row_end, F, r, d_X, ldx, d_W, ldw, d_H,
                                                                       partial);
    }
    }
    GRX_LAUNCH_CHECK();
    { GRX_PROF(GRX_K_REDUCE_PARTIALS, st);
    reduce_partials_kernel<<<1, 256, 0, st>>>(partial, grid, 1, d_out);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

// `iters` multiplicative updates over rows [row_begin, row_end); comm != NULL: the partial sums [W^T X | W^T W] of
// every pass are all-reduced before the H update that consumes them (one small collective per iteration, enqueued
// on the same stream between the launches -- the host does not wait)
static int nmf_iterate_impl(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                            int64_t row_begin, int64_t row_end, double *d_H, double *d_AB, int iters, grx_comm *comm,
                            void *d_workspace, size_t workspace_bytes, void *stream)
{
    const size_t nAB = (size_t)r * F + (size_t)r * r;
    if (F <= MAX_F && r <= MFMA_R && iters > 0) {
        // two launches per iteration instead of three: the W pass of iteration i starts by applying the H update
        // of iteration i - 1 (every workgroup recomputes the r x F entries; workgroup 0 stores them, ping-pong
        // between two scratch copies so that no workgroup reads what another is writing); one stand-alone
        // update closes the block
        const double *h_in = d_H;
        for (int it = 0; it < iters; ++it) {
            double *h_out = nmf_h_scratch(d_workspace, F, r, it & 1);
            int rc = w_pass_impl(n, F, r, d_X, ldx, d_W, ldw, row_begin, row_end, h_in, d_AB, it ? d_AB : nullptr,
                                 it ? h_out : nullptr, d_workspace, workspace_bytes, stream);
            if (rc != GRX_OK) return rc;
            if (comm) {
                rc = grx_comm_all_reduce(comm, d_AB, nAB, GRX_F64, GRX_SUM, stream);
                if (rc != GRX_OK) return rc;
            }
            if (it) h_in = h_out;
        }
        {
            GRX_PROF(GRX_K_NMF_H_UPDATE, grx_stream(stream));
            launch_h_update(F, r, h_in, d_H, d_AB, grx_stream(stream));
        }
        GRX_LAUNCH_CHECK();
    } else {
        for (int it = 0; it < iters; ++it) {
            int rc = grx_nmf_w_pass(n, F, r, d_X, ldx, d_W, ldw, row_begin, row_end, d_H, d_AB, d_workspace,
                                    workspace_bytes, stream);
            if (rc != GRX_OK) return rc;
            if (comm) {
                rc = grx_comm_all_reduce(comm, d_AB, nAB, GRX_F64, GRX_SUM, stream);
                if (rc != GRX_OK) return rc;
            }
            rc = grx_nmf_h_update(F, r, d_H, d_AB, stream);
            if (rc != GRX_OK) return rc;
        }
    }
    return GRX_OK;
}

int grx_nmf_iterate(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                    double *d_H, double *d_AB, double *d_err, int iters, void *d_workspace,
                    size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(iters >= 0, "grx_nmf_iterate: iters < 0");
    int rc = nmf_iterate_impl(n, F, r, d_X, ldx, d_W, ldw, 0, n, d_H, d_AB, iters, nullptr, d_workspace, workspace_bytes,
                              stream);
    if (rc != GRX_OK) return rc;
    if (d_err)
        return grx_nmf_residual(n, F, r, d_X, ldx, d_W, ldw, 0, n, d_H, d_err, d_workspace, workspace_bytes, stream);
    return GRX_OK;
}

int grx_nmf_iterate_rows(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                         int64_t row_begin, int64_t row_end, double *d_H, double *d_AB, int iters, grx_comm *comm,
                         void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(iters >= 0, "grx_nmf_iterate_rows: iters < 0");
    return nmf_iterate_impl(n, F, r, d_X, ldx, d_W, ldw, row_begin, row_end, d_H, d_AB, iters, comm, d_workspace,
                            workspace_bytes, stream);
}

}  // extern "C"

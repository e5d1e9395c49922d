// Gaussian MF half-sweep kernels (rows a8, a9 of SURVEY.md section 8).
//
// A factor half-sweep is two phases:
//   accumulate : per row r, S = sum_j (COV_other[o_j] + m_j m_j^T) (packed lower
//                triangle) and w = sum_j m_j * resid_j.  Pure streaming: every
//                rating gathers one packed covariance row (K(K+1)/2 values) and
//                one mean row.  One wavefront per task (run of <= PMF_GAUSS_CHUNK
//                ratings of one row), sums in registers, written once.  The
//                sums go IN PLACE into COV_side[r] / FACTOR_side[r] (nobody
//                gathers this side during its own sweep), so no row-sized
//                scratch is needed at 1M+ rows.
//   solve      : per non-empty row, COV[r] = inv(I/eta2 + S/sigma2) and
//                FACTOR[r] = COV[r] w / sigma2, one wavefront per row with the
//                whole K x K matrix in registers (lane = column).
// K = 64 / fp32 (the benchmark configuration) has a dedicated accumulate kernel
// that forms sum_j m_j m_j^T on the matrix cores (v_mfma_f32_32x32x2_f32: exact
// fp32 FMA chains) while the VALU only adds the gathered covariance rows.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "pmf_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T>
struct GaussParams {
    const PmfTask *tasks;
    int64_t n_tasks;
    const PmfSplitRow *split;
    const int32_t *other;
    const T *val;
    const T *factor_other;
    const T *cov_other;
    const T *bias_self;   // null when the model has no biases
    const T *bias_other;
    T *partial;           // [n_slots][cov_stride + kpad]
    // destination of a complete row's raw sums
    T *dst_s;
    int64_t dst_s_stride;
    T *dst_w;
    int64_t dst_w_stride;
    int K, kpad, kp, cov_stride;
};

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wavefront execute in order; this only stops the
    // compiler from moving LDS accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ PmfTask load_task_uniform(const PmfTask *tasks, int64_t id) {
    PmfTask t = tasks[id];
    PmfTask r;
    r.row = rfl(t.row);
    r.len = rfl(t.len);
    r.slot = rfl(t.slot);
    r.pad = 0;
    unsigned lo = (unsigned)rfl((int)(t.start & 0xFFFFFFFFll)), hi = (unsigned)rfl((int)(t.start >> 32));
    r.start = (int64_t)(((unsigned long long)hi << 32) | lo);
    return r;
}

// packed index p -> (row, col) of the lower triangle
__device__ __forceinline__ void tri_rc(int p, int &r, int &c) {
    r = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while ((r + 1) * (r + 2) / 2 <= p) ++r;
    while (r * (r + 1) / 2 > p) --r;
    c = p - r * (r + 1) / 2;
}

// ---------------------------------------------------------------------------
// accumulate, generic (any K <= 256, fp32 / fp64): VALU outer products
// ---------------------------------------------------------------------------
// One wavefront per task.  The packed range is covered in passes of 64 x CH
// 4-element chunks (lane owns chunks q = lane + 64 s of the pass); per pass the
// task's ratings are streamed two at a time: their mean rows are staged in LDS
// (the outer product needs m[r] m[c] for every packed entry), their covariance
// chunks are loaded as 16/32-byte accesses (2 x CH in flight per lane) and
// acc += V + m[r] m[c].  The (r, c) of a lane's entries are pass constants.
template <typename T, int KR>
__device__ __forceinline__ void solve_from_image(const T *img, T wj, int K, int kpad, T inv_sigma2, T inv_eta2,
                                                 T *vout, T *mout, int lane);

// KS > 0 (K <= 64): a task that is a whole row keeps its sums in an LDS image and is solved on the spot by the
// same wavefront (the KS-row register sweep), as in the MFMA kernels -- this is how fp64 contexts (parity mode)
// stop paying a separate solve launch.
template <typename T, int KS = 0>
__global__ __launch_bounds__(256) void gauss_accum_generic_kernel(GaussParams<T> p, T inv_sigma2 = (T)0, T inv_eta2 = (T)0,
                                                                  T *cov_self = nullptr, T *factor_self = nullptr) {
    constexpr int CH = sizeof(T) == 8 ? 5 : 4;  // chunks per lane per pass -> 64 * CH * 4 packed entries per pass (fp64, K = 64: 2 passes instead of 3)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t task_id = (int64_t)blockIdx.x * 4 + wave;
    if (task_id >= p.n_tasks) return;
    const PmfTask t = load_task_uniform(p.tasks, task_id);
    const int lds_per_wave = 2 * p.kpad + (KS > 0 ? p.cov_stride : 0);
    T *m0 = reinterpret_cast<T *>(smem_raw) + (int64_t)wave * lds_per_wave;
    T *m1 = m0 + p.kpad;
    T *img = m1 + p.kpad;                       // [cov_stride], KS > 0 only
    const bool solve_here = KS > 0 && t.slot < 0;
    const int32_t *col = p.other + t.start;
    const T *val = p.val + t.start;
    const T b_self = p.bias_self ? p.bias_self[t.row] : (T)0;
    const int chunks = p.cov_stride / PMF_VEC;
    T *out_s, *out_w;
    if (t.slot >= 0) {
        out_s = p.partial + (int64_t)t.slot * (p.cov_stride + p.kpad);
        out_w = out_s + p.cov_stride;
    } else {
        out_s = p.dst_s + (int64_t)t.row * p.dst_s_stride;
        out_w = p.dst_w + (int64_t)t.row * p.dst_w_stride;
    }
    T wacc[4] = {(T)0, (T)0, (T)0, (T)0};  // K <= 256: k = lane + 64 e
    for (int q0 = 0; q0 < chunks; q0 += 64 * CH) {
        Vec4<T> acc[CH];
        int rc[CH][PMF_VEC];  // r | c << 8 of every entry this lane owns in this pass
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            acc[s] = zero4<T>();
            const int q = q0 + lane + 64 * s;
#pragma unroll
            for (int e = 0; e < PMF_VEC; ++e) {
                int r = 0, c = 0;
                const int pi = q * PMF_VEC + e;
                if (q < chunks && pi < p.kp) tri_rc(pi, r, c);
                rc[s][e] = r | (c << 8);
            }
        }
        for (int j = 0; j < t.len; j += 2) {
            const bool two = j + 1 < t.len;
            const int o0 = col[j], o1 = two ? col[j + 1] : o0;
            for (int k = lane; k < p.kpad; k += 64) {
                m0[k] = p.factor_other[(int64_t)o0 * p.kpad + k];
                m1[k] = two ? p.factor_other[(int64_t)o1 * p.kpad + k] : (T)0;
            }
            const T *v0 = p.cov_other + (int64_t)o0 * p.cov_stride;
            const T *v1 = p.cov_other + (int64_t)o1 * p.cov_stride;
            Vec4<T> a[CH], b[CH];
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int q = min(q0 + lane + 64 * s, chunks - 1);  // clamped lanes are never stored
                a[s] = load4(v0 + (int64_t)q * PMF_VEC);
                b[s] = two ? load4(v1 + (int64_t)q * PMF_VEC) : zero4<T>();
            }
            wave_lds_fence();
            if (q0 == 0) {
                const T r0 = val[j] - b_self - (p.bias_other ? p.bias_other[o0] : (T)0);
                const T r1 = two ? val[j + 1] - b_self - (p.bias_other ? p.bias_other[o1] : (T)0) : (T)0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = lane + 64 * e;
                    if (k < p.K) wacc[e] += m0[k] * r0 + m1[k] * r1;
                }
            }
#pragma unroll
            for (int s = 0; s < CH; ++s)
#pragma unroll
                for (int e = 0; e < PMF_VEC; ++e) {
                    const int r = rc[s][e] & 255, c = rc[s][e] >> 8;
                    // reference order: (V_j + m_j m_j^T) added rating by rating
                    acc[s].v[e] += fma(m0[r], m0[c], a[s].v[e]);
                    acc[s].v[e] += fma(m1[r], m1[c], b[s].v[e]);
                }
            wave_lds_fence();
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int q = q0 + lane + 64 * s;
            if (q < chunks) {
#pragma unroll
                for (int e = 0; e < PMF_VEC; ++e)
                    if (q * PMF_VEC + e >= p.kp) acc[s].v[e] = (T)0;
                store4((solve_here ? img : out_s) + (int64_t)q * PMF_VEC, acc[s]);
            }
        }
    }
    if constexpr (KS > 0) {
        if (solve_here) {
            wave_lds_fence();
            solve_from_image<T, KS>(img, wacc[0], p.K, p.kpad, inv_sigma2, inv_eta2,
                                    cov_self + (int64_t)t.row * p.cov_stride, factor_self + (int64_t)t.row * p.kpad, lane);
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = lane + 64 * e;
        if (k < p.kpad) out_w[k] = k < p.K ? wacc[e] : (T)0;
    }
}

template <typename T>
struct SolveParams {
    const int32_t *rows;  // list of rows to solve, or null = rows row0 .. row0 + n (skip rows with S == 0)
    int64_t row0;
    int64_t n;
    const T *src_s;
    int64_t src_s_stride;
    const T *src_w;
    int64_t src_w_stride;
    T *cov;
    T *factor;
    T inv_sigma2, inv_eta2;
    int K, kpad, kp, cov_stride;
};

__device__ __forceinline__ float readlane_dyn(float x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ double readlane_dyn(double x, int lane) {
    long long q = __builtin_bit_cast(long long, x);
    int lo = __builtin_amdgcn_readlane((int)(q & 0xFFFFFFFFll), lane);
    int hi = __builtin_amdgcn_readlane((int)(q >> 32), lane);
    q = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, q);
}


// One wavefront per row, lane j = column j, B[i] = row (i + step) mod KR.
// Symmetric sweep on the Jacobi-scaled matrix (unit diagonal => pivots in
// (0, 1], which keeps the column update fma(-s, 1 - 1/d, s) = s/d free of
// cancellation).  The pivot row is always register 0 because every update
// writes row i into register i-1; after KR steps the rows are back in place
// and B = -inverse.  `img` is the packed lower triangle of S in LDS, `wj` lane
// j's right-hand side; writes the packed inverse and the mean.
template <typename T, int KR>
__device__ __forceinline__ void solve_from_image(const T *img, T wj, int K, int kpad, T inv_sigma2, T inv_eta2,
                                                 T *vout, T *mout, int lane) {
    const int j = lane;
    T B[KR];
    const int jc = j < K ? j : 0;
#pragma unroll
    for (int i = 0; i < KR; ++i) {
        const int ic = i < K ? i : 0;
        const int lo = ic < jc ? ic : jc, hi = ic < jc ? jc : ic;
        T s = img[hi * (hi + 1) / 2 + lo] * inv_sigma2;
        if (!(i < K && j < K)) s = (T)0;
        if (i == j) s += (i < K) ? inv_eta2 : (T)1;
        B[i] = s;
    }
    // Jacobi scaling g_j = 1/sqrt(P_jj)
    T diag = (T)1;
#pragma unroll
    for (int i = 0; i < KR; ++i) {
        const T dii = readlane_dyn(B[i], i);
        if (j == i) diag = dii;
    }
    const T g = (T)1 / sqrt(diag);
#pragma unroll
    for (int i = 0; i < KR; ++i) B[i] = B[i] * g * readlane_dyn(g, i);

    for (int k = 0; k < KR; ++k) {
        const T v = B[0];
        const T pinv = (T)1 / readlane_dyn(v, k);
        const T u = v * pinv;
        const T uc = (j == k) ? ((T)1 - pinv) : u;
        // column k as scalars first (one batch of v_readlane into SGPRs), then the
        // rank-1 update: back-to-back readlane -> use pairs cost a wait state each
        constexpr int SB = sizeof(T) == 8 ? (KR < 16 ? KR : 16) : (KR < 32 ? KR : 32);   // fp64: 16 scalars = 32 SGPRs per batch (32 would spill)
#pragma unroll
        for (int i0 = 1; i0 < KR; i0 += SB) {
            T sc[SB];
#pragma unroll
            for (int q = 0; q < SB; ++q)
                if (i0 + q < KR) sc[q] = readlane_dyn(B[i0 + q], k);
#pragma unroll
            for (int q = 0; q < SB; ++q)
                if (i0 + q < KR) B[i0 + q - 1] = fma(-sc[q], uc, B[i0 + q]);
        }
        B[KR - 1] = (j == k) ? -pinv : u;
    }
    // V = -(g_i g_j) B ;  m_j = inv_sigma2 * sum_i V[i][j] w_i
    // (g2: an opaque copy, so the 64 per-row scale scalars are re-read here instead of
    //  being kept alive in SGPRs across the whole sweep loop and spilled)
    T g2 = g;
    asm volatile("" : "+v"(g2));
    T mj = (T)0;
#pragma unroll
    for (int i = 0; i < KR; ++i) {
        const T vij = -B[i] * g2 * readlane_dyn(g2, i);
        mj = fma(vij, readlane_dyn(wj, i), mj);
        if (i < K && j <= i) vout[i * (i + 1) / 2 + j] = vij;
    }
    if (j < kpad) mout[j] = (j < K) ? mj * inv_sigma2 : (T)0;
}

template <typename T, int KR>
__global__ __launch_bounds__(256) void gauss_solve_reg_kernel(SolveParams<T> p) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
    if (idx >= p.n) return;
    const int row = p.rows ? rfl(p.rows[idx]) : (int)(p.row0 + idx);
    const T *S = p.src_s + (int64_t)row * p.src_s_stride;
    if (!p.rows && S[0] == (T)0) return;  // no rating anywhere for this row
    T *img = reinterpret_cast<T *>(smem_raw) + (int64_t)wave * p.cov_stride;
    for (int q = lane * PMF_VEC; q < p.cov_stride; q += 64 * PMF_VEC) store4(img + q, load4(S + q));
    wave_lds_fence();
    const T wj = (lane < p.K) ? p.src_w[(int64_t)row * p.src_w_stride + lane] : (T)0;
    solve_from_image<T, KR>(img, wj, p.K, p.kpad, p.inv_sigma2, p.inv_eta2,
                            p.cov + (int64_t)row * p.cov_stride, p.factor + (int64_t)row * p.kpad, lane);
}

// 64 < K <= 128, fp32: a row is handled by the TWO wavefronts of a 128-thread block.
// LDS layout shared by the accumulate kernel and the solve:
//   img [8256]  full 128-row packed lower triangle: S for rows < K; for the padding rows
//               >= K zeros with a diagonal chosen so that P_ii = S_ii/sigma2 + 1/eta2 = 1,
//               which removes every K-dependent mask from the register build
//   xbuf [1024] the MFMA sweep's two 128 x 4 pivot panels (double-buffered by step)
//   gbuf [128] (Jacobi scales), wbuf [128] (right-hand side)
// (Rounds 1-2 swept the matrix on the VALU, split between the two waves by rows (K <= 96) or by columns, one scalar
//  pivot at a time with the pivot row broadcast through LDS -- 85 VALU instructions and a 128-scalar LDS round trip per
//  pivot; the block sweep below replaced them in round 3: profiles/r03_solve_mfma_vs_valu.jsonl.)
#define PAIR_IMG 8256
#define PAIR_XBUF 1024   // two 128 x 4 pivot panels
#define PAIR_LDS_FLOATS (PAIR_IMG + PAIR_XBUF + 128 + 128)

__device__ __forceinline__ float pair_pad_diag(float inv_sigma2, float inv_eta2) { return (1.f - inv_eta2) / inv_sigma2; }

// ---- 64 < K <= 128 on the matrix cores: block sweep, four pivots per step ------------------------------------------
// The symmetric sweep of the K x K matrix as RANK-4 updates on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).  The
// Jacobi-scaled matrix B (padded to 16 TT rows with the identity) lives in the accumulator tiles of the block's two
// wavefronts: wave 0 owns tile rows [0, TH), wave 1 the rest, every tile column; lane l of a tile holds column l & 15 of
// rows 4 (l >> 4) + 0..3.  One step sweeps the pivot set Kb = {p .. p + 3}: with R = B[Kb, :] (4 x n, the pivot rows),
// D = B[Kb, Kb] and U = R^T D^-1 (n x 4),
//     B[i][j] -= U[i,:] R[:,j]  (i, j not in Kb),   B[i, Kb] = U[i,:],   B[Kb, j] = U[j,:]^T,   B[Kb, Kb] = -D^-1
// -- four scalar sweeps in one.  All four cases come out of ONE MFMA per tile, without cancellation, when
//   * the accumulator entries in the pivot rows and pivot columns are zeroed first,
//   * the A operand (lane: row i, pivot index c) is -U[i][c], but D^-1[c][i - p] for the pivot rows themselves,
//   * the B operand (lane: pivot index c, column j) is R[c][j], but -delta(c, j - p) for the pivot columns themselves:
// a pivot row then receives  sum_c D^-1[c][k'] R[c][j] = U[j][k'],  a pivot column  sum_c (-U[i][c]) (-delta) = U[i][c'],
// and the pivot block  sum_c D^-1[c][k'] (-delta(c, c')) = -D^-1.  The only data exchanged per step is the pivot panel R
// (128 x 4 floats, double-buffered in LDS): its owner -- the 16 lanes that hold those four rows in their four
// accumulator registers -- writes it as one float4 per column as soon as that tile row has been updated (the tile row
// of the NEXT pivots is updated first), so the panel's write -> barrier -> read trip runs under the remaining MFMAs.
// Every wave inverts the 4 x 4 pivot block itself (scalar sweep in registers, identical on all lanes).  Per step a
// wave issues TH x TT MFMAs (32 at K = 128: 1024 matrix-pipe cycles) and about 150 VALU instructions, against the
// 128 LDS-broadcast-bound scalar pivots x 85 VALU instructions of the VALU splits it replaced.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp_nr(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);   // one Newton step: within an ulp of 1 / x
}

// Precondition: img / wbuf are complete and the block has synchronised.  rt = PAIR_XBUF floats.
// W = the wavefront (0 / 1) as a compile-time constant: its tile rows are then known statically, the image build and the
// packed write-back know for every tile whether it lies below, on or above the diagonal (no per-element compare; the
// tiles above the diagonal are not written at all), and the two waves run their own copy of the code with the same
// sequence of barriers.
template <int TT, int W>
__device__ __forceinline__ void pair_solve_mfma_wave(float *img, float *rt, float *gbuf, const float *wbuf, int K, int kpad,
                                                     int cov_stride, float inv_sigma2, float inv_eta2, float *vout, float *mout,
                                                     int lane) {
    constexpr int TH = (TT + 1) / 2;
    constexpr int wave = W;
    const int tid = 64 * wave + lane;
    gbuf[tid] = 1.f / sqrtf(img[tid * (tid + 3) / 2] * inv_sigma2 + inv_eta2);   // (rows >= K: the padding diagonal gives 1)
    __syncthreads();
    const int lc = lane & 15, lg = lane >> 4;
    constexpr int base = W ? TH : 0, nrows = W ? TT - TH : TH;
    f32x4 D[TH][TT];
    float gj[TT];
    int tj[TT];                               // packed offset of this lane's column j as a ROW
#pragma unroll
    for (int J = 0; J < TT; ++J) {
        gj[J] = gbuf[16 * J + lc];
        tj[J] = (16 * J + lc) * (16 * J + lc + 1) / 2;
    }
#pragma unroll
    for (int ii = 0; ii < TH; ++ii) {
        if (ii < nrows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (base + ii) + 4 * lg + r;
                const float gi = gbuf[i] * inv_sigma2;
                const int ti = i * (i + 1) / 2;
#pragma unroll
                for (int J = 0; J < TT; ++J) {
                    const int j = 16 * J + lc;
                    float v;
                    if (base + ii > J) v = img[ti + j];                                  // a tile below the diagonal
                    else if (base + ii < J) v = img[tj[J] + i];                          // above: the transposed entry
                    else v = img[i >= j ? ti + j : tj[J] + i];
                    v *= gi;
                    if (base + ii == J && i == j) v = fmaf(inv_eta2, gbuf[i], v);
                    D[ii][J][r] = v * gj[J];
                }
            }
        } else {
#pragma unroll
            for (int J = 0; J < TT; ++J) D[ii][J] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the 16 lanes holding pivot rows 4 qg .. 4 qg + 3 of local tile row `ii` write them as the panel of a step
    auto publish = [&](float *dst, int ii_local, int qg) {
#pragma unroll
        for (int ii = 0; ii < TH; ++ii)
            if (ii == ii_local && lg == qg) {
#pragma unroll
                for (int J = 0; J < TT; ++J)
                    *reinterpret_cast<float4 *>(dst + (16 * J + lc) * 4) = make_float4(D[ii][J][0], D[ii][J][1], D[ii][J][2], D[ii][J][3]);
            }
    };
    const int steps = (K + 3) >> 2;
    if (wave == 0) publish(rt, 0, 0);
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        const int p = 4 * s, Ip = p >> 4, q = p & 15, qg = q >> 2;
        const float *rb = rt + (s & 1) * 512;
        __syncthreads();   // the panel of step s is visible; everybody has finished reading the other buffer
        // D = B[Kb, Kb] (4 x 4, symmetric) -> -D^-1 by four scalar sweeps, identical on every lane
        float m[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 col = *reinterpret_cast<const float4 *>(rb + (p + c) * 4);
            m[0][c] = col.x; m[1][c] = col.y; m[2][c] = col.z; m[3][c] = col.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float pinv = rcp_nr(m[k][k]);
            float u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = m[i][k] * pinv;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i != k && j != k) m[i][j] = fmaf(-u[i], m[k][j], m[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i != k) m[i][k] = m[k][i] = u[i];
            m[k][k] = -pinv;
        }
        // dcol[k] = D^-1[k][lg] (this lane's pivot index as an A operand) = -m[k][lg]
        float dcol[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dcol[k] = -(lg == 0 ? m[k][0] : lg == 1 ? m[k][1] : lg == 2 ? m[k][2] : m[k][3]);
        const int cq = lc - q;                      // 0..3 when this lane's row / column index is a pivot
        const bool piv = cq >= 0 && cq < 4;
        const float apiv = cq == 0 ? dcol[0] : cq == 1 ? dcol[1] : cq == 2 ? dcol[2] : dcol[3];   // D^-1[lg][cq]
        float bop[TT];
#pragma unroll
        for (int J = 0; J < TT; ++J) {
            bop[J] = rb[(16 * J + lc) * 4 + lg];
            if (J == Ip && piv) bop[J] = (lg == cq) ? -1.f : 0.f;
        }
        auto update_row = [&](auto ii_tag) {
            constexpr int ii = decltype(ii_tag)::value;
            const int I = base + ii;
            const float4 x = *reinterpret_cast<const float4 *>(rb + (16 * I + lc) * 4);
            float a = -fmaf(x.w, dcol[3], fmaf(x.z, dcol[2], fmaf(x.y, dcol[1], x.x * dcol[0])));
            const bool prow = I == Ip;              // wave-uniform
            if (prow && piv) a = apiv;
            // the accumulator entries of the pivot rows (tile row Ip) and pivot columns (tile column Ip) start from zero.
            // Wave-uniform BRANCHES around the selects (the empty asm keeps the compiler from turning them back into
            // selects on every tile): vector instructions take their issue cycles from the fp32 matrix pipe
            if (prow) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int J = 0; J < TT; ++J)
                    if (lg == qg) D[ii][J] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int J = 0; J < TT; ++J) {
                if (J == Ip) {
                    asm volatile("" ::: "memory");
                    if (piv) D[ii][J] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                D[ii][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bop[J], D[ii][J], 0, 0, 0);
            }
        };
        // the tile row that holds the next pivots first, then its panel, then the rest
        const int pn = p + 4, inext = (s + 1 < steps) ? (pn >> 4) - base : -1;
        auto for_rows = [&](auto &&body) {
            if constexpr (TH > 0) body(std::integral_constant<int, 0>{});
            if constexpr (TH > 1) body(std::integral_constant<int, 1>{});
            if constexpr (TH > 2) body(std::integral_constant<int, 2>{});
            if constexpr (TH > 3) body(std::integral_constant<int, 3>{});
        };
        for_rows([&](auto tag) {
            if (decltype(tag)::value == inext) update_row(tag);
        });
        if (inext >= 0 && inext < nrows) publish(rt + ((s + 1) & 1) * 512, inext, (pn & 15) >> 2);
        for_rows([&](auto tag) {
            if (decltype(tag)::value != inext && decltype(tag)::value < nrows) update_row(tag);
        });
    }
    // V = -(g_i g_j) B (packed lower triangle, staged in the LDS image and written out coalesced);
    // m_i = inv_sigma2 * sum_j V[i][j] w_j: each wave has its rows complete
    __syncthreads();   // (the image is free since the build; the last panel reads are done)
    float xj[TT];
#pragma unroll
    for (int J = 0; J < TT; ++J) xj[J] = gj[J] * wbuf[16 * J + lc];
#pragma unroll
    for (int ii = 0; ii < TH; ++ii) {
        if (ii < nrows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (base + ii) + 4 * lg + r;
                const float gi = -gbuf[i];
                float acc = 0.f;
#pragma unroll
                for (int J = 0; J < TT; ++J) {
                    const int j = 16 * J + lc;
                    const float b = D[ii][J][r] * gi;
                    acc = fmaf(b, xj[J], acc);
                    if (base + ii > J) {                       // below the diagonal: every entry is stored
                        if (i < K) img[i * (i + 1) / 2 + j] = b * gj[J];
                    } else if (base + ii == J) {
                        if (j <= i && i < K) img[i * (i + 1) / 2 + j] = b * gj[J];
                    }
                }
                acc = group_sum<16>(acc);
                if (lc == 0 && i < kpad) mout[i] = i < K ? acc * inv_sigma2 : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    for (int qv = threadIdx.x * PMF_VEC; qv < cov_stride; qv += 128 * PMF_VEC) store4(vout + qv, load4(img + qv));
}

template <int TT>
__device__ __forceinline__ void pair_solve_mfma(float *img, float *rt, float *gbuf, const float *wbuf, int K, int kpad,
                                                int cov_stride, float inv_sigma2, float inv_eta2, float *vout, float *mout,
                                                int wave, int lane) {
    if (wave == 0) pair_solve_mfma_wave<TT, 0>(img, rt, gbuf, wbuf, K, kpad, cov_stride, inv_sigma2, inv_eta2, vout, mout, lane);
    else pair_solve_mfma_wave<TT, 1>(img, rt, gbuf, wbuf, K, kpad, cov_stride, inv_sigma2, inv_eta2, vout, mout, lane);
}

template <int MT>
__global__ __launch_bounds__(128, 2) void gauss_solve_pair_kernel(SolveParams<float> p) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *img = reinterpret_cast<float *>(smem_raw);
    float *xbuf = img + PAIR_IMG, *gbuf = xbuf + PAIR_XBUF, *wbuf = gbuf + 128;
    const int wave = rfl(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t idx = blockIdx.x;
    const int row = p.rows ? p.rows[idx] : (int)(p.row0 + idx);
    const float *S = p.src_s + (int64_t)row * p.src_s_stride;
    if (!p.rows && S[0] == 0.f) return;  // uniform for the whole block
    const int K = p.K, j = 64 * wave + lane;
    const float pad_diag = pair_pad_diag(p.inv_sigma2, p.inv_eta2);
    for (int q = threadIdx.x * PMF_VEC; q < PAIR_IMG; q += 128 * PMF_VEC) {
        Vec4<float> v = q < p.cov_stride ? load4(S + q) : zero4<float>();
        if (q + PMF_VEC > p.kp) {
#pragma unroll
            for (int e = 0; e < PMF_VEC; ++e)
                if (q + e >= p.kp) {
                    int r, c;
                    tri_rc(q + e, r, c);
                    v.v[e] = (r == c) ? pad_diag : 0.f;
                }
        }
        store4(img + q, v);
    }
    wbuf[j] = j < K ? p.src_w[(int64_t)row * p.src_w_stride + j] : 0.f;
    __syncthreads();
    pair_solve_mfma<MT>(img, xbuf, gbuf, wbuf, K, p.kpad, p.cov_stride, p.inv_sigma2, p.inv_eta2,
                        p.cov + (int64_t)row * p.cov_stride, p.factor + (int64_t)row * p.kpad, wave, lane);
}

// ---------------------------------------------------------------------------
// accumulate (+ fused solve), fp32, 64 < K <= 128: two wavefronts per task
// ---------------------------------------------------------------------------
// Wave w streams half of the packed covariance chunks ([w * H, (w+1) * H), up to 17
// 16-byte chunks per lane) and owns five of the ten lower 32x32 blocks of sum m m^T
// (wave 0: (0,0) (1,0) (1,1) (2,0) (2,1); wave 1: (2,2) (3,0) (3,1) (3,2) (3,3)).
// Both waves fold their blocks into the shared LDS image; a complete row is then
// solved in place by the same two waves (pair_solve_mfma).
// NT = chunk columns per wave (host picks the smallest that covers ceil(chunks / 2) / 64):
// 17 for K = 128 (1032 chunks per wave), 13 for K <= 112, 9 for K <= 92.
template <int NT, bool FUSE, int MT = 0>
__global__ __launch_bounds__(128, 2) void gauss_accum_mfma128_kernel(GaussParams<float> p, float inv_sigma2, float inv_eta2,
                                                                     float *cov_self, float *factor_self) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *img = reinterpret_cast<float *>(smem_raw);
    float *xbuf = img + PAIR_IMG, *gbuf = xbuf + PAIR_XBUF, *wbuf = gbuf + 128;
    const int wave = rfl(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const PmfTask t = load_task_uniform(p.tasks, blockIdx.x);
    const int h = lane >> 5, c = lane & 31;
    const int K = p.K, kpad = p.kpad, stride = p.cov_stride, chunks = p.cov_stride / PMF_VEC;
    const int half = (chunks + 1) / 2;
    const int q_begin = wave ? half : 0, q_end = wave ? chunks : half;
    const int32_t *col = p.other + t.start;
    const float *val = p.val + t.start;
    const float b_self = p.bias_self ? p.bias_self[t.row] : 0.f;

    f32x16 d[5];
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[b][r] = 0.f;
    float4 acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float wA = 0.f, wB = 0.f;  // rhs segments 2*wave and 2*wave + 1

    // The task's row ids and ratings are fetched 64 at a time (one coalesced load each) and handed
    // out with v_readlane: the ids are SGPRs (scalar address arithmetic, SGPR-base loads) and no
    // trip waits for an index.  Every load of a trip is unconditional (clamped index, value
    // selected afterwards), so the m / rating / bias loads and the 17 covariance chunks leave
    // back to back and the trip pays ONE memory latency per rating.  (The first version loaded
    // the ids per trip and predicated the small loads with branches: index -> second index ->
    // m/rating/bias -> chunks of rating 0 -> chunks of rating 1 were five dependent latencies
    // per pair.)
    const bool has_bias = p.bias_other != nullptr;
    const float *bias_ptr = has_bias ? p.bias_other : p.factor_other;   // always readable
    const int bias_mul = has_bias ? 1 : 0;
    int idx_b = 0;
    float val_b = 0.f;
    for (int j = 0; j < t.len; j += 2) {
        if ((j & 63) == 0) {
            const int jj = min(j + lane, t.len - 1);
            idx_b = col[jj];
            val_b = val[jj];
        }
        const bool two = j + 1 < t.len;
        const int l0 = j & 63;   // even: l0 + 1 is in the same batch
        const int o0 = __builtin_amdgcn_readlane(idx_b, l0);
        const int o1 = two ? __builtin_amdgcn_readlane(idx_b, l0 + 1) : o0;
        const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val_b), l0));
        const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val_b), two ? l0 + 1 : l0));
        const int oh = h ? o1 : o0;
        const bool live = (h == 0) || two;
        const float *mrow = p.factor_other + (int64_t)oh * kpad;
        float m[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) m[b] = mrow[min(32 * b + c, kpad - 1)];
        const float bo = bias_ptr[(int64_t)oh * bias_mul];
#pragma unroll
        for (int b = 0; b < 4; ++b) m[b] = (live && 32 * b + c < K) ? m[b] : 0.f;
        const float res = live ? (h ? x1 : x0) - b_self - (has_bias ? bo : 0.f) : 0.f;
        const float4 *v0 = reinterpret_cast<const float4 *>(p.cov_other + (int64_t)o0 * stride);
        // one rating's chunks in flight at a time (68 VGPRs): with the MFMA blocks the kernel then
        // fits 256 registers, i.e. two blocks' worth of waves per SIMD, so one block can solve
        // while the other streams
        // every column is loaded, clamped to the wave's last chunk: no uniform branches around the
        // loads, and the clamped lanes (same cache line again) are never stored.  The lane base is
        // made opaque each trip so that the 17 clamped offsets are recomputed (two VALU ops per
        // load) instead of being kept live -- and spilled -- across the loop
        int qb = q_begin + lane;
        asm volatile("" : "+v"(qb));
        // NT <= 9 (K <= 92) leaves registers for BOTH ratings of the pair in flight (2 x 36 load registers):
        // more bytes outstanding per streaming wave, which matters while the CU's other blocks are solving
        constexpr bool BOTH = NT <= 9;
        float4 a[NT], a2[BOTH ? NT : 1];
#pragma unroll
        for (int s = 0; s < NT; ++s)   // uniform base + 32-bit unsigned byte offset: the SGPR-base load form
            a[s] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(v0) +
                                                     (unsigned)min(qb + 64 * s, q_end - 1) * 16u);
        if constexpr (BOTH) {
            // (o1 == o0 when the pair has one rating: the same lines again, never added)
            const float *v1 = p.cov_other + (int64_t)o1 * stride;
#pragma unroll
            for (int s = 0; s < NT; ++s)
                a2[s] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(v1) +
                                                          (unsigned)min(qb + 64 * s, q_end - 1) * 16u);
        }
        wA = fmaf(wave ? m[2] : m[0], res, wA);
        wB = fmaf(wave ? m[3] : m[1], res, wB);
        // operand pairs of this wave's five blocks (wave is uniform: scalar selects)
        const float A0 = wave ? m[2] : m[0], B0 = wave ? m[2] : m[0];
        const float A1 = wave ? m[3] : m[1], B1 = m[0];
        const float A2 = wave ? m[3] : m[1], B2 = m[1];
        const float A3 = wave ? m[3] : m[2], B3 = wave ? m[2] : m[0];
        const float A4 = wave ? m[3] : m[2], B4 = wave ? m[3] : m[1];
        d[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, d[0], 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, d[1], 0, 0, 0);
        d[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B2, d[2], 0, 0, 0);
        d[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A3, B3, d[3], 0, 0, 0);
        d[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(A4, B4, d[4], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            acc[s].x += a[s].x;
            acc[s].y += a[s].y;
            acc[s].z += a[s].z;
            acc[s].w += a[s].w;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BOTH) {
            if (two) {
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    acc[s].x += a2[s].x;
                    acc[s].y += a2[s].y;
                    acc[s].z += a2[s].z;
                    acc[s].w += a2[s].w;
                }
            }
        } else if (two) {
            // (the id is read again here so that the base stays a scalar inside this block)
            const float *v1 = p.cov_other + (int64_t)__builtin_amdgcn_readlane(idx_b, l0 + 1) * stride;
#pragma unroll
            for (int s = 0; s < NT; ++s)
                a[s] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(v1) +
                                                         (unsigned)min(qb + 64 * s, q_end - 1) * 16u);
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                acc[s].x += a[s].x;
                acc[s].y += a[s].y;
                acc[s].z += a[s].z;
                acc[s].w += a[s].w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- fold: image = zeros (+ padding diagonal), then the ten MFMA blocks ----
    const float pad_diag = pair_pad_diag(inv_sigma2, inv_eta2);
    for (int q = threadIdx.x * PMF_VEC; q < PAIR_IMG; q += 128 * PMF_VEC) {
        Vec4<float> v = zero4<float>();
        if (FUSE && q + PMF_VEC > p.kp) {
#pragma unroll
            for (int e = 0; e < PMF_VEC; ++e)
                if (q + e >= p.kp) {
                    int r, cc;
                    tri_rc(q + e, r, cc);
                    if (r == cc) v.v[e] = pad_diag;
                }
        }
        store4(img + q, v);
    }
    __syncthreads();
    // block (bi, bj) of d[b]: wave 0: (0,0) (1,0) (1,1) (2,0) (2,1); wave 1: (2,2) (3,0) (3,1) (3,2) (3,3)
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const int bi = wave ? (b == 0 ? 2 : 3) : (b == 0 ? 0 : (b <= 2 ? 1 : 2));
        const int bj = wave ? (b == 0 ? 2 : b - 1) : (b == 0 ? 0 : (b == 2 ? 1 : (b == 4 ? 1 : 0)));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int C = 32 * bj + c;
            if (R < K && C <= R) img[R * (R + 1) / 2 + C] = d[b][r];
        }
    }
    wA += __shfl_xor(wA, 32, 64);
    wB += __shfl_xor(wB, 32, 64);
    if (h == 0) {
        wbuf[64 * wave + c] = wA;
        wbuf[64 * wave + 32 + c] = wB;
    }
    __syncthreads();
    if constexpr (FUSE && MT > 0) if (t.slot < 0) {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int q = q_begin + lane + 64 * s;
            if (q < q_end) {
                float4 mm = reinterpret_cast<float4 *>(img)[q];
                mm.x += acc[s].x;
                mm.y += acc[s].y;
                mm.z += acc[s].z;
                mm.w += acc[s].w;
                reinterpret_cast<float4 *>(img)[q] = mm;
            }
        }
        __syncthreads();
        pair_solve_mfma<MT>(img, xbuf, gbuf, wbuf, K, kpad, stride, inv_sigma2, inv_eta2,
                            cov_self + (int64_t)t.row * stride, factor_self + (int64_t)t.row * kpad, wave, lane);
        return;
    }
    float *out_s, *out_w;
    if (t.slot >= 0) {
        out_s = p.partial + (int64_t)t.slot * (stride + kpad);
        out_w = out_s + stride;
    } else {
        out_s = p.dst_s + (int64_t)t.row * p.dst_s_stride;
        out_w = p.dst_w + (int64_t)t.row * p.dst_w_stride;
    }
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const int q = q_begin + lane + 64 * s;
        if (q < q_end) {
            const float4 mm = reinterpret_cast<const float4 *>(img)[q];
            float4 o = acc[s];
            o.x += mm.x;
            o.y += mm.y;
            o.z += mm.z;
            o.w += mm.w;
            reinterpret_cast<float4 *>(out_s)[q] = o;
        }
    }
    const int jj = threadIdx.x;
    if (jj < kpad) out_w[jj] = jj < K ? wbuf[jj] : 0.f;
}

// ---------------------------------------------------------------------------
// accumulate (+ fused solve), fp32, K <= 64: covariance rows on the VALU, m m^T on
// the MFMA pipe
// ---------------------------------------------------------------------------
// KB = 32 (K <= 32: one 32x32 block) or 64 (the three lower 32x32 blocks).  The
// packed row is cov_stride/4 16-byte chunks: chunk q = lane + 64 s, s < NT.  Two
// ratings feed one v_mfma_f32_32x32x2_f32 (k = 2): lanes 0-31 carry rating j,
// lanes 32-63 rating j+1 (operand lanes >= K hold zeros, so K need not be a
// multiple of 32; storage stays the exact K(K+1)/2 packing).  PU pairs are in
// flight per loop trip so that short rows (small K) still keep ~18 16-byte loads
// per lane outstanding.  The MFMA blocks are folded into the packed image through
// LDS once per task; a task that is a whole row is solved on the spot (FUSE).
// KS = register rows of the fused solve (8 / 16 for K <= 8 / 16: a quarter / half of the 32-row sweep).
template <int KB, int NT, bool FUSE, int KS = KB>
__global__ __launch_bounds__(256, KB == 64 ? 2 : 3) void gauss_accum_mfma_kernel(GaussParams<float> p, float inv_sigma2,
                                                                                float inv_eta2, float *cov_self,
                                                                                float *factor_self) {
    constexpr int NB = KB == 64 ? 3 : 1;
    // (the scheduler interleaves the trip's waits and adds with its loads -- about ten of the 18 are in
    //  flight at a time; forcing all 18 out before the first use: item sweep 68.4 -> 66.8 ms, user sweep
    //  56.6 -> 60.4 ms, worse overall -- and selecting that form for the item sweep only changed the
    //  epoch by less than the box-to-box noise (128.2 / 128.9 against 128.1 / 128.3 ms);
    //  issuing pair t+1 before consuming pair t needs 2 x 75 load registers and spills at NT >= 8)
    // (measured at K = 64 / NT = 9: two pairs in flight 1-2 % slower; one rating at a time at 3 waves
    //  per SIMD 2.5 % slower than one pair at 2 waves per SIMD; non-temporal loads of the item side's
    //  streamed-once covariance rows: no difference)
    constexpr int PU = NT >= 5 ? 1 : (NT >= 3 ? 2 : (NT == 2 ? 4 : 8));
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t task_id = (int64_t)blockIdx.x * 4 + wave;
    if (task_id >= p.n_tasks) return;
    const PmfTask t = load_task_uniform(p.tasks, task_id);
    const int h = lane >> 5, c = lane & 31;
    const int K = p.K, kpad = p.kpad, stride = p.cov_stride, chunks = p.cov_stride / PMF_VEC;
    const int32_t *col = p.other + t.start;
    const float *val = p.val + t.start;
    const float b_self = p.bias_self ? p.bias_self[t.row] : 0.f;
    const bool lo_ok = c < K, hi_ok = (KB == 64) && (32 + c < K);

    f32x16 d[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[b][r] = 0.f;
    float4 acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float wlo = 0.f, whi = 0.f;

    // one loop trip = PU pairs of ratings; FULL trips carry no validity tests
    // The task's row ids and ratings come 64 at a time (one coalesced load each per 64 ratings) and
    // are handed out with v_readlane: ids in SGPRs without a dependent scalar load per pair (the
    // PU = 8 trip of K <= 16 used to wait for eight of them in turn).
    int idx_b = 0;
    float val_b = 0.f;
    auto trip = [&](int j, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float4 a[PU][NT], b[PU][NT];
        float mlo[PU], mhi[PU], res[PU];
        if ((j & 63) == 0) {   // a trip never straddles a batch: 2 PU divides 64
            const int jj = min(j + lane, t.len - 1);
            idx_b = col[jj];
            val_b = val[jj];
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int j0 = j + 2 * u, l0 = j0 & 63;
            const bool has0 = FULL || j0 < t.len, has1 = FULL || j0 + 1 < t.len;
            const int o0 = __builtin_amdgcn_readlane(idx_b, l0);                 // lanes past the task hold
            const int o1 = __builtin_amdgcn_readlane(idx_b, l0 + 1);             // its last (valid) id
            const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val_b), l0));
            const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val_b), l0 + 1));
            const int oh = h ? o1 : o0;
            const bool live = h ? has1 : has0;
            const float *mrow = p.factor_other + (int64_t)oh * kpad;
            mlo[u] = (live && lo_ok) ? mrow[c] : 0.f;
            mhi[u] = (live && hi_ok) ? mrow[32 + c] : 0.f;
            const float xh = h ? x1 : x0;
            res[u] = live ? xh - b_self - (p.bias_other ? p.bias_other[oh] : 0.f) : 0.f;
            const float4 *v0 = reinterpret_cast<const float4 *>(p.cov_other + (int64_t)o0 * stride);
            const float4 *v1 = reinterpret_cast<const float4 *>(p.cov_other + (int64_t)o1 * stride);
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                // only the last chunk column can run past the row: those lanes re-read the row's
                // last chunk (same cache line, no extra traffic) and their sums are never stored,
                // which keeps the whole trip free of divergent branches
                const int q = (s + 1 < NT) ? lane + 64 * s : min(lane + 64 * s, chunks - 1);
                a[u][s] = make_float4(0.f, 0.f, 0.f, 0.f);
                b[u][s] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has0) a[u][s] = v0[q];
                if (has1) b[u][s] = v1[q];
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            wlo = fmaf(mlo[u], res[u], wlo);
            whi = fmaf(mhi[u], res[u], whi);
            d[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(mlo[u], mlo[u], d[0], 0, 0, 0);
            if constexpr (KB == 64) {
                d[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(mhi[u], mlo[u], d[1], 0, 0, 0);
                d[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(mhi[u], mhi[u], d[2], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                acc[s].x += a[u][s].x + b[u][s].x;
                acc[s].y += a[u][s].y + b[u][s].y;
                acc[s].z += a[u][s].z + b[u][s].z;
                acc[s].w += a[u][s].w + b[u][s].w;
            }
        }
    };
    int j = 0;
    for (; j + 2 * PU <= t.len; j += 2 * PU) trip(j, std::true_type{});
    if (j < t.len) trip(j, std::false_type{});

    // fold the outer-product blocks into the packed image via LDS
    float *img = reinterpret_cast<float *>(smem_raw) + (int64_t)wave * stride;
    // (the fold below writes every packed entry but not the row padding: clear first)
    for (int q = lane; q < chunks; q += 64) reinterpret_cast<float4 *>(img)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D layout of the 32x32 MFMA
        if (row < K && c <= row) img[row * (row + 1) / 2 + c] = d[0][r];
        if constexpr (KB == 64) {
            const int R1 = 32 + row;
            if (R1 < K) {
                if (c < K) img[R1 * (R1 + 1) / 2 + c] = d[1][r];
                if (c <= row) img[R1 * (R1 + 1) / 2 + 32 + c] = d[2][r];
            }
        }
    }
    wave_lds_fence();
    wlo += __shfl_xor(wlo, 32, 64);
    whi += __shfl_xor(whi, 32, 64);
    if (FUSE && t.slot < 0) {
        // the row is complete: finish it here (S = image + covariance sums stays in
        // LDS) while the other wavefronts of the CU keep streaming
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int q = lane + 64 * s;
            if (q < chunks) {
                float4 m = reinterpret_cast<float4 *>(img)[q];
                m.x += acc[s].x;
                m.y += acc[s].y;
                m.z += acc[s].z;
                m.w += acc[s].w;
                reinterpret_cast<float4 *>(img)[q] = m;
            }
        }
        wave_lds_fence();
        solve_from_image<float, KS>(img, h ? whi : wlo, K, kpad, inv_sigma2, inv_eta2,
                                    cov_self + (int64_t)t.row * stride, factor_self + (int64_t)t.row * kpad, lane);
        return;
    }
    float *out_s, *out_w;
    if (t.slot >= 0) {
        out_s = p.partial + (int64_t)t.slot * (stride + kpad);
        out_w = out_s + stride;
    } else {
        out_s = p.dst_s + (int64_t)t.row * p.dst_s_stride;
        out_w = p.dst_w + (int64_t)t.row * p.dst_w_stride;
    }
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const int q = lane + 64 * s;
        if (q < chunks) {
            const float4 m = reinterpret_cast<const float4 *>(img)[q];
            float4 o = acc[s];
            o.x += m.x;
            o.y += m.y;
            o.z += m.z;
            o.w += m.w;
            reinterpret_cast<float4 *>(out_s)[q] = o;
        }
    }
    if (h == 0) {
        if (c < kpad) out_w[c] = wlo;
        if (KB == 64 && 32 + c < kpad) out_w[32 + c] = whi;
    }
}

// ---------------------------------------------------------------------------
// combine the partial slots of split rows (slot order => deterministic)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gauss_combine_kernel(GaussParams<T> p) {
    constexpr int UN = 8;
    const PmfSplitRow sr = p.split[blockIdx.x];
    const int width = p.cov_stride + p.kpad;  // multiple of PMF_VEC
    T *out_s = p.dst_s + (int64_t)sr.row * p.dst_s_stride;
    T *out_w = p.dst_w + (int64_t)sr.row * p.dst_w_stride;
    const T *base = p.partial + (int64_t)sr.first_slot * width;
    for (int e = threadIdx.x * PMF_VEC; e < width; e += 256 * PMF_VEC) {
        Vec4<T> acc = zero4<T>();
        int k = 0;
        for (; k + UN <= sr.n_slots; k += UN) {
            Vec4<T> v[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) v[q] = load4(base + (int64_t)(k + q) * width + e);
#pragma unroll
            for (int q = 0; q < UN; ++q)
#pragma unroll
                for (int c = 0; c < PMF_VEC; ++c) acc.v[c] += v[q].v[c];
        }
        for (; k < sr.n_slots; ++k) {
            Vec4<T> v = load4(base + (int64_t)k * width + e);
#pragma unroll
            for (int c = 0; c < PMF_VEC; ++c) acc.v[c] += v.v[c];
        }
        if (e < p.cov_stride) store4(out_s + e, acc);
        else store4(out_w + (e - p.cov_stride), acc);
    }
}

// ---------------------------------------------------------------------------
// solve: V = inv(I/eta2 + S/sigma2), m = V w / sigma2
// ---------------------------------------------------------------------------

// Generic solve for K > 64 (fp64) / K > 128: one block per row, Gauss-Jordan sweep on the full matrix.  The matrix
// lives in LDS while K (K + 1) + 3 K elements fit the CU's 160 KB (fp32: K <= 200, fp64: K <= 141); beyond that
// (`scratch` != null) every block keeps it in its own slice of a global scratch buffer -- L2-resident, slow, and
// only there so that the Gaussian model has no K limit below the context's 256 (the reference has none at all).
template <typename T>
__global__ __launch_bounds__(256) void gauss_solve_lds_kernel(SolveParams<T> p, T *scratch) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int K = p.K, ld = K + 1;
    T *lds = reinterpret_cast<T *>(smem_raw);
    T *A = scratch ? scratch + (int64_t)blockIdx.x * K * ld : lds;   // [K][ld]
    T *g = scratch ? lds : lds + K * ld;                             // [K] scaling
    T *prow = g + K;                                                 // [K] scaled pivot row
    T *pcol = prow + K;                                              // [K] pivot column
    const int tid = threadIdx.x;
    for (int64_t idx = blockIdx.x; idx < p.n; idx += gridDim.x) {
    const int row = p.rows ? p.rows[idx] : (int)(p.row0 + idx);
    const T *S = p.src_s + (int64_t)row * p.src_s_stride;
    if (!p.rows && S[0] == (T)0) continue;   // uniform for the block
    __syncthreads();
    for (int e = tid; e < K * K; e += 256) {
        const int i = e / K, j = e % K;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        T s = S[hi * (hi + 1) / 2 + lo] * p.inv_sigma2;
        if (i == j) s += p.inv_eta2;
        A[i * ld + j] = s;
    }
    __syncthreads();
    for (int i = tid; i < K; i += 256) g[i] = (T)1 / sqrt(A[i * ld + i]);
    __syncthreads();
    for (int e = tid; e < K * K; e += 256) {
        const int i = e / K, j = e % K;
        A[i * ld + j] *= g[i] * g[j];
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const T pinv = (T)1 / A[k * ld + k];
        __syncthreads();
        for (int i = tid; i < K; i += 256) {
            prow[i] = A[k * ld + i] * pinv;
            pcol[i] = A[i * ld + k];
        }
        __syncthreads();
        for (int e = tid; e < K * K; e += 256) {
            const int i = e / K, j = e % K;
            T x;
            if (i == k) x = (j == k) ? -pinv : prow[j];
            else if (j == k) x = pcol[i] * pinv;
            else x = fma(-pcol[i], prow[j], A[i * ld + j]);
            A[i * ld + j] = x;
        }
        __syncthreads();
    }
    // V = -g g A ; m = inv_sigma2 V w
    T *vout = p.cov + (int64_t)row * p.cov_stride;
    for (int e = tid; e < K * K; e += 256) {
        const int i = e / K, j = e % K;
        const T v = -A[i * ld + j] * g[i] * g[j];
        A[i * ld + j] = v;
        if (j <= i) vout[i * (i + 1) / 2 + j] = v;
    }
    for (int i = tid; i < K; i += 256) prow[i] = p.src_w[(int64_t)row * p.src_w_stride + i];
    __syncthreads();
    for (int i = tid; i < p.kpad; i += 256) {
        T m = (T)0;
        if (i < K)
            for (int jj = 0; jj < K; ++jj) m = fma(A[i * ld + jj], prow[jj], m);
        p.factor[(int64_t)row * p.kpad + i] = m * p.inv_sigma2;
    }
    }
}

// ---------------------------------------------------------------------------
// bias half-sweep (lane group per task, same decomposition as the gamma sweep)
// ---------------------------------------------------------------------------
template <typename T>
struct BiasParams {
    const PmfTask *tasks;
    int64_t n_tasks;
    const PmfSplitRow *split;
    const int32_t *other;
    const T *val;
    const T *factor_self;
    const T *factor_other;
    T *bias_self;
    const T *bias_other;
    T *partial;  // [n_slots]
    T *stats;    // [rows][2] (residual sum, count), STATS mode
    T inv_sigma2, inv_eta_bias2;
    int kpad;
    int64_t row0, rows;  // finalize-from-stats covers rows [row0, rows)
};

template <typename T>
__device__ __forceinline__ T bias_from_sum(const BiasParams<T> &p, T sum, T count) {
    // gaussian_mf_cavi_bias.py:222-230: var = 1/(1/eta_b2 + n/sigma2); b = var/sigma2 * sum
    const T var = (T)1 / (p.inv_eta_bias2 + count * p.inv_sigma2);
    return (var * p.inv_sigma2) * sum;
}

template <typename T, int LPR, bool STATS>
__global__ __launch_bounds__(256) void gauss_bias_kernel(BiasParams<T> p) {
    constexpr int G = 256 / LPR;
    constexpr int UN = LPR < 4 ? LPR : 4;
    const int c = threadIdx.x % LPR;
    const int64_t task_id = (int64_t)blockIdx.x * G + threadIdx.x / LPR;
    if (task_id >= p.n_tasks) return;
    const PmfTask t = p.tasks[task_id];
    const int koff = c * PMF_VEC;
    const bool active = koff < p.kpad;
    const Vec4<T> self = active ? load4(p.factor_self + (int64_t)t.row * p.kpad + koff) : zero4<T>();
    const int32_t *col = p.other + t.start;
    const T *val = p.val + t.start;
    T sum = (T)0;
    for (int base = 0; base < t.len; base += LPR) {
        const int n = min(LPR, t.len - base);
        int my_o = 0;
        T my_r = (T)0;
        if (c < n) {
            my_o = col[base + c];
            my_r = val[base + c] - p.bias_other[my_o];
        }
        for (int tt = 0; tt < n; tt += UN) {
            int o[UN];
            T rv[UN];
            Vec4<T> b[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                o[q] = __shfl(my_o, tt + q, LPR);
                rv[q] = __shfl(my_r, tt + q, LPR);
            }
#pragma unroll
            for (int q = 0; q < UN; ++q)
                b[q] = active ? load4(p.factor_other + (int64_t)o[q] * p.kpad + koff) : zero4<T>();
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                if (tt + q < n) {
                    T d = b[q].v[0] * self.v[0];
                    d = fma(b[q].v[1], self.v[1], d);
                    d = fma(b[q].v[2], self.v[2], d);
                    d = fma(b[q].v[3], self.v[3], d);
                    d = group_sum<LPR>(d);
                    sum += rv[q] - d;
                }
            }
        }
    }
    if (c != 0) return;
    if (t.slot >= 0) {
        p.partial[t.slot] = sum;
    } else if (STATS) {
        p.stats[(int64_t)t.row * 2] = sum;
        p.stats[(int64_t)t.row * 2 + 1] = (T)t.len;
    } else {
        p.bias_self[t.row] = bias_from_sum(p, sum, (T)t.len);
    }
}

template <typename T, bool STATS>
__global__ void gauss_bias_split_kernel(BiasParams<T> p, int64_t n_split, const int64_t *ptr) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_split) return;
    const PmfSplitRow sr = p.split[s];
    T sum = (T)0;
    for (int k = 0; k < sr.n_slots; ++k) sum += p.partial[sr.first_slot + k];
    const T cnt = (T)(ptr[sr.row + 1] - ptr[sr.row]);
    if (STATS) {
        p.stats[(int64_t)sr.row * 2] = sum;
        p.stats[(int64_t)sr.row * 2 + 1] = cnt;
    } else {
        p.bias_self[sr.row] = bias_from_sum(p, sum, cnt);
    }
}

template <typename T>
__global__ void gauss_bias_finalize_all_kernel(BiasParams<T> p) {
    const int64_t r = p.row0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.rows) return;
    const T cnt = p.stats[r * 2 + 1];
    if (cnt > (T)0) p.bias_self[r] = bias_from_sum(p, p.stats[r * 2], cnt);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
#define GAUSS_PROLOGUE(fn)                                                                              \
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, fn ": null context");                                       \
    PMF_REQUIRE(side == PMF_SIDE_USER || side == PMF_SIDE_ITEM, PMF_EINVAL, fn ": bad side %d", side);  \
    PMF_REQUIRE(ctx->K <= 256, PMF_ERANGE, fn ": the Gaussian path supports n_factors <= 256 (got %d)", ctx->K); \
    PMF_HIP_CHECK(hipSetDevice(ctx->device));

static bool use_bias(const pmf_ctx *ctx) {
    return ctx->arr[0][PMF_ARR_BIAS] != nullptr && ctx->arr[1][PMF_ARR_BIAS] != nullptr;
}

// mode 0: fused (sums in place, then solve)   mode 1: accumulate into stats
template <int KB, int NT, int KS = KB>
static void launch_accum_mfma_nt(pmf_ctx *ctx, const GaussParams<float> &p, dim3 grid, bool fuse, float is2, float ie2,
                                 float *cov, float *fac) {
    const size_t smem = (size_t)4 * ctx->cov_stride * sizeof(float);
    if (fuse)
        hipLaunchKernelGGL((gauss_accum_mfma_kernel<KB, NT, true, KS>), grid, dim3(256), smem, ctx->stream, p, is2, ie2, cov, fac);
    else
        hipLaunchKernelGGL((gauss_accum_mfma_kernel<KB, NT, false>), grid, dim3(256), smem, ctx->stream, p, 0.f, 0.f,
                           (float *)nullptr, (float *)nullptr);
}

// K <= 64, fp32: pick the instantiation by MFMA block count and packed-row chunk count
static void launch_accum_mfma(pmf_ctx *ctx, const GaussParams<float> &p, dim3 grid, bool fuse, float is2, float ie2,
                              float *cov, float *fac) {
    const int nt = (ctx->cov_stride / PMF_VEC + 63) / 64;  // 1..9
    if (ctx->K <= 8) {
        launch_accum_mfma_nt<32, 1, 8>(ctx, p, grid, fuse, is2, ie2, cov, fac);
    } else if (ctx->K <= 16) {
        launch_accum_mfma_nt<32, 1, 16>(ctx, p, grid, fuse, is2, ie2, cov, fac);
    } else if (ctx->K <= 32) {
        switch (nt) {
            case 1: launch_accum_mfma_nt<32, 1>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 2: launch_accum_mfma_nt<32, 2>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            default: launch_accum_mfma_nt<32, 3>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
        }
    } else if (ctx->K <= 48) {   // Kp <= 1176: 3..5 chunk columns; a 48-row sweep instead of the 64-row one
        switch (nt) {
            case 3: launch_accum_mfma_nt<64, 3, 48>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 4: launch_accum_mfma_nt<64, 4, 48>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            default: launch_accum_mfma_nt<64, 5, 48>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
        }
    } else if (ctx->K <= 56) {   // Kp <= 1596: 5..7 chunk columns; a 56-row sweep (K = 50 is in the reference's grid)
        switch (nt) {
            case 5: launch_accum_mfma_nt<64, 5, 56>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 6: launch_accum_mfma_nt<64, 6, 56>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            default: launch_accum_mfma_nt<64, 7, 56>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
        }
    } else {
        switch (nt) {
            case 3: launch_accum_mfma_nt<64, 3>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 4: launch_accum_mfma_nt<64, 4>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 5: launch_accum_mfma_nt<64, 5>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 6: launch_accum_mfma_nt<64, 6>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 7: launch_accum_mfma_nt<64, 7>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            case 8: launch_accum_mfma_nt<64, 8>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
            default: launch_accum_mfma_nt<64, 9>(ctx, p, grid, fuse, is2, ie2, cov, fac); break;
        }
    }
}

// MT = 16-row tiles per dimension of the fused MFMA block sweep (ceil(K / 16): 5..8)
template <int NT>
static void launch_accum_mfma128(pmf_ctx *ctx, const GaussParams<float> &p, dim3 grid, size_t smem, bool fuse, float is2,
                                 float ie2, float *cov, float *fac) {
    if (!fuse) {
        hipLaunchKernelGGL((gauss_accum_mfma128_kernel<NT, false>), grid, dim3(128), smem, ctx->stream, p, 0.f, 0.f,
                           (float *)nullptr, (float *)nullptr);
        return;
    }
    const int mt = (ctx->K + 15) / 16;
    if constexpr (NT == 9) {          // K <= 92
        if (mt <= 5) hipLaunchKernelGGL((gauss_accum_mfma128_kernel<NT, true, 5>), grid, dim3(128), smem, ctx->stream, p, is2, ie2, cov, fac);
        else hipLaunchKernelGGL((gauss_accum_mfma128_kernel<NT, true, 6>), grid, dim3(128), smem, ctx->stream, p, is2, ie2, cov, fac);
    } else if constexpr (NT == 13) {  // K <= 112
        if (mt <= 6) hipLaunchKernelGGL((gauss_accum_mfma128_kernel<NT, true, 6>), grid, dim3(128), smem, ctx->stream, p, is2, ie2, cov, fac);
        else hipLaunchKernelGGL((gauss_accum_mfma128_kernel<NT, true, 7>), grid, dim3(128), smem, ctx->stream, p, is2, ie2, cov, fac);
    } else {
        hipLaunchKernelGGL((gauss_accum_mfma128_kernel<NT, true, 8>), grid, dim3(128), smem, ctx->stream, p, is2, ie2, cov, fac);
    }
}

// *fused is set when the kernel also solved every single-task row (K = 64 fp32,
// not in stats mode); the caller then only solves the split rows.
template <typename T>
static int run_factor_accumulate(pmf_ctx *ctx, int side, void *stats, double sigma2, double eta2, bool *fused) {
    *fused = false;
    const int other = 1 - side;
    const PmfSideIndex &ix = ctx->index[side];
    const PmfTaskView tl = pmf_task_view(ctx, side, ix.gauss_tasks, stats != nullptr);
    int rc;
    PMF_REQUIRE(ix.d_ptr, PMF_EINVAL, "pmf_gauss_factor_sweep: ratings have not been set");
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_FACTOR, "pmf_gauss_factor_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_COV, "pmf_gauss_factor_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, other, PMF_ARR_FACTOR, "pmf_gauss_factor_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, other, PMF_ARR_COV, "pmf_gauss_factor_sweep"))) return rc;
    const int width = ctx->cov_stride + ctx->kpad;
    if (tl.n_slots > 0)
        if ((rc = pmf_ensure_partial(ctx, (size_t)tl.n_slots * width * sizeof(T)))) return rc;
    GaussParams<T> p;
    p.tasks = tl.d_tasks;
    p.n_tasks = tl.n_tasks;
    p.split = tl.d_split;
    p.other = ix.d_other;
    p.val = (const T *)ix.d_val;
    p.factor_other = (const T *)ctx->arr[other][PMF_ARR_FACTOR];
    p.cov_other = (const T *)ctx->arr[other][PMF_ARR_COV];
    const bool bias = use_bias(ctx);
    p.bias_self = bias ? (const T *)ctx->arr[side][PMF_ARR_BIAS] : nullptr;
    p.bias_other = bias ? (const T *)ctx->arr[other][PMF_ARR_BIAS] : nullptr;
    p.partial = (T *)ctx->d_partial;
    if (stats) {
        if (tl.row1 > tl.row0)  // rows without ratings on this rank contribute zeros
            PMF_HIP_CHECK(hipMemsetAsync((T *)stats + tl.row0 * width, 0, (size_t)(tl.row1 - tl.row0) * width * sizeof(T),
                                         ctx->stream));
        p.dst_s = (T *)stats;
        p.dst_s_stride = width;
        p.dst_w = (T *)stats + ctx->cov_stride;
        p.dst_w_stride = width;
    } else {
        p.dst_s = (T *)ctx->arr[side][PMF_ARR_COV];
        p.dst_s_stride = ctx->cov_stride;
        p.dst_w = (T *)ctx->arr[side][PMF_ARR_FACTOR];
        p.dst_w_stride = ctx->kpad;
    }
    p.K = ctx->K;
    p.kpad = ctx->kpad;
    p.kp = ctx->kp;
    p.cov_stride = ctx->cov_stride;
    if (tl.n_tasks > 0) {
        PmfProfScope prof(ctx, PMF_KERNEL_GAUSS_ACCUM);
        dim3 grid((unsigned)((tl.n_tasks + 3) / 4));
        bool fast = false;
        if constexpr (std::is_same<T, float>::value) {
            if (!ctx->gauss_generic) {
                fast = true;
                const bool fuse = !stats && !ctx->gauss_unfused;
                *fused = fuse;
                const float is2 = (float)(1.0 / sigma2), ie2 = (float)(1.0 / eta2);
                float *cov = (float *)ctx->arr[side][PMF_ARR_COV], *fac = (float *)ctx->arr[side][PMF_ARR_FACTOR];
                if (ctx->K > 128) {
                    fast = false;       // 128 < K <= 256: the generic kernels (no reference configuration is this large)
                    *fused = false;
                } else if (ctx->K <= 64) {
                    launch_accum_mfma(ctx, p, grid, fuse, is2, ie2, cov, fac);
                } else {  // 64 < K <= 128: one 128-thread block (two wavefronts) per task
                    const size_t smem = (size_t)PAIR_LDS_FLOATS * sizeof(float);
                    dim3 g2((unsigned)tl.n_tasks);
                    const int chunks = ctx->cov_stride / PMF_VEC, nt = ((chunks + 1) / 2 + 63) / 64;
                    if (nt <= 9) launch_accum_mfma128<9>(ctx, p, g2, smem, fuse, is2, ie2, cov, fac);
                    else if (nt <= 13) launch_accum_mfma128<13>(ctx, p, g2, smem, fuse, is2, ie2, cov, fac);
                    else launch_accum_mfma128<17>(ctx, p, g2, smem, fuse, is2, ie2, cov, fac);
                }
            }
        }
        if (!fast) {
            // fp64 contexts (and fp32 with PMF_GAUSS_GENERIC): K <= 64 rows that are one task are solved in the kernel
            const bool fuse = !stats && !ctx->gauss_unfused && ctx->K <= 64 && std::is_same<T, double>::value;
            T *cov = (T *)ctx->arr[side][PMF_ARR_COV], *fac = (T *)ctx->arr[side][PMF_ARR_FACTOR];
            const T is2 = (T)(1.0 / sigma2), ie2 = (T)(1.0 / eta2);
            const size_t plain = (size_t)4 * 2 * ctx->kpad * sizeof(T), fused_lds = plain + (size_t)4 * ctx->cov_stride * sizeof(T);
            *fused = fuse;
            bool launched = false;
            if constexpr (std::is_same<T, double>::value) {
                if (fuse) {
                    launched = true;
                    if (ctx->K <= 8)
                        hipLaunchKernelGGL((gauss_accum_generic_kernel<T, 8>), grid, dim3(256), fused_lds, ctx->stream, p, is2, ie2, cov, fac);
                    else if (ctx->K <= 16)
                        hipLaunchKernelGGL((gauss_accum_generic_kernel<T, 16>), grid, dim3(256), fused_lds, ctx->stream, p, is2, ie2, cov, fac);
                    else if (ctx->K <= 32)
                        hipLaunchKernelGGL((gauss_accum_generic_kernel<T, 32>), grid, dim3(256), fused_lds, ctx->stream, p, is2, ie2, cov, fac);
                    else
                        hipLaunchKernelGGL((gauss_accum_generic_kernel<T, 64>), grid, dim3(256), fused_lds, ctx->stream, p, is2, ie2, cov, fac);
                }
            }
            if (!launched)
                hipLaunchKernelGGL((gauss_accum_generic_kernel<T, 0>), grid, dim3(256), plain, ctx->stream, p, (T)0, (T)0,
                                   (T *)nullptr, (T *)nullptr);
        }
    }
    if (tl.n_split > 0) {
        PmfProfScope prof(ctx, PMF_KERNEL_GAUSS_COMBINE);
        hipLaunchKernelGGL((gauss_combine_kernel<T>), dim3((unsigned)tl.n_split), dim3(256), 0, ctx->stream, p);
    }
    PMF_HIP_CHECK(hipGetLastError());
    return PMF_OK;
}

template <typename T, int KR>
static void launch_solve_reg(pmf_ctx *ctx, const SolveParams<T> &sp) {
    dim3 grid((unsigned)((sp.n + 3) / 4));
    hipLaunchKernelGGL((gauss_solve_reg_kernel<T, KR>), grid, dim3(256),
                       (size_t)4 * ctx->cov_stride * sizeof(T), ctx->stream, sp);
}

template <typename T>
static int run_factor_solve(pmf_ctx *ctx, int side, const void *stats, double sigma2, double eta2,
                            bool split_rows_only = false) {
    const PmfSideIndex &ix = ctx->index[side];
    PMF_REQUIRE(sigma2 > 0 && eta2 > 0, PMF_EINVAL, "pmf_gauss_factor_sweep: variances must be positive");
    int rc;
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_FACTOR, "pmf_gauss_factor_finalize"))) return rc;
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_COV, "pmf_gauss_factor_finalize"))) return rc;
    SolveParams<T> sp;
    const int width = ctx->cov_stride + ctx->kpad;
    const PmfTaskView tl = pmf_task_view(ctx, side, ix.gauss_tasks, stats != nullptr);
    if (stats) {
        sp.rows = nullptr;
        sp.row0 = tl.row0;
        sp.n = tl.row1 - tl.row0;
        sp.src_s = (const T *)stats;
        sp.src_s_stride = width;
        sp.src_w = (const T *)stats + ctx->cov_stride;
        sp.src_w_stride = width;
    } else {
        sp.rows = split_rows_only ? tl.d_split_rows : tl.d_nonempty;
        sp.row0 = 0;
        sp.n = split_rows_only ? tl.n_split : tl.n_nonempty;
        sp.src_s = (const T *)ctx->arr[side][PMF_ARR_COV];
        sp.src_s_stride = ctx->cov_stride;
        sp.src_w = (const T *)ctx->arr[side][PMF_ARR_FACTOR];
        sp.src_w_stride = ctx->kpad;
    }
    sp.cov = (T *)ctx->arr[side][PMF_ARR_COV];
    sp.factor = (T *)ctx->arr[side][PMF_ARR_FACTOR];
    sp.inv_sigma2 = (T)(1.0 / sigma2);
    sp.inv_eta2 = (T)(1.0 / eta2);
    sp.K = ctx->K;
    sp.kpad = ctx->kpad;
    sp.kp = ctx->kp;
    sp.cov_stride = ctx->cov_stride;
    if (sp.n == 0) return PMF_OK;
    PmfProfScope prof(ctx, PMF_KERNEL_GAUSS_SOLVE);
    if (ctx->K <= 8) launch_solve_reg<T, 8>(ctx, sp);
    else if (ctx->K <= 16) launch_solve_reg<T, 16>(ctx, sp);
    else if (ctx->K <= 32) launch_solve_reg<T, 32>(ctx, sp);
    else if (ctx->K <= 64) launch_solve_reg<T, 64>(ctx, sp);
    else if (std::is_same<T, float>::value && !ctx->gauss_lds_solve && ctx->K <= 128) {
        if constexpr (std::is_same<T, float>::value) {
            const size_t smem = (size_t)PAIR_LDS_FLOATS * sizeof(float);
            switch ((ctx->K + 15) / 16) {
                case 5: hipLaunchKernelGGL((gauss_solve_pair_kernel<5>), dim3((unsigned)sp.n), dim3(128), smem, ctx->stream, sp); break;
                case 6: hipLaunchKernelGGL((gauss_solve_pair_kernel<6>), dim3((unsigned)sp.n), dim3(128), smem, ctx->stream, sp); break;
                case 7: hipLaunchKernelGGL((gauss_solve_pair_kernel<7>), dim3((unsigned)sp.n), dim3(128), smem, ctx->stream, sp); break;
                default: hipLaunchKernelGGL((gauss_solve_pair_kernel<8>), dim3((unsigned)sp.n), dim3(128), smem, ctx->stream, sp); break;
            }
        }
    } else {
        const int K = ctx->K;
        const size_t mat = (size_t)K * (K + 1) * sizeof(T), vecs = (size_t)3 * K * sizeof(T);
        const bool in_lds = mat + vecs <= (size_t)160 * 1024;
        const unsigned blocks = (unsigned)std::min<int64_t>(sp.n, in_lds ? sp.n : 2048);
        T *scratch = nullptr;
        if (!in_lds) {   // one matrix per resident block in global scratch (<= 2048 x 264 KB)
            if ((rc = pmf_ensure_scratch(ctx, (size_t)blocks * mat))) return rc;
            scratch = (T *)ctx->d_scratch;
        }
        const size_t smem = in_lds ? mat + vecs : vecs;
        hipError_t e = hipFuncSetAttribute((const void *)gauss_solve_lds_kernel<T>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            pmf_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", smem, hipGetErrorString(e));
            return PMF_EHIP;
        }
        hipLaunchKernelGGL((gauss_solve_lds_kernel<T>), dim3(blocks), dim3(256), smem, ctx->stream, sp, scratch);
    }
    PMF_HIP_CHECK(hipGetLastError());
    return PMF_OK;
}

// several ranks: accumulate -> all-reduce -> finalize on the library's statistics buffer (pmf_comm.hip)
template <typename T>
static int run_factor_dist(pmf_ctx *ctx, int side, double sigma2, double eta2) {
    const size_t width = (size_t)ctx->cov_stride + ctx->kpad;
    void *stats = nullptr;
    int rc = pmf_comm_stats(ctx, 0, (size_t)ctx->rows[side] * width * sizeof(T), &stats);
    if (rc) return rc;
    // finalize = one K x K solve per row, and the finalised state ([Kp + Kpad] per row) is as wide as the statistics:
    // reduce-scatter -> solve 1/N of the rows -> all-gather moves the same bytes and divides the solves by N
    PmfExchange ex;
    ex.prefer_scatter = true;
    ex.arrays[ex.n_arrays++] = PMF_ARR_FACTOR;
    ex.arrays[ex.n_arrays++] = PMF_ARR_COV;
    return pmf_comm_half_sweep(
        ctx, side, width, stats, true,
        [&] {
            bool fused = false;
            return run_factor_accumulate<T>(ctx, side, stats, 1.0, 1.0, &fused);
        },
        [&] { return run_factor_solve<T>(ctx, side, stats, sigma2, eta2); }, ex);
}

extern "C" int pmf_gauss_factor_sweep(pmf_ctx *ctx, int side, double sigma2, double eta2) {
    GAUSS_PROLOGUE("pmf_gauss_factor_sweep");
    PMF_REQUIRE(sigma2 > 0 && eta2 > 0, PMF_EINVAL, "pmf_gauss_factor_sweep: variances must be positive");
    if (side == PMF_SIDE_ITEM && pmf_comm_active(ctx))
        return ctx->dtype == PMF_F64 ? run_factor_dist<double>(ctx, side, sigma2, eta2)
                                     : run_factor_dist<float>(ctx, side, sigma2, eta2);
    int rc;
    bool fused = false;
    if (ctx->dtype == PMF_F64) {
        if ((rc = run_factor_accumulate<double>(ctx, side, nullptr, sigma2, eta2, &fused))) return rc;
        return run_factor_solve<double>(ctx, side, nullptr, sigma2, eta2, fused);
    }
    if ((rc = run_factor_accumulate<float>(ctx, side, nullptr, sigma2, eta2, &fused))) return rc;
    return run_factor_solve<float>(ctx, side, nullptr, sigma2, eta2, fused);
}

extern "C" int pmf_gauss_factor_accumulate(pmf_ctx *ctx, int side, void *stats_dev) {
    GAUSS_PROLOGUE("pmf_gauss_factor_accumulate");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gauss_factor_accumulate: null stats buffer");
    bool fused = false;
    if (ctx->dtype == PMF_F64) return run_factor_accumulate<double>(ctx, side, stats_dev, 1.0, 1.0, &fused);
    return run_factor_accumulate<float>(ctx, side, stats_dev, 1.0, 1.0, &fused);
}

extern "C" int pmf_gauss_factor_finalize(pmf_ctx *ctx, int side, const void *stats_dev, double sigma2,
                                         double eta2) {
    GAUSS_PROLOGUE("pmf_gauss_factor_finalize");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gauss_factor_finalize: null stats buffer");
    if (ctx->dtype == PMF_F64) return run_factor_solve<double>(ctx, side, stats_dev, sigma2, eta2);
    return run_factor_solve<float>(ctx, side, stats_dev, sigma2, eta2);
}

// ---- bias ------------------------------------------------------------------
template <typename T, int LPR>
static void launch_bias(pmf_ctx *ctx, const BiasParams<T> &p, bool stats) {
    constexpr int G = 256 / LPR;
    dim3 grid((unsigned)((p.n_tasks + G - 1) / G));
    if (stats) hipLaunchKernelGGL((gauss_bias_kernel<T, LPR, true>), grid, dim3(256), 0, ctx->stream, p);
    else hipLaunchKernelGGL((gauss_bias_kernel<T, LPR, false>), grid, dim3(256), 0, ctx->stream, p);
}

// mode 0 fused, 1 accumulate to stats, 2 finalize from stats
template <typename T>
static int run_bias(pmf_ctx *ctx, int side, int mode, void *stats, double sigma2, double eta_bias2) {
    const int other = 1 - side;
    const PmfSideIndex &ix = ctx->index[side];
    const PmfTaskView tl = pmf_task_view(ctx, side, ix.bias_tasks, mode != 0);
    int rc;
    PMF_REQUIRE(ix.d_ptr, PMF_EINVAL, "pmf_gauss_bias_sweep: ratings have not been set");
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_FACTOR, "pmf_gauss_bias_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, other, PMF_ARR_FACTOR, "pmf_gauss_bias_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_BIAS, "pmf_gauss_bias_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, other, PMF_ARR_BIAS, "pmf_gauss_bias_sweep"))) return rc;
    if (mode != 1) PMF_REQUIRE(sigma2 > 0 && eta_bias2 > 0, PMF_EINVAL, "pmf_gauss_bias_sweep: variances must be positive");
    if (mode != 2 && tl.n_slots > 0)
        if ((rc = pmf_ensure_partial(ctx, (size_t)tl.n_slots * sizeof(T)))) return rc;
    BiasParams<T> p;
    p.tasks = tl.d_tasks;
    p.n_tasks = tl.n_tasks;
    p.split = tl.d_split;
    p.other = ix.d_other;
    p.val = (const T *)ix.d_val;
    p.factor_self = (const T *)ctx->arr[side][PMF_ARR_FACTOR];
    p.factor_other = (const T *)ctx->arr[other][PMF_ARR_FACTOR];
    p.bias_self = (T *)ctx->arr[side][PMF_ARR_BIAS];
    p.bias_other = (const T *)ctx->arr[other][PMF_ARR_BIAS];
    p.partial = (T *)ctx->d_partial;
    p.stats = (T *)stats;
    p.inv_sigma2 = mode == 1 ? (T)1 : (T)(1.0 / sigma2);
    p.inv_eta_bias2 = mode == 1 ? (T)1 : (T)(1.0 / eta_bias2);
    p.kpad = ctx->kpad;
    p.row0 = tl.row0;
    p.rows = tl.row1;
    PmfProfScope prof(ctx, PMF_KERNEL_GAUSS_BIAS);
    if (mode == 2) {
        if (p.rows > p.row0) {
            dim3 grid((unsigned)((p.rows - p.row0 + 255) / 256));
            hipLaunchKernelGGL((gauss_bias_finalize_all_kernel<T>), grid, dim3(256), 0, ctx->stream, p);
        }
        PMF_HIP_CHECK(hipGetLastError());
        return PMF_OK;
    }
    if (mode == 1 && p.rows > p.row0)
        PMF_HIP_CHECK(hipMemsetAsync((T *)stats + p.row0 * 2, 0, (size_t)(p.rows - p.row0) * 2 * sizeof(T), ctx->stream));
    if (tl.n_tasks > 0) {
        switch (pmf_lanes_per_row(ctx->kpad)) {
            case 1: launch_bias<T, 1>(ctx, p, mode == 1); break;
            case 2: launch_bias<T, 2>(ctx, p, mode == 1); break;
            case 4: launch_bias<T, 4>(ctx, p, mode == 1); break;
            case 8: launch_bias<T, 8>(ctx, p, mode == 1); break;
            case 16: launch_bias<T, 16>(ctx, p, mode == 1); break;
            case 32: launch_bias<T, 32>(ctx, p, mode == 1); break;
            default: launch_bias<T, 64>(ctx, p, mode == 1); break;
        }
    }
    if (tl.n_split > 0) {
        dim3 grid((unsigned)((tl.n_split + 255) / 256));
        if (mode == 1)
            hipLaunchKernelGGL((gauss_bias_split_kernel<T, true>), grid, dim3(256), 0, ctx->stream, p, tl.n_split, ix.d_ptr);
        else
            hipLaunchKernelGGL((gauss_bias_split_kernel<T, false>), grid, dim3(256), 0, ctx->stream, p, tl.n_split, ix.d_ptr);
    }
    PMF_HIP_CHECK(hipGetLastError());
    return PMF_OK;
}

template <typename T>
static int run_bias_dist(pmf_ctx *ctx, int side, double sigma2, double eta_bias2) {
    void *stats = nullptr;   // [rows x 2]: latency-bound, one message
    int rc = pmf_comm_stats(ctx, 1, (size_t)ctx->rows[side] * 2 * sizeof(T), &stats);
    if (rc) return rc;
    PmfExchange ex;
    ex.arrays[ex.n_arrays++] = PMF_ARR_BIAS;
    return pmf_comm_half_sweep(
        ctx, side, 2, stats, false, [&] { return run_bias<T>(ctx, side, 1, stats, 1, 1); },
        [&] { return run_bias<T>(ctx, side, 2, stats, sigma2, eta_bias2); }, ex);
}

extern "C" int pmf_gauss_bias_sweep(pmf_ctx *ctx, int side, double sigma2, double eta_bias2) {
    GAUSS_PROLOGUE("pmf_gauss_bias_sweep");
    if (side == PMF_SIDE_ITEM && pmf_comm_active(ctx))
        return ctx->dtype == PMF_F64 ? run_bias_dist<double>(ctx, side, sigma2, eta_bias2)
                                     : run_bias_dist<float>(ctx, side, sigma2, eta_bias2);
    if (ctx->dtype == PMF_F64) return run_bias<double>(ctx, side, 0, nullptr, sigma2, eta_bias2);
    return run_bias<float>(ctx, side, 0, nullptr, sigma2, eta_bias2);
}

extern "C" int pmf_gauss_bias_accumulate(pmf_ctx *ctx, int side, void *stats_dev) {
    GAUSS_PROLOGUE("pmf_gauss_bias_accumulate");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gauss_bias_accumulate: null stats buffer");
    if (ctx->dtype == PMF_F64) return run_bias<double>(ctx, side, 1, stats_dev, 1, 1);
    return run_bias<float>(ctx, side, 1, stats_dev, 1, 1);
}

extern "C" int pmf_gauss_bias_finalize(pmf_ctx *ctx, int side, const void *stats_dev, double sigma2,
                                       double eta_bias2) {
    GAUSS_PROLOGUE("pmf_gauss_bias_finalize");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gauss_bias_finalize: null stats buffer");
    if (ctx->dtype == PMF_F64) return run_bias<double>(ctx, side, 2, (void *)stats_dev, sigma2, eta_bias2);
    return run_bias<float>(ctx, side, 2, (void *)stats_dev, sigma2, eta_bias2);
}

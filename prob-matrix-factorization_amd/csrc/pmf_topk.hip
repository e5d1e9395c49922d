// Top-k items per user from the dense reconstruction FACTOR_user . FACTOR_item^T
// (+ biases): the one dense contraction of the product, so it runs on the matrix
// cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains) for fp32 contexts.
// Scores follow `predict` (hpf_cavi.py:215-231); ties go to the lower item id.
//
// Two phases per batch of Q query users: (1) score tile kernel writes
// scores[Q x I]; (2) one wavefront per user selects the k best in k passes
// (each pass = strided scan + wave arg-max restricted to entries ranked after
// the previous pick), which is deterministic and needs no sorting network.
#include <algorithm>
#include <type_traits>

#include "pmf_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TopkParams {
    const int32_t *users;  // [nq] query user ids (device)
    int nq;
    int64_t n_items;
    int K, kpad;
};

// 32 users x 32 items per wavefront, 4 item tiles per block.
__global__ __launch_bounds__(256) void topk_scores_f32_kernel(TopkParams p, const float *fu, const float *fi,
                                                              const float *bu, const float *bi, float *scores) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, c = lane & 31;
    const int q0 = blockIdx.x * 32;
    const int64_t i0 = ((int64_t)blockIdx.y * 4 + wave) * 32;
    if (i0 >= p.n_items) return;
    // k is split between the two half-waves: half h covers [h*H, (h+1)*H)
    const int H = ((p.kpad / 2) + 3) / 4 * 4;
    const int q = q0 + c;
    const int64_t it = i0 + c;
    const int user = (q < p.nq) ? p.users[q] : -1;
    const float *arow = (user >= 0) ? fu + (int64_t)user * p.kpad : nullptr;
    const float *brow = (it < p.n_items) ? fi + it * p.kpad : nullptr;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int t = 0; t < H; t += 4) {
        const int k = h * H + t;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (k < p.kpad) {
            if (arow) a = *reinterpret_cast<const float4 *>(arow + k);
            if (brow) b = *reinterpret_cast<const float4 *>(brow + k);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    // C/D layout: col = lane & 31 (item), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (user)
    const float bcol = (bi && it < p.n_items) ? bi[it] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int qq = q0 + row;
        if (qq < p.nq && it < p.n_items) {
            float s = acc[r];
            if (bu) s = bu[p.users[qq]] + bcol + s;
            scores[(int64_t)qq * p.n_items + it] = s;
        }
    }
}

// fp64 contexts: plain dot products (parity mode, not a throughput path)
__global__ void topk_scores_f64_kernel(TopkParams p, const double *fu, const double *fi, const double *bu,
                                       const double *bi, double *scores) {
    const int64_t total = (int64_t)p.nq * p.n_items;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e / p.n_items);
        const int64_t it = e % p.n_items;
        const int user = p.users[q];
        const double *a = fu + (int64_t)user * p.kpad, *b = fi + it * p.kpad;
        double s = 0.0;
        for (int k = 0; k < p.K; ++k) s = fma(a[k], b[k], s);
        if (bu) s = bu[user] + bi[it] + s;
        scores[e] = s;
    }
}

// (value desc, index asc) order helpers
template <typename S>
__device__ __forceinline__ bool ranks_before(S v1, int i1, S v2, int i2) {
    return v1 > v2 || (v1 == v2 && i1 < i2);
}

// wave arg-best under (value desc, index asc); every lane returns the winner
template <typename S>
__device__ __forceinline__ void wave_best(S &bv, int &bidx, bool &found) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const S ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        const int of = __shfl_xor((int)found, off, 64);
        if (of && (!found || ranks_before(ov, oi, bv, bidx))) {
            bv = ov;
            bidx = oi;
            found = true;
        }
    }
}

// One wavefront per user, k <= 64.  Two passes over the user's score row instead of k:
//  (1) per-lane maxima; the k-th largest of the 64 lane maxima is a lower bound tau of the
//      k-th best score (the k best lane maxima are k distinct entries >= tau);
//  (2) every entry >= tau is compacted (ballot + prefix) into an LDS candidate list, from
//      which the k winners are drawn with the deterministic (value desc, index asc) rule.
// A list that would overflow (massive ties) falls back to the k-pass scan of the row.
#define TOPK_CAP 1024
template <typename S>
__global__ __launch_bounds__(256) void topk_select_kernel(const S *scores, int nq, int64_t n_items, int k,
                                                          int32_t *out_items, double *out_scores) {
    __shared__ S c_val[4][TOPK_CAP];
    __shared__ int c_idx[4][TOPK_CAP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const S *row = scores + (int64_t)q * n_items;
    S *cv = c_val[wave];
    int *ci = c_idx[wave];
    int count = -1;  // -1: candidate list unusable -> scan the row itself
    if (k <= 64) {
        // pass 1: lane maxima
        bool have = false;
        S lm = (S)0;
        for (int64_t i = lane; i < n_items; i += 64) {
            const S v = row[i];
            if (v == v && (!have || v > lm)) {
                lm = v;
                have = true;
            }
        }
        // k-th largest lane maximum (values only; ties are harmless for a lower bound)
        S tau = (S)0;
        bool tau_ok = false;
        {
            bool alive = have;
            for (int t = 0; t < k; ++t) {
                S bv = lm;
                int bidx = lane;
                bool found = alive;
                wave_best(bv, bidx, found);
                tau_ok = found;
                if (!found) break;
                tau = bv;
                if (lane == bidx) alive = false;
            }
        }
        if (tau_ok) {
            // pass 2: compact entries >= tau
            count = 0;
            const int64_t rounds = (n_items + 63) / 64;
            for (int64_t r = 0; r < rounds && count >= 0; ++r) {
                const int64_t i = r * 64 + lane;
                S v = (S)0;
                bool pred = false;
                if (i < n_items) {
                    v = row[i];
                    pred = (v == v) && v >= tau;
                }
                const unsigned long long mask = __ballot(pred);
                const int n = __popcll(mask);
                if (n) {
                    if (count + n > TOPK_CAP) {
                        count = -1;
                    } else {
                        if (pred) {
                            const int at = count + __popcll(mask & ((1ull << lane) - 1ull));
                            cv[at] = v;
                            ci[at] = (int)i;
                        }
                        count += n;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    bool have_prev = false;
    S pv = (S)0;
    int pi = -1;
    for (int t = 0; t < k; ++t) {
        bool found = false;
        S bv = (S)0;
        int bidx = 0x7fffffff;
        if (count >= 0) {
            for (int e = lane; e < count; e += 64) {
                const S v = cv[e];
                const int i = ci[e];
                if (have_prev && !ranks_before(pv, pi, v, i)) continue;
                if (!found || ranks_before(v, i, bv, bidx)) {
                    bv = v;
                    bidx = i;
                    found = true;
                }
            }
        } else {
            for (int64_t i = lane; i < n_items; i += 64) {
                const S v = row[i];
                if (!(v == v)) continue;  // NaN never ranks
                if (have_prev && !ranks_before(pv, pi, v, (int)i)) continue;
                if (!found || ranks_before(v, (int)i, bv, bidx)) {
                    bv = v;
                    bidx = (int)i;
                    found = true;
                }
            }
        }
        wave_best(bv, bidx, found);
        if (lane == 0) {
            out_items[(int64_t)q * k + t] = found ? bidx : -1;
            out_scores[(int64_t)q * k + t] = found ? (double)bv : 0.0;
        }
        if (!found) {  // fewer than k rankable items: fill the rest
            for (int r = t + 1; r < k && lane == 0; ++r) {
                out_items[(int64_t)q * k + r] = -1;
                out_scores[(int64_t)q * k + r] = 0.0;
            }
            break;
        }
        have_prev = true;
        pv = bv;
        pi = bidx;
    }
}

template <typename T>
static int run_topk(pmf_ctx *ctx, int64_t n_query, const int32_t *user_ids, int k, int use_bias,
                    int32_t *out_items, double *out_scores) {
    const int64_t I = ctx->rows[PMF_SIDE_ITEM];
    int rc;
    if ((rc = pmf_require_array(ctx, PMF_SIDE_USER, PMF_ARR_FACTOR, "pmf_topk_items"))) return rc;
    if ((rc = pmf_require_array(ctx, PMF_SIDE_ITEM, PMF_ARR_FACTOR, "pmf_topk_items"))) return rc;
    if (use_bias) {
        if ((rc = pmf_require_array(ctx, PMF_SIDE_USER, PMF_ARR_BIAS, "pmf_topk_items"))) return rc;
        if ((rc = pmf_require_array(ctx, PMF_SIDE_ITEM, PMF_ARR_BIAS, "pmf_topk_items"))) return rc;
    }
    const T *fu = (const T *)ctx->arr[PMF_SIDE_USER][PMF_ARR_FACTOR];
    const T *fi = (const T *)ctx->arr[PMF_SIDE_ITEM][PMF_ARR_FACTOR];
    const T *bu = use_bias ? (const T *)ctx->arr[PMF_SIDE_USER][PMF_ARR_BIAS] : nullptr;
    const T *bi = use_bias ? (const T *)ctx->arr[PMF_SIDE_ITEM][PMF_ARR_BIAS] : nullptr;
    // batch size: scores buffer of at most ~512 MB
    int64_t Q = std::max<int64_t>(32, (512ll << 20) / (I * (int64_t)sizeof(T)));
    Q = std::min<int64_t>(Q / 32 * 32, (n_query + 31) / 32 * 32);
    const size_t score_bytes = (size_t)Q * I * sizeof(T);
    const size_t id_bytes = (size_t)Q * sizeof(int32_t);
    const size_t out_bytes = (size_t)Q * k * (sizeof(int32_t) + sizeof(double));
    if ((rc = pmf_ensure_scratch(ctx, score_bytes + id_bytes + out_bytes + 64))) return rc;
    char *base = (char *)ctx->d_scratch;
    T *d_scores = (T *)base;
    double *d_out_scores = (double *)(base + score_bytes);
    int32_t *d_out_items = (int32_t *)(base + score_bytes + (size_t)Q * k * sizeof(double));
    int32_t *d_users = d_out_items + (size_t)Q * k;
    for (int64_t at = 0; at < n_query; at += Q) {
        const int nq = (int)std::min<int64_t>(Q, n_query - at);
        PMF_HIP_CHECK(hipMemcpyAsync(d_users, user_ids + at, (size_t)nq * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        TopkParams p;
        p.users = d_users;
        p.nq = nq;
        p.n_items = I;
        p.K = ctx->K;
        p.kpad = ctx->kpad;
        {
            PmfProfScope prof(ctx, PMF_KERNEL_TOPK);
            if constexpr (std::is_same<T, float>::value) {
                dim3 grid((unsigned)((nq + 31) / 32), (unsigned)((I + 127) / 128));
                hipLaunchKernelGGL(topk_scores_f32_kernel, grid, dim3(256), 0, ctx->stream, p, fu, fi, bu, bi, d_scores);
            } else {
                const int64_t total = (int64_t)nq * I;
                hipLaunchKernelGGL(topk_scores_f64_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)),
                                   dim3(256), 0, ctx->stream, p, fu, fi, bu, bi, d_scores);
            }
            hipLaunchKernelGGL((topk_select_kernel<T>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, ctx->stream,
                               d_scores, nq, I, k, d_out_items, d_out_scores);
        }
        PMF_HIP_CHECK(hipGetLastError());
        PMF_HIP_CHECK(hipMemcpyAsync(out_items + at * k, d_out_items, (size_t)nq * k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        PMF_HIP_CHECK(hipMemcpyAsync(out_scores + at * k, d_out_scores, (size_t)nq * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PMF_OK;
}

extern "C" int pmf_topk_items(pmf_ctx *ctx, int64_t n_query, const int32_t *user_ids, int k, int use_bias,
                              int32_t *out_items, double *out_scores) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_topk_items: null context");
    PMF_REQUIRE(n_query >= 0, PMF_EINVAL, "pmf_topk_items: negative n_query");
    if (n_query == 0) return PMF_OK;
    PMF_REQUIRE(user_ids && out_items && out_scores, PMF_EINVAL, "pmf_topk_items: null argument");
    PMF_REQUIRE(k >= 1 && k <= 1024 && k <= ctx->rows[PMF_SIDE_ITEM], PMF_ERANGE,
                "pmf_topk_items: k=%d outside [1, min(1024, n_items)]", k);
    for (int64_t n = 0; n < n_query; ++n)
        PMF_REQUIRE(user_ids[n] >= 0 && user_ids[n] < ctx->rows[PMF_SIDE_USER], PMF_ERANGE,
                    "pmf_topk_items: user id %d at position %lld outside [0, %lld)", user_ids[n], (long long)n,
                    (long long)ctx->rows[PMF_SIDE_USER]);
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->dtype == PMF_F64) return run_topk<double>(ctx, n_query, user_ids, k, use_bias, out_items, out_scores);
    return run_topk<float>(ctx, n_query, user_ids, k, use_bias, out_items, out_scores);
}

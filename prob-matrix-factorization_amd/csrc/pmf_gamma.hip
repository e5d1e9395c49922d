// Poisson MF / HPF half-sweep kernels (rows a5, a6, a7 of SURVEY.md section 8).
//
// Work decomposition: a *lane group* of LPR = pow2(Kpad/4) lanes owns one task
// (a run of at most PMF_GAMMA_CHUNK ratings of one row); lane c of the group
// owns factor elements [4c, 4c+4) of every K-vector it touches, so one gathered
// factor row is one 16-byte access per lane and a wavefront gathers 64/LPR rows
// per instruction (K = 64 fp32: 4 rows x 256 B = one full 1-KiB wave access).
// The K-length dot product is a 4-FMA partial per lane + a DPP reduction inside
// the group; shape/rate sums live in registers for the whole task, so every
// rating costs exactly one index, one value and one gathered row of HBM traffic.
// No atomics: rows that exceed one chunk write per-chunk partial sums that a
// second kernel adds in chunk order, so results are bitwise reproducible.
#include "pmf_device.h"

template <typename T>
struct GammaParams {
    const PmfTask *tasks;
    int64_t n_tasks;
    const PmfSplitRow *split;
    const int32_t *other;
    const T *val;
    T *factor_self;
    const T *factor_other;
    T *shape;
    T *rate;
    T *prior_rate_vec;  // E_xi / E_eta (HPF) or null
    T *hyper_rate;      // gamma_b_xi / gamma_b_eta (HPF) or null
    // extended model (EXT): per-row scalar factors phi / psi
    const T *scale_other;
    T *scale_self;
    T *scale_shape;
    T *scale_rate;
    int pw;             // partial slot width: 2*kpad (+ PMF_VEC carrying sum x when EXT)
    T *partial;         // [n_slots][pw]
    T *stats;           // [rows][2][kpad], STATS mode only
    T shape_prior, rate_prior, hyper_shape, hyper_rate_prior;
    int hierarchical;
    int K, kpad;
    int64_t row0;       // first row of a finalize-from-stats launch
    int64_t rows;       // one past its last row
};

// shape = prior + sum, rate = prior_rate + sum, E = shape / rate, plus the
// xi / eta update (hpf_cavi.py:155-159) when hierarchical.
template <typename T, int LPR, bool EXT>
__device__ __forceinline__ void gamma_finalize_row(const GammaParams<T> &p, int row, int c, bool active,
                                                   const Vec4<T> &sum_a, const Vec4<T> &sum_b, T xsum, bool empty) {
    const int koff = c * PMF_VEC;
    const T rp = p.hierarchical ? p.prior_rate_vec[row] : p.rate_prior;
    Vec4<T> sh, rt, ex;
    T esum = (T)0;
#pragma unroll
    for (int e = 0; e < PMF_VEC; ++e) {
        const bool ok = active && (koff + e) < p.K;
        T s = p.shape_prior + sum_a.v[e];
        T r = rp + sum_b.v[e];
        T x = s / r;
        sh.v[e] = ok ? s : (T)0;
        rt.v[e] = ok ? r : (T)0;
        ex.v[e] = ok ? x : (T)0;
        esum += EXT ? ex.v[e] * sum_b.v[e] : ex.v[e];
    }
    if (active) {
        const int64_t at = (int64_t)row * p.kpad + koff;
        store4(p.shape + at, sh);
        store4(p.rate + at, rt);
        if (!(EXT && empty)) store4(p.factor_self + at, ex);
    }
    if (EXT) {
        // phi / psi: shape a0 + sum x, rate b0 + sum_j s_j (other_j . FACTOR_new[row]) = b0 + FACTOR_new . sum_b
        esum = group_sum<LPR>(esum);
        if (c == 0) {
            const T ss = p.shape_prior + xsum;
            const T sr = p.rate_prior + (empty ? (T)0 : esum);
            p.scale_shape[row] = ss;
            p.scale_rate[row] = sr;
            if (!empty) p.scale_self[row] = ss / sr;
        }
    } else if (p.hierarchical) {
        esum = group_sum<LPR>(esum);
        if (c == 0) {
            T hr = p.hyper_rate_prior + esum;
            p.hyper_rate[row] = hr;
            p.prior_rate_vec[row] = p.hyper_shape / hr;
        }
    }
}

template <typename T, int LPR, bool STATS, bool EXT>
__global__ __launch_bounds__(256) void gamma_sweep_kernel(GammaParams<T> p) {
    constexpr int G = 256 / LPR;
    constexpr int UN = LPR < 4 ? LPR : 4;
    const int c = threadIdx.x % LPR;
    const int64_t task_id = (int64_t)blockIdx.x * G + threadIdx.x / LPR;
    if (task_id >= p.n_tasks) return;
    const PmfTask t = p.tasks[task_id];
    const int koff = c * PMF_VEC;
    const bool active = koff < p.kpad;
    const int kpad = p.kpad;

    Vec4<T> self = active ? load4(p.factor_self + (int64_t)t.row * kpad + koff) : zero4<T>();
    Vec4<T> acc_a = zero4<T>(), acc_b = zero4<T>();
    T xsum = (T)0;
    const int32_t *col = p.other + t.start;
    const T *val = p.val + t.start;

    for (int base = 0; base < t.len; base += LPR) {
        const int n = min(LPR, t.len - base);
        int my_o = 0;
        T my_x = (T)0;
        if (c < n) {
            my_o = col[base + c];
            my_x = val[base + c];
        }
        for (int tt = 0; tt < n; tt += UN) {
            int o[UN];
            T xv[UN], sv[UN];
            Vec4<T> b[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                o[q] = __shfl(my_o, tt + q, LPR);
                xv[q] = __shfl(my_x, tt + q, LPR);
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                b[q] = active ? load4(p.factor_other + (int64_t)o[q] * kpad + koff) : zero4<T>();
                sv[q] = EXT ? p.scale_other[o[q]] : (T)1;
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                if (tt + q < n) {
                    T d = b[q].v[0] * self.v[0];
                    d = fma(b[q].v[1], self.v[1], d);
                    d = fma(b[q].v[2], self.v[2], d);
                    d = fma(b[q].v[3], self.v[3], d);
                    d = group_sum<LPR>(d);
                    if (!EXT) d = vmax(d, (T)PMF_RATE_FLOOR);
                    const T w = xv[q] / d;
                    if (EXT) xsum += xv[q];
#pragma unroll
                    for (int e = 0; e < PMF_VEC; ++e) {
                        acc_a.v[e] += (w * b[q].v[e]) * self.v[e];
                        acc_b.v[e] += EXT ? b[q].v[e] * sv[q] : b[q].v[e];
                    }
                }
            }
        }
    }

    if (t.slot >= 0) {
        T *slot = p.partial + (int64_t)t.slot * p.pw;
        if (active) {
            store4(slot + koff, acc_a);
            store4(slot + kpad + koff, acc_b);
        }
        if (EXT && c == 0) slot[2 * kpad] = xsum;
    } else if (STATS) {
        if (active) {
            T *dst = p.stats + (int64_t)t.row * 2 * kpad + koff;
            store4(dst, acc_a);
            store4(dst + kpad, acc_b);
        }
    } else {
        gamma_finalize_row<T, LPR, EXT>(p, t.row, c, active, acc_a, acc_b, xsum, t.len == 0);
    }
}

// Gather ceiling of this access shape (profiling aid, pmf_prof_gather_ceiling): the sweep kernel's
// memory side only -- the same tasks, the same coalesced index / rating loads, the same 16-byte-per-lane
// row gathers with UN in flight -- with the arithmetic reduced to one add per loaded value and no row
// output.  Its time is what the L2 / Infinity Cache / HBM deliver for this gather pattern; the real
// kernel cannot be faster.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void gamma_gather_probe_kernel(GammaParams<T> p, T *sink) {
    constexpr int G = 256 / LPR;
    constexpr int UN = LPR < 4 ? LPR : 4;
    const int c = threadIdx.x % LPR;
    const int64_t task_id = (int64_t)blockIdx.x * G + threadIdx.x / LPR;
    if (task_id >= p.n_tasks) return;
    const PmfTask t = p.tasks[task_id];
    const int koff = c * PMF_VEC;
    const bool active = koff < p.kpad;
    const int kpad = p.kpad;
    Vec4<T> acc = active ? load4(p.factor_self + (int64_t)t.row * kpad + koff) : zero4<T>();
    const int32_t *col = p.other + t.start;
    const T *val = p.val + t.start;
    for (int base = 0; base < t.len; base += LPR) {
        const int n = min(LPR, t.len - base);
        int my_o = 0;
        T my_x = (T)0;
        if (c < n) {
            my_o = col[base + c];
            my_x = val[base + c];
        }
        acc.v[0] += my_x;
        for (int tt = 0; tt < n; tt += UN) {
            int o[UN];
            Vec4<T> b[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) o[q] = __shfl(my_o, tt + q, LPR);
#pragma unroll
            for (int q = 0; q < UN; ++q) b[q] = active ? load4(p.factor_other + (int64_t)o[q] * kpad + koff) : zero4<T>();
#pragma unroll
            for (int q = 0; q < UN; ++q)
                if (tt + q < n) {
#pragma unroll
                    for (int e = 0; e < PMF_VEC; ++e) acc.v[e] += b[q].v[e];
                }
        }
    }
    // never true for finite data; keeps every load alive without an output stream
    if (acc.v[0] + acc.v[1] + acc.v[2] + acc.v[3] == (T)-1.2345678e30) sink[0] = acc.v[0];
}

// One block per split row: group g adds slots g, g+G, ... in order, the G
// group sums are then added in group order by group 0 (fixed order => bitwise
// reproducible), which finalises the row (or writes its raw sums in STATS mode).
template <typename T, int LPR, bool STATS, bool EXT>
__global__ __launch_bounds__(256) void gamma_split_kernel(GammaParams<T> p) {
    constexpr int G = 256 / LPR;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);  // [G][2][kpad]
    const PmfSplitRow sr = p.split[blockIdx.x];
    const int c = threadIdx.x % LPR;
    const int g = threadIdx.x / LPR;
    const int koff = c * PMF_VEC;
    const int kpad = p.kpad;
    const bool active = koff < kpad;
    Vec4<T> sa = zero4<T>(), sb = zero4<T>();
    T xsum = (T)0;
    if (EXT && threadIdx.x == 0)  // G*2*kpad sums are staged in LDS; sum x is tiny: one lane adds it in order
        for (int s = 0; s < sr.n_slots; ++s) xsum += p.partial[(int64_t)(sr.first_slot + s) * p.pw + 2 * kpad];
    if (active) {
        for (int s = g; s < sr.n_slots; s += G) {
            const T *src = p.partial + (int64_t)(sr.first_slot + s) * p.pw + koff;
            Vec4<T> a = load4(src), b = load4(src + kpad);
#pragma unroll
            for (int e = 0; e < PMF_VEC; ++e) {
                sa.v[e] += a.v[e];
                sb.v[e] += b.v[e];
            }
        }
        store4(smem + (int64_t)g * 2 * kpad + koff, sa);
        store4(smem + (int64_t)g * 2 * kpad + kpad + koff, sb);
    }
    __syncthreads();
    if (g != 0) return;
    sa = zero4<T>();
    sb = zero4<T>();
    if (active) {
        const int ng = min(G, sr.n_slots);
        for (int s = 0; s < ng; ++s) {
            Vec4<T> a = load4(smem + (int64_t)s * 2 * kpad + koff);
            Vec4<T> b = load4(smem + (int64_t)s * 2 * kpad + kpad + koff);
#pragma unroll
            for (int e = 0; e < PMF_VEC; ++e) {
                sa.v[e] += a.v[e];
                sb.v[e] += b.v[e];
            }
        }
    }
    if (STATS) {
        if (active) {
            T *dst = p.stats + (int64_t)sr.row * 2 * kpad + koff;
            store4(dst, sa);
            store4(dst + kpad, sb);
        }
    } else {
        if (EXT) xsum = __shfl(xsum, 0, LPR);  // group 0 = lanes [0, LPR): lane 0 holds the total
        gamma_finalize_row<T, LPR, EXT>(p, sr.row, c, active, sa, sb, xsum, false);
    }
}

// STATS mode, after the all-reduce: every row from its summed statistics.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void gamma_finalize_all_kernel(GammaParams<T> p) {
    constexpr int G = 256 / LPR;
    const int c = threadIdx.x % LPR;
    const int64_t row = p.row0 + (int64_t)blockIdx.x * G + threadIdx.x / LPR;
    if (row >= p.rows) return;
    const int koff = c * PMF_VEC;
    const bool active = koff < p.kpad;
    Vec4<T> sa = zero4<T>(), sb = zero4<T>();
    if (active) {
        const T *src = p.stats + row * 2 * p.kpad + koff;
        sa = load4(src);
        sb = load4(src + p.kpad);
    }
    gamma_finalize_row<T, LPR, false>(p, (int)row, c, active, sa, sb, (T)0, false);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, int LPR>
static int launch_gamma(pmf_ctx *ctx, int side, GammaParams<T> &p, const PmfTaskView &tl,
                        int mode /*0 fused, 1 accumulate, 2 finalize, 3 extended, 4 gather probe*/) {
    constexpr int G = 256 / LPR;
    if (mode == 4) {
        if (tl.n_tasks > 0)
            hipLaunchKernelGGL((gamma_gather_probe_kernel<T, LPR>), dim3((unsigned)((tl.n_tasks + G - 1) / G)), dim3(256), 0,
                               ctx->stream, p, (T *)ctx->d_scratch);
        PMF_HIP_CHECK(hipGetLastError());
        return PMF_OK;
    }
    if (mode != 2) {
        if (tl.n_tasks > 0) {
            PmfProfScope prof(ctx, PMF_KERNEL_GAMMA_SWEEP);
            dim3 grid((unsigned)((tl.n_tasks + G - 1) / G));
            if (mode == 0)
                hipLaunchKernelGGL((gamma_sweep_kernel<T, LPR, false, false>), grid, dim3(256), 0, ctx->stream, p);
            else if (mode == 3)
                hipLaunchKernelGGL((gamma_sweep_kernel<T, LPR, false, true>), grid, dim3(256), 0, ctx->stream, p);
            else
                hipLaunchKernelGGL((gamma_sweep_kernel<T, LPR, true, false>), grid, dim3(256), 0, ctx->stream, p);
        }
        if (tl.n_split > 0) {
            PmfProfScope prof(ctx, PMF_KERNEL_GAMMA_FINAL);
            size_t smem = (size_t)G * 2 * ctx->kpad * sizeof(T);
            if (mode == 0)
                hipLaunchKernelGGL((gamma_split_kernel<T, LPR, false, false>), dim3((unsigned)tl.n_split), dim3(256), smem, ctx->stream, p);
            else if (mode == 3)
                hipLaunchKernelGGL((gamma_split_kernel<T, LPR, false, true>), dim3((unsigned)tl.n_split), dim3(256), smem, ctx->stream, p);
            else
                hipLaunchKernelGGL((gamma_split_kernel<T, LPR, true, false>), dim3((unsigned)tl.n_split), dim3(256), smem, ctx->stream, p);
        }
    } else {
        PmfProfScope prof(ctx, PMF_KERNEL_GAMMA_FINAL);
        if (p.rows > p.row0) {
            dim3 grid((unsigned)((p.rows - p.row0 + G - 1) / G));
            hipLaunchKernelGGL((gamma_finalize_all_kernel<T, LPR>), grid, dim3(256), 0, ctx->stream, p);
        }
    }
    PMF_HIP_CHECK(hipGetLastError());
    return PMF_OK;
}

template <typename T>
static int run_gamma(pmf_ctx *ctx, int side, int mode, void *stats, double shape_prior, double rate_prior,
                     int hierarchical, double hyper_shape, double hyper_rate_prior) {
    const int other = 1 - side;
    const PmfSideIndex &ix = ctx->index[side];
    const PmfTaskView tl = pmf_task_view(ctx, side, ix.gamma_tasks, mode == 1 || mode == 2);
    int rc;
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_FACTOR, "pmf_gamma_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, other, PMF_ARR_FACTOR, "pmf_gamma_sweep"))) return rc;
    PMF_REQUIRE(ix.d_ptr, PMF_EINVAL, "pmf_gamma_sweep: ratings have not been set");
    if (mode == 4) {
        if ((rc = pmf_ensure_scratch(ctx, 64))) return rc;
    } else if (mode != 1) {
        if ((rc = pmf_alloc_array(ctx, side, PMF_ARR_SHAPE))) return rc;
        if ((rc = pmf_alloc_array(ctx, side, PMF_ARR_RATE))) return rc;
        if (hierarchical) {
            if ((rc = pmf_require_array(ctx, side, PMF_ARR_PRIOR_RATE, "pmf_gamma_sweep (hierarchical)"))) return rc;
            if ((rc = pmf_alloc_array(ctx, side, PMF_ARR_HYPER_RATE))) return rc;
        }
    }
    const int pw = 2 * ctx->kpad + (mode == 3 ? PMF_VEC : 0);
    if (mode != 2 && mode != 4 && tl.n_slots > 0)
        if ((rc = pmf_ensure_partial(ctx, (size_t)tl.n_slots * pw * sizeof(T)))) return rc;
    if (mode == 3) {
        if ((rc = pmf_require_array(ctx, side, PMF_ARR_SCALE, "pmf_gamma_ext_sweep"))) return rc;
        if ((rc = pmf_require_array(ctx, other, PMF_ARR_SCALE, "pmf_gamma_ext_sweep"))) return rc;
        if ((rc = pmf_alloc_array(ctx, side, PMF_ARR_SCALE_SHAPE))) return rc;
        if ((rc = pmf_alloc_array(ctx, side, PMF_ARR_SCALE_RATE))) return rc;
    }

    GammaParams<T> p;
    p.tasks = tl.d_tasks;
    p.n_tasks = tl.n_tasks;
    p.split = tl.d_split;
    p.other = ix.d_other;
    p.val = (const T *)ix.d_val;
    p.factor_self = (T *)ctx->arr[side][PMF_ARR_FACTOR];
    p.factor_other = (const T *)ctx->arr[other][PMF_ARR_FACTOR];
    p.shape = (T *)ctx->arr[side][PMF_ARR_SHAPE];
    p.rate = (T *)ctx->arr[side][PMF_ARR_RATE];
    p.prior_rate_vec = (T *)ctx->arr[side][PMF_ARR_PRIOR_RATE];
    p.hyper_rate = (T *)ctx->arr[side][PMF_ARR_HYPER_RATE];
    p.scale_other = (const T *)ctx->arr[other][PMF_ARR_SCALE];
    p.scale_self = (T *)ctx->arr[side][PMF_ARR_SCALE];
    p.scale_shape = (T *)ctx->arr[side][PMF_ARR_SCALE_SHAPE];
    p.scale_rate = (T *)ctx->arr[side][PMF_ARR_SCALE_RATE];
    p.pw = pw;
    p.partial = (T *)ctx->d_partial;
    p.stats = (T *)stats;
    p.shape_prior = (T)shape_prior;
    p.rate_prior = (T)rate_prior;
    p.hyper_shape = (T)hyper_shape;
    p.hyper_rate_prior = (T)hyper_rate_prior;
    p.hierarchical = hierarchical;
    p.K = ctx->K;
    p.kpad = ctx->kpad;
    p.row0 = tl.row0;   // finalize-from-stats covers rows [row0, rows)
    p.rows = tl.row1;

    switch (pmf_lanes_per_row(ctx->kpad)) {
        case 1: return launch_gamma<T, 1>(ctx, side, p, tl, mode);
        case 2: return launch_gamma<T, 2>(ctx, side, p, tl, mode);
        case 4: return launch_gamma<T, 4>(ctx, side, p, tl, mode);
        case 8: return launch_gamma<T, 8>(ctx, side, p, tl, mode);
        case 16: return launch_gamma<T, 16>(ctx, side, p, tl, mode);
        case 32: return launch_gamma<T, 32>(ctx, side, p, tl, mode);
        case 64: return launch_gamma<T, 64>(ctx, side, p, tl, mode);
    }
    pmf_set_error("pmf_gamma_sweep: unsupported n_factors %d", ctx->K);
    return PMF_ERANGE;
}

#define GAMMA_PROLOGUE(fn)                                                                              \
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, fn ": null context");                                       \
    PMF_REQUIRE(side == PMF_SIDE_USER || side == PMF_SIDE_ITEM, PMF_EINVAL, fn ": bad side %d", side);  \
    PMF_HIP_CHECK(hipSetDevice(ctx->device));

// several ranks: accumulate -> all-reduce -> finalize on the library's statistics buffer (pmf_comm.hip)
template <typename T>
static int run_gamma_dist(pmf_ctx *ctx, int side, double shape_prior, double rate_prior, int hierarchical,
                          double hyper_shape, double hyper_rate_prior) {
    const size_t width = (size_t)2 * ctx->kpad;
    void *stats = nullptr;
    int rc = pmf_comm_stats(ctx, 0, (size_t)ctx->rows[side] * width * sizeof(T), &stats);
    if (rc) return rc;
    // finalize is element-wise (cheap) and writes 3 Kpad + 2 values per row for 2 Kpad of statistics: the plain
    // all-reduce is the default exchange here
    PmfExchange ex;
    ex.arrays[ex.n_arrays++] = PMF_ARR_FACTOR;
    ex.arrays[ex.n_arrays++] = PMF_ARR_SHAPE;
    ex.arrays[ex.n_arrays++] = PMF_ARR_RATE;
    if (hierarchical) {
        ex.arrays[ex.n_arrays++] = PMF_ARR_PRIOR_RATE;
        ex.arrays[ex.n_arrays++] = PMF_ARR_HYPER_RATE;
    }
    if ((rc = pmf_alloc_array(ctx, side, PMF_ARR_SHAPE))) return rc;   // (the gathers address them before finalize runs)
    if ((rc = pmf_alloc_array(ctx, side, PMF_ARR_RATE))) return rc;
    if (hierarchical && (rc = pmf_alloc_array(ctx, side, PMF_ARR_HYPER_RATE))) return rc;
    return pmf_comm_half_sweep(
        ctx, side, width, stats, true, [&] { return run_gamma<T>(ctx, side, 1, stats, 0, 0, 0, 0, 0); },
        [&] { return run_gamma<T>(ctx, side, 2, stats, shape_prior, rate_prior, hierarchical, hyper_shape, hyper_rate_prior); },
        ex);
}

extern "C" int pmf_gamma_sweep(pmf_ctx *ctx, int side, double shape_prior, double rate_prior,
                               int hierarchical, double hyper_shape, double hyper_rate_prior) {
    GAMMA_PROLOGUE("pmf_gamma_sweep");
    if (side == PMF_SIDE_ITEM && pmf_comm_active(ctx)) {
        if (ctx->dtype == PMF_F64)
            return run_gamma_dist<double>(ctx, side, shape_prior, rate_prior, hierarchical, hyper_shape, hyper_rate_prior);
        return run_gamma_dist<float>(ctx, side, shape_prior, rate_prior, hierarchical, hyper_shape, hyper_rate_prior);
    }
    if (ctx->dtype == PMF_F64)
        return run_gamma<double>(ctx, side, 0, nullptr, shape_prior, rate_prior, hierarchical, hyper_shape, hyper_rate_prior);
    return run_gamma<float>(ctx, side, 0, nullptr, shape_prior, rate_prior, hierarchical, hyper_shape, hyper_rate_prior);
}

extern "C" int pmf_gamma_ext_sweep(pmf_ctx *ctx, int side, double shape_prior, double rate_prior) {
    GAMMA_PROLOGUE("pmf_gamma_ext_sweep");
    PMF_REQUIRE(!pmf_comm_active(ctx), PMF_EINVAL, "pmf_gamma_ext_sweep: the extended model is not available on several ranks");
    if (ctx->dtype == PMF_F64) return run_gamma<double>(ctx, side, 3, nullptr, shape_prior, rate_prior, 0, 0, 0);
    return run_gamma<float>(ctx, side, 3, nullptr, shape_prior, rate_prior, 0, 0, 0);
}

extern "C" int pmf_gamma_accumulate(pmf_ctx *ctx, int side, void *stats_dev) {
    GAMMA_PROLOGUE("pmf_gamma_accumulate");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gamma_accumulate: null stats buffer");
    if (ctx->dtype == PMF_F64) return run_gamma<double>(ctx, side, 1, stats_dev, 0, 0, 0, 0, 0);
    return run_gamma<float>(ctx, side, 1, stats_dev, 0, 0, 0, 0, 0);
}

extern "C" int pmf_gamma_finalize(pmf_ctx *ctx, int side, const void *stats_dev, double shape_prior,
                                  double rate_prior, int hierarchical, double hyper_shape,
                                  double hyper_rate_prior) {
    GAMMA_PROLOGUE("pmf_gamma_finalize");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gamma_finalize: null stats buffer");
    if (ctx->dtype == PMF_F64)
        return run_gamma<double>(ctx, side, 2, (void *)stats_dev, shape_prior, rate_prior, hierarchical, hyper_shape, hyper_rate_prior);
    return run_gamma<float>(ctx, side, 2, (void *)stats_dev, shape_prior, rate_prior, hierarchical, hyper_shape, hyper_rate_prior);
}

// Profiling aid (no reference counterpart): average device time of `repeats` launches of the
// gather-only twin of the Poisson/HPF half-sweep of `side` -- the ceiling the cache hierarchy sets
// for this context's gather pattern (bench.py reports the sweep kernel against it).
extern "C" int pmf_prof_gather_ceiling(pmf_ctx *ctx, int side, int repeats, double *ms_per_launch) {
    GAMMA_PROLOGUE("pmf_prof_gather_ceiling");
    PMF_REQUIRE(ms_per_launch != nullptr && repeats >= 1, PMF_EINVAL, "pmf_prof_gather_ceiling: bad arguments");
    hipEvent_t a = nullptr, b = nullptr;
    PMF_HIP_CHECK(hipEventCreate(&a));
    hipError_t e = hipEventCreate(&b);
    if (e != hipSuccess) {
        (void)hipEventDestroy(a);
        pmf_set_error("hipEventCreate failed: %s", hipGetErrorString(e));
        return PMF_EHIP;
    }
    int rc = PMF_OK;
    for (int k = 0; k <= repeats && !rc; ++k) {   // launch 0 warms the caches and is not timed
        if (k == 1) (void)hipEventRecord(a, ctx->stream);
        rc = ctx->dtype == PMF_F64 ? run_gamma<double>(ctx, side, 4, nullptr, 0, 0, 0, 0, 0)
                                   : run_gamma<float>(ctx, side, 4, nullptr, 0, 0, 0, 0, 0);
    }
    float ms = 0.f;
    if (!rc) {
        e = hipEventRecord(b, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (e != hipSuccess) {
            pmf_set_error("pmf_prof_gather_ceiling: %s", hipGetErrorString(e));
            rc = PMF_EHIP;
        }
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_per_launch = (double)ms / repeats;
    return rc;
}

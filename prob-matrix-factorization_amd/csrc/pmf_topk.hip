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
#include <climits>
#include <type_traits>

#include "pmf_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TopkParams {
    const int32_t *users;  // [nq] query user ids (device)
    int nq;
    int64_t n_items;
    int K, kpad;
};

// 32 users x 32 items per wavefront, 4 item tiles per block.
// (bu / bi: BIAS arrays when mode == PMF_PREDICT_BIAS, SCALE arrays when mode == PMF_PREDICT_SCALE, else null)
__global__ __launch_bounds__(256) void topk_scores_f32_kernel(TopkParams p, const float *fu, const float *fi,
                                                              const float *bu, const float *bi, int mode, float *scores) {
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
    const bool scale = mode == PMF_PREDICT_SCALE;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int qq = q0 + row;
        if (qq < p.nq && it < p.n_items) {
            float s = acc[r];
            if (bu) s = scale ? s * (bu[p.users[qq]] * bcol) : bu[p.users[qq]] + bcol + s;
            scores[(int64_t)qq * p.n_items + it] = s;
        }
    }
}

// fp64 contexts: plain dot products (parity mode, not a throughput path)
__global__ void topk_scores_f64_kernel(TopkParams p, const double *fu, const double *fi, const double *bu,
                                       const double *bi, int mode, double *scores) {
    const int64_t total = (int64_t)p.nq * p.n_items;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e / p.n_items);
        const int64_t it = e % p.n_items;
        const int user = p.users[q];
        const double *a = fu + (int64_t)user * p.kpad, *b = fi + it * p.kpad;
        double s = 0.0;
        for (int k = 0; k < p.K; ++k) s = fma(a[k], b[k], s);
        if (bu) s = mode == PMF_PREDICT_SCALE ? s * (bu[user] * bi[it]) : bu[user] + bi[it] + s;
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


// ---------------------------------------------------------------------------
// fused score + select, fp32, Kpad <= 128, k <= 64: scores never touch HBM
// ---------------------------------------------------------------------------
// One wavefront owns 32 query users for a whole segment of the item range; the four wavefronts of a block
// walk that range together.  A wave's A operand -- its 32 users' factor rows -- is loaded ONCE into KH
// registers per lane (lane (i = l & 31, h = l >> 5) holds the 16-byte pieces 2 q + h, q = 0 .. KH / 4 - 1, of
// user i: the k order of the MFMA steps is a permutation, the same one for both operands) and stays there.
// The B operand is staged by the block: ST item rows per stage are fetched with fully coalesced 16-byte
// loads (a thread's piece is contiguous with its neighbours': 2 lines per 256-byte row, where a lane-per-row
// load touches 64 lines per instruction and left the kernel bound by the texture addresser, 0.58 of the MFMA
// peak), parked in registers while the previous stage is multiplied, and written to one of two LDS buffers
// with an odd row pitch (KH / 2 + 1 pieces), from which every wave reads its MFMA layout with conflict-free
// ds_read_b128 -- each item row leaves the L2 once per 128 users instead of once per 32.  One barrier per
// stage.  Per 32-item tile a wave runs KH v_mfma_f32_32x32x2_f32 and compares its 16 scores with the users'
// current k-th best.  Almost every tile ends there (a user's list changes about k ln(N / k) times over N
// items); a score that does qualify is inserted into that user's sorted list in LDS by the lanes of the
// wave in parallel (lane e holds entry e: one compare, one ballot, one shift), in ascending item order, so
// ties keep the lower item id.  With few query users the item range is cut into segments (gridDim.y) so
// that the chip is filled; a merge kernel then picks the k best of the segments' lists.
#define TOPK_NEG_INF (-__builtin_inff())
#ifndef PRIO_SLICE
#define PRIO_SLICE 16      // stages between priority rotations (tools/probe_topk_stamps.py sweeps it: 1 .. 64 are equivalent)
#endif

#ifdef PMF_TOPK_STAMPS
// DIAGNOSTIC BUILD ONLY (tools/probe_topk_stamps.py compiles this file with -DPMF_TOPK_STAMPS into a library of its
// own; the product library has no stamp): per (wavefront, user tile) the 100 MHz real-time stamps of the scan's begin
// and end, the block's grid size and its XCC id -- the residency timeline of a launch -- and the shader-clock cycles
// the wave spent ranking candidates and waiting at the stage barriers, with the number of candidates.
__device__ long long g_topk_stamps[16384 * 8];
extern "C" int pmf_debug_topk_stamps(long long *host, int n_waves) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_topk_stamps), (size_t)n_waves * 8 * sizeof(long long));
}
#endif

// v_writelane_b32: (value, lane, old) -> old with `value` in lane `lane` (both wave-uniform).  This clang has no
// __builtin for it; a declaration carrying the intrinsic's name binds to it.
extern "C" __device__ int pmf_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane");

// list keys of the fused kernel: (score bits mapped so that unsigned order is the scores' order) << 32 | ~item.  An empty
// entry is -inf's key with a zero low word: below every candidate's key (no item id has ~item == 0), and its score
// reads back as -inf without a special case.
#define TOPK_EMPTY_KEY 0x007fffff00000000ull
__device__ __forceinline__ unsigned topk_key_score_bits(unsigned hi) {   // (integer throughout: stays on the scalar unit)
    return hi ^ ((int)hi < 0 ? 0x80000000u : 0xffffffffu);
}
__device__ __forceinline__ float topk_key_score(unsigned hi) { return __uint_as_float(topk_key_score_bits(hi)); }

template <int KH>
struct TopkStage {
    static constexpr int ST = KH == 8 ? 64 : 32;      // item rows per stage (one MFMA tile; two where a tile is too short a load)
    static constexpr int PR = KH / 2;                 // 16-byte pieces per (padded) factor row
    static constexpr int PQ = PR + 1;                 // LDS row pitch in pieces; odd -> the 16 lanes of a b128 group hit 16 bank quads
    static constexpr int LPT = ST * PR / 256;         // pieces fetched per thread per stage
    static constexpr size_t buffer_bytes = (size_t)ST * PQ * 16;   // the kernel takes one or two of these (nbuf)
};

// (waves per SIMD the register allocation must leave room for: four for the plain K <= 64 scan, whose lists leave room for
//  four blocks per CU; the bias / scale modes and K > 64 need more registers than that)
template <int KH, int MODE, int NBUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KH <= 32 ? (MODE == 0 ? 4 : 3) : 2))) void topk_fused_kernel(TopkParams p, const float *fu, const float *fi, const float *cu,
                                                         const float *ci, int k, int64_t seg_items, int nseg,
                                                         float *cand_val, int32_t *cand_idx, int32_t *out_items,
                                                         double *out_scores) {
    using S = TopkStage<KH>;
    constexpr int ST = S::ST, PR = S::PR, PQ = S::PQ, LPT = S::LPT;
    constexpr int nbuf = NBUF;                        // stage buffers: two, or one where that keeps another block on the CU
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // (readfirstlane: the wave index is uniform, and the compiler only knows it once told -- what hangs off q0 then
    //  branches on SCC instead of masking exec)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int h = lane >> 5, c = lane & 31;
    f32x4 *stage = reinterpret_cast<f32x4 *>(smem_raw);                                          // [nbuf][ST][PQ]
    // [32][k] list entries of this wave's users, best first: 64-bit keys (insert() below)
    unsigned long long *le = reinterpret_cast<unsigned long long *>(smem_raw + nbuf * S::buffer_bytes) + (size_t)wave * 32 * k;
    const int seg = blockIdx.y;
    // (item indices are ints in here -- the host sends more than 2^31 - 64 items to the two-phase path: 64-bit compares
    //  have no scalar form, and every instruction of this loop is paid for)
    const int i_begin = (int)((int64_t)seg * seg_items);
    const int i_end = (int)((int64_t)i_begin + seg_items < p.n_items ? (int64_t)i_begin + seg_items : p.n_items);
    const int kpad = p.kpad;

    // PERSISTENT blocks: the grid is sized to what the chip holds at once (launch_topk_fused_mode) and a block walks
    // the 128-user tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    const int n_user_tiles = (p.nq + 127) / 128;
    for (int ut = blockIdx.x; ut < n_user_tiles; ut += gridDim.x) {
    const int q0 = (ut * 4 + wave) * 32;
    const bool active = q0 < p.nq;                     // a wave without users still stages item rows
#ifdef PMF_TOPK_STAMPS
    const long long st_begin = __builtin_amdgcn_s_memrealtime();
    long long st_drain = 0, st_barrier = 0, st_cand = 0, st_tiles = 0;
#define STAMP_BARRIER() do { const long long t_ = __builtin_amdgcn_s_memtime(); __syncthreads(); st_barrier += __builtin_amdgcn_s_memtime() - t_; } while (0)
#else
#define STAMP_BARRIER() __syncthreads()
#endif

    // A operand: this lane's pieces of user (q0 + c)'s row, resident for the whole scan
    const int user = (q0 + c < p.nq) ? p.users[q0 + c] : -1;
    float a[KH];
#pragma unroll
    for (int t = 0; t < KH; t += 4) {
        const int kk = 2 * t + 4 * h;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (user >= 0 && kk < kpad) v = *reinterpret_cast<const float4 *>(fu + (int64_t)user * kpad + kk);
        a[t] = v.x; a[t + 1] = v.y; a[t + 2] = v.z; a[t + 3] = v.w;
    }
    // per accumulator register r: the user of row (r & 3) + 8 (r >> 2) + 4 h, its bias / scale, its k-th best
    float tau[16], ucst[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        tau[r] = qq < p.nq ? TOPK_NEG_INF : __builtin_inff();   // rows past the last query never qualify
        ucst[r] = MODE == PMF_PREDICT_SCALE ? 1.f : 0.f;
        if (MODE != 0 && qq < p.nq) ucst[r] = cu[p.users[qq]];
    }
    for (int e = lane; e < 32 * k; e += 64) le[e] = TOPK_EMPTY_KEY;

    // Loads are unconditional (no branch, nothing for the loop's wait counters to merge): a row past the
    // segment's end re-reads the last row (its scores are never ranked), a piece past Kpad re-reads the
    // last piece (the A operand is zero there).
    // (The scan is bound by the fp32 matrix pipe, and on gfx950 the fp32 MFMA shares the SIMD's fp32 lanes with the VALU:
    //  every vector instruction of ANY resident wave costs the pipe its issue cycles -- profiles/r03_topk_wave_stamps.jsonl:
    //  9.5 k cycles per tile for four waves whose MFMAs need 8.2 k.  Hence the running pointers here instead of 64-bit
    //  row-address arithmetic per stage, and the scalar masks in the threshold test below.)
    f32x4 g[LPT];
    const int last_piece = kpad / 4 - 1;
    // full stages: one scalar base for the block, advanced per stage, plus this thread's constant 32-bit byte offsets
    // (global_load with an SGPR base: no per-thread pointer arithmetic in the loop)
    const char *next_stage = reinterpret_cast<const char *>(fi + (int64_t)i_begin * kpad);
    const unsigned stage_bytes = (unsigned)ST * kpad * 4;
    unsigned goff[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        const int idx = j * 256 + (int)threadIdx.x;
        const int r = idx / PR, pc = idx % PR;
        goff[j] = ((unsigned)r * kpad + 4u * (pc < last_piece ? pc : last_piece)) * 4u;
    }
    auto fetch_full = [&]() __attribute__((always_inline)) {          // (out of line, g[] would live in scratch)
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            unsigned off = goff[j];
            asm volatile("" : "+v"(off));             // (keeps the zero-extension in the loop: SGPR base + 32-bit VGPR offset)
            g[j] = *reinterpret_cast<const f32x4 *>(next_stage + off);
        }
        next_stage += stage_bytes;
    };
    auto fetch_tail = [&](int i0) __attribute__((always_inline)) {    // the segment's last, partial stage
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int idx = j * 256 + (int)threadIdx.x;
            const int r = idx / PR, pc = idx % PR;
            const int it = i0 + r < i_end ? i0 + r : i_end - 1;
            g[j] = *reinterpret_cast<const f32x4 *>(fi + (int64_t)it * kpad + 4 * (pc < last_piece ? pc : last_piece));
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int idx = j * 256 + (int)threadIdx.x;
            stage[(size_t)buf * ST * PQ + (idx / PR) * PQ + idx % PR] = g[j];
        }
    };
    // insert the candidate (score bits vb, ~item) into the list of local user `ul`; returns that list's new k-th best value
    // (its bits).  Lane e holds entry e.  An entry is one 64-bit key, (score bits made monotone) << 32 | ~item: a bigger key
    // ranks first, so the candidate's place is ONE 64-bit compare and a ballot, its key is built with scalar instructions
    // and dropped into its lane with v_writelane, and the entries behind it move down one lane by DPP under an EXEC mask
    // the scalar unit derives from the ballot.  EVERY instruction counts here, scalar ones as much as vector ones: beside
    // waves that stream MFMAs an instruction of a ranking wave costs the SIMD about five cycles whatever unit runs it
    // (profiles/r03_topk_instruction_cost.log) -- hence the one asm block (no EXEC save / restore pairs, no branch
    // around the masked store, lane select straight from an SGPR, both words stored by one ds_write2).
    const unsigned long long kmask = k >= 64 ? ~0ull : (1ull << k) - 1;   // the lanes that hold list entries
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    const unsigned le_lane = (unsigned)(size_t)(lds_u64 *)(le + lane);    // LDS byte address of this lane's entry of list 0
    auto insert = [&](int ul, unsigned vb, unsigned nitem) __attribute__((always_inline)) -> unsigned {
        if (vb == 0x80000000u) vb = 0u;                              // -0.0 ranks as +0.0 does
        const unsigned ov = vb ^ ((unsigned)((int)vb >> 31) | 0x80000000u);
        const unsigned long long ck = ((unsigned long long)ov << 32) | nitem;
        // (every lane reads: past entry k - 1 it is the next lists' entries, or the padding behind the last one; they are
        //  masked out of the ballot and never written back -- no divergent region, the k-th entry is read as a scalar)
        const unsigned long long e = le[ul * k + lane];
        unsigned lo = (unsigned)e, hi = (unsigned)(e >> 32);
        const int pos = __popcll(__builtin_amdgcn_ballot_w64(e > ck) & kmask);   // entries that rank before the candidate
        if (pos < k) {                                               // (an earlier candidate of this tile may have filled the list)
            // lanes >= pos take the entry of the lane below (lane pos has no active source: it keeps its entry until the
            // writelane), the candidate goes into lane pos, lanes < k store.  EXEC is all ones here (no divergent region).
            asm volatile("s_mov_b32 m0, %[pos]\n\t"               // (one SGPR per vector instruction: the lane select rides in M0,
                         "s_mov_b64 exec, %[from]\n\t"             //  which nothing else in this kernel uses -- tests/test_abi_cpu.py)
                         "s_nop 4\n\t"
                         "v_mov_b32_dpp %[lo], %[lo] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %[hi], %[hi] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_mov_b64 exec, %[kmask]\n\t"
                         "v_writelane_b32 %[lo], %[nitem], m0\n\t"
                         "v_writelane_b32 %[hi], %[ov], m0\n\t"
                         "ds_write2_b32 %[addr], %[lo], %[hi] offset1:1\n\t"
                         "s_mov_b64 exec, -1"
                         : [lo] "+v"(lo), [hi] "+v"(hi)
                         : [from] "s"(~0ull << pos), [kmask] "s"(kmask), [nitem] "s"(nitem), [ov] "s"(ov), [pos] "s"(pos),
                           [addr] "v"(le_lane + (unsigned)(ul * k) * 8u)
                         : "memory");
        }
        const unsigned kth = (unsigned)__builtin_amdgcn_readlane((int)hi, k - 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return topk_key_score_bits(kth);
    };

    // the KH MFMA steps of one 32-item tile, B operand from the staged rows
    auto mfma_all = [&](const f32x4 *rows, f32x16 &acc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const f32x4 *mine = rows + c * PQ + h;
#pragma unroll
        for (int t = 0; t < KH; t += 4) {
            const f32x4 b = mine[t / 2];             // piece 2 (t / 4) + h of item (i0 + c)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t + 1], b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t + 2], b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t + 3], b.w, acc, 0, 0, 0);
        }
    };
    // dot products -> scores (predict's arithmetic order); TAIL: the stage holds rows past the segment's end
    auto scores = [&](auto TAIL, int i0, f32x16 &acc) __attribute__((always_inline)) {
        if (MODE == 0) return;
        const int it = i0 + c;
        float ccst = MODE == PMF_PREDICT_SCALE ? 1.f : 0.f;
        if (!decltype(TAIL)::value || it < i_end) ccst = ci[it];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == PMF_PREDICT_BIAS) acc[r] = ucst[r] + ccst + acc[r];  // b_u + b_i + dot
            if (MODE == PMF_PREDICT_SCALE) acc[r] = acc[r] * (ucst[r] * ccst);
        }
    };
    // Does any score of the tile reach its user's list?  Almost never (a list changes ~k ln(N / k) times in N items), so
    // the test itself is what a tile pays: 16 v_cmpx narrow EXEC to the lanes whose rows are ALL below their thresholds
    // -- no mask per row, no OR chain (16 scalar instructions the compare-and-collect form needs) -- and two scalar
    // instructions see whether that is every lane.  (Two asm blocks: an asm statement takes 30 operands.  Each leaves
    // EXEC as it found it.  The s_nop pair is the MFMA -> VALU read distance the compiler cannot see into an asm for.)
    auto all_below = [&](const f32x16 &acc, unsigned long long &m1, unsigned long long &m2) __attribute__((always_inline)) -> bool {
        asm volatile("s_nop 15\n\ts_nop 3\n\t"
                     "v_cmpx_lt_f32 vcc, %[a0], %[t0]\n\tv_cmpx_lt_f32 vcc, %[a1], %[t1]\n\t"
                     "v_cmpx_lt_f32 vcc, %[a2], %[t2]\n\tv_cmpx_lt_f32 vcc, %[a3], %[t3]\n\t"
                     "v_cmpx_lt_f32 vcc, %[a4], %[t4]\n\tv_cmpx_lt_f32 vcc, %[a5], %[t5]\n\t"
                     "v_cmpx_lt_f32 vcc, %[a6], %[t6]\n\tv_cmpx_lt_f32 vcc, %[a7], %[t7]\n\t"
                     "s_mov_b64 %[m], exec\n\ts_mov_b64 exec, -1"
                     : [m] "=s"(m1)
                     : [a0] "v"(acc[0]), [t0] "v"(tau[0]), [a1] "v"(acc[1]), [t1] "v"(tau[1]), [a2] "v"(acc[2]), [t2] "v"(tau[2]),
                       [a3] "v"(acc[3]), [t3] "v"(tau[3]), [a4] "v"(acc[4]), [t4] "v"(tau[4]), [a5] "v"(acc[5]), [t5] "v"(tau[5]),
                       [a6] "v"(acc[6]), [t6] "v"(tau[6]), [a7] "v"(acc[7]), [t7] "v"(tau[7])
                     : "vcc");
        asm volatile("v_cmpx_lt_f32 vcc, %[a0], %[t0]\n\tv_cmpx_lt_f32 vcc, %[a1], %[t1]\n\t"
                     "v_cmpx_lt_f32 vcc, %[a2], %[t2]\n\tv_cmpx_lt_f32 vcc, %[a3], %[t3]\n\t"
                     "v_cmpx_lt_f32 vcc, %[a4], %[t4]\n\tv_cmpx_lt_f32 vcc, %[a5], %[t5]\n\t"
                     "v_cmpx_lt_f32 vcc, %[a6], %[t6]\n\tv_cmpx_lt_f32 vcc, %[a7], %[t7]\n\t"
                     "s_mov_b64 %[m], exec\n\ts_mov_b64 exec, -1"
                     : [m] "=s"(m2)
                     : [a0] "v"(acc[8]), [t0] "v"(tau[8]), [a1] "v"(acc[9]), [t1] "v"(tau[9]), [a2] "v"(acc[10]), [t2] "v"(tau[10]),
                       [a3] "v"(acc[11]), [t3] "v"(tau[11]), [a4] "v"(acc[12]), [t4] "v"(tau[12]), [a5] "v"(acc[13]), [t5] "v"(tau[13]),
                       [a6] "v"(acc[14]), [t6] "v"(tau[14]), [a7] "v"(acc[15]), [t7] "v"(tau[15])
                     : "vcc");
        return (m1 & m2) == ~0ull;
    };
    // the 16 threshold compares with their lane masks left in SGPRs (v_cmp -> s[..]): the scan for candidates is scalar
    auto thresholds = [&](int r0, int r1, const f32x16 &acc, unsigned long long (&mk)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r >= r0 && r < r1) mk[r] = __builtin_amdgcn_ballot_w64(acc[r] >= tau[r]);
    };
    // the tile's candidates go into their lists in ascending item order (lane order within the tile), so equal scores
    // keep the lower item id in front
    auto drain = [&](int r0, int r1, int i0, const f32x16 &acc, const unsigned long long (&mk)[16], unsigned long long okm)
                     __attribute__((always_inline)) {
        const unsigned ni0 = ~(unsigned)i0;           // ~(i0 + j) == ~i0 - j
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < r0 || r >= r1) continue;
            unsigned long long m = mk[r] & okm;
            while (m) {
                const int L = __builtin_ctzll(m);
                asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(L));
#ifdef PMF_TOPK_STAMPS
                ++st_cand;
#endif
                const int hh = L >> 5;
                // (a row past the last query user never gets here: its threshold is +inf -- and if an infinite score did
                //  bring it here, the list it lands in exists in LDS and is never handed over)
                const int ul = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const unsigned vb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(acc[r]), L);
#ifdef PMF_TOPK_PAD_SCALAR   // (diagnostic: what an extra scalar / vector instruction per candidate costs the launch --
#pragma unroll                // profiles/r03_topk_instruction_cost.log: about five cycles of SIMD time each, either kind)
                for (int pad = 0; pad < PMF_TOPK_PAD_SCALAR; ++pad) { int d_; asm volatile("s_mov_b32 %0, 0" : "=s"(d_)); }
#endif
#ifdef PMF_TOPK_PAD_VECTOR
#pragma unroll
                for (int pad = 0; pad < PMF_TOPK_PAD_VECTOR; ++pad) { int d_; asm volatile("v_mov_b32 %0, 0" : "=v"(d_)); }
#endif
                // (an earlier candidate of this tile may have raised the list's threshold past v: insert() re-checks)
                const unsigned nt = insert(ul, vb, ni0 - (unsigned)(L & 31));
                // tau[r] = nt in the half-wave (h == hh) that holds this user's row: one v_mov under a scalar EXEC
                asm volatile("s_mov_b64 exec, %[half]\n\t"
                             "v_mov_b32 %[t], %[nt]\n\t"
                             "s_mov_b64 exec, -1"
                             : [t] "+v"(tau[r])
                             : [half] "s"(0xffffffffull << (32 * hh)), [nt] "s"(nt));
            }
        }
    };
    // Time-sliced priority.  Left alone, the SIMD's arbitration favours the same resident wave for a whole scan: the
    // four blocks of a CU then finish one after the other (9.4 .. 16.5 ms for the same work) and the last quarter of
    // the launch runs at 3, 2, 1 blocks per CU.  Rotating s_setprio over the SIMD's wave slots every PRIO_SLICE stages
    // makes them finish together (12.3 .. 12.7 ms): profiles/r03_topk_wave_stamps.jsonl.
    const int slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3;   // HW_ID.WAVE_ID: this wave's slot on its SIMD
    auto slice_priority = [&](int stage_no) __attribute__((always_inline)) {
        switch ((slot + stage_no / PRIO_SLICE) & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
    auto rotate_priority = [&](int stage_no) __attribute__((always_inline)) {
        if ((stage_no & (PRIO_SLICE - 1)) == 0) slice_priority(stage_no);
    };
    auto tile = [&](auto TAIL, int i0, const f32x4 *rows, int stage_no) __attribute__((always_inline)) {
        f32x16 acc;
        unsigned long long mk[16], okm;
        mfma_all(rows, acc);
        scores(TAIL, i0, acc);
        // which half of the rows holds a candidate (the v_cmpx blocks' lane masks); a partial tile takes both
        unsigned long long m1 = 0ull, m2 = 0ull;
        if (!decltype(TAIL)::value && all_below(acc, m1, m2)) return;
        okm = decltype(TAIL)::value ? __builtin_amdgcn_ballot_w64(i0 + c < i_end) : ~0ull;
#ifdef PMF_TOPK_STAMPS
        const long long t_ = __builtin_amdgcn_s_memtime();
        ++st_tiles;
#endif
#ifdef PMF_TOPK_RANK_PRIORITY   // (experiment: rank at the top priority, then step back to the time slice's -- no gain)
        __builtin_amdgcn_s_setprio(3);
#endif
        if (m1 != ~0ull) {
            thresholds(0, 8, acc, mk);
            drain(0, 8, i0, acc, mk, okm);
        }
        if (m2 != ~0ull) {
            thresholds(8, 16, acc, mk);
            drain(8, 16, i0, acc, mk, okm);
        }
#ifdef PMF_TOPK_RANK_PRIORITY
        slice_priority(stage_no);
#endif
#ifdef PMF_TOPK_STAMPS
        st_drain += __builtin_amdgcn_s_memtime() - t_;
#endif
    };
    auto tiles = [&](auto TAIL, int i0, const f32x4 *rows, int stage_no) __attribute__((always_inline)) {
        if (!active) return;                          // a wave without users only stages item rows
        tile(TAIL, i0, rows, stage_no);
        if (ST == 64 && (!decltype(TAIL)::value || i0 + 32 < i_end)) tile(TAIL, i0 + 32, rows + 32 * PQ, stage_no);
    };
    // the stage in g[] becomes the one the block multiplies next (every wave runs the same stages: the barriers)
    int buf = 0;
    auto publish = [&]() __attribute__((always_inline)) {
        if (nbuf == 2) {
            stash(buf ^ 1);                           // last read there: the stage before this one, behind the barrier
            STAMP_BARRIER();
            buf ^= 1;
        } else {                                      // one buffer (long lists: the LDS saved keeps another block resident)
            STAMP_BARRIER();                          // every wave has read this stage
            stash(0);
            STAMP_BARRIER();
        }
    };
    // The loop proper has no case to tell apart: it runs while the stage after the current one is a full one; the
    // segment's last full stage and its partial one (rows past the end re-read the last row and are never ranked) follow.
    const int n_full = (i_end - i_begin) / ST;
    const bool has_tail = (i_end - i_begin) % ST != 0;
    const std::integral_constant<bool, false> FULL;
    const std::integral_constant<bool, true> PARTIAL;
    int stage_no = 0, i0 = i_begin;
    if (n_full > 0) fetch_full(); else fetch_tail(i_begin);            // (segments are never empty)
    stash(0);
    __syncthreads();
    const int n_steady = n_full > 0 ? n_full - 1 : 0;
    for (int s0 = 0; s0 < n_steady; s0 += PRIO_SLICE) {               // (one priority slice per trip: nothing to test per stage)
        slice_priority(s0);
        const int s1 = s0 + PRIO_SLICE < n_steady ? s0 + PRIO_SLICE : n_steady;
        for (int st = s0; st < s1; ++st, i0 += ST) {
            fetch_full();
            tiles(FULL, i0, stage + (size_t)buf * ST * PQ, st);
            publish();
        }
    }
    stage_no = n_steady;
    if (n_full > 0) {
        rotate_priority(stage_no);
        if (has_tail) fetch_tail(i0 + ST);
        tiles(FULL, i0, stage + (size_t)buf * ST * PQ, stage_no);
        if (has_tail) publish();
        ++stage_no;
        i0 += ST;
    }
    if (has_tail) tiles(PARTIAL, i0, stage + (size_t)buf * ST * PQ, stage_no);
    STAMP_BARRIER();                                  // the next user tile's first stage goes where this one is read
    // hand the lists over: final result when the item range was not segmented, else this segment's candidates
    for (int e = lane; e < 32 * k; e += 64) {
        const int ul = e / k, t = e % k;
        const int qq = q0 + ul;
        if (qq >= p.nq) continue;
        const unsigned long long ent = le[e];
        const float v = topk_key_score((unsigned)(ent >> 32));
        const int idx = (unsigned)ent == 0u ? 0x7fffffff : (int)~(unsigned)ent;
        if (nseg == 1) {
            out_items[(int64_t)qq * k + t] = idx == 0x7fffffff ? -1 : idx;
            out_scores[(int64_t)qq * k + t] = idx == 0x7fffffff ? 0.0 : (double)v;
        } else {
            cand_val[((int64_t)qq * nseg + seg) * k + t] = v;
            cand_idx[((int64_t)qq * nseg + seg) * k + t] = idx;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the lists are re-initialised for the next user tile
    __builtin_amdgcn_wave_barrier();
#ifdef PMF_TOPK_STAMPS
    if (lane == 0 && seg == 0) {
        long long *o = g_topk_stamps + (size_t)((ut * 4 + wave) & 16383) * 8;
        o[0] = st_begin;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = gridDim.x;
        o[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
        o[4] = st_drain;
        o[5] = st_barrier;
        o[6] = st_cand;
        o[7] = st_tiles;
    }
#endif
    }   // user tiles
    __builtin_amdgcn_s_setprio(0);
}

// k best of a user's nseg * k segment candidates, (value desc, item id asc); one wavefront per user
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *cand_val, const int32_t *cand_idx, int nq, int n_cand,
                                                         int k, int32_t *out_items, double *out_scores) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const float *cv = cand_val + (int64_t)q * n_cand;
    const int32_t *ci = cand_idx + (int64_t)q * n_cand;
    bool have_prev = false;
    float pv = 0.f;
    int pi = -1;
    for (int t = 0; t < k; ++t) {
        bool found = false;
        float bv = 0.f;
        int bidx = 0x7fffffff;
        for (int e = lane; e < n_cand; e += 64) {
            const float v = cv[e];
            const int i = ci[e];
            if (i == 0x7fffffff) continue;   // an unused list slot
            if (have_prev && !ranks_before(pv, pi, v, i)) continue;
            if (!found || ranks_before(v, i, bv, bidx)) {
                bv = v;
                bidx = i;
                found = true;
            }
        }
        wave_best(bv, bidx, found);
        if (lane == 0) {
            out_items[(int64_t)q * k + t] = found ? bidx : -1;
            out_scores[(int64_t)q * k + t] = found ? (double)bv : 0.0;
        }
        if (!found) {
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

template <int KH, int MODE>
static hipError_t launch_topk_fused_mode(pmf_ctx *ctx, const TopkParams &p, dim3 grid, size_t list_bytes, const float *fu,
                                         const float *fi, const float *cu, const float *ci, int k, int64_t seg_items,
                                         int nseg, float *cand_val, int32_t *cand_idx, int32_t *out_items,
                                         double *out_scores) {
    // One stage buffer or two.  Two (one barrier per stage) is the faster loop at equal residency, but the lists take
    // 1 KB of LDS per k and block: from k = 23 at K = 64 the second buffer costs the CU a resident block
    // (profiles/r03_topk_long_lists.jsonl: 34.3 -> 30.8 ms at k = 24, 72.9 -> 49.3 ms at k = 64).  Take one buffer where
    // it keeps more blocks on the CU.  (ctx->topk_stage_buffers pins it: PMF_TOPK_STAGE_BUFFERS, for the probes.)
    const void *fn2 = (const void *)topk_fused_kernel<KH, MODE, 2>, *fn1 = (const void *)topk_fused_kernel<KH, MODE, 1>;
    const size_t smem2 = 2 * TopkStage<KH>::buffer_bytes + list_bytes, smem1 = TopkStage<KH>::buffer_bytes + list_bytes;
    if (smem2 > (64u << 10)) {   // long lists at K > 64: past the default dynamic-LDS limit
        hipError_t e = hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        if (e == hipSuccess) e = hipFuncSetAttribute(fn1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        if (e != hipSuccess) return e;
    }
    int per_cu2 = 0, per_cu1 = 0, dev = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, fn2, 256, smem2);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu1, fn1, 256, smem1);
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    int nbuf = per_cu1 > per_cu2 ? 1 : 2;
    if (ctx->topk_stage_buffers == 1 || ctx->topk_stage_buffers == 2) nbuf = ctx->topk_stage_buffers;
    const size_t smem = nbuf == 2 ? smem2 : smem1;
    // persistent in x: at most as many blocks as are resident at once (the occupancy query x CUs), each walking its
    // share of the 128-user tiles; with a segmented item range (few users) every (tile, segment) keeps its own block
    if (grid.y == 1) {
        unsigned resident = (unsigned)std::max(1, nbuf == 2 ? per_cu2 : per_cu1) * (unsigned)std::max(1, cus);
        if (ctx->topk_max_blocks > 0) resident = std::min(resident, (unsigned)ctx->topk_max_blocks);   // (tests: few blocks, many tiles each)
        if (grid.x > resident) grid.x = resident;
    }
    if (nbuf == 2)
        hipLaunchKernelGGL((topk_fused_kernel<KH, MODE, 2>), grid, dim3(256), smem, ctx->stream, p, fu, fi, cu, ci, k, seg_items,
                           nseg, cand_val, cand_idx, out_items, out_scores);
    else
        hipLaunchKernelGGL((topk_fused_kernel<KH, MODE, 1>), grid, dim3(256), smem, ctx->stream, p, fu, fi, cu, ci, k, seg_items,
                           nseg, cand_val, cand_idx, out_items, out_scores);
    return hipSuccess;
}

template <int KH>
static hipError_t launch_topk_fused(pmf_ctx *ctx, const TopkParams &p, dim3 grid, size_t list_bytes, int mode,
                                    const float *fu, const float *fi, const float *cu, const float *ci, int k,
                                    int64_t seg_items, int nseg, float *cand_val, int32_t *cand_idx, int32_t *out_items,
                                    double *out_scores) {
    if (mode == PMF_PREDICT_BIAS)
        return launch_topk_fused_mode<KH, PMF_PREDICT_BIAS>(ctx, p, grid, list_bytes, fu, fi, cu, ci, k, seg_items, nseg,
                                                            cand_val, cand_idx, out_items, out_scores);
    if (mode == PMF_PREDICT_SCALE)
        return launch_topk_fused_mode<KH, PMF_PREDICT_SCALE>(ctx, p, grid, list_bytes, fu, fi, cu, ci, k, seg_items, nseg,
                                                             cand_val, cand_idx, out_items, out_scores);
    return launch_topk_fused_mode<KH, 0>(ctx, p, grid, list_bytes, fu, fi, cu, ci, k, seg_items, nseg, cand_val, cand_idx,
                                         out_items, out_scores);
}

// fp32, Kpad <= 128, k <= 64
static int run_topk_fused(pmf_ctx *ctx, int64_t n_query, const int32_t *user_ids, int k, int mode, const float *cu,
                          const float *ci, int32_t *out_items, double *out_scores) {
    const int64_t I = ctx->rows[PMF_SIDE_ITEM];
    const float *fu = (const float *)ctx->arr[PMF_SIDE_USER][PMF_ARR_FACTOR];
    const float *fi = (const float *)ctx->arr[PMF_SIDE_ITEM][PMF_ARR_FACTOR];
    const int64_t Q = std::min<int64_t>(n_query, 1 << 20);   // query users per launch
    // few users: cut the item range so that the grid still has a few thousand wavefronts
    const int64_t waves = (Q + 31) / 32;
    int nseg = (int)std::min<int64_t>(64, std::max<int64_t>(1, 4096 / waves));
    nseg = (int)std::min<int64_t>(nseg, std::max<int64_t>(1, I / 2048));
    int64_t seg_items = ((I + nseg - 1) / nseg + 31) / 32 * 32;
    nseg = (int)((I + seg_items - 1) / seg_items);
    const size_t n_cand = nseg > 1 ? (size_t)nseg * k : 0;
    const size_t id_bytes = ((size_t)Q * sizeof(int32_t) + 15) / 16 * 16;
    const size_t out_i = ((size_t)Q * k * sizeof(int32_t) + 15) / 16 * 16, out_s = (size_t)Q * k * sizeof(double);
    const size_t cv_bytes = ((size_t)Q * n_cand * sizeof(float) + 15) / 16 * 16;
    int rc;
    if ((rc = pmf_ensure_scratch(ctx, id_bytes + out_s + out_i + 2 * cv_bytes + 64))) return rc;
    char *base = (char *)ctx->d_scratch;
    double *d_out_scores = (double *)base;
    int32_t *d_out_items = (int32_t *)(base + out_s);
    int32_t *d_users = (int32_t *)(base + out_s + out_i);
    float *d_cv = (float *)(base + out_s + out_i + id_bytes);
    int32_t *d_ci = (int32_t *)(base + out_s + out_i + id_bytes + cv_bytes);
    const size_t list_bytes = (size_t)8 * 32 * k * sizeof(float) + 512;   // + 64 entries: insert() reads a full wave's width
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
            dim3 grid((unsigned)((nq + 127) / 128), (unsigned)nseg);
            const int kh = ctx->kpad <= 16 ? 8 : ctx->kpad <= 32 ? 16 : ctx->kpad <= 64 ? 32 : 64;
            hipError_t le;
            switch (kh) {
                case 8: le = launch_topk_fused<8>(ctx, p, grid, list_bytes, mode, fu, fi, cu, ci, k, seg_items, nseg, d_cv, d_ci, d_out_items, d_out_scores); break;
                case 16: le = launch_topk_fused<16>(ctx, p, grid, list_bytes, mode, fu, fi, cu, ci, k, seg_items, nseg, d_cv, d_ci, d_out_items, d_out_scores); break;
                case 32: le = launch_topk_fused<32>(ctx, p, grid, list_bytes, mode, fu, fi, cu, ci, k, seg_items, nseg, d_cv, d_ci, d_out_items, d_out_scores); break;
                default: le = launch_topk_fused<64>(ctx, p, grid, list_bytes, mode, fu, fi, cu, ci, k, seg_items, nseg, d_cv, d_ci, d_out_items, d_out_scores); break;
            }
            PMF_HIP_CHECK(le);
            if (nseg > 1)
                hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, ctx->stream, d_cv, d_ci, nq,
                                   (int)n_cand, k, d_out_items, d_out_scores);
        }
        PMF_HIP_CHECK(hipGetLastError());
        PMF_HIP_CHECK(hipMemcpyAsync(out_items + at * k, d_out_items, (size_t)nq * k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        PMF_HIP_CHECK(hipMemcpyAsync(out_scores + at * k, d_out_scores, (size_t)nq * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PMF_OK;
}

template <typename T>
static int run_topk(pmf_ctx *ctx, int64_t n_query, const int32_t *user_ids, int k, int use_bias,
                    int32_t *out_items, double *out_scores) {
    const int64_t I = ctx->rows[PMF_SIDE_ITEM];
    int rc;
    if ((rc = pmf_require_array(ctx, PMF_SIDE_USER, PMF_ARR_FACTOR, "pmf_topk_items"))) return rc;
    if ((rc = pmf_require_array(ctx, PMF_SIDE_ITEM, PMF_ARR_FACTOR, "pmf_topk_items"))) return rc;
    PMF_REQUIRE(use_bias == 0 || use_bias == PMF_PREDICT_BIAS || use_bias == PMF_PREDICT_SCALE, PMF_EINVAL,
                "pmf_topk_items: use_bias must be 0, PMF_PREDICT_BIAS or PMF_PREDICT_SCALE (got %d)", use_bias);
    const int carr = use_bias == PMF_PREDICT_SCALE ? PMF_ARR_SCALE : PMF_ARR_BIAS;
    if (use_bias) {
        if ((rc = pmf_require_array(ctx, PMF_SIDE_USER, carr, "pmf_topk_items"))) return rc;
        if ((rc = pmf_require_array(ctx, PMF_SIDE_ITEM, carr, "pmf_topk_items"))) return rc;
    }
    const T *fu = (const T *)ctx->arr[PMF_SIDE_USER][PMF_ARR_FACTOR];
    const T *fi = (const T *)ctx->arr[PMF_SIDE_ITEM][PMF_ARR_FACTOR];
    const T *bu = use_bias ? (const T *)ctx->arr[PMF_SIDE_USER][carr] : nullptr;
    const T *bi = use_bias ? (const T *)ctx->arr[PMF_SIDE_ITEM][carr] : nullptr;
    if constexpr (std::is_same<T, float>::value) {
        if (ctx->kpad <= 128 && k <= 64 && I <= (int64_t)INT32_MAX - 64 && !ctx->topk_two_phase)   // (the fused scan counts items in ints)
            return run_topk_fused(ctx, n_query, user_ids, k, use_bias, bu, bi, out_items, out_scores);
    }
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
                hipLaunchKernelGGL(topk_scores_f32_kernel, grid, dim3(256), 0, ctx->stream, p, fu, fi, bu, bi, use_bias, d_scores);
            } else {
                const int64_t total = (int64_t)nq * I;
                hipLaunchKernelGGL(topk_scores_f64_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)),
                                   dim3(256), 0, ctx->stream, p, fu, fi, bu, bi, use_bias, d_scores);
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

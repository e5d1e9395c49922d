"""Control experiment for the top-k roofline: what does a BARE fp32-MFMA loop
deliver on this box, at the top-k kernel's occupancy and LDS footprint, and at which clock?

The fused top-k runs at about 0.72 of the fp32-MFMA peak (157.3 TFLOP/s = 256 FLOP / cycle / CU at 2.4 GHz).  Round 2
read that as "MFMA issue-saturated at the 1.62 GHz the chip holds under this load" from GRBM_GUI_ACTIVE; this program
checks the claim from the other side.  It builds (hipcc, on the box) a small library of loops on random operands:

    bare32      v_mfma_f32_32x32x2_f32 back to back, operands in registers          (the guide's 155 TF case)
    lds32       the same, with the B operand re-read from LDS by ds_read_b128 per 32-item tile, as top-k does
    cmp32       lds32 + top-k's per-tile epilogue (16 compares against per-row thresholds, one ballot)
    stage32     cmp32 + top-k's staging (two coalesced global loads per thread and tile, LDS double buffer, one barrier per tile)
    bare16 / lds16   the same contraction on v_mfma_f32_16x16x4_f32 (four 16 x 16 tiles per step)

each at 1 wave per SIMD and at top-k's residency (256-thread blocks with 27.6 KB of LDS: 5 per CU), timed with
hipEvents over launches of >= 20 ms after 2 s of warm-up, with the in-kernel clock stamped per wave:
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6; median over waves).

    python tools/probe_mfma_clock.py            # one JSON line per variant
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One "tile" = the contraction top-k does per 32 users x 32 items at K = 64: 2 * 32 * 32 * 64 flop.
//   SHAPE 32: 32 x v_mfma_f32_32x32x2_f32 into one f32x16 accumulator
//   SHAPE 16: 64 x v_mfma_f32_16x16x4_f32 into four f32x4 accumulators (16 k-steps x 4 tiles)
// MODE 0 bare (operands in registers), 1 B operand from LDS (8 ds_read_b128 per tile), 2 = 1 + compare epilogue,
// 3 = 2 + top-k's staging: per tile two coalesced 16-byte global loads per thread (a 25.6 MB table), parked in registers
// during the tile, written to the other LDS buffer, one __syncthreads per tile.
template <int SHAPE, int MODE>
__global__ __launch_bounds__(256) void mfma_loop(const float *seed, float *sink, long long *stamps, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    f32x4 *lds = reinterpret_cast<f32x4 *>(smem);                 // [32 rows][17 pieces]: top-k's stage pitch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float a[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) a[t] = seed[(gid * 32 + t) & 0xFFFFF];
    for (int e = threadIdx.x; e < 32 * 17; e += 256)
        lds[e] = f32x4{seed[(e * 4) & 0xFFFFF], seed[(e * 4 + 1) & 0xFFFFF], seed[(e * 4 + 2) & 0xFFFFF], seed[(e * 4 + 3) & 0xFFFFF]};
    __syncthreads();
    float tau[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tau[r] = 1e30f;
    float total = 0.f;
    const f32x4 *mine = lds + (lane & 31) * 17 + (lane >> 5);
    const f32x4 *table = reinterpret_cast<const f32x4 *>(seed);      // MODE 3: 1M floats = 16,384 rows of 16 pieces (L2 resident)
    int buf = 0;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 g0, g1;
        if (MODE == 3) {
            const int row0 = ((it * 32 + (int)blockIdx.x * 7) & 16383);
            g0 = table[((row0 + (threadIdx.x >> 4)) & 16383) * 16 + (threadIdx.x & 15)];
            g1 = table[((row0 + 16 + (threadIdx.x >> 4)) & 16383) * 16 + (threadIdx.x & 15)];
            mine = lds + buf * 32 * 17 + (lane & 31) * 17 + (lane >> 5);
        }
        f32x4 b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE >= 1) b[q] = mine[2 * q];
            else b[q] = f32x4{a[4 * q], a[(4 * q + 5) & 31], a[(4 * q + 10) & 31], a[(4 * q + 15) & 31]};
        }
        if constexpr (SHAPE == 32) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q], b[q][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], b[q][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], b[q][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], b[q][3], acc, 0, 0, 0);
            }
            if (MODE >= 2) {
                bool any = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) any |= acc[r] >= tau[r];
                if (__ballot(any) != 0ull) total += acc[0];
            } else {
                total += acc[0] + acc[15];
            }
        } else {
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(4 * q + 2 * e + u) & 31], b[q][(2 * e + u) & 3], acc[u], 0, 0, 0);
            if (MODE >= 2) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) any |= acc[u][r] >= tau[4 * u + r];
                if (__ballot(any) != 0ull) total += acc[0][0];
            } else {
                total += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
            }
        }
        a[0] += 1e-30f * total;    // a loop-carried operand: nothing can be hoisted; numerically a no-op
        if (MODE == 3) {
            f32x4 *dst = lds + (buf ^ 1) * 32 * 17;
            dst[(threadIdx.x >> 4) * 17 + (threadIdx.x & 15)] = g0;
            dst[(16 + (threadIdx.x >> 4)) * 17 + (threadIdx.x & 15)] = g1;
            __syncthreads();
            buf ^= 1;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
    }
    if (total == 123.456f) sink[gid] = total;
}

template <int SHAPE, int MODE>
static int run(int blocks, int lds_bytes, int iters, int reps, const float *seed, float *sink, long long *stamps, float *ms) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    if (lds_bytes > 64 * 1024) hipFuncSetAttribute((const void *)mfma_loop<SHAPE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_loop<SHAPE, MODE>), dim3(blocks), dim3(256), lds_bytes, 0, seed, sink, stamps, iters);
    hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    hipEventElapsedTime(ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return (int)e;
}

extern "C" int probe(int shape, int mode, int blocks, int lds_bytes, int iters, int reps, const float *seed, float *sink,
                     long long *stamps, float *ms) {
    if (shape == 32 && mode == 0) return run<32, 0>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    if (shape == 32 && mode == 1) return run<32, 1>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    if (shape == 32 && mode == 2) return run<32, 2>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    if (shape == 32 && mode == 3) return run<32, 3>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    if (shape == 16 && mode == 3) return run<16, 3>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    if (shape == 16 && mode == 0) return run<16, 0>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    if (shape == 16 && mode == 1) return run<16, 1>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    if (shape == 16 && mode == 2) return run<16, 2>(blocks, lds_bytes, iters, reps, seed, sink, stamps, ms);
    return -1;
}
// A one-wave observer that runs BESIDE another kernel (its own stream) and samples the shader clock: every
// `interval` ticks of the 100 MHz real-time counter it stores (s_memtime, s_memrealtime).
__global__ void clock_observer(long long *out, int samples, long long interval) {
    if (threadIdx.x != 0) return;
    for (int s = 0; s < samples; ++s) {
        const long long r = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - r < interval) __builtin_amdgcn_s_sleep(32);
        out[2 * s] = __builtin_amdgcn_s_memtime();
        out[2 * s + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
static hipStream_t g_obs = nullptr;
extern "C" int observe_begin(long long *out, int samples, long long interval) {
    if (!g_obs && hipStreamCreateWithFlags(&g_obs, hipStreamNonBlocking) != hipSuccess) return -1;
    hipLaunchKernelGGL(clock_observer, dim3(1), dim3(64), 0, g_obs, out, samples, interval);
    return (int)hipGetLastError();
}
extern "C" int observe_end() { return (int)hipStreamSynchronize(g_obs); }
extern "C" void *dev_alloc(size_t bytes) { void *p = nullptr; return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr; }
extern "C" int h2d(void *d, const void *h, size_t n) { return (int)hipMemcpy(d, h, n, hipMemcpyHostToDevice); }
extern "C" int d2h(void *h, const void *d, size_t n) { return (int)hipMemcpy(h, d, n, hipMemcpyDeviceToHost); }
'''


def main():
    work = tempfile.mkdtemp(prefix="mfma_probe_")
    src, lib = os.path.join(work, "probe.hip"), os.path.join(work, "libprobe.so")
    open(src, "w").write(SRC)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
                    "-fno-slp-vectorize", "-Wno-unused-value", "-Wno-unused-result", src, "-o", lib], check=True)
    L = C.CDLL(lib)
    L.dev_alloc.restype = C.c_void_p
    L.dev_alloc.argtypes = [C.c_size_t]
    L.h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.probe.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    rng = np.random.default_rng(0)
    seed_h = rng.standard_normal(1 << 20).astype(np.float32)
    max_blocks = 256 * 8
    seed = L.dev_alloc(seed_h.nbytes)
    sink = L.dev_alloc(max_blocks * 256 * 4)
    stamps = L.dev_alloc(max_blocks * 4 * 16)
    assert seed and sink and stamps
    L.h2d(seed, seed_h.ctypes.data, seed_h.nbytes)
    flop_per_tile = 2.0 * 32 * 32 * 64
    topk_lds = 2 * 32 * 17 * 16 + 8 * 32 * 10 * 4          # stage buffers + four waves' k = 10 lists: 27,648 B -> 5 blocks per CU
    variants = [("bare32", 32, 0), ("lds32", 32, 1), ("cmp32", 32, 2), ("stage32", 32, 3), ("bare16", 16, 0), ("lds16", 16, 1),
                ("cmp16", 16, 2), ("stage16", 16, 3)]
    for name, shape, mode in variants:
        for label, blocks, lds in (("1 wave per SIMD", 256, 2 * 32 * 17 * 16), ("4 blocks per CU", 256 * 4, topk_lds),
                                   ("top-k's LDS footprint: 5 blocks per CU", 256 * 5, topk_lds)):
            waves_per_simd = blocks // 256
            iters = 12000 // waves_per_simd                   # ~ 25 M MFMA-pipe cycles per SIMD and launch
            ms = C.c_float(0.0)
            rc = L.probe(shape, mode, blocks, lds, iters, 100, seed, sink, stamps, C.byref(ms))     # warm-up: >= 1 s
            assert rc == 0, rc
            rc = L.probe(shape, mode, blocks, lds, iters, 20, seed, sink, stamps, C.byref(ms))
            assert rc == 0, rc
            st = np.empty(blocks * 4 * 2, dtype=np.int64)
            L.d2h(st.ctypes.data, stamps, st.nbytes)
            st = st.reshape(-1, 2)
            clock_ghz = float(np.median(st[:, 0] / np.maximum(st[:, 1], 1))) * 0.1
            per_launch_ms = ms.value / 20
            tf = blocks * 4 * iters * flop_per_tile / (per_launch_ms * 1e-3) / 1e12
            cycles_per_tile = float(np.median(st[:, 0])) / iters / waves_per_simd
            print(json.dumps({"variant": name, "occupancy": label, "mfma": f"v_mfma_f32_{'32x32x2' if shape == 32 else '16x16x4'}_f32",
                              "launch_ms": round(per_launch_ms, 3), "TFLOP/s": round(tf, 1), "frac_of_157.3": round(tf / 157.3, 3),
                              "in_kernel_clock_GHz": round(clock_ghz, 3),
                              "SIMD_cycles_per_tile": round(cycles_per_tile, 1), "ideal_cycles_per_tile": 2048}), flush=True)


    # ---- the real top-k launch, with the clock sampled beside it ------------------------------------------------------
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "prob-matrix-factorization_amd")]
    import time
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ITEM, USER
    L.observe_begin.argtypes = [C.c_void_p, C.c_int, C.c_longlong]
    U, I, K, k, Q = 1_000_000, 100_000, 64, 10, 262_144
    ctx = pmf_hip.Context(U, I, K, dtype="f32")
    ctx.set_array(USER, ARR_FACTOR, rng.gamma(0.5, 1.0, (U, K)))
    ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(0.5, 1.0, (I, K)))
    users = rng.permutation(U)[:Q].astype(np.int32)
    samples, interval = 1600, 25_000                           # 0.25 ms apart, 400 ms in all
    obs = L.dev_alloc(samples * 16)

    def observed(label, work):
        """clock samples while `work()` runs (after 1.5 s of the same work as warm-up)"""
        t0 = time.time()
        while time.time() - t0 < 1.5:
            work()
        assert L.observe_begin(obs, samples, interval) == 0
        t0 = time.time()
        n = 0
        while time.time() - t0 < 0.38:
            work()
            n += 1
        assert L.observe_end() == 0
        st = np.empty(samples * 2, dtype=np.int64)
        L.d2h(st.ctypes.data, obs, st.nbytes)
        st = st.reshape(-1, 2)
        clk = np.diff(st[:, 0]) / np.maximum(np.diff(st[:, 1]), 1) * 0.1
        busy = clk[: int(0.36 / 0.00025)]                     # the samples taken while the work was running
        return {"work": label, "calls": n, "clock_GHz_median": round(float(np.median(busy)), 3),
                "clock_GHz_p10": round(float(np.percentile(busy, 10)), 3), "clock_GHz_p90": round(float(np.percentile(busy, 90)), 3)}

    ms = C.c_float(0.0)
    print(json.dumps(observed("idle (observer alone)", lambda: time.sleep(0.01))), flush=True)
    print(json.dumps(observed("bare32, 5 blocks per CU", lambda: L.probe(32, 0, 1280, topk_lds, 2400, 2, seed, sink, stamps, C.byref(ms)))), flush=True)
    print(json.dumps(observed("cmp32, 5 blocks per CU", lambda: L.probe(32, 2, 1280, topk_lds, 2400, 2, seed, sink, stamps, C.byref(ms)))), flush=True)
    ctx.prof_enable(True)
    ctx.prof_reset()
    res = observed("pmf_topk_items, 262,144 users x 100,000 items, K = 64, k = 10", lambda: ctx.topk_items(users, k))
    tms, tn = ctx.prof_get()["topk"]
    res["kernel_ms"] = round(tms / max(tn, 1), 3)
    res["TFLOP/s"] = round(2.0 * Q * I * K / (tms / max(tn, 1) * 1e-3) / 1e12, 1)
    res["mfma_cycles_needed_per_SIMD_M"] = round(Q / 32 * (I / 32) * 32 * 64 / 1024 / 1e6, 1)
    res["cycles_elapsed_at_observed_clock_M"] = round(res["clock_GHz_median"] * 1e9 * tms / max(tn, 1) * 1e-3 / 1e6, 1)
    print(json.dumps(res), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()

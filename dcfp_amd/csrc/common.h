// common.h — shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/dcfp_hip.h"

#define DCFP_WAVE 64

// launch check: a kernel launch reports configuration errors through hipGetLastError.
#define DCFP_RETURN_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        return (e__ == hipSuccess) ? DCFP_OK : (int)e__; \
    } while (0)

// the BatchNorm-backward-sums side output of the fan-in dgrad (Igemm2Params::red_*), as dcfp_igemm2_run takes it
struct Igemm2Red {
    const float* x;
    const unsigned long long* mask;
    const float* mean;
    float* part;
};

static inline hipStream_t dcfp_s(dcfp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool dcfp_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// compute units of the current device (256 on MI355X; also the answer when no device is present)
static inline int dcfp_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        n = (hipGetDevice(&dev) == hipSuccess &&
             hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over a 256-thread block in a fixed (deterministic) order; result valid in thread 0.
__device__ __forceinline__ float block_sum_256(float v, float* smem /* >= 4 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    __syncthreads();
    return r;
}

// nn.BatchNorm2d's training-mode bookkeeping for channel c, done by the thread that finalises it (bn.hip, syncbn_p2p.hip)
__device__ __forceinline__ void running_update(const DcfpBnRunning& r, int c, float mean, float var, float n) {
    if (r.running_mean) {
        // every product and sum rounded on its own (hipcc contracts a*b + c*d into an FMA either way round, and the
        // __f*_rn intrinsics do not stop it): the kernels that end in this update - per-rank statistics, the SyncBN
        // combine, the peer-to-peer exchange - must leave the same bits
#pragma clang fp contract(off)
        const float unb = n / fmaxf(n - 1.0f, 1.0f);
        const float keep = 1.0f - r.momentum;
        const float a = r.running_mean[c] * keep, b = r.momentum * mean;
        r.running_mean[c] = a + b;
        const float vu = var * unb;
        const float d = r.running_var[c] * keep, e = r.momentum * vu;
        r.running_var[c] = d + e;
    }
    if (c == 0 && r.num_batches_tracked) r.num_batches_tracked[0] += 1;
}

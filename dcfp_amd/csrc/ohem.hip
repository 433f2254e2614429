// ohem.hip — OHEM threshold on the device (loss/ohem.py:20-48 find_threshold, :51-78
// generate_new_target): the k-th smallest ground-truth probability of the valid positions of the
// 1/8-zoomed grid by an exact radix select, threshold = max(thresh, kth) written to a device
// scalar, and the kept-pixel mask `gt_prob <= threshold` that the fused CE kernels consume.
// The reference copies the full-resolution probabilities to the host and runs np.partition; here
// nothing leaves the device and the training step has no host synchronisation for it.
//
// The zoomed grid is small (N*H*W/64 = 131 072 positions at 4 x 1024 x 2048), so ONE workgroup of
// 1024 threads runs the whole select: latency-bound, four 8-bit passes over L2-resident data.
#include "common.h"

namespace {

constexpr int kSelThreads = 1024;

// order-preserving map float -> uint32 (handles negative values too; probabilities are >= 0)
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void __launch_bounds__(kSelThreads)
ohem_threshold_kernel(const float* __restrict__ pred, const int* __restrict__ lab, long long n,
                      int ignore_label, float thresh, long long min_kept, float* __restrict__ out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long s_cnt;
    __shared__ unsigned s_prefix, s_k;
    const int tid = threadIdx.x;
    // ---- number of valid positions (ohem.py:32-35)
    if (tid == 0) s_cnt = 0ull;
    __syncthreads();
    unsigned long long c = 0;
    for (long long i = tid; i < n; i += kSelThreads) c += (lab[i] != ignore_label);
    // wave reduction, then one atomic per wave (integer: exact in any order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((tid & 63) == 0) atomicAdd(&s_cnt, c);
    __syncthreads();
    const long long num_valid = (long long)s_cnt;
    if (min_kept >= num_valid) {              // ohem.py:36-37
        if (tid == 0) out[0] = 1.0f;
        return;
    }
    if (min_kept <= 0) {                      // ohem.py:41-42: threshold stays self.thresh
        if (tid == 0) out[0] = thresh;
        return;
    }
    // ---- k-th smallest (0-based rank k) of the valid values: ohem.py:43-45
    if (tid == 0) {
        s_prefix = 0u;
        s_k = (unsigned)((num_valid < min_kept ? num_valid : min_kept) - 1);
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (long long i = tid; i < n; i += kSelThreads) {
            if (lab[i] == ignore_label) continue;
            const unsigned key = f2key(pred[i]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        if (tid == 0) {                        // 256-entry scan: negligible next to the passes
            unsigned k = s_k, b = 0;
            for (; b < 255; ++b) {
                if (k < hist[b]) break;
                k -= hist[b];
            }
            s_k = k;
            s_prefix = prefix | (b << shift);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float kth = key2f(s_prefix);
        out[0] = kth > thresh ? kth : thresh;  // ohem.py:46-47
    }
}

// keep[i] = gt_prob[i] <= *threshold  (ohem.py:69 kept_flag; ignored pixels carry gt_prob = 1 and are
// dropped by the label test inside the CE kernels)
__global__ void __launch_bounds__(256)
ohem_keep_mask_kernel(const float* __restrict__ gtp, const float* __restrict__ thr, long long n,
                      unsigned char* __restrict__ keep) {
    const float t = thr[0];
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(gtp)[i];
        uchar4 m;
        m.x = v.x <= t; m.y = v.y <= t; m.z = v.z <= t; m.w = v.w <= t;
        reinterpret_cast<uchar4*>(keep)[i] = m;
    }
    if (blockIdx.x == 0) {
        const long long i = (n4 << 2) + threadIdx.x;
        if (i < n) keep[i] = gtp[i] <= t;
    }
}

}  // namespace

extern "C" int dcfp_ohem_threshold_f32(const float* pred8, const int32_t* lab8, int64_t n, int ignore_label,
                                       float thresh, int64_t min_kept, float* threshold,
                                       dcfp_stream_t stream) {
    if (!threshold || n < 0 || (n > 0 && (!pred8 || !lab8))) return DCFP_E_BADDESC;
    if (n >= (1LL << 32)) return DCFP_E_UNSUPPORTED;          // 32-bit ranks in the select
    hipLaunchKernelGGL(ohem_threshold_kernel, dim3(1), dim3(kSelThreads), 0, dcfp_s(stream), pred8, lab8,
                       (long long)n, ignore_label, thresh, (long long)min_kept, threshold);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_ohem_keep_mask_u8(const float* gt_prob, const float* threshold, int64_t n,
                                      uint8_t* keep, dcfp_stream_t stream) {
    if (!gt_prob || !threshold || !keep || n < 0) return DCFP_E_BADDESC;
    if (n == 0) return DCFP_OK;
    if (!dcfp_aligned16(gt_prob) || (reinterpret_cast<uintptr_t>(keep) & 3u)) return DCFP_E_BADDESC;
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ohem_keep_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, dcfp_s(stream), gt_prob,
                       threshold, (long long)n, keep);
    DCFP_RETURN_LAUNCH();
}

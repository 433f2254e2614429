// igemm2_common.h — types shared by the implicit-GEMM conv kernels (conv_igemm2.hip and the persistent
// 1x1 LDS-DMA kernel in conv_igemm2p.hip).
#pragma once
#include "common.h"
#include <type_traits>

// (global scope: passed between translation units)
struct Igemm2Params {
    const float* in;
    const float* wp;    // permuted weights [T][CkP][Mpad]
    const float* bias;
    const float* scale;     // inference epilogue: y = act(acc*scale[m] + shift[m] (+ residual))
    const float* shift;
    const float* residual;  // same layout as out
    int relu;
    float* out;
    long long in_nstride, out_nstride;
    int N, M, Mpad, Ck, CkP;
    int Hi, Wi, Ho, Wo, P, tiles_per_img, tiles_n_total, tiles_m;
    int sn, sd, off0, offstep;
    int accumulate, vec_store;
    int tile2d;         // igemm2_dma_kernel<9, true>: 0, or log2(columns) of a 2-D pixel tile (8 rows x 32 or 16 x 16)
    int Hc, Wc, tiles_per_phase;   // strided dgrad (sd > 1): coarse grid ceil(Ho/sd) x ceil(Wo/sd) of one output phase
    int in_pitch;       // row pitch (floats) of the B source; > Wi: rows carry a zero tail of in_pitch - Wi floats, so a
                        // shifted 16-byte quad may hang over either row end and still read zeros (no border handling)
    int zfold;          // strided dgrad of a 1x1 conv: only phase `zfold - 1` has a tap; its tiles also write the zeros
    const float* fan_src;   // igemm2_dma1p_kernel, gradient fan-in of a residual block: out = result + fan_src where the bit of
    const unsigned long long* fan_mask;   // fan_mask is set (layout of dcfp_bn_apply_relu_mask_f32), fan_src laid out as out
    long long wp_nstride;   // igemm2_dma1p_kernel: floats between the weight copies of consecutive images (0: shared) - the
                            // batched GEMM of conv_winograd.hip, where "image" xi has its own transformed filter
    int nt_store;       // igemm2_dma1p_kernel, plain epilogue: cache policy of the output stores (A/B knob DCFP_IGEMM_NT: 0 default,
                        // 1 nt, 2 nt + sc1) - the conv output is next read by a BatchNorm pass streaming it from HBM
    int tapskip;        // 9-tap LDS-DMA kernels: K-steps of kernel rows that lie wholly in the padding for a tile are skipped
    float* stat_part;   // nullable: per-(pixel-tile, wave-column) row statistics [slot][M][2] = (mean, M2)
    // igemm2_dma1p_kernel with fan_src (nullable as a group): the fan-in's OUTPUT is the gradient `g_out` arriving at the
    // previous residual block's output; with these the epilogue also emits that block's BatchNorm-backward sums
    //   red_part[slot][M][2] = (sum g, sum g * (red_x - red_mean[m]))  over the slot's 128 pixels,  g = g_out * red_mask bit
    // (slot = pixel tile * 2 + wave column, as stat_part) - bn_bwd_reduce without re-reading g_out
    const float* red_x;                    // that block's bn3 input, laid out as out
    const unsigned long long* red_mask;    // that block's ReLU bit mask (layout of fan_mask)
    const float* red_mean;                 // [M]
    float* red_part;
};

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;      // default K-step depth
constexpr unsigned kOob = 0x80000000u;

// compile-time loop: every index is a constant expression, so register arrays stay in registers
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// One (input channel c, output channel m) pair of the fused Winograd kernels' transformed filters (conv_winograd2.hip):
// Ug[cb][mb][xi][wm][lane][kk] = (G g G^T)[xi] of g = w[m * sAm + c * sAc + tap] (flip: 8 - tap), c = 8 cb + 2 kk + lane / 32,
// m = 64 mb + 32 wm + lane % 32; zeros past M / Ck.  Shared by wino_filter2_kernel (one conv per launch) and the
// multi-tensor refresh (permute_weights_multi_kernel, conv_igemm2.hip: every kept copy of a model in one launch).
__device__ __forceinline__ void wino_filter2_pair(const float* __restrict__ w, int sAm, int sAc, int flip, int M, int Ck,
                                                  int Mpad, long long idx, float* __restrict__ Ug) {
    const int mblocks = Mpad / 64;
    const int c = (int)(idx / Mpad), m = (int)(idx - (long long)c * Mpad);
    float g[9];
    if (m < M && c < Ck) {
        const float* src = w + (long long)m * sAm + (long long)c * sAc;
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t] = src[flip ? 8 - t : t];
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t] = 0.f;
    }
    float r[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g0 = g[k], g1 = g[3 + k], g2 = g[6 + k];
        r[0][k] = g0;
        r[1][k] = 0.5f * ((g0 + g2) + g1);
        r[2][k] = 0.5f * ((g0 + g2) - g1);
        r[3][k] = g2;
    }
    const int cb = c >> 3, cl = c & 7, mb = m >> 6, ml = m & 63;
    float* dst = Ug + ((((long long)(cb * mblocks + mb) * 16) * 2 + (ml >> 5)) * 64 + (cl & 1) * 32 + (ml & 31)) * 4 + (cl >> 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = r[i][0], b = r[i][1], cc = r[i][2];
        dst[(4 * i + 0) * 512] = a;
        dst[(4 * i + 1) * 512] = 0.5f * ((a + cc) + b);
        dst[(4 * i + 2) * 512] = 0.5f * ((a + cc) - b);
        dst[(4 * i + 3) * 512] = cc;
    }
}

template <int T>
struct Frag;
template <>
struct Frag<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
};
template <>
struct Frag<2> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[2]) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        f[0] = v.x; f[1] = v.y;
    }
};
template <>
struct Frag<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[1]) { f[0] = p[0]; }
};

}  // namespace

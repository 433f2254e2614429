// conv_gemv.hip — 1x1 convs on a 1 x 1 map: the ASPP image-pool branch (networks/tools/aspp.py:56-61: AdaptiveAvgPool2d(1)
// then a 2048 -> 256 1x1 conv on N x 2048 x 1 x 1).  A matrix-vector product per image; the 256 x 256 MFMA tiles of the
// general kernels spent 159 us per launch on it (0.03 TF: VERDICT r02 weak 9).  Plain VALU kernels, fixed summation order.
#include "common.h"

namespace {

// y[n][m] = sum_c w[m][c] x[n][c] (+ bias[m]).  One wave per output channel m, lanes stride over c; up to 8 images per pass.
__global__ void __launch_bounds__(256) gemv_fwd_kernel(const float* __restrict__ x, long long x_nstride,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, long long y_nstride, int N, int M, int C) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* wr = w + (long long)m * C;
    for (int n0 = 0; n0 < N; n0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float wv = wr[c];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n0 + j < N) acc[j] = fmaf(wv, x[(long long)(n0 + j) * x_nstride + c], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (n0 + j >= N) break;
            const float s = wave_sum(acc[j]);
            if (lane == 0) y[(long long)(n0 + j) * y_nstride + m] = s + (bias ? bias[m] : 0.f);
        }
    }
}

// dx[n][c] (+)= sum_m w[m][c] dy[n][m].  Block = 64 input channels x 4 waves, wave q sums the output channels m = q, q + 4, ...;
// the four partial sums are added in wave order through LDS.  Up to 8 images per pass.
__global__ void __launch_bounds__(256) gemv_dgrad_kernel(const float* __restrict__ dy, long long dy_nstride,
                                                         const float* __restrict__ w, float* __restrict__ dx,
                                                         long long dx_nstride, int N, int M, int C, int accumulate) {
    __shared__ float part[4][8][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    for (int n0 = 0; n0 < N; n0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (c < C)
            for (int m = q; m < M; m += 4) {
                const float wv = w[(long long)m * C + c];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (n0 + j < N) acc[j] = fmaf(wv, dy[(long long)(n0 + j) * dy_nstride + m], acc[j]);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) part[q][j][lane] = acc[j];
        __syncthreads();
        if (q == 0 && c < C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (n0 + j >= N) break;
                const float s = (part[0][j][lane] + part[1][j][lane]) + (part[2][j][lane] + part[3][j][lane]);
                float* o = dx + (long long)(n0 + j) * dx_nstride + c;
                *o = accumulate ? *o + s : s;
            }
        }
        __syncthreads();
    }
}

// dw[m][c] = sum_n dy[n][m] x[n][c]
__global__ void __launch_bounds__(256) gemv_wgrad_kernel(const float* __restrict__ dy, long long dy_nstride,
                                                         const float* __restrict__ x, long long x_nstride,
                                                         float* __restrict__ dw, int N, int M, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * C) return;
    const int m = (int)(i / C), c = (int)(i - (long long)m * C);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(dy[(long long)n * dy_nstride + m], x[(long long)n * x_nstride + c], s);
    dw[i] = s;
}

}  // namespace

// a 1x1 stride-1 conv whose input (= output) map is a single pixel
bool dcfp_gemv_shape(const DcfpConvDesc* d) {
    static const bool on = [] { const char* e = getenv("DCFP_CONV_GEMV"); return !e || atoi(e) != 0; }();   // =0: the general kernels
    return on && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == 1 && d->W == 1;
}

int dcfp_gemv_fwd(const DcfpConvDesc* d, const float* x, const float* w, const float* bias, float* y, long long y_nstride,
                  hipStream_t stream) {
    hipLaunchKernelGGL(gemv_fwd_kernel, dim3((unsigned)((d->Cout + 3) / 4)), dim3(256), 0, stream, x, (long long)d->Cin, w, bias,
                       y, y_nstride ? y_nstride : (long long)d->Cout, d->N, d->Cout, d->Cin);
    DCFP_RETURN_LAUNCH();
}

int dcfp_gemv_dgrad(const DcfpConvDesc* d, const float* dy, long long dy_nstride, const float* w, float* dx, int accumulate,
                    hipStream_t stream) {
    hipLaunchKernelGGL(gemv_dgrad_kernel, dim3((unsigned)((d->Cin + 63) / 64)), dim3(256), 0, stream, dy,
                       dy_nstride ? dy_nstride : (long long)d->Cout, w, dx, (long long)d->Cin, d->N, d->Cout, d->Cin, accumulate);
    DCFP_RETURN_LAUNCH();
}

int dcfp_gemv_wgrad(const DcfpConvDesc* d, const float* dy, long long dy_nstride, const float* x, float* dw, hipStream_t stream) {
    const long long mc = (long long)d->Cout * d->Cin;
    hipLaunchKernelGGL(gemv_wgrad_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, stream, dy,
                       dy_nstride ? dy_nstride : (long long)d->Cout, x, (long long)d->Cin, dw, d->N, d->Cout, d->Cin);
    DCFP_RETURN_LAUNCH();
}

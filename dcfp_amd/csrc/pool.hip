// pool.hip — HBM-bound element-wise / pooling kernels on the hot path:
//   MaxPool2d(3,2,1)            networks/backbone/resnet.py:100,149
//   AdaptiveAvgPool2d(1)        networks/tools/aspp.py:56        (dcfp_rowsum_f32)
//   bilinear 1x1 -> HxW         networks/tools/aspp.py:76        (dcfp_broadcast_hw_f32)
//   gradient fan-in add         autograd of resnet.py:55, aspp.py:70-77
//   Dropout2d with host mask    networks/deeplabv3.py:40
#include "common.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;

// One thread per output element; -inf padding; first maximum wins; NaN propagates
// (ATen max_pool2d semantics).  argmax = h*W + w inside the input plane.
__global__ void __launch_bounds__(kThreads)
maxpool3x3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                        int32_t* __restrict__ argmax, long long total, int H, int W, int Hout,
                        int Wout) {
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int ow = (int)(idx % Wout);
        const long long t = idx / Wout;
        const int oh = (int)(t % Hout);
        const long long plane = t / Hout;
        const float* xp = x + plane * (long long)H * W;
        const int h0 = oh * 2 - 1, w0 = ow * 2 - 1;
        const int hs = h0 < 0 ? 0 : h0, ws = w0 < 0 ? 0 : w0;
        const int he = h0 + 3 > H ? H : h0 + 3, we = w0 + 3 > W ? W : w0 + 3;
        float best = -INFINITY;
        int bi = hs * W + ws;
        for (int h = hs; h < he; ++h) {
            for (int w = ws; w < we; ++w) {
                const float v = xp[h * W + w];
                if (v > best || v != v) {
                    best = v;
                    bi = h * W + w;
                }
            }
        }
        y[idx] = best;
        argmax[idx] = bi;
    }
}

// Four consecutive outputs of one output row per thread (W % 8 == 0, so Wout = W/2 is a multiple of 4):
// per input row two 16-byte loads (columns 2*ow0 .. 2*ow0+7) and one scalar (column 2*ow0-1, the previous
// thread's last element: an L1 hit), 16-byte stores of values and argmax.  One block = one output row
// segment, so consecutive blocks re-read the shared input row while it is still in L2.  The compare
// sequence per output is the scalar kernel's (rows top to bottom, columns left to right, first maximum
// wins, NaN propagates): results are bit-identical.
__global__ void __launch_bounds__(kThreads)
maxpool3x3s2_fwd_vec4_kernel(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ argmax,
                             int H, int W, int Hout, int Wout, int segs) {
    const long long rowid = blockIdx.x / segs;              // plane * Hout + oh
    const int seg = blockIdx.x - (int)(rowid * segs);
    const long long plane = rowid / Hout;
    const int oh = (int)(rowid - plane * Hout);
    const int ow0 = 4 * (seg * kThreads + threadIdx.x);
    if (ow0 >= Wout) return;
    const float* xp = x + plane * (long long)H * W;
    const int h0 = 2 * oh - 1, c0 = 2 * ow0;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int hs = h0 < 0 ? 0 : h0;
    int bi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bi[j] = hs * W + ((c0 + 2 * j - 1) < 0 ? 0 : (c0 + 2 * j - 1));
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int h = h0 + r;
        if (h < 0 || h >= H) continue;                      // block-uniform
        const float* row = xp + (long long)h * W;
        const float4 a = *reinterpret_cast<const float4*>(row + c0);
        const float4 b = *reinterpret_cast<const float4*>(row + c0 + 4);
        const float v[9] = {c0 > 0 ? row[c0 - 1] : 0.f, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k == 0 && j == 0 && c0 == 0) continue;  // left padding column
                const float t = v[2 * j + k];
                if (t > best[j] || t != t) { best[j] = t; bi[j] = h * W + c0 + 2 * j + k - 1; }
            }
        }
    }
    const long long o = rowid * Wout + ow0;
    *reinterpret_cast<float4*>(y + o) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<int4*>(argmax + o) = make_int4(bi[0], bi[1], bi[2], bi[3]);
}

// Gather form of the backward: every input pixel looks at the (at most 4) windows
// that contain it — deterministic, no atomics.
__global__ void __launch_bounds__(kThreads)
maxpool3x3s2_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ argmax,
                        float* __restrict__ dx, long long total, int H, int W, int Hout,
                        int Wout) {
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int w = (int)(idx % W);
        const long long t = idx / W;
        const int h = (int)(t % H);
        const long long plane = t / H;
        const float* dyp = dy + plane * (long long)Hout * Wout;
        const int32_t* ap = argmax + plane * (long long)Hout * Wout;
        const int me = h * W + w;
        // windows oh with 2*oh-1 <= h <= 2*oh+1
        const int oh_lo = h / 2, oh_hi = (h + 1) / 2;
        const int ow_lo = w / 2, ow_hi = (w + 1) / 2;
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            if (oh >= Hout) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                if (ow >= Wout) continue;
                if (ap[oh * Wout + ow] == me) acc += dyp[oh * Wout + ow];
            }
        }
        dx[idx] = acc;
    }
}

// Same gather, four input columns per thread and no integer divisions: block = one input row segment.
// Columns w0..w0+3 (w0 % 4 == 0) lie in the windows ow = w0/2 .. w0/2+2, row h in oh = h/2 .. (h+1)/2;
// contributions are added in the same (oh, ow) order as above, so both kernels agree bit for bit.
__global__ void __launch_bounds__(kThreads)
maxpool3x3s2_bwd_vec4_kernel(const float* __restrict__ dy, const int32_t* __restrict__ argmax,
                             float* __restrict__ dx, int H, int W, int Hout, int Wout, int segs) {
    const long long rowid = blockIdx.x / segs;              // plane * H + h
    const int seg = blockIdx.x - (int)(rowid * segs);
    const long long plane = rowid / H;
    const int h = (int)(rowid - plane * H);
    const int w0 = 4 * (seg * kThreads + threadIdx.x);
    if (w0 >= W) return;
    const float* dyp = dy + plane * (long long)Hout * Wout;
    const int32_t* ap = argmax + plane * (long long)Hout * Wout;
    const int oh_lo = h / 2, oh_hi = (h + 1) / 2, c0 = w0 / 2;
    const int me = h * W + w0;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        if (oh >= Hout) continue;
        const int base = oh * Wout + c0;
        const int i0 = ap[base];
        const float g0 = dyp[base];
        int i1 = -1, i2 = -1;
        float g1 = 0.f, g2 = 0.f;
        if (c0 + 1 < Wout) { i1 = ap[base + 1]; g1 = dyp[base + 1]; }
        if (c0 + 2 < Wout) { i2 = ap[base + 2]; g2 = dyp[base + 2]; }
        if (i0 == me) a0 += g0;
        if (i0 == me + 1) a1 += g0;
        if (i1 == me + 1) a1 += g1;
        if (i1 == me + 2) a2 += g1;
        if (i1 == me + 3) a3 += g1;
        if (i2 == me + 3) a3 += g2;
    }
    *reinterpret_cast<float4*>(dx + rowid * W + w0) = make_float4(a0, a1, a2, a3);
}

// y[row] = scale * sum_i x[row, i]; one block per (n,c) row.
__global__ void __launch_bounds__(kThreads)
rowsum_kernel(const float* __restrict__ x, long long x_nstride, float* __restrict__ y, float scale,
              int C, int HW, int vec) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int n = row / C, c = row - n * C;
    const float* xr = x + (long long)n * x_nstride + (long long)c * HW;
    float s = 0.f;
    if (vec) {
        for (int i = 4 * threadIdx.x; i < HW; i += 4 * kThreads) {
            const float4 v = *reinterpret_cast<const float4*>(xr + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += kThreads) s += xr[i];
    }
    const float t = block_sum_256(s, red);
    if (threadIdx.x == 0) y[row] = t * scale;
}

__global__ void __launch_bounds__(kThreads)
broadcast_hw_kernel(const float* __restrict__ v, float scale, float* __restrict__ y,
                    long long y_nstride, int accumulate, int C, int HW, int colchunks,
                    int cols_per_block, int vec) {
    const int row = blockIdx.x / colchunks;
    const int chunk = blockIdx.x - row * colchunks;
    const int n = row / C, c = row - n * C;
    const float val = v[row] * scale;
    float* yr = y + (long long)n * y_nstride + (long long)c * HW;
    const int i0 = chunk * cols_per_block;
    int i1 = i0 + cols_per_block;
    if (i1 > HW) i1 = HW;
    if (vec) {
        for (int i = i0 + 4 * threadIdx.x; i < i1; i += 4 * kThreads) {
            float4 o = make_float4(val, val, val, val);
            if (accumulate) {
                const float4 p = *reinterpret_cast<const float4*>(yr + i);
                o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
            }
            *reinterpret_cast<float4*>(yr + i) = o;
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += kThreads) yr[i] = accumulate ? yr[i] + val : val;
    }
}

__global__ void __launch_bounds__(kThreads)
add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
           long long n, int vec) {
    if (vec) {
        const long long n4 = n / 4;
        for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4;
             i += (long long)gridDim.x * kThreads) {
            const float4 p = reinterpret_cast<const float4*>(a)[i];
            const float4 q = reinterpret_cast<const float4*>(b)[i];
            reinterpret_cast<float4*>(out)[i] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        }
        for (long long i = n4 * 4 + (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
             i += (long long)gridDim.x * kThreads)
            out[i] = a[i] + b[i];
    } else {
        for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
             i += (long long)gridDim.x * kThreads)
            out[i] = a[i] + b[i];
    }
}

__global__ void __launch_bounds__(kThreads)
channel_scale_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                     float* __restrict__ y, int HW, int colchunks, int cols_per_block, int vec) {
    const int row = blockIdx.x / colchunks;
    const int chunk = blockIdx.x - row * colchunks;
    const float m = mask[row];
    const float* xr = x + (long long)row * HW;
    float* yr = y + (long long)row * HW;
    const int i0 = chunk * cols_per_block;
    int i1 = i0 + cols_per_block;
    if (i1 > HW) i1 = HW;
    if (vec) {
        for (int i = i0 + 4 * threadIdx.x; i < i1; i += 4 * kThreads) {
            const float4 p = *reinterpret_cast<const float4*>(xr + i);
            *reinterpret_cast<float4*>(yr + i) = make_float4(p.x * m, p.y * m, p.z * m, p.w * m);
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += kThreads) yr[i] = xr[i] * m;
    }
}

constexpr int kCols = 4096;
inline unsigned stream_grid(long long total) {
    long long b = (total + kThreads - 1) / kThreads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int dcfp_maxpool3x3s2_fwd_f32(const float* x, float* y, int32_t* argmax, int N, int C,
                                         int H, int W, int Hout, int Wout, dcfp_stream_t stream) {
    if (!x || !y || !argmax || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DCFP_E_BADDESC;
    if (Hout != (H + 2 - 3) / 2 + 1 || Wout != (W + 2 - 3) / 2 + 1) return DCFP_E_BADDESC;
    if ((long long)H * W > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    const long long total = (long long)N * C * Hout * Wout;
    const long long orows = (long long)N * C * Hout;
    const int segs = (Wout / 4 + kThreads - 1) / kThreads;
    if (W % 8 == 0 && dcfp_aligned16(x) && dcfp_aligned16(y) && dcfp_aligned16(argmax) &&
        orows * segs <= 0x7fffffffLL)
        hipLaunchKernelGGL(maxpool3x3s2_fwd_vec4_kernel, dim3((unsigned)(orows * segs)), dim3(kThreads), 0,
                           dcfp_s(stream), x, y, argmax, H, W, Hout, Wout, segs);
    else
        hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), x, y, argmax, total, H, W, Hout, Wout);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_maxpool3x3s2_bwd_f32(const float* dy, const int32_t* argmax, float* dx, int N,
                                         int C, int H, int W, int Hout, int Wout,
                                         dcfp_stream_t stream) {
    if (!dy || !dx || !argmax || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DCFP_E_BADDESC;
    if (Hout != (H + 2 - 3) / 2 + 1 || Wout != (W + 2 - 3) / 2 + 1) return DCFP_E_BADDESC;
    const long long total = (long long)N * C * H * W;
    const long long rows = (long long)N * C * H;
    const int segs = (W / 4 + kThreads - 1) / kThreads;
    if (W % 4 == 0 && dcfp_aligned16(dx) && rows * segs <= 0x7fffffffLL)
        hipLaunchKernelGGL(maxpool3x3s2_bwd_vec4_kernel, dim3((unsigned)(rows * segs)), dim3(kThreads), 0,
                           dcfp_s(stream), dy, argmax, dx, H, W, Hout, Wout, segs);
    else
        hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), dy, argmax, dx, total, H, W, Hout, Wout);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_rowsum_f32(const float* x, int64_t x_nstride, float* y, float scale, int N,
                               int C, int HW, dcfp_stream_t stream) {
    if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return DCFP_E_BADDESC;
    if (x_nstride == 0) x_nstride = (int64_t)C * HW;
    const int vec = (HW % 4 == 0) && (x_nstride % 4 == 0) && dcfp_aligned16(x);
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)(N * C)), dim3(kThreads), 0, dcfp_s(stream), x,
                       (long long)x_nstride, y, scale, C, HW, vec);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_broadcast_hw_f32(const float* v, float scale, float* y, int64_t y_nstride,
                                     int accumulate, int N, int C, int HW, dcfp_stream_t stream) {
    if (!v || !y || N <= 0 || C <= 0 || HW <= 0) return DCFP_E_BADDESC;
    if (y_nstride == 0) y_nstride = (int64_t)C * HW;
    const int colchunks = (HW + kCols - 1) / kCols;
    const int vec = (HW % 4 == 0) && (y_nstride % 4 == 0) && dcfp_aligned16(y);
    hipLaunchKernelGGL(broadcast_hw_kernel, dim3((unsigned)(N * C * colchunks)), dim3(kThreads), 0,
                       dcfp_s(stream), v, scale, y, (long long)y_nstride, accumulate, C, HW,
                       colchunks, kCols, vec);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_add_f32(const float* a, const float* b, float* out, int64_t n,
                            dcfp_stream_t stream) {
    if (!a || !b || !out || n < 0) return DCFP_E_BADDESC;
    if (n == 0) return DCFP_OK;
    const int vec = dcfp_aligned16(a) && dcfp_aligned16(b) && dcfp_aligned16(out);
    hipLaunchKernelGGL(add_kernel, dim3(stream_grid((n + 3) / 4)), dim3(kThreads), 0, dcfp_s(stream),
                       a, b, out, (long long)n, vec);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_channel_scale_f32(const float* x, const float* mask, float* y, int N, int C,
                                      int HW, dcfp_stream_t stream) {
    if (!x || !mask || !y || N <= 0 || C <= 0 || HW <= 0) return DCFP_E_BADDESC;
    const int colchunks = (HW + kCols - 1) / kCols;
    const int vec = (HW % 4 == 0) && dcfp_aligned16(x) && dcfp_aligned16(y);
    hipLaunchKernelGGL(channel_scale_kernel, dim3((unsigned)(N * C * colchunks)), dim3(kThreads), 0,
                       dcfp_s(stream), x, mask, y, HW, colchunks, kCols, vec);
    DCFP_RETURN_LAUNCH();
}

// upsample_ce.hip — bilinear upsample (F.interpolate, networks/deeplabv3.py:47,50) and its
// fusion with nn.CrossEntropyLoss(ignore_index, 'mean') (loss/criterion.py:60,65-67).
//
// The fused pair never materialises the N x C x H x W logits (637 MB per head at
// 4x19x1024x2048): forward reads the low-resolution logits (L1/L2 resident) and the
// labels, and keeps one float per pixel (log-sum-exp); backward GATHERS, per
// low-resolution cell and class, the contributions of the <= ~16x16 full-resolution
// pixels that touch that cell, so there are no float atomics and the result is
// run-to-run deterministic.
//
// Coordinate rules follow ATen's area_pixel_compute_scale / _source_index:
//   align_corners: scale = (in-1)/(out-1) (0 if out==1), src = scale*dst
//   otherwise    : scale = in/out,        src = max(scale*(dst+0.5)-0.5, 0)
//   i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kLossBlocks = 2048;

struct Lerp {
    int i0, i1;
    float l0, l1;
};

template <bool ALIGN>
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_size) {
    float src;
    if (ALIGN) {
        src = scale * (float)dst;
    } else {
        src = scale * ((float)dst + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
    }
    Lerp r;
    int i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    r.i0 = i0;
    r.i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    float l1 = src - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    r.l1 = l1;
    r.l0 = 1.f - l1;
    return r;
}

inline float host_scale(int in, int out, int align) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

// Conservative [lo, hi] range of destination indices whose taps may include source cell i.
template <bool ALIGN>
__device__ __forceinline__ void dst_range(int i, float scale, int out_size, int& lo, int& hi) {
    if (!(scale > 0.f)) {  // out_size == 1 under align_corners
        lo = 0;
        hi = out_size - 1;
        return;
    }
    float a, b;
    if (ALIGN) {
        a = ((float)i - 1.f) / scale;
        b = ((float)i + 1.f) / scale;
    } else {
        a = ((float)i - 0.5f) / scale - 0.5f;
        b = ((float)i + 1.5f) / scale - 0.5f;
    }
    int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
    lo = l < 0 ? 0 : l;
    hi = h > out_size - 1 ? out_size - 1 : h;
}

__device__ __forceinline__ float tap_weight(const Lerp& L, int i) {
    return (L.i0 == i ? L.l0 : 0.f) + (L.i1 == i ? L.l1 : 0.f);
}

// ------------------------------------------------------------ plain upsample
template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int h,
                    int w, int H, int W, float sh, float sw) {
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int X = (int)(idx % W);
        const long long t = idx / W;
        const int Y = (int)(t % H);
        const long long plane = t / H;
        const Lerp Lh = lerp_of<ALIGN>(Y, sh, h), Lw = lerp_of<ALIGN>(X, sw, w);
        const float* p = x + plane * (long long)h * w;
        const float top = Lw.l0 * p[Lh.i0 * w + Lw.i0] + Lw.l1 * p[Lh.i0 * w + Lw.i1];
        const float bot = Lw.l0 * p[Lh.i1 * w + Lw.i0] + Lw.l1 * p[Lh.i1 * w + Lw.i1];
        y[idx] = Lh.l0 * top + Lh.l1 * bot;
    }
}

template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long total, int h,
                    int w, int H, int W, float sh, float sw) {
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int j = (int)(idx % w);
        const long long t = idx / w;
        const int i = (int)(t % h);
        const long long plane = t / h;
        const float* g = dy + plane * (long long)H * W;
        int ylo, yhi, xlo, xhi;
        dst_range<ALIGN>(i, sh, H, ylo, yhi);
        dst_range<ALIGN>(j, sw, W, xlo, xhi);
        float acc = 0.f;
        for (int Y = ylo; Y <= yhi; ++Y) {
            const float wy = tap_weight(lerp_of<ALIGN>(Y, sh, h), i);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int X = xlo; X <= xhi; ++X) {
                const float wx = tap_weight(lerp_of<ALIGN>(X, sw, w), j);
                if (wx != 0.f) row += wx * g[(long long)Y * W + X];
            }
            acc += wy * row;
        }
        dx[idx] = acc;
    }
}

// ------------------------------------------------------- fused upsample + CE
template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                       const uint8_t* __restrict__ keep, int ignore_index, int N, int C, int h,
                       int w, int H, int W, float sh, float sw, float* __restrict__ lse_out,
                       float* __restrict__ gt_prob, float* __restrict__ part) {
    __shared__ float red[4];
    const long long total = (long long)N * H * W;
    const long long plane = (long long)h * w;
    float loss = 0.f, cnt = 0.f;
    for (long long pix = (long long)blockIdx.x * kThreads + threadIdx.x; pix < total;
         pix += (long long)gridDim.x * kThreads) {
        const int X = (int)(pix % W);
        const long long t = pix / W;
        const int Y = (int)(t % H);
        const int n = (int)(t / H);
        const Lerp Lh = lerp_of<ALIGN>(Y, sh, h), Lw = lerp_of<ALIGN>(X, sw, w);
        const int o00 = Lh.i0 * w + Lw.i0, o01 = Lh.i0 * w + Lw.i1;
        const int o10 = Lh.i1 * w + Lw.i0, o11 = Lh.i1 * w + Lw.i1;
        const float* base = logits + (long long)n * C * plane;
        const long long label = labels[pix];
        float m = -INFINITY, s = 0.f, zl = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* p = base + c * plane;
            const float z = Lh.l0 * (Lw.l0 * p[o00] + Lw.l1 * p[o01]) +
                            Lh.l1 * (Lw.l0 * p[o10] + Lw.l1 * p[o11]);
            if (z > m) {
                s = s * expf(m - z) + 1.f;
                m = z;
            } else {
                s += expf(z - m);
            }
            if (c == label) zl = z;
        }
        const float lse = m + logf(s);
        const bool labelled = label != ignore_index;
        const bool valid = labelled && (!keep || keep[pix]);
        if (lse_out) lse_out[pix] = lse;
        if (gt_prob) gt_prob[pix] = labelled ? expf(zl - lse) : 1.f;
        if (valid) {
            loss += lse - zl;
            cnt += 1.f;
        }
    }
    const float t1 = block_sum_256(loss, red);
    const float t2 = block_sum_256(cnt, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = t1;
        part[blockIdx.x * 2 + 1] = t2;
    }
}

// One 256-thread block: thread t sums the partials t, t + 256, ... in fp64, then a fixed-order tree over the 256
// sums (deterministic; the single-thread loop this replaces took 134 us for 32 k partials).
__global__ void __launch_bounds__(256) loss_final_kernel(const float* __restrict__ part, int nblocks,
                                                         float* __restrict__ out2) {
    __shared__ double sa[256], sb[256];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 256) {
        const float2 v = *reinterpret_cast<const float2*>(part + 2 * k);
        a += (double)v.x;
        b += (double)v.y;
    }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = (float)sa[0];
        out2[1] = (float)sb[0];
    }
}

// thread <-> one element of dlogits[n,c,i,j]
template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                       const uint8_t* __restrict__ keep, int ignore_index, int N, int C, int h,
                       int w, int H, int W, float sh, float sw, const float* __restrict__ lse,
                       const float* __restrict__ grad_scale, float* __restrict__ dlogits) {
    const long long total = (long long)N * C * h * w;
    const float gs = grad_scale[0];
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int j = (int)(idx % w);
        long long t = idx / w;
        const int i = (int)(t % h);
        t /= h;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        const float* p = logits + ((long long)n * C + c) * h * w;
        const long long* lab = labels + (long long)n * H * W;
        const float* ls = lse + (long long)n * H * W;
        const uint8_t* kp = keep ? keep + (long long)n * H * W : nullptr;
        int ylo, yhi, xlo, xhi;
        dst_range<ALIGN>(i, sh, H, ylo, yhi);
        dst_range<ALIGN>(j, sw, W, xlo, xhi);
        float acc = 0.f;
        for (int Y = ylo; Y <= yhi; ++Y) {
            const Lerp Lh = lerp_of<ALIGN>(Y, sh, h);
            const float wy = tap_weight(Lh, i);
            if (wy == 0.f) continue;
            const float* r0 = p + Lh.i0 * w;
            const float* r1 = p + Lh.i1 * w;
            float row = 0.f;
            for (int X = xlo; X <= xhi; ++X) {
                const Lerp Lw = lerp_of<ALIGN>(X, sw, w);
                const float wx = tap_weight(Lw, j);
                if (wx == 0.f) continue;
                const long long q = (long long)Y * W + X;
                const long long label = lab[q];
                if (label == ignore_index || (kp && !kp[q])) continue;
                const float z = Lh.l0 * (Lw.l0 * r0[Lw.i0] + Lw.l1 * r0[Lw.i1]) +
                                Lh.l1 * (Lw.l0 * r1[Lw.i0] + Lw.l1 * r1[Lw.i1]);
                const float prob = expf(z - ls[q]);
                row += wx * (prob - (label == c ? 1.f : 0.f));
            }
            acc += wy * row;
        }
        dlogits[idx] = acc * gs;
    }
}

// Same gradient, thread <-> one low-resolution pixel (n,i,j) with ALL classes in registers (CT of
// them, compile-time): the per-destination-pixel work that does not depend on the class — the two
// interpolation stencils, the tap weights, label, LSE — is done once instead of once per class
// (3.0 -> ~1 ms per head at 19 classes).  The logit interpolation keeps the forward's expression.
template <bool ALIGN, int CT>
__global__ void __launch_bounds__(kThreads)
upsample_ce_bwd_classes_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                               const uint8_t* __restrict__ keep, int ignore_index, int N, int h, int w,
                               int H, int W, float sh, float sw, const float* __restrict__ lse,
                               const float* __restrict__ grad_scale, float* __restrict__ dlogits,
                               const float* __restrict__ pix_weight /* GSRL weights, nullable */,
                               int scale_per_image /* grad_scale[n] instead of grad_scale[0] */) {
    // four lanes share one low-resolution pixel (destination rows Y = ylo + part, +4, ...) and
    // combine their partial sums with two butterfly shuffles: 4x the waves for latency hiding
    const long long total = (long long)N * h * w * 4;
    const long long plane = (long long)h * w;
    for (long long idx4 = (long long)blockIdx.x * kThreads + threadIdx.x; idx4 < total;
         idx4 += (long long)gridDim.x * kThreads) {
        const int part = (int)(idx4 & 3);
        const long long idx = idx4 >> 2;
        const int j = (int)(idx % w);
        const long long t = idx / w;
        const int i = (int)(t % h);
        const int n = (int)(t / h);
        const float* base = logits + (long long)n * CT * plane;
        const long long* lab = labels + (long long)n * H * W;
        const float* ls = lse + (long long)n * H * W;
        const uint8_t* kp = keep ? keep + (long long)n * H * W : nullptr;
        const float* wp = pix_weight ? pix_weight + (long long)n * H * W : nullptr;
        const float gs = grad_scale[scale_per_image ? n : 0];
        int ylo, yhi, xlo, xhi;
        dst_range<ALIGN>(i, sh, H, ylo, yhi);
        dst_range<ALIGN>(j, sw, W, xlo, xhi);
        float acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.f;
        for (int Y = ylo + part; Y <= yhi; Y += 4) {
            const Lerp Lh = lerp_of<ALIGN>(Y, sh, h);
            const float wy = tap_weight(Lh, i);
            if (wy == 0.f) continue;
            for (int X = xlo; X <= xhi; ++X) {
                const Lerp Lw = lerp_of<ALIGN>(X, sw, w);
                float wgt = wy * tap_weight(Lw, j);
                if (wgt == 0.f) continue;
                const long long q = (long long)Y * W + X;
                const long long label = lab[q];
                if (label == ignore_index || (kp && !kp[q])) continue;
                if (wp) {
                    const float pwq = wp[q];
                    if (pwq == 0.f) continue;
                    wgt *= pwq;
                }
                const int o00 = Lh.i0 * w + Lw.i0, o01 = Lh.i0 * w + Lw.i1;
                const int o10 = Lh.i1 * w + Lw.i0, o11 = Lh.i1 * w + Lw.i1;
                const float lq = ls[q];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const float* p = base + c * plane;
                    const float z = Lh.l0 * (Lw.l0 * p[o00] + Lw.l1 * p[o01]) +
                                    Lh.l1 * (Lw.l0 * p[o10] + Lw.l1 * p[o11]);
                    acc[c] += wgt * (expf(z - lq) - (label == c ? 1.f : 0.f));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {   // fixed combination order: (p0+p1)+(p2+p3)
            acc[c] += __shfl_xor(acc[c], 1, 64);
            acc[c] += __shfl_xor(acc[c], 2, 64);
        }
        if (part == 0) {
            float* o = dlogits + (long long)n * CT * plane + (long long)i * w + j;
#pragma unroll
            for (int c = 0; c < CT; ++c) o[c * plane] = acc[c] * gs;
        }
    }
}

// The same gradient organised by CELLS (19 classes, the training path).  The cell of a destination pixel is the
// low-resolution position (i0, j0) its stencil starts at; every destination pixel lies in exactly one cell and
// touches only that cell's four corners (i0|i1, j0|j1).  The kernel above visits every destination pixel once per
// corner it touches - four softmax evaluations per pixel; here a lane pair owns a cell, holds the four corner
// logit vectors in registers, evaluates each of its ~8 x 8 destination pixels ONCE and accumulates the four
// corner sums (4 x 19 registers).  A workgroup covers a 7 x 15 tile of low-resolution outputs = 8 x 16 cells (one
// halo row / column above and left is recomputed by the neighbour: 1.2x instead of 4x), parks the corner sums in
// LDS and gathers, per output and class, the <= 4 (cell, corner) terms that land on it in a fixed order - no
// atomics, deterministic.  Interpolation and softmax expressions are the forward's.
constexpr int kCellTH = 7, kCellTW = 15, kCellThreads = 256;      // 8 x 16 cells x 2 lanes: four full waves, two
                                                                   // workgroups per CU at 219 registers (an 8 x 16 tile
                                                                   // with 320 threads left 3 of 8 wave slots empty)
constexpr int kCellCount = (kCellTH + 1) * (kCellTW + 1);

template <bool ALIGN, int CT>
__global__ void __launch_bounds__(kCellThreads, 2)
upsample_ce_bwd_cells_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                             const uint8_t* __restrict__ keep, int ignore_index, int N, int h, int w, int H,
                             int W, float sh, float sw, const float* __restrict__ lse,
                             const float* __restrict__ grad_scale, float* __restrict__ dlogits,
                             const float* __restrict__ pix_weight, int scale_per_image, int tiles_w, int tiles_h) {
    __shared__ float corner[kCellCount][4][CT + 1];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h;
    const int n = b / tiles_h;
    const int I0 = th * kCellTH, J0 = tw * kCellTW;
    const long long plane = (long long)h * w;
    const float* base = logits + (long long)n * CT * plane;
    const long long* lab = labels + (long long)n * H * W;
    const float* ls = lse + (long long)n * H * W;
    const uint8_t* kp = keep ? keep + (long long)n * H * W : nullptr;
    const float* wp = pix_weight ? pix_weight + (long long)n * H * W : nullptr;

    const int cell = tid >> 1, part = tid & 1;
    if (cell < kCellCount) {
        const int ci = I0 - 1 + cell / (kCellTW + 1), cj = J0 - 1 + cell % (kCellTW + 1);
        float a00[CT], a01[CT], a10[CT], a11[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) a00[c] = a01[c] = a10[c] = a11[c] = 0.f;
        if (ci >= 0 && ci < h && cj >= 0 && cj < w) {
            const int i1 = ci + (ci < h - 1 ? 1 : 0), j1 = cj + (cj < w - 1 ? 1 : 0);
            float p00[CT], p01[CT], p10[CT], p11[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float* p = base + c * plane;
                p00[c] = p[ci * w + cj]; p01[c] = p[ci * w + j1];
                p10[c] = p[i1 * w + cj]; p11[c] = p[i1 * w + j1];
            }
            int ylo, yhi, xlo, xhi;
            dst_range<ALIGN>(ci, sh, H, ylo, yhi);
            dst_range<ALIGN>(cj, sw, W, xlo, xhi);
            for (int Y = ylo + part; Y <= yhi; Y += 2) {
                const Lerp Lh = lerp_of<ALIGN>(Y, sh, h);
                if (Lh.i0 != ci) continue;
                for (int X = xlo; X <= xhi; ++X) {
                    const Lerp Lw = lerp_of<ALIGN>(X, sw, w);
                    if (Lw.i0 != cj) continue;
                    const long long q = (long long)Y * W + X;
                    const long long label = lab[q];
                    if (label == ignore_index || (kp && !kp[q])) continue;
                    float pw = 1.f;
                    if (wp) {
                        pw = wp[q];
                        if (pw == 0.f) continue;
                    }
                    const float lq = ls[q];
                    const float w00 = Lh.l0 * Lw.l0 * pw, w01 = Lh.l0 * Lw.l1 * pw;
                    const float w10 = Lh.l1 * Lw.l0 * pw, w11 = Lh.l1 * Lw.l1 * pw;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const float z = Lh.l0 * (Lw.l0 * p00[c] + Lw.l1 * p01[c]) +
                                        Lh.l1 * (Lw.l0 * p10[c] + Lw.l1 * p11[c]);
                        const float g = expf(z - lq) - (label == c ? 1.f : 0.f);
                        a00[c] += w00 * g; a01[c] += w01 * g;
                        a10[c] += w10 * g; a11[c] += w11 * g;
                    }
                }
            }
        }
        // the two lanes of a cell: even rows + odd rows, always in that order
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            a00[c] += __shfl_xor(a00[c], 1, 64); a01[c] += __shfl_xor(a01[c], 1, 64);
            a10[c] += __shfl_xor(a10[c], 1, 64); a11[c] += __shfl_xor(a11[c], 1, 64);
        }
        if (part == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                corner[cell][0][c] = a00[c]; corner[cell][1][c] = a01[c];
                corner[cell][2][c] = a10[c]; corner[cell][3][c] = a11[c];
            }
        }
    }
    __syncthreads();
    // gather: output (i, j) collects corner (a, b) of cell (ci, cj) wherever ci + (a && ci < h-1) == i and
    // cj + (b && cj < w-1) == j; candidates are ci in {i-1, i}, cj in {j-1, j}; fixed visiting order
    const float gs = grad_scale[scale_per_image ? n : 0];
    for (int o = tid; o < kCellTH * kCellTW * CT; o += kCellThreads) {
        const int c = o / (kCellTH * kCellTW);
        const int r = o - c * (kCellTH * kCellTW);
        const int li = r / kCellTW, lj = r - li * kCellTW;
        const int i = I0 + li, j = J0 + lj;
        if (i >= h || j >= w) continue;
        float acc = 0.f;
#pragma unroll
        for (int di = 0; di < 2; ++di) {
            const int ci = i - 1 + di;
            if (ci < 0) continue;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (ci + ((a && ci < h - 1) ? 1 : 0) != i) continue;
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    const int cj = j - 1 + dj;
                    if (cj < 0) continue;
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        if (cj + ((bb && cj < w - 1) ? 1 : 0) != j) continue;
                        acc += corner[(li + di) * (kCellTW + 1) + (lj + dj)][a * 2 + bb][c];
                    }
                }
            }
        }
        dlogits[((long long)n * CT + c) * plane + (long long)i * w + j] = acc * gs;
    }
}

// OHEM threshold search input (loss/ohem.py:20-33): the reference zooms the full-resolution
// softmax to 1/factor with scipy.ndimage.zoom(order=1) and the labels with order=0, then
// gathers the zoomed probability of the zoomed label.  Per zoomed position that is a bilinear
// blend of prob_L at four full-resolution pixels, L = nearest label — computed here from the
// low-resolution logits and the per-pixel LSE without ever forming the probability tensor.
// scipy's coordinate rule (grid_mode=False): src = o * (in-1)/(out-1), evaluated in double.
template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
ohem_zoom_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                 const float* __restrict__ lse, int N, int C, int h, int w, int H, int W,
                 float sh, float sw, int H8, int W8, float* __restrict__ pred8,
                 int* __restrict__ lab8) {
    const long long total = (long long)N * H8 * W8;
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int ox = (int)(idx % W8);
        const long long t = idx / W8;
        const int oy = (int)(t % H8);
        const int n = (int)(t / H8);
        const double cy = H8 > 1 ? (double)oy * (double)(H - 1) / (double)(H8 - 1) : 0.0;
        const double cx = W8 > 1 ? (double)ox * (double)(W - 1) / (double)(W8 - 1) : 0.0;
        int ny = (int)floor(cy + 0.5), nx = (int)floor(cx + 0.5);
        ny = ny > H - 1 ? H - 1 : ny; nx = nx > W - 1 ? W - 1 : nx;
        const long long L = labels[((long long)n * H + ny) * W + nx];
        lab8[idx] = (int)L;
        float out = 0.f;
        if (L >= 0 && L < C) {
            const int y0 = (int)floor(cy), x0 = (int)floor(cx);
            const double ty = cy - y0, tx = cx - x0;
            const float* p = logits + ((long long)n * C + L) * h * w;
            double acc = 0.0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int Y = y0 + dy, X = x0 + dx;
                    const double wgt = (dy ? ty : 1.0 - ty) * (dx ? tx : 1.0 - tx);
                    if (Y > H - 1 || X > W - 1 || wgt == 0.0) continue;   // mode='constant', cval=0
                    const Lerp Lh = lerp_of<ALIGN>(Y, sh, h), Lw = lerp_of<ALIGN>(X, sw, w);
                    const float z = Lh.l0 * (Lw.l0 * p[Lh.i0 * w + Lw.i0] + Lw.l1 * p[Lh.i0 * w + Lw.i1]) +
                                    Lh.l1 * (Lw.l0 * p[Lh.i1 * w + Lw.i0] + Lw.l1 * p[Lh.i1 * w + Lw.i1]);
                    acc += wgt * (double)expf(z - lse[((long long)n * H + Y) * W + X]);
                }
            }
            out = (float)acc;
        }
        pred8[idx] = out;
    }
}

// ---- GSRL (loss/criterion.py:77-101): margin map, k x k stride-1 max filter, per-pixel-weighted CE
// margin[pix] = p1 - p2 (two largest softmax probabilities of the interpolated logits)
template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_margin_kernel(const float* __restrict__ logits, int N, int C, int h, int w, int H, int W,
                       float sh, float sw, float* __restrict__ margin) {
    const long long total = (long long)N * H * W;
    const long long plane = (long long)h * w;
    for (long long pix = (long long)blockIdx.x * kThreads + threadIdx.x; pix < total;
         pix += (long long)gridDim.x * kThreads) {
        const int X = (int)(pix % W);
        const long long t = pix / W;
        const int Y = (int)(t % H);
        const int n = (int)(t / H);
        const Lerp Lh = lerp_of<ALIGN>(Y, sh, h), Lw = lerp_of<ALIGN>(X, sw, w);
        const int o00 = Lh.i0 * w + Lw.i0, o01 = Lh.i0 * w + Lw.i1;
        const int o10 = Lh.i1 * w + Lw.i0, o11 = Lh.i1 * w + Lw.i1;
        const float* base = logits + (long long)n * C * plane;
        float m1 = -INFINITY, m2 = -INFINITY, s = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* p = base + c * plane;
            const float z = Lh.l0 * (Lw.l0 * p[o00] + Lw.l1 * p[o01]) +
                            Lh.l1 * (Lw.l0 * p[o10] + Lw.l1 * p[o11]);
            if (z > m1) {
                s = s * expf(m1 - z) + 1.f;
                m2 = m1;
                m1 = z;
            } else {
                s += expf(z - m1);
                m2 = z > m2 ? z : m2;
            }
        }
        // p1 = exp(m1 - lse) = 1/s ; p2 = exp(m2 - m1)/s
        margin[pix] = (1.f - (C > 1 ? expf(m2 - m1) : 0.f)) / s;
    }
}

// y = max over the k x k window (stride 1, pad k/2, -inf padding) of a [planes,H,W] map
__global__ void __launch_bounds__(kThreads)
maxfilter_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int H, int W,
                 int k) {
    const int r = k / 2;
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int X = (int)(idx % W);
        const long long t = idx / W;
        const int Y = (int)(t % H);
        const float* xp = x + (t / H) * (long long)H * W;
        const int y0 = Y - r < 0 ? 0 : Y - r, y1 = Y + (k - 1 - r) > H - 1 ? H - 1 : Y + (k - 1 - r);
        const int x0 = X - r < 0 ? 0 : X - r, x1 = X + (k - 1 - r) > W - 1 ? W - 1 : X + (k - 1 - r);
        float best = -INFINITY;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) {
                const float v = xp[yy * W + xx];
                best = (v > best || v != v) ? v : best;
            }
        y[idx] = best;
    }
}

// per-image sums  out[n] = (sum_pix w*ce, sum_pix w)  with w = pix_weight (0 where ignored)
template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_wce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                        const float* __restrict__ pw, int ignore_index, int C, int h, int w, int H,
                        int W, float sh, float sw, float* __restrict__ lse_out,
                        float* __restrict__ part /* [N][gridDim.x][2] */) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const long long HWl = (long long)H * W;
    const long long plane = (long long)h * w;
    const float* base = logits + (long long)n * C * plane;
    float a = 0.f, b = 0.f;
    for (long long q = (long long)blockIdx.x * kThreads + threadIdx.x; q < HWl;
         q += (long long)gridDim.x * kThreads) {
        const int X = (int)(q % W), Y = (int)(q / W);
        const long long pix = n * HWl + q;
        const Lerp Lh = lerp_of<ALIGN>(Y, sh, h), Lw = lerp_of<ALIGN>(X, sw, w);
        const int o00 = Lh.i0 * w + Lw.i0, o01 = Lh.i0 * w + Lw.i1;
        const int o10 = Lh.i1 * w + Lw.i0, o11 = Lh.i1 * w + Lw.i1;
        const long long label = labels[pix];
        float m = -INFINITY, s = 0.f, zl = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* p = base + c * plane;
            const float z = Lh.l0 * (Lw.l0 * p[o00] + Lw.l1 * p[o01]) +
                            Lh.l1 * (Lw.l0 * p[o10] + Lw.l1 * p[o11]);
            if (z > m) { s = s * expf(m - z) + 1.f; m = z; } else { s += expf(z - m); }
            if (c == label) zl = z;
        }
        const float lse = m + logf(s);
        if (lse_out) lse_out[pix] = lse;
        const float wgt = pw[pix];
        if (label != ignore_index) a += wgt * (lse - zl);   // CE(reduction='none') is 0 where ignored
        b += wgt;
    }
    const float t1 = block_sum_256(a, red);
    const float t2 = block_sum_256(b, red);
    if (threadIdx.x == 0) {
        part[((long long)n * gridDim.x + blockIdx.x) * 2 + 0] = t1;
        part[((long long)n * gridDim.x + blockIdx.x) * 2 + 1] = t2;
    }
}

__global__ void wce_final_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
    const int n = blockIdx.x;
    if (threadIdx.x != 0) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nblocks; ++k) {
        a += (double)part[((long long)n * nblocks + k) * 2];
        b += (double)part[((long long)n * nblocks + k) * 2 + 1];
    }
    out[2 * n] = (float)a;
    out[2 * n + 1] = (float)b;
}

// dlogits[n,c,i,j] = gs[n] * sum_pix wt * pw[pix] * (softmax - onehot)   (gather form)
template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_wce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                        const float* __restrict__ pw, int ignore_index, int N, int C, int h, int w,
                        int H, int W, float sh, float sw, const float* __restrict__ lse,
                        const float* __restrict__ gs, float* __restrict__ dlogits) {
    const long long total = (long long)N * C * h * w;
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const int j = (int)(idx % w);
        long long t = idx / w;
        const int i = (int)(t % h);
        t /= h;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        const float* p = logits + ((long long)n * C + c) * h * w;
        const long long* lab = labels + (long long)n * H * W;
        const float* ls = lse + (long long)n * H * W;
        const float* wp = pw + (long long)n * H * W;
        int ylo, yhi, xlo, xhi;
        dst_range<ALIGN>(i, sh, H, ylo, yhi);
        dst_range<ALIGN>(j, sw, W, xlo, xhi);
        float acc = 0.f;
        for (int Y = ylo; Y <= yhi; ++Y) {
            const Lerp Lh = lerp_of<ALIGN>(Y, sh, h);
            const float wy = tap_weight(Lh, i);
            if (wy == 0.f) continue;
            const float* r0 = p + Lh.i0 * w;
            const float* r1 = p + Lh.i1 * w;
            float row = 0.f;
            for (int X = xlo; X <= xhi; ++X) {
                const Lerp Lw = lerp_of<ALIGN>(X, sw, w);
                const float wx = tap_weight(Lw, j);
                if (wx == 0.f) continue;
                const long long q = (long long)Y * W + X;
                const long long label = lab[q];
                const float wgt = wp[q];
                if (label == ignore_index || wgt == 0.f) continue;
                const float z = Lh.l0 * (Lw.l0 * r0[Lw.i0] + Lw.l1 * r0[Lw.i1]) +
                                Lh.l1 * (Lw.l0 * r1[Lw.i0] + Lw.l1 * r1[Lw.i1]);
                row += wx * wgt * (expf(z - ls[q]) - (label == c ? 1.f : 0.f));
            }
            acc += wy * row;
        }
        dlogits[idx] = acc * gs[n];
    }
}

template <bool ALIGN>
__global__ void __launch_bounds__(kThreads)
upsample_argmax_kernel(const float* __restrict__ logits, int N, int C, int h, int w, int H, int W,
                       float sh, float sw, int* __restrict__ pred) {
    const long long total = (long long)N * H * W;
    const long long plane = (long long)h * w;
    for (long long pix = (long long)blockIdx.x * kThreads + threadIdx.x; pix < total;
         pix += (long long)gridDim.x * kThreads) {
        const int X = (int)(pix % W);
        const long long t = pix / W;
        const int Y = (int)(t % H);
        const int n = (int)(t / H);
        const Lerp Lh = lerp_of<ALIGN>(Y, sh, h), Lw = lerp_of<ALIGN>(X, sw, w);
        const int o00 = Lh.i0 * w + Lw.i0, o01 = Lh.i0 * w + Lw.i1;
        const int o10 = Lh.i1 * w + Lw.i0, o11 = Lh.i1 * w + Lw.i1;
        const float* base = logits + (long long)n * C * plane;
        float best = -INFINITY;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float* p = base + c * plane;
            const float z = Lh.l0 * (Lw.l0 * p[o00] + Lw.l1 * p[o01]) +
                            Lh.l1 * (Lw.l0 * p[o10] + Lw.l1 * p[o11]);
            if (z > best) { best = z; bi = c; }     // first maximum wins, like torch.argmax / np.argmax
        }
        pred[pix] = bi;
    }
}

// per-block LDS histogram (C*C <= 4096 bins), then one integer atomic per non-empty bin
__global__ void __launch_bounds__(kThreads)
confusion_kernel(const int* __restrict__ pred, const long long* __restrict__ gt, int ignore_index,
                 long long n, int C, unsigned long long* __restrict__ conf) {
    extern __shared__ unsigned int hist[];
    const int bins = C * C;
    for (int i = threadIdx.x; i < bins; i += kThreads) hist[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kThreads) {
        const long long g = gt[i];
        const int p = pred[i];
        if (g != ignore_index && g >= 0 && g < C && p >= 0 && p < C) atomicAdd(&hist[(int)g * C + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += kThreads)
        if (hist[i]) atomicAdd(&conf[i], (unsigned long long)hist[i]);
}

__global__ void __launch_bounds__(kThreads)
confusion_global_kernel(const int* __restrict__ pred, const long long* __restrict__ gt, int ignore_index,
                        long long n, int C, unsigned long long* __restrict__ conf) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kThreads) {
        const long long g = gt[i];
        const int p = pred[i];
        if (g != ignore_index && g >= 0 && g < C && p >= 0 && p < C) atomicAdd(&conf[g * C + p], 1ull);
    }
}

inline unsigned stream_grid(long long total) {
    long long b = (total + kThreads - 1) / kThreads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int dcfp_upsample_bilinear_fwd_f32(const float* x, float* y, int N, int C, int h, int w,
                                              int H, int W, int align_corners,
                                              dcfp_stream_t stream) {
    if (!x || !y || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DCFP_E_BADDESC;
    const long long total = (long long)N * C * H * W;
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    if (align_corners)
        hipLaunchKernelGGL(upsample_fwd_kernel<true>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), x, y, total, h, w, H, W, sh, sw);
    else
        hipLaunchKernelGGL(upsample_fwd_kernel<false>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), x, y, total, h, w, H, W, sh, sw);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_upsample_bilinear_bwd_f32(const float* dy, float* dx, int N, int C, int h,
                                              int w, int H, int W, int align_corners,
                                              dcfp_stream_t stream) {
    if (!dy || !dx || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DCFP_E_BADDESC;
    const long long total = (long long)N * C * h * w;
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    if (align_corners)
        hipLaunchKernelGGL(upsample_bwd_kernel<true>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), dy, dx, total, h, w, H, W, sh, sw);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel<false>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), dy, dx, total, h, w, H, W, sh, sw);
    DCFP_RETURN_LAUNCH();
}

extern "C" size_t dcfp_upsample_ce_workspace_bytes(int N, int H, int W) {
    (void)N; (void)H; (void)W;
    return (size_t)kLossBlocks * 2 * sizeof(float);
}

extern "C" int dcfp_upsample_ce_fwd_f32(const float* logits, const int64_t* labels,
                                        const uint8_t* pixel_keep, int ignore_index, int N, int C,
                                        int h, int w, int H, int W, int align_corners, float* lse,
                                        float* gt_prob, float* out2, void* workspace,
                                        size_t workspace_bytes, dcfp_stream_t stream) {
    if (!logits || !labels || !out2 || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
        return DCFP_E_BADDESC;
    if (!workspace || workspace_bytes < (size_t)kLossBlocks * 2 * sizeof(float))
        return DCFP_E_WORKSPACE;
    const long long total = (long long)N * H * W;
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > kLossBlocks) blocks = kLossBlocks;
    float* part = static_cast<float*>(workspace);
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    const long long* lab = reinterpret_cast<const long long*>(labels);
    if (align_corners)
        hipLaunchKernelGGL(upsample_ce_fwd_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, pixel_keep, ignore_index, N, C, h, w, H, W,
                           sh, sw, lse, gt_prob, part);
    else
        hipLaunchKernelGGL(upsample_ce_fwd_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, pixel_keep, ignore_index, N, C, h, w, H, W,
                           sh, sw, lse, gt_prob, part);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, dcfp_s(stream), part, (int)blocks,
                       out2);
    DCFP_RETURN_LAUNCH();
}

// 19-class backward: the cell-organised kernel (DCFP_CE_BWD_CELLS=0: the per-output kernel, for A/B)
template <bool ALIGN>
static void launch_ce_bwd_19(const float* logits, const long long* lab, const uint8_t* keep, int ignore_index, int N,
                             int h, int w, int H, int W, float sh, float sw, const float* lse, const float* gscale,
                             float* dlogits, const float* pix_weight, int per_image, hipStream_t st) {
    static const bool cells = [] { const char* e = getenv("DCFP_CE_BWD_CELLS"); return !e || atoi(e) != 0; }();
    if (cells) {
        const int tiles_w = (w + kCellTW - 1) / kCellTW, tiles_h = (h + kCellTH - 1) / kCellTH;
        hipLaunchKernelGGL((upsample_ce_bwd_cells_kernel<ALIGN, 19>), dim3((unsigned)(N * tiles_h * tiles_w)),
                           dim3(kCellThreads), 0, st, logits, lab, keep, ignore_index, N, h, w, H, W, sh, sw, lse,
                           gscale, dlogits, pix_weight, per_image, tiles_w, tiles_h);
        return;
    }
    const long long tot2 = (long long)N * h * w * 4;   // 4 lanes per low-resolution pixel
    const unsigned b2 = (unsigned)((tot2 + kThreads - 1) / kThreads);
    hipLaunchKernelGGL((upsample_ce_bwd_classes_kernel<ALIGN, 19>), dim3(b2), dim3(kThreads), 0, st, logits, lab,
                       keep, ignore_index, N, h, w, H, W, sh, sw, lse, gscale, dlogits, pix_weight, per_image);
}

extern "C" int dcfp_upsample_ce_bwd_f32(const float* logits, const int64_t* labels,
                                        const uint8_t* pixel_keep, int ignore_index, int N, int C,
                                        int h, int w, int H, int W, int align_corners,
                                        const float* lse, const float* grad_scale, float* dlogits,
                                        dcfp_stream_t stream) {
    if (!logits || !labels || !lse || !grad_scale || !dlogits || N <= 0 || C <= 0 || h <= 0 ||
        w <= 0 || H <= 0 || W <= 0)
        return DCFP_E_BADDESC;
    const long long total = (long long)N * C * h * w;
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    const long long* lab = reinterpret_cast<const long long*>(labels);
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    if (C == 19 && (long long)N * ((h + kCellTH - 1) / kCellTH) * ((w + kCellTW - 1) / kCellTW) <= 0x7fffffffLL) {
        // Cityscapes (datasets/CSdatasets.py:13): all classes in registers
        if (align_corners)
            launch_ce_bwd_19<true>(logits, lab, pixel_keep, ignore_index, N, h, w, H, W, sh, sw, lse, grad_scale,
                                   dlogits, nullptr, 0, dcfp_s(stream));
        else
            launch_ce_bwd_19<false>(logits, lab, pixel_keep, ignore_index, N, h, w, H, W, sh, sw, lse, grad_scale,
                                    dlogits, nullptr, 0, dcfp_s(stream));
        DCFP_RETURN_LAUNCH();
    }
    if (align_corners)
        hipLaunchKernelGGL(upsample_ce_bwd_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, pixel_keep, ignore_index, N, C, h, w, H, W,
                           sh, sw, lse, grad_scale, dlogits);
    else
        hipLaunchKernelGGL(upsample_ce_bwd_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, pixel_keep, ignore_index, N, C, h, w, H, W,
                           sh, sw, lse, grad_scale, dlogits);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_ohem_zoom_gt_prob_f32(const float* logits, const int64_t* labels, const float* lse,
                                          int N, int C, int h, int w, int H, int W,
                                          int align_corners, int H8, int W8, float* pred8,
                                          int32_t* lab8, dcfp_stream_t stream) {
    if (!logits || !labels || !lse || !pred8 || !lab8 || N <= 0 || C <= 0 || h <= 0 || w <= 0 ||
        H <= 0 || W <= 0 || H8 <= 0 || W8 <= 0)
        return DCFP_E_BADDESC;
    const long long total = (long long)N * H8 * W8;
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    const long long* lab = reinterpret_cast<const long long*>(labels);
    if (align_corners)
        hipLaunchKernelGGL(ohem_zoom_kernel<true>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, lse, N, C, h, w, H, W, sh, sw, H8, W8, pred8, lab8);
    else
        hipLaunchKernelGGL(ohem_zoom_kernel<false>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, lse, N, C, h, w, H, W, sh, sw, H8, W8, pred8, lab8);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_upsample_margin_f32(const float* logits, int N, int C, int h, int w, int H, int W,
                                        int align_corners, float* margin, dcfp_stream_t stream) {
    if (!logits || !margin || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DCFP_E_BADDESC;
    const long long total = (long long)N * H * W;
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    if (align_corners)
        hipLaunchKernelGGL(upsample_margin_kernel<true>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), logits, N, C, h, w, H, W, sh, sw, margin);
    else
        hipLaunchKernelGGL(upsample_margin_kernel<false>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), logits, N, C, h, w, H, W, sh, sw, margin);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_maxfilter2d_s1_f32(const float* x, float* y, int planes, int H, int W, int k,
                                       dcfp_stream_t stream) {
    if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || k <= 0 || (k & 1) == 0) return DCFP_E_BADDESC;
    const long long total = (long long)planes * H * W;
    hipLaunchKernelGGL(maxfilter_kernel, dim3(stream_grid(total)), dim3(kThreads), 0, dcfp_s(stream), x, y,
                       total, H, W, k);
    DCFP_RETURN_LAUNCH();
}

constexpr int kWceBlocks = 512;   // per image

extern "C" size_t dcfp_upsample_wce_workspace_bytes(int N, int H, int W) {
    (void)H; (void)W;
    return (size_t)(N > 0 ? N : 0) * kWceBlocks * 2 * sizeof(float);
}

extern "C" int dcfp_upsample_wce_fwd_f32(const float* logits, const int64_t* labels,
                                         const float* pix_weight, int ignore_index, int N, int C, int h,
                                         int w, int H, int W, int align_corners, float* lse,
                                         float* out_per_image, void* workspace, size_t workspace_bytes,
                                         dcfp_stream_t stream) {
    if (!logits || !labels || !pix_weight || !out_per_image || N <= 0 || C <= 0 || h <= 0 || w <= 0 ||
        H <= 0 || W <= 0 || N > 65535)
        return DCFP_E_BADDESC;
    if (!workspace || workspace_bytes < (size_t)N * kWceBlocks * 2 * sizeof(float)) return DCFP_E_WORKSPACE;
    long long blocks = ((long long)H * W + kThreads - 1) / kThreads;
    if (blocks > kWceBlocks) blocks = kWceBlocks;
    float* part = static_cast<float*>(workspace);
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    const long long* lab = reinterpret_cast<const long long*>(labels);
    dim3 grid((unsigned)blocks, (unsigned)N);
    if (align_corners)
        hipLaunchKernelGGL(upsample_wce_fwd_kernel<true>, grid, dim3(kThreads), 0, dcfp_s(stream), logits, lab,
                           pix_weight, ignore_index, C, h, w, H, W, sh, sw, lse, part);
    else
        hipLaunchKernelGGL(upsample_wce_fwd_kernel<false>, grid, dim3(kThreads), 0, dcfp_s(stream), logits, lab,
                           pix_weight, ignore_index, C, h, w, H, W, sh, sw, lse, part);
    hipLaunchKernelGGL(wce_final_kernel, dim3((unsigned)N), dim3(64), 0, dcfp_s(stream), part, (int)blocks,
                       out_per_image);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_upsample_wce_bwd_f32(const float* logits, const int64_t* labels,
                                         const float* pix_weight, int ignore_index, int N, int C, int h,
                                         int w, int H, int W, int align_corners, const float* lse,
                                         const float* grad_scale_per_image, float* dlogits,
                                         dcfp_stream_t stream) {
    if (!logits || !labels || !pix_weight || !lse || !grad_scale_per_image || !dlogits || N <= 0 ||
        C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
        return DCFP_E_BADDESC;
    const long long total = (long long)N * C * h * w;
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    const long long* lab = reinterpret_cast<const long long*>(labels);
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    if (C == 19) {   // all classes in registers (see upsample_ce_bwd_cells_kernel)
        if (align_corners)
            launch_ce_bwd_19<true>(logits, lab, nullptr, ignore_index, N, h, w, H, W, sh, sw, lse,
                                   grad_scale_per_image, dlogits, pix_weight, 1, dcfp_s(stream));
        else
            launch_ce_bwd_19<false>(logits, lab, nullptr, ignore_index, N, h, w, H, W, sh, sw, lse,
                                    grad_scale_per_image, dlogits, pix_weight, 1, dcfp_s(stream));
        DCFP_RETURN_LAUNCH();
    }
    if (align_corners)
        hipLaunchKernelGGL(upsample_wce_bwd_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, pix_weight, ignore_index, N, C, h, w, H, W, sh, sw,
                           lse, grad_scale_per_image, dlogits);
    else
        hipLaunchKernelGGL(upsample_wce_bwd_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), logits, lab, pix_weight, ignore_index, N, C, h, w, H, W, sh, sw,
                           lse, grad_scale_per_image, dlogits);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_upsample_argmax_f32(const float* logits, int N, int C, int h, int w, int H, int W,
                                        int align_corners, int32_t* pred, dcfp_stream_t stream) {
    if (!logits || !pred || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DCFP_E_BADDESC;
    const long long total = (long long)N * H * W;
    const float sh = host_scale(h, H, align_corners), sw = host_scale(w, W, align_corners);
    if (align_corners)
        hipLaunchKernelGGL(upsample_argmax_kernel<true>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), logits, N, C, h, w, H, W, sh, sw, pred);
    else
        hipLaunchKernelGGL(upsample_argmax_kernel<false>, dim3(stream_grid(total)), dim3(kThreads), 0,
                           dcfp_s(stream), logits, N, C, h, w, H, W, sh, sw, pred);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_confusion_matrix_i64(const int32_t* pred, const int64_t* gt, int ignore_index,
                                         int64_t n_pixels, int C, int64_t* conf, dcfp_stream_t stream) {
    if (!pred || !gt || !conf || n_pixels < 0 || C <= 0 || C > 1024) return DCFP_E_BADDESC;
    if (n_pixels == 0) return DCFP_OK;
    long long b = (n_pixels + kThreads - 1) / kThreads;
    if (b > 1024) b = 1024;
    if (C <= 64)
        hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)b), dim3(kThreads), (size_t)C * C * sizeof(unsigned),
                           dcfp_s(stream), pred, reinterpret_cast<const long long*>(gt), ignore_index,
                           (long long)n_pixels, C, reinterpret_cast<unsigned long long*>(conf));
    else   // ADE (150) / COCO-Stuff (171): the histogram does not fit LDS, count in global memory
        hipLaunchKernelGGL(confusion_global_kernel, dim3((unsigned)b), dim3(kThreads), 0, dcfp_s(stream), pred,
                           reinterpret_cast<const long long*>(gt), ignore_index, (long long)n_pixels, C,
                           reinterpret_cast<unsigned long long*>(conf));
    DCFP_RETURN_LAUNCH();
}

// bn.hip — BatchNorm2d(+ReLU)(+residual) training forward/backward for NCHW fp32.
// Replaces nn.BatchNorm2d / nn.ReLU(inplace) as used in the reference at
// networks/backbone/resnet.py:26-33,41-56 and networks/tools/aspp.py:15-16,22-24.
//
// All four kernels are HBM-bound streams: 16-byte loads per lane, one (n,c) row
// segment or one (c, chunk) slice per 256-thread block, wave-shuffle reductions,
// per-block partials combined in a fixed order in fp64 (deterministic, no atomics).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;

// The BN output value, written ONE way so that the forward kernel and the backward kernels
// that re-derive the ReLU mask from x round identically.
__device__ __forceinline__ float bn_val(float v, float mu, float istd, float g, float b) {
    return fmaf((v - mu) * istd, g, b);
}

struct BnPlan {
    int chunks;       // blocks per channel
    int chunk_elems;  // elements (of the N*HW per-channel population) per block, multiple of 4
};

// One plan for every per-channel reduction (statistics, backward sums, the fused backward): a block owns at most
// kChunkElems = 8192 elements of a channel's N*HW population - 8 float4 per thread and tensor, which is what the fused
// backward can keep in registers between its two phases (round 4; rounds 1-3 capped the grid at ~4096 blocks instead,
// up to 32768 elements per block).  Fused and two-kernel backward share the plan, so their sums are the same bits.
constexpr int kChunkElems = 8192;

BnPlan bn_plan(int N, int C, int HW) {
    const long long E = (long long)N * HW;
    long long chunks = (E + kChunkElems - 1) / kChunkElems;
    if (E <= kChunkElems) {
        // small populations (up to one full chunk): the split of rounds 1-3 - ~2048 elements per block, at most ~4096
        // blocks in all; chunks of at most 8192 elements either way, which is all the fused backward asks for
        const long long by_size = (E + 2047) / 2048;
        long long by_grid = 4096 / (C > 0 ? C : 1);
        if (by_grid < 1) by_grid = 1;
        chunks = by_size < by_grid ? by_size : by_grid;
    }
    if (chunks < 1) chunks = 1;
    long long ce = (E + chunks - 1) / chunks;
    ce = (ce + 1023) / 1024 * 1024;   // whole wave iterations (256 elements per wave): the bit-mask ReLU path needs them
    chunks = (E + ce - 1) / ce;
    if (chunks < 1) chunks = 1;
    BnPlan p;
    p.chunks = (int)chunks;
    p.chunk_elems = (int)ce;
    return p;
}

// The two backward sums of one float4 and the value of dx, written ONE way: explicit fused multiply-adds under
// `fp contract(off)` - the rounding is then the same in every kernel that inlines these (the two-kernel path and the fused
// kernel must agree bit for bit: the data-parallel path runs the former, the plain step the latter, and the tests hold them
// identical), and no product is rounded on its own where hipcc's default contraction would have fused it.
__device__ __forceinline__ void bwd_accumulate(const float4& g, const float4& xv, float mu, float& s1, float& s2) {
#pragma clang fp contract(off)
    s1 += (g.x + g.y) + (g.z + g.w);
    const float a = fmaf(g.x, xv.x - mu, g.y * (xv.y - mu));
    const float b = fmaf(g.z, xv.z - mu, g.w * (xv.w - mu));
    s2 += a + b;
}
__device__ __forceinline__ float bwd_dx(float g, float xv, float mu, float mean_dy, float k, float gi) {
#pragma clang fp contract(off)
    return fmaf(-(xv - mu), k, g - mean_dy) * gi;
}

// ---- stats: shifted sums  S1 = sum(x-K), S2 = sum((x-K)^2), K = x[0,c,0]
template <bool VEC>
__global__ void __launch_bounds__(kThreads)
bn_stats_partial_kernel(const float* __restrict__ x, long long nstride, int HW, long long E,
                        int chunk_elems, int chunks, float* __restrict__ part) {
    __shared__ float red[4];
    const int c = blockIdx.y, chunk = blockIdx.x;
    const float* xc = x + (long long)c * HW;
    const float K = xc[0];
    const long long e0 = (long long)chunk * chunk_elems;
    long long e1 = e0 + chunk_elems;
    if (e1 > E) e1 = E;
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
        for (long long e = e0 + 4 * threadIdx.x; e < e1; e += 4 * kThreads) {
            const long long n = e / HW;
            const int i = (int)(e - n * HW);
            const float4 v = *reinterpret_cast<const float4*>(xc + n * nstride + i);
            const float a = v.x - K, b = v.y - K, cc = v.z - K, d = v.w - K;
            s1 += (a + b) + (cc + d);
            s2 += (a * a + b * b) + (cc * cc + d * d);
        }
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += kThreads) {
            const long long n = e / HW;
            const int i = (int)(e - n * HW);
            const float a = xc[n * nstride + i] - K;
            s1 += a;
            s2 += a * a;
        }
    }
    const float t1 = block_sum_256(s1, red);
    const float t2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        part[((long long)c * chunks + chunk) * 2 + 0] = t1;
        part[((long long)c * chunks + chunk) * 2 + 1] = t2;
    }
}

__global__ void bn_stats_final_kernel(const float* __restrict__ x, int HW, long long E, int C,
                                      int chunks, const float* __restrict__ part,
                                      float* __restrict__ mean, float* __restrict__ var, DcfpBnRunning run) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double S1 = 0.0, S2 = 0.0;
    for (int k = 0; k < chunks; ++k) {
        S1 += (double)part[((long long)c * chunks + k) * 2 + 0];
        S2 += (double)part[((long long)c * chunks + k) * 2 + 1];
    }
    const double K = (double)x[(long long)c * HW];
    const double m1 = S1 / (double)E;
    double v = S2 / (double)E - m1 * m1;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)(K + m1);
    var[c] = (float)v;
    running_update(run, c, (float)(K + m1), (float)v, (float)E);
}

// ReLU bit mask of a residual BatchNorm (one bit per element instead of re-reading the 4-byte output
// in both backward kernels).  Layout per (n,c) row, HW % 256 == 0: a wave iteration covers 256
// consecutive pixels (lane l holds pixels 4l..4l+3); word 4*(i/256) + k holds, at bit l, the mask of
// pixel 256*(i/256) + 4l + k.
__device__ __forceinline__ bool mask_bit(const unsigned long long* __restrict__ row_words, int i_wave, int k) {
    const unsigned long long wv = row_words[(i_wave >> 8) * 4 + k];
    return (wv >> (threadIdx.x & 63)) & 1ull;
}

// ---- apply: one (n,c) row segment per block
template <bool VEC, bool MASK = false>
__global__ void __launch_bounds__(kThreads)
bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                const float* __restrict__ var, const float* __restrict__ gamma,
                const float* __restrict__ beta, float eps, const float* __restrict__ res,
                int relu, float* __restrict__ y, long long y_nstride, int C, int HW,
                int colchunks, int cols_per_block, unsigned long long* __restrict__ mask = nullptr,
                int W = 0, int y_pitch = 0) {
    const long long row = blockIdx.x / colchunks;
    const int chunk = blockIdx.x - (int)(row * colchunks);
    const int n = (int)(row / C), c = (int)(row - (long long)n * C);
    const float istd = 1.0f / sqrtf(var[c] + eps);
    const float mu = mean[c], g = gamma[c], b = beta[c];
    const float* xr = x + row * HW;
    const float* rr = res ? res + row * HW : nullptr;
    // y_pitch != 0: rows of W floats are written y_pitch floats apart (the tail behind each row is the caller's
    // zero padding and is never touched); element i of the (n,c) plane then lives at (i / W) * y_pitch + i % W
    float* yr = y + (long long)n * y_nstride + (long long)c * (y_pitch ? (long long)(HW / W) * y_pitch : HW);
    const int i0 = chunk * cols_per_block;
    int i1 = i0 + cols_per_block;
    if (i1 > HW) i1 = HW;
    if (VEC) {
        for (int i = i0 + 4 * threadIdx.x; i < i1; i += 4 * kThreads) {
            const float4 v = *reinterpret_cast<const float4*>(xr + i);
            float4 o;
            o.x = bn_val(v.x, mu, istd, g, b);
            o.y = bn_val(v.y, mu, istd, g, b);
            o.z = bn_val(v.z, mu, istd, g, b);
            o.w = bn_val(v.w, mu, istd, g, b);
            if (rr) {
                const float4 r = *reinterpret_cast<const float4*>(rr + i);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            if (relu) {
                o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
                o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
            }
            if constexpr (MASK) {
                const unsigned long long b0 = __ballot(o.x > 0.f), b1 = __ballot(o.y > 0.f);
                const unsigned long long b2 = __ballot(o.z > 0.f), b3 = __ballot(o.w > 0.f);
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* mw = mask + row * (HW >> 6) + (i >> 8) * 4;
                    mw[0] = b0; mw[1] = b1; mw[2] = b2; mw[3] = b3;
                }
            }
            if (y_pitch) { const int h = i / W; *reinterpret_cast<float4*>(yr + h * y_pitch + (i - h * W)) = o; }
            else *reinterpret_cast<float4*>(yr + i) = o;
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += kThreads) {
            float o = bn_val(xr[i], mu, istd, g, b);
            if (rr) o += rr[i];
            if (relu) o = o > 0.f ? o : 0.f;
            if (y_pitch) { const int h = i / W; yr[h * y_pitch + (i - h * W)] = o; }
            else yr[i] = o;
        }
    }
}

// ---- backward stage 1: S1 = sum g, S2 = sum g*(x-mean),  g = dy*(y>0)
// RELU: 0 none, 1 mask from the saved output y, 2 mask re-derived from x (no residual: y is a
// function of x alone, so the 4 B/element read of y is skipped)
template <bool VEC, int RELU>
__global__ void __launch_bounds__(kThreads)
bn_bwd_reduce_partial_kernel(const float* __restrict__ dy, long long dy_nstride,
                             const float* __restrict__ x, const float* __restrict__ y,
                             long long y_nstride, const float* __restrict__ mean,
                             const float* __restrict__ var, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float eps, int C, int HW,
                             long long E, int chunk_elems, int chunks, float* __restrict__ part) {
    __shared__ float red[4];
    const int c = blockIdx.y, chunk = blockIdx.x;
    const float mu = mean[c];
    float istd = 0.f, gm = 0.f, bt = 0.f;
    if (RELU == 2) { istd = 1.0f / sqrtf(var[c] + eps); gm = gamma[c]; bt = beta[c]; }
    const long long e0 = (long long)chunk * chunk_elems;
    long long e1 = e0 + chunk_elems;
    if (e1 > E) e1 = E;
    const long long coff = (long long)c * HW;
    const long long x_nstride = (long long)C * HW;
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
        for (long long e = e0 + 4 * threadIdx.x; e < e1; e += 4 * kThreads) {
            const long long n = e / HW;
            const int i = (int)(e - n * HW);
            float4 g = *reinterpret_cast<const float4*>(dy + n * dy_nstride + coff + i);
            const float4 xv = *reinterpret_cast<const float4*>(x + n * x_nstride + coff + i);
            if (RELU == 1) {
                const float4 yv = *reinterpret_cast<const float4*>(y + n * y_nstride + coff + i);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            } else if (RELU == 2) {
                g.x = bn_val(xv.x, mu, istd, gm, bt) > 0.f ? g.x : 0.f;
                g.y = bn_val(xv.y, mu, istd, gm, bt) > 0.f ? g.y : 0.f;
                g.z = bn_val(xv.z, mu, istd, gm, bt) > 0.f ? g.z : 0.f;
                g.w = bn_val(xv.w, mu, istd, gm, bt) > 0.f ? g.w : 0.f;
            } else if (RELU == 3) {   // y carries the bit mask of dcfp_bn_apply_relu_mask_f32
                const unsigned long long* mw = reinterpret_cast<const unsigned long long*>(y) +
                                               (n * C + c) * (long long)(HW >> 6);
                const int iw = i - 4 * (int)(threadIdx.x & 63);
                g.x = mask_bit(mw, iw, 0) ? g.x : 0.f; g.y = mask_bit(mw, iw, 1) ? g.y : 0.f;
                g.z = mask_bit(mw, iw, 2) ? g.z : 0.f; g.w = mask_bit(mw, iw, 3) ? g.w : 0.f;
            }
            bwd_accumulate(g, xv, mu, s1, s2);
        }
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += kThreads) {
            const long long n = e / HW;
            const int i = (int)(e - n * HW);
            float g = dy[n * dy_nstride + coff + i];
            const float xv = x[n * x_nstride + coff + i];
            if (RELU == 1) g = y[n * y_nstride + coff + i] > 0.f ? g : 0.f;
            else if (RELU == 2) g = bn_val(xv, mu, istd, gm, bt) > 0.f ? g : 0.f;
            s1 += g;
            s2 += g * (xv - mu);
        }
    }
    const float t1 = block_sum_256(s1, red);
    const float t2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        part[((long long)c * chunks + chunk) * 2 + 0] = t1;
        part[((long long)c * chunks + chunk) * 2 + 1] = t2;
    }
}

// The per-channel totals of `chunks` (s1, s2) partial pairs, by ONE wave, in an order both backward paths share:
// value q = 2 * chunk + which (which = 0: s1, 1: s2) goes to lane q % 64, which adds its values in ascending q in fp64;
// the 32 lanes of a parity class are then combined by an xor butterfly (offsets 2 .. 32: every lane of the class ends
// with the same bits).  Returns the class total: s1's in even lanes, s2's in odd lanes.  value_of(j) = value q = lane + 64 j.
template <class F>
__device__ __forceinline__ double wave_pair_totals(int n2, int lane, F value_of) {
    double acc = 0.0;
    for (int j = 0; lane + 64 * j < n2; ++j) acc += (double)value_of(j);
#pragma unroll
    for (int off = 2; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
    return acc;
}

// o3 (nullable): dgamma = sum(dy*(x-mean)) * rsqrt(var + eps), saving the caller three launches.  One wave per channel.
__global__ void __launch_bounds__(64) bn_pair_final_kernel(int C, int chunks, const float* __restrict__ part,
                                     float* __restrict__ o1, float* __restrict__ o2,
                                     const float* __restrict__ var, float eps, float* __restrict__ o3,
                                     float* __restrict__ o4) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* pc = part + (long long)c * chunks * 2;
    const double t = wave_pair_totals(2 * chunks, lane, [&](int j) { return pc[lane + 64 * j]; });
    const double S1 = __shfl(t, 0, 64), S2 = __shfl(t, 1, 64);
    if (lane == 0) {
        o1[c] = (float)S1;
        o2[c] = (float)S2;
        if (o3) o3[c] = (float)S2 * rsqrtf(var[c] + eps);
        if (o4) o4[c] = (float)S1;
    }
}

// SyncBatchNorm forward combine: allv[r] = (mean_r[C], var_r[C], count_r) gathered from every rank ->
// pooled mean / biased variance over all ranks' pixels (parallel-variance combination), one launch
__global__ void syncbn_combine_kernel(const float* __restrict__ allv, int world, int C,
                                      float* __restrict__ gmean, float* __restrict__ gvar,
                                      float* __restrict__ total, DcfpBnRunning run) {
#pragma clang fp contract(off)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int S = 2 * C + 1;
    // fp64, rank order, every product and sum rounded on its own (no FMA contraction): the host path
    // (ops.syncbn_combine_reference) evaluates the same expression with torch and must agree bit for bit
    double tot = 0.0, m = 0.0, v = 0.0;
    for (int r = 0; r < world; ++r) {
        const double n = (double)allv[(long long)r * S + 2 * C];
        tot = __dadd_rn(tot, n);
        m = __dadd_rn(m, __dmul_rn((double)allv[(long long)r * S + c], n));
    }
    m /= tot;
    for (int r = 0; r < world; ++r) {
        const double n = (double)allv[(long long)r * S + 2 * C];
        const double d = __dadd_rn((double)allv[(long long)r * S + c], -m);
        v = __dadd_rn(v, __dmul_rn(__dadd_rn((double)allv[(long long)r * S + C + c], __dmul_rn(d, d)), n));
    }
    gmean[c] = (float)m;
    gvar[c] = (float)(v / tot);
    if (c == 0) total[0] = (float)tot;
    running_update(run, c, (float)m, (float)(v / tot), (float)tot);
}

// Batch statistics from the conv epilogue's partials: part[s][c] = (mean_s, M2_s) of `cnt` values each, Chan's
// parallel combination in fp64:  mean = avg(mean_s),  var = (sum M2_s + cnt * sum (mean_s - mean)^2) / (S * cnt).
// Work split (round 2: the one-block-per-channel version read 8 bytes at a stride of C*8 - one 64-byte line per
// load - and took 13.8 us per BatchNorm): a block owns 32 consecutive channels; thread (g = t / 32, l = t % 32)
// sums the slots s = g, g + 8, ... of channel c0 + l, so a wave's loads cover 2 x 256 contiguous bytes, and the 8
// partial sums of a channel are added in a fixed order (deterministic).
constexpr int kSfpThreads = 1024;
// (1024 threads: a channel's slots are spread over 32 threads, each keeping 4 loads in flight - with 8 threads per
// channel and one load at a time the kernel was latency-bound at ~70 us)
// kSfpCh channels per block (8 for narrow layers with many slots - the 64-channel stem has 16 384 - else 32)
template <int kSfpCh>
__global__ void __launch_bounds__(kSfpThreads)
bn_stats_from_partials_kernel(const float* __restrict__ part, long long S, int cnt, int C,
                              float* __restrict__ mean, float* __restrict__ var, DcfpBnRunning run) {
    constexpr int kSfpGroups = kSfpThreads / kSfpCh;
    __shared__ double red[kSfpGroups][kSfpCh];
    __shared__ double mu_s[kSfpCh];
    const int l = threadIdx.x % kSfpCh, g = threadIdx.x / kSfpCh;
    const int c = blockIdx.x * kSfpCh + l;
    const bool on = c < C;
    const float2* p2 = reinterpret_cast<const float2*>(part);
    constexpr int U = 4;
    double a[U] = {0.0, 0.0, 0.0, 0.0};
    if (on) {
        long long s = g;
        for (; s + (U - 1) * kSfpGroups < S; s += U * kSfpGroups) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p2[(s + u * kSfpGroups) * C + c].x;
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] += (double)v[u];
        }
        for (; s < S; s += kSfpGroups) a[0] += (double)p2[s * C + c].x;
    }
    red[g][l] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (g == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < kSfpGroups; ++k) t += red[k][l];
        mu_s[l] = t / (double)S;
    }
    __syncthreads();
    const double mu = mu_s[l];
    double b[U] = {0.0, 0.0, 0.0, 0.0};
    if (on) {
        long long s = g;
        for (; s + (U - 1) * kSfpGroups < S; s += U * kSfpGroups) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p2[(s + u * kSfpGroups) * C + c];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double d = (double)v[u].x - mu;
                b[u] += (double)v[u].y + (double)cnt * d * d;
            }
        }
        for (; s < S; s += kSfpGroups) {
            const float2 v = p2[s * C + c];
            const double d = (double)v.x - mu;
            b[0] += (double)v.y + (double)cnt * d * d;
        }
    }
    __syncthreads();
    red[g][l] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    if (g == 0 && on) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < kSfpGroups; ++k) t += red[k][l];
        const float v = (float)(t / ((double)S * (double)cnt));
        mean[c] = (float)mu;
        var[c] = v;
        running_update(run, c, (float)mu, v, (float)((double)S * (double)cnt));
    }
}

// BatchNorm-backward sums from the fan-in dgrad's side output (igemm2_dma1p_kernel<..., RED>): part[s][c] = (sum g,
// sum g*(x - mean)) over 128 pixels each -> S1[c], S2[c] in fp64, slots in a fixed order (thread (g, l) takes the
// slots s = g, g + groups, ...; the groups' sums are added in order), then what bn_pair_final_kernel writes.
__global__ void __launch_bounds__(kSfpThreads)
bn_bwd_sums_from_partials_kernel(const float* __restrict__ part, long long S, int C, const float* __restrict__ var,
                                 float eps, float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o3,
                                 float* __restrict__ o4) {
    constexpr int kCh = 32, kGroups = kSfpThreads / kCh;
    __shared__ double r1[kGroups][kCh], r2[kGroups][kCh];
    const int l = threadIdx.x % kCh, g = threadIdx.x / kCh;
    const int c = blockIdx.x * kCh + l;
    const bool on = c < C;
    const float2* p2 = reinterpret_cast<const float2*>(part);
    constexpr int U = 4;
    double a[U] = {0.0, 0.0, 0.0, 0.0}, b[U] = {0.0, 0.0, 0.0, 0.0};
    if (on) {
        long long s = g;
        for (; s + (U - 1) * kGroups < S; s += U * kGroups) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p2[(s + u * kGroups) * C + c];
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] += (double)v[u].x; b[u] += (double)v[u].y; }
        }
        for (; s < S; s += kGroups) {
            const float2 v = p2[s * C + c];
            a[0] += (double)v.x; b[0] += (double)v.y;
        }
    }
    r1[g][l] = (a[0] + a[1]) + (a[2] + a[3]);
    r2[g][l] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    if (g == 0 && on) {
        double S1 = 0.0, S2 = 0.0;
#pragma unroll
        for (int k = 0; k < kGroups; ++k) { S1 += r1[k][l]; S2 += r2[k][l]; }
        o1[c] = (float)S1;
        o2[c] = (float)S2;
        if (o3) o3[c] = (float)S2 * rsqrtf(var[c] + eps);
        if (o4) o4[c] = (float)S1;
    }
}

// running = (1-m)*running + m*stat, variance unbiased by count/(count-1) (nn.BatchNorm2d training)
__global__ void bn_running_kernel(int C, const float* __restrict__ mean, const float* __restrict__ var,
                                  float momentum, float count, const float* __restrict__ count_dev,
                                  float* __restrict__ rmean, float* __restrict__ rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float n = count_dev ? count_dev[0] : count;
    const DcfpBnRunning r = {rmean, rvar, nullptr, momentum, 0};
    running_update(r, c, mean[c], var[c], n);      // (one rounding recipe for every kernel: common.h)
}

// ---- backward stage 2
template <bool VEC, int RELU>
__global__ void __launch_bounds__(kThreads)
bn_bwd_apply_kernel(const float* __restrict__ dy, long long dy_nstride,
                    const float* __restrict__ x, const float* __restrict__ y,
                    long long y_nstride, const float* __restrict__ mean,
                    const float* __restrict__ var, const float* __restrict__ gamma,
                    const float* __restrict__ beta, float eps,
                    const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_xmu,
                    float inv_count_host, const float* __restrict__ count_dev,
                    float* __restrict__ dx, float* __restrict__ dres, int C,
                    int HW, int colchunks, int cols_per_block, int W, int dx_pitch) {
    const float inv_count = count_dev ? 1.0f / count_dev[0] : inv_count_host;
    const long long row = blockIdx.x / colchunks;
    const int chunk = blockIdx.x - (int)(row * colchunks);
    const int n = (int)(row / C), c = (int)(row - (long long)n * C);
    const float istd = 1.0f / sqrtf(var[c] + eps);
    const float mu = mean[c];
    const float mean_dy = sum_dy[c] * inv_count;
    const float k = sum_dy_xmu[c] * inv_count * istd * istd;
    const float gm = gamma[c];
    const float gi = gm * istd;
    const float bt = RELU == 2 ? beta[c] : 0.f;
    const float* dyr = dy + (long long)n * dy_nstride + (long long)c * HW;
    const float* xr = x + row * HW;
    const float* yr = RELU == 1 ? y + (long long)n * y_nstride + (long long)c * HW : nullptr;
    float* dxr = dx + row * (dx_pitch ? (long long)(HW / W) * dx_pitch : HW);   // dx_pitch: as y_pitch of bn_apply
    float* drr = dres ? dres + row * HW : nullptr;
    const int i0 = chunk * cols_per_block;
    int i1 = i0 + cols_per_block;
    if (i1 > HW) i1 = HW;
    if (VEC) {
        for (int i = i0 + 4 * threadIdx.x; i < i1; i += 4 * kThreads) {
            float4 g = *reinterpret_cast<const float4*>(dyr + i);
            const float4 xv = *reinterpret_cast<const float4*>(xr + i);
            if (RELU == 1) {
                const float4 yv = *reinterpret_cast<const float4*>(yr + i);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            } else if (RELU == 2) {
                g.x = bn_val(xv.x, mu, istd, gm, bt) > 0.f ? g.x : 0.f;
                g.y = bn_val(xv.y, mu, istd, gm, bt) > 0.f ? g.y : 0.f;
                g.z = bn_val(xv.z, mu, istd, gm, bt) > 0.f ? g.z : 0.f;
                g.w = bn_val(xv.w, mu, istd, gm, bt) > 0.f ? g.w : 0.f;
            } else if (RELU == 3) {
                const unsigned long long* mw = reinterpret_cast<const unsigned long long*>(y) + row * (HW >> 6);
                const int iw = i - 4 * (int)(threadIdx.x & 63);
                g.x = mask_bit(mw, iw, 0) ? g.x : 0.f; g.y = mask_bit(mw, iw, 1) ? g.y : 0.f;
                g.z = mask_bit(mw, iw, 2) ? g.z : 0.f; g.w = mask_bit(mw, iw, 3) ? g.w : 0.f;
            }
            float4 o;
            o.x = bwd_dx(g.x, xv.x, mu, mean_dy, k, gi);
            o.y = bwd_dx(g.y, xv.y, mu, mean_dy, k, gi);
            o.z = bwd_dx(g.z, xv.z, mu, mean_dy, k, gi);
            o.w = bwd_dx(g.w, xv.w, mu, mean_dy, k, gi);
            if (dx_pitch) { const int h = i / W; *reinterpret_cast<float4*>(dxr + h * dx_pitch + (i - h * W)) = o; }
            else *reinterpret_cast<float4*>(dxr + i) = o;
            if (drr) *reinterpret_cast<float4*>(drr + i) = g;
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += kThreads) {
            float g = dyr[i];
            if (RELU == 1) g = yr[i] > 0.f ? g : 0.f;
            else if (RELU == 2) g = bn_val(xr[i], mu, istd, gm, bt) > 0.f ? g : 0.f;
            const float o = (g - mean_dy - (xr[i] - mu) * k) * gi;
            if (dx_pitch) { const int h = i / W; dxr[h * dx_pitch + (i - h * W)] = o; }
            else dxr[i] = o;
            if (drr) drr[i] = g;
        }
    }
}

// ---- backward, both stages in ONE launch (round 4).  The two-kernel backward reads dy and x twice (20 B/element:
// 8 for the sums, 12 for dx); here block (c, chunk) loads its <= 8192 elements of dy and x ONCE into registers (8 float4
// per thread and tensor), publishes its two partial sums, waits for the other blocks of its channel, and computes dx
// from the registers: 12 B/element.  Hand-off between the blocks of a channel (they may sit on different XCDs, whose
// L2s are not coherent): every partial is ONE 8-byte granule {tag = epoch, value}, stored and polled with relaxed
// agent-scope atomics (write-through / L2-bypassing) - the data is the flag, no fence, no counter
// (cdna_hip_programming.md section 6, Guideline 16, form R2).  `sync` holds 2 granules per (channel, chunk); the caller
// zero-fills it once and passes a strictly increasing epoch (never 0) with every call, so a granule of an earlier
// call can never pass for this one's.  Forward progress: blocks are dispatched in blockIdx order and a block only
// waits for the `chunks` blocks of its own channel (consecutive ids, <= kFusedMaxChunks of them, far below what the
// chip holds), so the oldest unfinished channel always has all its blocks resident; the spin is bounded all the same
// (spin_limit polls of ~0.3 us: give up -> dx of this block = NaN, *status = epoch).
// Arithmetic and summation order are those of bn_bwd_reduce_partial_kernel + bn_pair_final_kernel + bn_bwd_apply_kernel
// (same plan, same helpers): the results are the same bits as the two-kernel path's.
typedef __attribute__((address_space(1))) unsigned long long bn_gu64;
constexpr int kFusedMaxChunks = 256;                        // blocks per channel: 2M elements (the stem at 4 x 512 x 1024); two granules per polling thread
constexpr int kFusedFlatChunks = 16;                        // up to here every block reads all partials of its channel
constexpr int kFusedIters = kChunkElems / (4 * kThreads);   // 8 float4 per thread and tensor

struct BnFusedParams {
    const float* dy; long long dy_nstride;
    const float* x; const float* y; long long y_nstride;
    const float* mean; const float* var; const float* gamma; const float* beta;
    float eps, inv_count;
    float* dx; float* dres;
    int C, HW, E, chunk_elems, chunks, W, dx_pitch;
    unsigned long long* sync; unsigned epoch, spin_limit; int* status;
    float* sum_dy; float* sum_dy_xmu; float* dgamma; float* dbeta;
};

// The hand-off of the fused backward: block (c, chunk) publishes its two partial sums (t1, t2 valid in thread 0), the
// channel's totals come back in tot[0..1] (as floats: what bn_pair_final_kernel writes); *failed is set when the wait gave up.
// parts: 2 * kFusedMaxChunks floats of LDS.  Ends in a __syncthreads().
// Channels of more than kFusedFlatChunks blocks meet in two levels: only the block of the LAST chunk (dispatched last, so it
// rarely waits) reads all the partials; it publishes the two totals as granules of their own and the others poll just those -
// every block reading every partial is 2 x chunks^2 granule loads per channel and round of polling (4 KB per block and round at
// 256 blocks).  The block that reads the partials does so with ALL its threads, at most two granules each (both loads in
// flight together; four registers - a polling wave holding eight granules per lane cost the kernel its fifth block per CU),
// parks the values in LDS, and one wave adds them in bn_pair_final_kernel's order.
__device__ __forceinline__ void bn_fused_handoff(const BnFusedParams& p, int c, int chunk, float t1, float t2, float* tot,
                                                 int* failed, float* parts) {
    const int n2 = 2 * p.chunks;
    bn_gu64* gr = (bn_gu64*)(p.sync + (long long)c * (n2 + 2));
    const unsigned long long tag = (unsigned long long)p.epoch << 32;
    const bool two_level = p.chunks > kFusedFlatChunks;
    const bool summer = !two_level || chunk == p.chunks - 1;          // (block-uniform)
    const int tid = (int)threadIdx.x;
    if (tid == 0) {
        __hip_atomic_store(gr + chunk * 2 + 0, tag | __float_as_uint(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gr + chunk * 2 + 1, tag | __float_as_uint(t2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (summer) {
        const int q0 = tid, q1 = tid + kThreads;                       // n2 <= 2 * kThreads
        unsigned long long v0 = q0 < n2 ? __hip_atomic_load(gr + q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
        unsigned long long v1 = q1 < n2 ? __hip_atomic_load(gr + q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
        unsigned spins = 0;
        while ((v0 >> 32) != p.epoch || (v1 >> 32) != p.epoch) {
            if (++spins > p.spin_limit) { *failed = 1; break; }
            __builtin_amdgcn_s_sleep(4);
            if ((v0 >> 32) != p.epoch) v0 = __hip_atomic_load(gr + q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v1 >> 32) != p.epoch) v1 = __hip_atomic_load(gr + q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (q0 < n2) parts[q0] = __uint_as_float((unsigned)v0);
        if (q1 < n2) parts[q1] = __uint_as_float((unsigned)v1);
        __syncthreads();
        if (tid < 64) {               // the channel's sums: bn_pair_final_kernel's order (wave_pair_totals)
            const double acc = wave_pair_totals(n2, tid, [&](int j) { return parts[tid + 64 * j]; });
            if (tid < 2) {
                tot[tid] = (float)acc;
                if (two_level && !*failed)
                    __hip_atomic_store(gr + n2 + tid, tag | __float_as_uint((float)acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (tid < 2) {             // the others: the two totals only
        unsigned long long v = __hip_atomic_load(gr + n2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((v >> 32) != p.epoch) {
            if (++spins > p.spin_limit) { *failed = 1; break; }
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(gr + n2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        tot[tid] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
    if (tid == 0 && *failed && p.status) atomicExch(p.status, (int)p.epoch);
}

template <int RELU, bool DRES>
__global__ void __launch_bounds__(kThreads) bn_bwd_fused_kernel(const BnFusedParams p) {
    __shared__ float red[4];
    __shared__ float tot[2];
    __shared__ int failed;
    __shared__ float parts[2 * kFusedMaxChunks];
    const int c = blockIdx.x / p.chunks, chunk = blockIdx.x - c * p.chunks;
    const int HW = p.HW, C = p.C;
    const float mu = p.mean[c];
    const float istd = 1.0f / sqrtf(p.var[c] + p.eps);
    const float gm = p.gamma[c];
    const float bt = RELU == 2 ? p.beta[c] : 0.f;
    const int e0 = chunk * p.chunk_elems;
    const int e1 = e0 + p.chunk_elems < p.E ? e0 + p.chunk_elems : p.E;
    const long long coff = (long long)c * HW;
    const long long x_nstride = (long long)C * HW;
    if (threadIdx.x == 0) failed = 0;

    // ---- phase 1: the block's elements into registers, g = dy * mask in place of dy
    float4 g[kFusedIters], xv[kFusedIters];
    int nn[kFusedIters], ii[kFusedIters];
#pragma unroll
    for (int k = 0; k < kFusedIters; ++k) {
        const int e = e0 + 4 * (int)threadIdx.x + 4 * kThreads * k;
        const int n = (int)((unsigned)e / (unsigned)HW);
        nn[k] = n;
        ii[k] = e - n * HW;
        if (e < e1) {
            g[k] = *reinterpret_cast<const float4*>(p.dy + (long long)n * p.dy_nstride + coff + ii[k]);
            xv[k] = *reinterpret_cast<const float4*>(p.x + (long long)n * x_nstride + coff + ii[k]);
        } else {
            g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kFusedIters; ++k) {
        const int e = e0 + 4 * (int)threadIdx.x + 4 * kThreads * k;
        if (e < e1) {
            if (RELU == 1) {
                const float4 yv = *reinterpret_cast<const float4*>(p.y + (long long)nn[k] * p.y_nstride + coff + ii[k]);
                g[k].x = yv.x > 0.f ? g[k].x : 0.f; g[k].y = yv.y > 0.f ? g[k].y : 0.f;
                g[k].z = yv.z > 0.f ? g[k].z : 0.f; g[k].w = yv.w > 0.f ? g[k].w : 0.f;
            } else if (RELU == 2) {
                g[k].x = bn_val(xv[k].x, mu, istd, gm, bt) > 0.f ? g[k].x : 0.f;
                g[k].y = bn_val(xv[k].y, mu, istd, gm, bt) > 0.f ? g[k].y : 0.f;
                g[k].z = bn_val(xv[k].z, mu, istd, gm, bt) > 0.f ? g[k].z : 0.f;
                g[k].w = bn_val(xv[k].w, mu, istd, gm, bt) > 0.f ? g[k].w : 0.f;
            } else if (RELU == 3) {
                const unsigned long long* mw = reinterpret_cast<const unsigned long long*>(p.y) +
                                               ((long long)nn[k] * C + c) * (long long)(HW >> 6);
                const int iw = ii[k] - 4 * (int)(threadIdx.x & 63);
                g[k].x = mask_bit(mw, iw, 0) ? g[k].x : 0.f; g[k].y = mask_bit(mw, iw, 1) ? g[k].y : 0.f;
                g[k].z = mask_bit(mw, iw, 2) ? g[k].z : 0.f; g[k].w = mask_bit(mw, iw, 3) ? g[k].w : 0.f;
            }
            bwd_accumulate(g[k], xv[k], mu, s1, s2);
        }
    }
    const float t1 = block_sum_256(s1, red);
    const float t2 = block_sum_256(s2, red);

    bn_fused_handoff(p, c, chunk, t1, t2, tot, &failed, parts);
    const bool bad = failed != 0;
    const float S1f = tot[0], S2f = tot[1];
    if (chunk == 0 && threadIdx.x == 0) {
        const float nanv = __uint_as_float(0x7fc00000u);
        p.sum_dy[c] = bad ? nanv : S1f;
        p.sum_dy_xmu[c] = bad ? nanv : S2f;
        if (p.dgamma) p.dgamma[c] = bad ? nanv : S2f * rsqrtf(p.var[c] + p.eps);
        if (p.dbeta) p.dbeta[c] = bad ? nanv : S1f;
    }

    // ---- phase 2: dx from the registers (bn_bwd_apply_kernel's expressions)
    const float mean_dy = S1f * p.inv_count;
    const float kk = S2f * p.inv_count * istd * istd;
    const float gi = bad ? __uint_as_float(0x7fc00000u) : gm * istd;
    const long long plane = p.dx_pitch ? (long long)(HW / p.W) * p.dx_pitch : HW;
#pragma unroll
    for (int k = 0; k < kFusedIters; ++k) {
        const int e = e0 + 4 * (int)threadIdx.x + 4 * kThreads * k;
        if (e < e1) {
            const long long row = (long long)nn[k] * C + c;
            float4 o;
            o.x = bwd_dx(g[k].x, xv[k].x, mu, mean_dy, kk, gi);
            o.y = bwd_dx(g[k].y, xv[k].y, mu, mean_dy, kk, gi);
            o.z = bwd_dx(g[k].z, xv[k].z, mu, mean_dy, kk, gi);
            o.w = bwd_dx(g[k].w, xv[k].w, mu, mean_dy, kk, gi);
            float* dxr = p.dx + row * plane;
            if (p.dx_pitch) { const int h = ii[k] / p.W; *reinterpret_cast<float4*>(dxr + h * p.dx_pitch + (ii[k] - h * p.W)) = o; }
            else *reinterpret_cast<float4*>(dxr + ii[k]) = o;
            if (DRES) *reinterpret_cast<float4*>(p.dres + row * HW + ii[k]) = g[k];
        }
    }
}

// ---- the same for blocks that lie inside ONE image (HW a multiple of the chunk, every chunk full: every BatchNorm of the
// model at 1024 x 2048): image, channel and chunk give block-uniform base pointers, a thread's eight float4 sit 4 KB
// apart from one lane offset, nothing is predicated - under 96 registers, FIVE blocks per CU instead of four.
template <int RELU, bool DRES, bool PITCH>
__global__ void __launch_bounds__(kThreads, 5) bn_bwd_fusedu_kernel(const BnFusedParams p) {
    __shared__ float red[4];
    __shared__ float tot[2];
    __shared__ int failed;
    __shared__ float parts[2 * kFusedMaxChunks];
    const int c = blockIdx.x / p.chunks, chunk = blockIdx.x - c * p.chunks;
    const int HW = p.HW, C = p.C;
    const float mu = p.mean[c];
    const float istd = 1.0f / sqrtf(p.var[c] + p.eps);
    const float gm = p.gamma[c];
    const float bt = RELU == 2 ? p.beta[c] : 0.f;
    const int e0 = chunk * kChunkElems;
    const int n = e0 / HW, i0 = e0 - n * HW;                     // block-uniform: the chunk lies inside image n
    const long long row = (long long)n * C + c;
    const float* dyb = p.dy + (long long)n * p.dy_nstride + (long long)c * HW + i0;
    const float* xb = p.x + row * HW + i0;
    const unsigned lo = 4u * threadIdx.x;
    if (threadIdx.x == 0) failed = 0;

    float4 g[kFusedIters], xv[kFusedIters];
#pragma unroll
    for (int k = 0; k < kFusedIters; ++k) {
        g[k] = *reinterpret_cast<const float4*>(dyb + 4 * kThreads * k + lo);
        xv[k] = *reinterpret_cast<const float4*>(xb + 4 * kThreads * k + lo);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kFusedIters; ++k) {
        if (RELU == 1) {
            const float4 yv = *reinterpret_cast<const float4*>(p.y + (long long)n * p.y_nstride + (long long)c * HW + i0 +
                                                               4 * kThreads * k + lo);
            g[k].x = yv.x > 0.f ? g[k].x : 0.f; g[k].y = yv.y > 0.f ? g[k].y : 0.f;
            g[k].z = yv.z > 0.f ? g[k].z : 0.f; g[k].w = yv.w > 0.f ? g[k].w : 0.f;
        } else if (RELU == 2) {
            g[k].x = bn_val(xv[k].x, mu, istd, gm, bt) > 0.f ? g[k].x : 0.f;
            g[k].y = bn_val(xv[k].y, mu, istd, gm, bt) > 0.f ? g[k].y : 0.f;
            g[k].z = bn_val(xv[k].z, mu, istd, gm, bt) > 0.f ? g[k].z : 0.f;
            g[k].w = bn_val(xv[k].w, mu, istd, gm, bt) > 0.f ? g[k].w : 0.f;
        } else if (RELU == 3) {
            // a wave iteration covers 256 consecutive pixels: words 4 * (i / 256) + 0..3 of the (n, c) row, bit = lane
            const unsigned long long* mw = reinterpret_cast<const unsigned long long*>(p.y) + row * (long long)(HW >> 6) +
                                           ((i0 + 4 * kThreads * k + 256 * (int)(threadIdx.x >> 6)) >> 8) * 4;
            const int sh = (int)(threadIdx.x & 63);
            g[k].x = ((mw[0] >> sh) & 1ull) ? g[k].x : 0.f; g[k].y = ((mw[1] >> sh) & 1ull) ? g[k].y : 0.f;
            g[k].z = ((mw[2] >> sh) & 1ull) ? g[k].z : 0.f; g[k].w = ((mw[3] >> sh) & 1ull) ? g[k].w : 0.f;
        }
        bwd_accumulate(g[k], xv[k], mu, s1, s2);
    }
    const float t1 = block_sum_256(s1, red);
    const float t2 = block_sum_256(s2, red);
    bn_fused_handoff(p, c, chunk, t1, t2, tot, &failed, parts);
    const bool bad = failed != 0;
    const float S1f = tot[0], S2f = tot[1];
    const float nanv = __uint_as_float(0x7fc00000u);
    if (chunk == 0 && threadIdx.x == 0) {
        p.sum_dy[c] = bad ? nanv : S1f;
        p.sum_dy_xmu[c] = bad ? nanv : S2f;
        if (p.dgamma) p.dgamma[c] = bad ? nanv : S2f * rsqrtf(p.var[c] + p.eps);
        if (p.dbeta) p.dbeta[c] = bad ? nanv : S1f;
    }
    const float mean_dy = S1f * p.inv_count;
    const float kk = S2f * p.inv_count * istd * istd;
    const float gi = bad ? nanv : gm * istd;
    float* dxb = p.dx + row * (PITCH ? (long long)(HW / p.W) * p.dx_pitch : (long long)HW) + (PITCH ? 0 : i0);
    float* drb = DRES ? p.dres + row * HW + i0 : nullptr;
#pragma unroll
    for (int k = 0; k < kFusedIters; ++k) {
        float4 o;
        o.x = bwd_dx(g[k].x, xv[k].x, mu, mean_dy, kk, gi);
        o.y = bwd_dx(g[k].y, xv[k].y, mu, mean_dy, kk, gi);
        o.z = bwd_dx(g[k].z, xv[k].z, mu, mean_dy, kk, gi);
        o.w = bwd_dx(g[k].w, xv[k].w, mu, mean_dy, kk, gi);
        if (PITCH) {
            const int i = i0 + 4 * kThreads * k + (int)lo;
            const int h = i / p.W;
            *reinterpret_cast<float4*>(dxb + h * p.dx_pitch + (i - h * p.W)) = o;
        } else {
            *reinterpret_cast<float4*>(dxb + 4 * kThreads * k + lo) = o;
        }
        if (DRES) *reinterpret_cast<float4*>(drb + 4 * kThreads * k + lo) = g[k];
    }
}

// elements of an (n, c) row per block of the apply kernels (DCFP_BN_COLS: multiples of 1024).  2048 since the end of round 4:
// shorter blocks stream faster on this chip - forward apply 15.97 -> 15.31 ms per step (5.37 -> 5.60 TB/s; 1024: 14.97, the step
// no better), 8192 / 16384 slower (profiles/r04_bn_cols_ab.txt)
static const int kColsPerBlock = [] { const char* e = getenv("DCFP_BN_COLS"); const int v = e ? atoi(e) : 2048; return v >= 1024 && v % 1024 == 0 ? v : 2048; }();

}  // namespace

extern "C" size_t dcfp_bn_workspace_bytes(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    const BnPlan p = bn_plan(N, C, HW);
    return (size_t)C * p.chunks * 2 * sizeof(float);
}

static int run_arg(const DcfpBnRunning* run, DcfpBnRunning* out) {
    DcfpBnRunning none = {nullptr, nullptr, nullptr, 0.f, 0};
    *out = run ? *run : none;
    if (out->running_mean && !out->running_var) return DCFP_E_BADDESC;
    return DCFP_OK;
}

extern "C" int dcfp_bn_stats_f32(const float* x, int64_t x_nstride, int N, int C, int HW,
                                 float* mean, float* var, const DcfpBnRunning* run, void* workspace,
                                 size_t workspace_bytes, dcfp_stream_t stream) {
    if (!x || !mean || !var || N <= 0 || C <= 0 || HW <= 0) return DCFP_E_BADDESC;
    DcfpBnRunning rn;
    if (run_arg(run, &rn)) return DCFP_E_BADDESC;
    if (x_nstride == 0) x_nstride = (int64_t)C * HW;
    const BnPlan p = bn_plan(N, C, HW);
    if (!workspace || workspace_bytes < (size_t)C * p.chunks * 2 * sizeof(float))
        return DCFP_E_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    const long long E = (long long)N * HW;
    const bool vec = (HW % 4 == 0) && (x_nstride % 4 == 0) && dcfp_aligned16(x);
    dim3 grid(p.chunks, C);
    if (vec)
        hipLaunchKernelGGL(bn_stats_partial_kernel<true>, grid, dim3(kThreads), 0, dcfp_s(stream), x,
                           (long long)x_nstride, HW, E, p.chunk_elems, p.chunks, part);
    else
        hipLaunchKernelGGL(bn_stats_partial_kernel<false>, grid, dim3(kThreads), 0, dcfp_s(stream),
                           x, (long long)x_nstride, HW, E, p.chunk_elems, p.chunks, part);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 255) / 256), dim3(256), 0, dcfp_s(stream),
                       x, HW, E, C, p.chunks, part, mean, var, rn);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_bn_apply_f32(const float* x, const float* mean, const float* var,
                                 const float* gamma, const float* beta, float eps,
                                 const float* residual, int relu, float* y, int64_t y_nstride,
                                 int N, int C, int HW, int W, int y_pitch, dcfp_stream_t stream) {
    if (!x || !mean || !var || !gamma || !beta || !y || N <= 0 || C <= 0 || HW <= 0)
        return DCFP_E_BADDESC;
    if (y_pitch && (W <= 0 || HW % W != 0 || y_pitch < W)) return DCFP_E_BADDESC;
    if (y_pitch == W) y_pitch = 0;
    if (y_nstride == 0) y_nstride = (int64_t)C * (y_pitch ? (int64_t)(HW / W) * y_pitch : HW);
    const int colchunks = (HW + kColsPerBlock - 1) / kColsPerBlock;
    const long long blocks = (long long)N * C * colchunks;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    const bool vec = (HW % 4 == 0) && (y_nstride % 4 == 0) && dcfp_aligned16(x) &&
                     dcfp_aligned16(y) && (!residual || dcfp_aligned16(residual)) &&
                     (!y_pitch || (W % 4 == 0 && y_pitch % 4 == 0));
    if (vec)
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), x, mean, var, gamma, beta, eps, residual, relu, y,
                           (long long)y_nstride, C, HW, colchunks, kColsPerBlock, nullptr, W, y_pitch);
    else
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           dcfp_s(stream), x, mean, var, gamma, beta, eps, residual, relu, y,
                           (long long)y_nstride, C, HW, colchunks, kColsPerBlock, nullptr, W, y_pitch);
    DCFP_RETURN_LAUNCH();
}

// y = relu(BN(x) [+ residual]) and the 1-bit-per-element ReLU mask the backward kernels take with relu == 3
// (through their `y` argument) instead of re-reading y.  Needs HW % 256 == 0 and 16-byte aligned rows.
extern "C" int dcfp_bn_apply_relu_mask_f32(const float* x, const float* mean, const float* var,
                                           const float* gamma, const float* beta, float eps,
                                           const float* residual, float* y, void* relu_mask,
                                           int N, int C, int HW, dcfp_stream_t stream) {
    if (!x || !mean || !var || !gamma || !beta || !y || !relu_mask || N <= 0 || C <= 0 || HW <= 0)
        return DCFP_E_BADDESC;
    const bool vec = dcfp_aligned16(x) && dcfp_aligned16(y) && (!residual || dcfp_aligned16(residual));
    if (!vec || HW % 256 != 0 || (reinterpret_cast<uintptr_t>(relu_mask) & 7u)) return DCFP_E_UNSUPPORTED;
    const int colchunks = (HW + kColsPerBlock - 1) / kColsPerBlock;
    const long long blocks = (long long)N * C * colchunks;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    hipLaunchKernelGGL((bn_apply_kernel<true, true>), dim3((unsigned)blocks), dim3(kThreads), 0, dcfp_s(stream),
                       x, mean, var, gamma, beta, eps, residual, 1, y, (long long)C * HW, C, HW, colchunks,
                       kColsPerBlock, static_cast<unsigned long long*>(relu_mask));
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_bn_bwd_reduce_f32(const float* dy, int64_t dy_nstride, const float* x,
                                      const float* y, int64_t y_nstride, const float* mean,
                                      const float* var, const float* gamma, const float* beta,
                                      float eps, int relu, int N, int C, int HW, float* sum_dy,
                                      float* sum_dy_xmu, float* dgamma, float* dbeta, void* workspace,
                                      size_t workspace_bytes, dcfp_stream_t stream) {
    if (dgamma && !var) return DCFP_E_BADDESC;
    if (!dy || !x || !mean || !sum_dy || !sum_dy_xmu || N <= 0 || C <= 0 || HW <= 0)
        return DCFP_E_BADDESC;
    if (relu < 0 || relu > 3) return DCFP_E_BADDESC;
    if ((relu == 1 || relu == 3) && !y) return DCFP_E_BADDESC;
    if (relu == 2 && (!var || !gamma || !beta)) return DCFP_E_BADDESC;
    if (dy_nstride == 0) dy_nstride = (int64_t)C * HW;
    if (y_nstride == 0) y_nstride = (int64_t)C * HW;
    const BnPlan p = bn_plan(N, C, HW);
    if (!workspace || workspace_bytes < (size_t)C * p.chunks * 2 * sizeof(float))
        return DCFP_E_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    const long long E = (long long)N * HW;
    const bool vec = (HW % 4 == 0) && (dy_nstride % 4 == 0) && (y_nstride % 4 == 0) &&
                     dcfp_aligned16(dy) && dcfp_aligned16(x) && (relu != 1 || dcfp_aligned16(y));
    dim3 grid(p.chunks, C);
#define LAUNCH_RED(V, R)                                                                          \
    hipLaunchKernelGGL((bn_bwd_reduce_partial_kernel<V, R>), grid, dim3(kThreads), 0,             \
                       dcfp_s(stream), dy, (long long)dy_nstride, x, y, (long long)y_nstride,     \
                       mean, var, gamma, beta, eps, C, HW, E, p.chunk_elems, p.chunks, part)
    if (relu == 3 && !(vec && HW % 256 == 0 && dy_nstride == (int64_t)C * HW)) return DCFP_E_UNSUPPORTED;
    if (vec) { if (relu == 3) LAUNCH_RED(true, 3); else if (relu == 2) LAUNCH_RED(true, 2); else if (relu == 1) LAUNCH_RED(true, 1); else LAUNCH_RED(true, 0); }
    else     { if (relu == 2) LAUNCH_RED(false, 2); else if (relu == 1) LAUNCH_RED(false, 1); else LAUNCH_RED(false, 0); }
#undef LAUNCH_RED
    hipLaunchKernelGGL(bn_pair_final_kernel, dim3((unsigned)C), dim3(64), 0, dcfp_s(stream), C,
                       p.chunks, part, sum_dy, sum_dy_xmu, var, eps, dgamma, dbeta);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_bn_update_running_f32(const float* mean, const float* var, int C, float momentum,
                                          float count, const float* count_dev, float* running_mean,
                                          float* running_var, dcfp_stream_t stream) {
    if (!mean || !var || !running_mean || !running_var || C <= 0 || (!count_dev && !(count > 0.f)))
        return DCFP_E_BADDESC;
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, dcfp_s(stream), C, mean,
                       var, momentum, count, count_dev, running_mean, running_var);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_bn_bwd_apply_f32(const float* dy, int64_t dy_nstride, const float* x,
                                     const float* y, int64_t y_nstride, const float* mean,
                                     const float* var, const float* gamma, const float* beta,
                                     float eps, const float* sum_dy, const float* sum_dy_xmu,
                                     float count, const float* count_dev, int relu, float* dx,
                                     float* d_residual, int N, int C, int HW, int W, int dx_pitch,
                                     dcfp_stream_t stream) {
    if (!dy || !x || !mean || !var || !gamma || !sum_dy || !sum_dy_xmu || !dx || N <= 0 ||
        C <= 0 || HW <= 0 || (!count_dev && !(count > 0.f)))
        return DCFP_E_BADDESC;
    if (relu < 0 || relu > 3 || ((relu == 1 || relu == 3) && !y) || (relu == 2 && !beta)) return DCFP_E_BADDESC;
    if (dx_pitch && (W <= 0 || HW % W != 0 || dx_pitch < W)) return DCFP_E_BADDESC;
    if (dx_pitch == W) dx_pitch = 0;
    if (dy_nstride == 0) dy_nstride = (int64_t)C * HW;
    if (y_nstride == 0) y_nstride = (int64_t)C * HW;
    const int colchunks = (HW + kColsPerBlock - 1) / kColsPerBlock;
    const long long blocks = (long long)N * C * colchunks;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    const bool vec = (HW % 4 == 0) && (dy_nstride % 4 == 0) && (y_nstride % 4 == 0) &&
                     dcfp_aligned16(dy) && dcfp_aligned16(x) && dcfp_aligned16(dx) &&
                     (relu != 1 || dcfp_aligned16(y)) && (!d_residual || dcfp_aligned16(d_residual)) &&
                     (!dx_pitch || (W % 4 == 0 && dx_pitch % 4 == 0));
    const float inv_count = count > 0.f ? 1.0f / count : 0.f;
#define LAUNCH_APP(V, R)                                                                          \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<V, R>), dim3((unsigned)blocks), dim3(kThreads), 0,    \
                       dcfp_s(stream), dy, (long long)dy_nstride, x, y, (long long)y_nstride,     \
                       mean, var, gamma, beta, eps, sum_dy, sum_dy_xmu, inv_count, count_dev, dx, \
                       d_residual, C,                                                             \
                       HW, colchunks, kColsPerBlock, W, dx_pitch)
    if (relu == 3 && !(vec && HW % 256 == 0)) return DCFP_E_UNSUPPORTED;
    if (vec) { if (relu == 3) LAUNCH_APP(true, 3); else if (relu == 2) LAUNCH_APP(true, 2); else if (relu == 1) LAUNCH_APP(true, 1); else LAUNCH_APP(true, 0); }
    else     { if (relu == 2) LAUNCH_APP(false, 2); else if (relu == 1) LAUNCH_APP(false, 1); else LAUNCH_APP(false, 0); }
#undef LAUNCH_APP
    DCFP_RETURN_LAUNCH();
}

// Bytes of the hand-off buffer of dcfp_bn_bwd_fused_f32 for this shape (0: the shape is not supported by the fused kernel)
extern "C" size_t dcfp_bn_bwd_fused_sync_bytes(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || HW % 4 != 0) return 0;
    const long long E = (long long)N * HW;
    if (E >= 0x7fffffffLL - kChunkElems) return 0;
    const BnPlan p = bn_plan(N, C, HW);
    if (p.chunks > kFusedMaxChunks || p.chunk_elems > kChunkElems) return 0;
    if ((long long)C * p.chunks > 0x7fffffffLL) return 0;
    return (size_t)C * (p.chunks * 2 + 2) * sizeof(unsigned long long);     // per channel: the partials + the two totals
}

extern "C" int dcfp_bn_bwd_fused_f32(const float* dy, int64_t dy_nstride, const float* x, const float* y,
                                     int64_t y_nstride, const float* mean, const float* var, const float* gamma,
                                     const float* beta, float eps, float count, int relu, float* dx, float* d_residual,
                                     int N, int C, int HW, int W, int dx_pitch, float* sum_dy, float* sum_dy_xmu,
                                     float* dgamma, float* dbeta, void* sync, size_t sync_bytes, uint32_t epoch,
                                     uint32_t spin_limit, int32_t* status, dcfp_stream_t stream) {
    if (!dy || !x || !mean || !var || !gamma || !dx || !sum_dy || !sum_dy_xmu || !sync || N <= 0 || C <= 0 || HW <= 0 ||
        !(count > 0.f) || epoch == 0)
        return DCFP_E_BADDESC;
    if (relu < 0 || relu > 3 || ((relu == 1 || relu == 3) && !y) || (relu == 2 && !beta)) return DCFP_E_BADDESC;
    if (dx_pitch && (W <= 0 || HW % W != 0 || dx_pitch < W)) return DCFP_E_BADDESC;
    if (dx_pitch == W) dx_pitch = 0;
    if (dy_nstride == 0) dy_nstride = (int64_t)C * HW;
    if (y_nstride == 0) y_nstride = (int64_t)C * HW;
    const size_t need = dcfp_bn_bwd_fused_sync_bytes(N, C, HW);
    if (need == 0) return DCFP_E_UNSUPPORTED;
    if (sync_bytes < need || (reinterpret_cast<uintptr_t>(sync) & 7u)) return DCFP_E_WORKSPACE;
    const bool vec = (dy_nstride % 4 == 0) && (y_nstride % 4 == 0) && dcfp_aligned16(dy) && dcfp_aligned16(x) &&
                     dcfp_aligned16(dx) && (relu != 1 || dcfp_aligned16(y)) &&
                     (!d_residual || dcfp_aligned16(d_residual)) && (!dx_pitch || (W % 4 == 0 && dx_pitch % 4 == 0));
    if (!vec) return DCFP_E_UNSUPPORTED;
    if (relu == 3 && !(HW % 256 == 0 && dy_nstride == (int64_t)C * HW)) return DCFP_E_UNSUPPORTED;
    const BnPlan pl = bn_plan(N, C, HW);
    BnFusedParams p;
    p.dy = dy; p.dy_nstride = dy_nstride; p.x = x; p.y = y; p.y_nstride = y_nstride;
    p.mean = mean; p.var = var; p.gamma = gamma; p.beta = beta; p.eps = eps; p.inv_count = 1.0f / count;
    p.dx = dx; p.dres = d_residual; p.C = C; p.HW = HW; p.E = (int)((long long)N * HW);
    p.chunk_elems = pl.chunk_elems; p.chunks = pl.chunks; p.W = W; p.dx_pitch = dx_pitch;
    p.sync = static_cast<unsigned long long*>(sync); p.epoch = epoch;
    p.spin_limit = spin_limit ? spin_limit : (1u << 18); p.status = status;
    p.sum_dy = sum_dy; p.sum_dy_xmu = sum_dy_xmu; p.dgamma = dgamma; p.dbeta = dbeta;
    const size_t lds = 0;
    const dim3 grid((unsigned)((long long)C * pl.chunks));
    // blocks inside one image, every chunk full: the low-register kernel (DCFP_BN_FUSED_UNI=0: the general one)
    static const bool uni_on = [] { const char* e = getenv("DCFP_BN_FUSED_UNI"); return !e || atoi(e) != 0; }();
    if (uni_on && pl.chunk_elems == kChunkElems && HW % kChunkElems == 0) {
#define LAUNCH_FUSEDU2(R, D, P) hipLaunchKernelGGL((bn_bwd_fusedu_kernel<R, D, P>), grid, dim3(kThreads), lds, dcfp_s(stream), p)
#define LAUNCH_FUSEDU(R)                                                                                       \
    do {                                                                                                       \
        if (d_residual) { if (dx_pitch) LAUNCH_FUSEDU2(R, true, true); else LAUNCH_FUSEDU2(R, true, false); }  \
        else { if (dx_pitch) LAUNCH_FUSEDU2(R, false, true); else LAUNCH_FUSEDU2(R, false, false); }           \
    } while (0)
        if (relu == 3) LAUNCH_FUSEDU(3); else if (relu == 2) LAUNCH_FUSEDU(2); else if (relu == 1) LAUNCH_FUSEDU(1); else LAUNCH_FUSEDU(0);
#undef LAUNCH_FUSEDU
#undef LAUNCH_FUSEDU2
        DCFP_RETURN_LAUNCH();
    }
#define LAUNCH_FUSED(R)                                                                                        \
    do {                                                                                                       \
        if (d_residual) hipLaunchKernelGGL((bn_bwd_fused_kernel<R, true>), grid, dim3(kThreads), lds, dcfp_s(stream), p);  \
        else hipLaunchKernelGGL((bn_bwd_fused_kernel<R, false>), grid, dim3(kThreads), lds, dcfp_s(stream), p);            \
    } while (0)
    if (relu == 3) LAUNCH_FUSED(3); else if (relu == 2) LAUNCH_FUSED(2); else if (relu == 1) LAUNCH_FUSED(1); else LAUNCH_FUSED(0);
#undef LAUNCH_FUSED
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_syncbn_combine_f32(const float* gathered, int world, int C, float* mean, float* var,
                                       float* total_count, const DcfpBnRunning* run, dcfp_stream_t stream) {
    if (!gathered || !mean || !var || !total_count || world <= 0 || C <= 0) return DCFP_E_BADDESC;
    DcfpBnRunning rn;
    if (run_arg(run, &rn)) return DCFP_E_BADDESC;
    hipLaunchKernelGGL(syncbn_combine_kernel, dim3((C + 255) / 256), dim3(256), 0, dcfp_s(stream),
                       gathered, world, C, mean, var, total_count, rn);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_bn_stats_from_partials_f32(const float* partials, int64_t slots, int slot_count, int C,
                                               float* mean, float* var, const DcfpBnRunning* run,
                                               dcfp_stream_t stream) {
    if (!partials || !mean || !var || slots <= 0 || slot_count <= 0 || C <= 0) return DCFP_E_BADDESC;
    DcfpBnRunning rn;
    if (run_arg(run, &rn)) return DCFP_E_BADDESC;
    if (reinterpret_cast<uintptr_t>(partials) & 7u) return DCFP_E_BADDESC;
    // 8 channels per block at every width (round 4: 128 threads per channel, two loads deep - 13.4 -> 8.9 us on 256 channels
    // x 1024 slots against the 32-channel blocks, profiles/r04_sfp_ab.txt; 98 launches per step)
    hipLaunchKernelGGL(bn_stats_from_partials_kernel<8>, dim3((unsigned)((C + 7) / 8)), dim3(kSfpThreads), 0,
                       dcfp_s(stream), partials, (long long)slots, slot_count, C, mean, var, rn);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_bn_bwd_sums_from_partials_f32(const float* partials, int64_t slots, int C, const float* var, float eps,
                                                  float* sum_dy, float* sum_dy_xmu, float* dgamma, float* dbeta,
                                                  dcfp_stream_t stream) {
    if (!partials || slots <= 0 || C <= 0 || !sum_dy || !sum_dy_xmu || (dgamma && !var)) return DCFP_E_BADDESC;
    if (reinterpret_cast<uintptr_t>(partials) & 7u) return DCFP_E_BADDESC;
    hipLaunchKernelGGL(bn_bwd_sums_from_partials_kernel, dim3((unsigned)((C + 31) / 32)), dim3(kSfpThreads), 0,
                       dcfp_s(stream), partials, (long long)slots, C, var, eps, sum_dy, sum_dy_xmu, dgamma, dbeta);
    DCFP_RETURN_LAUNCH();
}

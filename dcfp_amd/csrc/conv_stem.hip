// conv_stem.hip — the first conv of the deep stem: 3 -> 64 channels, 3x3, stride 2, pad 1 on the full-resolution image
// (networks/backbone/resnet.py:88-90, `backbone.conv1.0`; SURVEY K3: "special-case the Cin = 3 stem, HBM-bound").
//
// K = Cin x 9 = 27: the general implicit-GEMM kernels pad it to their 16-deep K-steps and stage a stride-2 im2col through LDS
// (0.61 ms forward at 12 TF, 0.36 ms weight gradient at 20 TF for 4 x 3 x 1024 x 2048: profiles/r03_conv_entries_below_125tf.txt)
// although the conv moves 638 MB for 7.2 GFLOP, i.e. is bound by HBM (0.1 ms).  Here K = 27 (+1 zero) is 14 K-pairs of
// v_mfma_f32_32x32x2_f32 without im2col and without workgroup-level staging (a wave parks data only in its own LDS region):
//   forward   y[co][p] = sum_t w[co][t] x_t[p]:  A = the 64 x 28 weights, held in 28 registers per lane for the whole kernel
//             (lane = (co % 32, K-pair half)); the other operand = the 28 taps of 32 consecutive output pixels of a row,
//             lane = (pixel, half), read from the segment's input window that the wave stages once in LDS; 28 MFMAs per 32
//             pixels; pixels along M, so a lane holds 4 consecutive pixels of a channel: 16-byte stores.
//   wgrad     dw[co][t] = sum_p dy[co][p] x_t[p]:  K = pixels.  A = dy, lane = (co % 32, pixel half): one 16-byte load covers the
//             lane's 4 pixels of an 8-pixel group; B = x_t[p], lane = (tap, pixel half): 4 dword loads; 8 MFMAs per 8 pixels.
//             A wave owns a contiguous run of 8-pixel groups and leaves its 64 x 27 partial in a slab; stem_wgrad_reduce sums
//             the slabs in wave order (fixed order, no atomics).
// fp32 multiplicands, fp32 accumulation on the fp32 MFMA, as everywhere.  Padding = out-of-range buffer offsets (read zeros).
#include "igemm2_common.h"
#include <stdlib.h>

namespace {

struct StemParams {
    const float* x;     // [N][Cin][H][W] dense
    const float* w;     // [64][Cin][3][3]
    const float* dy;    // wgrad: [N][64][Ho][Wo], images dy_nstride floats apart
    float* y;           // forward: [N][64][Ho][Wo], images y_nstride floats apart
    float* part;        // wgrad: [waves][64][27]
    float* stat;        // forward, nullable: BatchNorm partials [slot][64][2] = (mean, M2) over 128 pixels
    const float* bias;  // nullable
    long long y_nstride, dy_nstride;
    int N, Cin, H, W, Ho, Wo, T;      // T = Cin * 9 <= 32
    int groups, per_wave;             // wgrad: 32-pixel chunks in all, chunks per wave
    unsigned x_bytes;
};

// tap t of a lane: channel plane offset + row + column shift (floats), and its (kh, kw)
__device__ __forceinline__ void tap_of(int t, int T, int H, int W, int& off, int& kh, int& kw, bool& live) {
    live = t < T;
    const int tt = live ? t : 0;
    const int ci = tt / 9, r = tt - ci * 9;
    kh = r / 3; kw = r - kh * 3;
    off = ci * H * W + kh * W + kw;
}

// Forward.  block = 4 waves; a wave takes 128 consecutive output pixels of a row in four sub-steps of 32, four such segments in a row.  MFMA roles: M = pixels (A = the taps of this lane's pixel), N = output channels (B = the weights): a
// lane then holds 4 CONSECUTIVE pixels of one channel per accumulator quad - 16-byte stores - and, over the wave's 128 pixels,
// everything the BatchNorm behind the conv needs from that channel: STATS writes (mean, M2) per channel and 128 pixels in the
// layout dcfp_bn_stats_from_partials_f32 merges (part[slot][64][2], slot = linear pixel / 128), combined sub-step by sub-step
// with Chan's formula in a fixed order.
constexpr int kStemSeg = 4;
constexpr int kXRow = 264;                 // staged floats per (channel, kernel row): columns 2 ox0 - 4 ... 2 ox0 + 259
constexpr int kXPitch = kXRow + 4;         // + 4: the 9 rows start 8 banks apart
// The taps of a segment's 128 pixels overlap (27 taps per pixel, 2.3 distinct input floats per pixel): the wave stages the
// segment's input window ONCE - 9 (channel, kernel row) rows of 264 floats, 16-byte loads along the row, 10 per lane - in its
// own 9.4 KB of LDS and reads the taps from there (a first version loaded every tap from global memory: 56 scattered dword
// loads per lane and segment, 0.30 ms - the texture path, not HBM, was the limit).  The next segment's loads are in flight
// while this one computes.
template <bool STATS>
__global__ void __launch_bounds__(256) stem_fwd_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(16))) float xs_all[4][9 * kXPitch];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    float* xs = xs_all[wid];
    // a wave walks kStemSeg segments of 128 output pixels (the 28 weights per lane are loaded once per wave): segment
    // s = (row, 128-pixel column block), rows = n * Ho + oy
    const int spr = (p.Wo + 127) >> 7;                          // segments per output row
    const long long nseg = (long long)p.N * p.Ho * spr;
    const long long seg0 = ((long long)blockIdx.x * 4 + wid) * kStemSeg;
    if (seg0 >= nseg) return;                                   // wave-uniform
    // B: weights of output channels l31 and 32 + l31, K-pair j = taps (2j, 2j + 1), this lane holds tap 2j + lhi
    float wb[2][14];
    int xoff[14];                                               // LDS offset of tap 2j + lhi for pixel 0 of a sub-step
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int t = 2 * j + lhi;
        wb[0][j] = t < p.T ? p.w[l31 * p.T + t] : 0.f;
        wb[1][j] = t < p.T ? p.w[(32 + l31) * p.T + t] : 0.f;
        const int tt = t < p.T ? t : 0;                         // (a dead tap multiplies a zero weight: any staged value will do)
        const int rc = tt / 3, kw = tt - rc * 3;                // rc = ci * 3 + kh
        xoff[j] = rc * kXPitch + 3 + kw + 2 * l31;              // column 2 (ox - ox0) - 1 + kw, window starting at 2 ox0 - 4
    }
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const long long plane = (long long)p.Ho * p.Wo;
    const int nrc = p.Cin * 3, nchunk = nrc * (kXRow / 4);      // 16-byte chunks of a segment's window
    f32x4 stage[10];
    auto fetch = [&](long long seg) {                           // this lane's chunks lane, lane + 64, ... of segment `seg`
        const int rowi = (int)(seg / spr), ox0 = (int)(seg - (long long)rowi * spr) * 128;
        const int n = rowi / p.Ho, oy = rowi - n * p.Ho;
        const int img = n * p.Cin * p.H * p.W;                  // (32-bit: the tensor is below 2^31 bytes, dcfp_stem_shape)
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = lane + 64 * i;
            const int rc = c / (kXRow / 4), q = c - rc * (kXRow / 4);
            const int ci = rc / 3, kh = rc - ci * 3;
            const int h = 2 * oy - 1 + kh, col = 2 * ox0 - 4 + 4 * q;          // W % 4 == 0: a chunk is inside or outside as a whole
            const bool ok = c < nchunk && (unsigned)h < (unsigned)p.H && (unsigned)col < (unsigned)p.W;
            const unsigned vo = ok ? (unsigned)(img + (ci * p.H + h) * p.W + col) * 4u : kOob;
            stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, vo, 0, 0));
        }
    };
    fetch(seg0);
  for (int sg = 0; sg < kStemSeg; ++sg) {
    const long long seg = seg0 + sg;
    if (seg >= nseg) break;                                     // wave-uniform
    const int rowi = (int)(seg / spr), ox0 = (int)(seg - (long long)rowi * spr) * 128;
    const int n = rowi / p.Ho, oy = rowi - n * p.Ho;
    // park the window (the wave's own LDS: no barrier, the previous segment's tap reads are complete - their values fed MFMAs)
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int c = lane + 64 * i;
        const int rc = c / (kXRow / 4), q = c - rc * (kXRow / 4);
        if (c < nchunk) *reinterpret_cast<f32x4*>(xs + rc * kXPitch + 4 * q) = stage[i];
    }
    if (sg + 1 < kStemSeg && seg + 1 < nseg) fetch(seg + 1);
    float* orow = p.y + (long long)n * p.y_nstride + (long long)oy * p.Wo;
    float rmean[2] = {0.f, 0.f}, rm2[2] = {0.f, 0.f};
    static_for<0, 4>([&](auto it_) {
        constexpr int it = decltype(it_)::value;
        float xa[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) xa[j] = xs[xoff[j] + 64 * it];
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j], wb[0][j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j], wb[1][j], acc[1], 0, 0, 0);
        }
        const int oxs = ox0 + 32 * it;                           // (Wo % 32 == 0: a quad is inside or outside as a whole)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = 32 * i + l31;
            const float b = p.bias ? p.bias[co] : 0.f;
            float* o = orow + (long long)co * plane + oxs + 4 * lhi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[i][4 * q] + b, acc[i][4 * q + 1] + b, acc[i][4 * q + 2] + b, acc[i][4 * q + 3] + b};
                if (oxs + 8 * q + 4 * lhi < p.Wo) *reinterpret_cast<f32x4*>(o + 8 * q) = v;
            }
            if constexpr (STATS) {           // (only launched with Wo % 128 == 0 and no bias: every pixel of the wave exists)
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][r];
                s += __shfl_xor(s, 32, 64);
                const float m32 = s * (1.0f / 32.0f);
                float m2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float dlt = acc[i][r] - m32; m2 = fmaf(dlt, dlt, m2); }
                m2 += __shfl_xor(m2, 32, 64);
                if constexpr (it == 0) { rmean[i] = m32; rm2[i] = m2; }
                else {                       // Chan: (it * 32 pixels so far) + (32 new ones)
                    constexpr float na = 32.0f * it, nb = 32.0f, nn = na + nb;
                    const float dlt = m32 - rmean[i];
                    rmean[i] += dlt * (nb / nn);
                    rm2[i] += m2 + dlt * dlt * (na * nb / nn);
                }
            }
        }
    });
    if constexpr (STATS) {
        if (lhi == 0) {
            const long long slot = ((long long)(n * p.Ho + oy) * p.Wo + ox0) >> 7;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float* sp = p.stat + (slot * 64 + 32 * i + l31) * 2;
                sp[0] = rmean[i]; sp[1] = rm2[i];
            }
        }
    }
  }
}

// Weight gradient.  block = 4 waves; wave gw = blockIdx.x * 4 + wid owns the 32-pixel chunks gw * per_wave ... (Wo % 32 == 0: a
// chunk = 32 consecutive output pixels of one row).  dy of a chunk ([64 channels][32 pixels], 8 KB) is loaded with the lanes
// ALONG the pixels (whole 128-byte lines: this 537 MB stream is the kernel's traffic), parked in the wave's own 9 KB of LDS
// and read back in the MFMA layout (lane = channel, 4 pixels per 16-byte read); the next chunk's loads are in flight meanwhile.
constexpr int kDyPitch = 36;                                      // floats per channel row in LDS (32 + 4: conflict-free 16-byte reads)
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(16))) float lds[4][64 * kDyPitch];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int gw = blockIdx.x * 4 + wid;
    int g = gw * p.per_wave;
    int gend = g + p.per_wave;
    if (gend > p.groups) gend = p.groups;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    int toff, kh, kw; bool live;
    tap_of(l31, p.T, p.H, p.W, toff, kh, kw, live);
    const int cpr = p.Wo >> 5;                                  // chunks per output row
    const long long plane = (long long)p.Ho * p.Wo;
    float* my = lds[wid];
    const int q8 = lane & 7, c8 = lane >> 3;                     // staging: pixel quad q8 of channels c8, c8 + 8, ...
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 stage[8];
    float xv[16];
    auto fetch = [&](int gg) {                                   // global loads of chunk gg: dy (coalesced) and this lane's tap of x
        const int row = gg / cpr, gx = gg - row * cpr;           // row = n * Ho + oy
        const int n = row / p.Ho, oy = row - n * p.Ho;
        const float* dyp = p.dy + (long long)n * p.dy_nstride + (long long)oy * p.Wo + gx * 32 + 4 * q8;
#pragma unroll
        for (int i = 0; i < 8; ++i) stage[i] = *reinterpret_cast<const f32x4*>(dyp + (long long)(c8 + 8 * i) * plane);
        const bool rok = live && (unsigned)(2 * oy - 1 + kh) < (unsigned)p.H;
        const long long base = (long long)n * p.Cin * p.H * p.W + toff + (long long)(2 * oy - 1) * p.W + (2 * (gx * 32) - 1);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int px = 8 * k4 + 4 * lhi + e;             // pixel of the chunk: K-group k4, this lane's half
                const int ww = 2 * (gx * 32 + px) - 1 + kw;
                const bool ok = rok && (unsigned)ww < (unsigned)p.W;
                const unsigned vo = ok ? (unsigned)((base + 2 * px) * 4) : kOob;
                xv[4 * k4 + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, vo, 0, 0));
            }
    };
    if (g < gend) fetch(g);
    for (; g < gend; ++g) {
        // park dy in LDS (the wave's own region: no barrier, only its own LDS counter)
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(my + (c8 + 8 * i) * kDyPitch + 4 * q8) = stage[i];
        float xc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) xc[k] = xv[k];
        if (g + 1 < gend) fetch(g + 1);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(my + l31 * kDyPitch + 8 * k4 + 4 * lhi);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(my + (32 + l31) * kDyPitch + 8 * k4 + 4 * lhi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0[e], xc[4 * k4 + e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[e], xc[4 * k4 + e], acc[1], 0, 0, 0);
            }
        }
    }
    if (l31 >= p.T) return;
    float* o = p.part + (long long)gw * 64 * p.T + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            o[co * p.T] = acc[i][r];
        }
}

// dw[i] = sum over the waves' slabs in two levels, each in ascending order (fixed order, no atomics): level 1 sums runs of 64
// slabs (grid.y = waves / 64; consecutive threads read consecutive floats, 8 loads in flight), level 2 the run sums.
__global__ void __launch_bounds__(256) stem_wgrad_reduce_kernel(const float* __restrict__ part, int count, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* src = part + (long long)blockIdx.y * count * n + i;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= count; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < count; ++k) s += src[(long long)k * n];
    out[(long long)blockIdx.y * n + i] = s;
}

constexpr int kStemSegHost = 4;
constexpr int kStemWaves = 4096;      // wgrad: 1024 workgroups of 4 waves, 4 waves per SIMD

}  // namespace

// DCFP_CONV_STEM: 0 off (the general kernels), 1 on (default)
bool dcfp_stem_shape(const DcfpConvDesc* d) {
    static const int on = [] { const char* e = getenv("DCFP_CONV_STEM"); return e ? atoi(e) : 1; }();
    if (!on) return false;
    if (d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1 || d->dil != 1) return false;
    if (d->Cin * 9 > 28 || d->Cout != 64) return false;
    if (d->x_pitch && d->x_pitch != d->W) return false;
    if (d->dy_pitch && d->dy_pitch != d->Wout) return false;
    if (d->Wout % 32 != 0 || d->W % 4 != 0) return false;
    return (long long)d->N * d->Cin * d->H * d->W * 4 < 0x7fffff00LL;
}

// (mean, M2) partials over 128 pixels each that the forward can emit for the BatchNorm behind the conv, or 0
long long dcfp_stem_stat_slots(const DcfpConvDesc* d, const float* y, long long y_nstride) {
    if (d->Wout % 128 != 0) return 0;
    (void)y; (void)y_nstride;
    return (long long)d->N * d->Hout * d->Wout / 128;
}

int dcfp_stem_fwd(const DcfpConvDesc* d, const float* x, const float* w, const float* bias, float* y, long long y_nstride,
                  hipStream_t stream, float* stat_part) {
    StemParams p = {};
    p.x = x; p.w = w; p.y = y; p.bias = bias; p.stat = stat_part;
    p.y_nstride = y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout;
    if (p.y_nstride % 4 != 0 || !dcfp_aligned16(y)) return DCFP_E_UNSUPPORTED;
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Ho = d->Hout; p.Wo = d->Wout; p.T = d->Cin * 9;
    p.x_bytes = (unsigned)((long long)d->N * d->Cin * d->H * d->W * 4);
    if (stat_part && (bias || dcfp_stem_stat_slots(d, y, y_nstride) <= 0)) return DCFP_E_UNSUPPORTED;
    const long long nseg = (long long)d->N * d->Hout * ((d->Wout + 127) / 128);
    const dim3 grid((unsigned)((nseg + 4 * kStemSegHost - 1) / (4 * kStemSegHost)));
    if (stat_part) hipLaunchKernelGGL(stem_fwd_kernel<true>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(stem_fwd_kernel<false>, grid, dim3(256), 0, stream, p);
    DCFP_RETURN_LAUNCH();
}

size_t dcfp_stem_wgrad_workspace_bytes(const DcfpConvDesc* d) {
    return (size_t)(kStemWaves + kStemWaves / 64) * 64 * d->Cin * 9 * sizeof(float);      // wave slabs + the level-1 sums
}

int dcfp_stem_wgrad(const DcfpConvDesc* d, const float* dy, long long dy_nstride, const float* x, float* dw, void* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
    if (!workspace || workspace_bytes < dcfp_stem_wgrad_workspace_bytes(d)) return DCFP_E_WORKSPACE;
    StemParams p = {};
    p.x = x; p.dy = dy; p.part = static_cast<float*>(workspace);
    p.dy_nstride = dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * d->Wout;
    if (p.dy_nstride % 4 != 0 || !dcfp_aligned16(dy)) return DCFP_E_UNSUPPORTED;
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Ho = d->Hout; p.Wo = d->Wout; p.T = d->Cin * 9;
    p.x_bytes = (unsigned)((long long)d->N * d->Cin * d->H * d->W * 4);
    const long long groups = (long long)d->N * d->Hout * (d->Wout / 32);
    if (groups >= (1LL << 31)) return DCFP_E_UNSUPPORTED;
    p.groups = (int)groups;
    p.per_wave = (int)((groups + kStemWaves - 1) / kStemWaves);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(kStemWaves / 4), dim3(256), 0, stream, p);
    const int n = 64 * p.T;
    float* lvl1 = p.part + (long long)kStemWaves * n;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256), kStemWaves / 64), dim3(256), 0, stream, p.part, 64,
                       n, lvl1);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, stream, lvl1, kStemWaves / 64, n, dw);
    DCFP_RETURN_LAUNCH();
}

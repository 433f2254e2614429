// conv_wgrad.hip — conv2d weight gradient on the fp32 matrix cores of gfx950.
// Replaces the wgrad half of autograd's convolution_backward for every nn.Conv2d on
// the hot path (networks/backbone/resnet.py:25-30,88-96; networks/tools/aspp.py:13-14;
// networks/deeplabv3.py:25-33,37-41).
//
//   dW[m][n] = sum_{img, p} dY[img][m][p] * X[img][ci][src(p, tap)]     m = co, n = ci*T + tap
// A GEMM with tiny M x N (e.g. 256 x 2304) and a huge reduction (K = N*Hout*Wout =
// 131072 pixels at 4x128x256): the K range is split across workgroups, each split
// writes its fp32 partial slab, and a second kernel sums the slabs in a FIXED order —
// no float atomics, so BN-gamma / weight gradients are run-to-run reproducible
// (SURVEY.md §7 hard parts: jitter would flip masks near the threshold).
//
// Both operands are pixel-contiguous in NCHW, so LDS keeps them K-contiguous:
// As[m][k], Bs[n][k] (row pitch 20 floats: conflict-free ds_read_b128), and one
// 16-byte read per lane feeds FOUR consecutive k-steps of a 32x32x2 MFMA tile.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// EXPERIMENTAL opt-in: DCFP_CONV_MATH=bf16x3 routes qualifying wgrad shapes to conv_wgrad3.hip
int dcfp_wgrad3_launch(const float* dy, long long dy_nstride, const float* x, long long x_nstride,
                       float* out, int N, int M, int Cin, int T, int H, int W, int Ho, int Wo, int pad,
                       int dil, int Kpix, int kchunk, int splits, int tiles_m, int tiles_n,
                       hipStream_t stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;   // pixels per K-step
constexpr int BKP = 20;  // LDS row pitch (floats)

struct WgradParams {
    const float* dy;
    const float* x;
    float* out;  // dw (splits == 1) or workspace slabs [splits][M][Nn]
    long long dy_nstride, x_nstride;
    int N, M, Cin, Nn;
    int H, W, Ho, Wo, P, stride, pad, dil;
    int Kpix;
    int kchunk, splits, tiles_m, tiles_n;
    int quad_ok;  // Wo % 4 == 0: the 4 pixels of a quad share (img, oh)
    int x_pitch, dy_pitch;   // row pitches (floats) of x / dy; > W / Wo: rows carry a zero tail (wgrad_dma_kernel, wgrad2_kernel)
    int batch;               // wgrad_dma_kernel<1>: independent problems in one launch (Winograd: 16 transformed components),
    long long dy_bstride, x_bstride;   // floats between their operands; outputs [split][batch][M][Nn]
};

constexpr unsigned kOob = 0x80000000u;      // > any record count: buffer loads return 0
constexpr unsigned kMaxRecords = 0x7ffffffcu;

// compile-time loop (all indices constant expressions: register arrays stay in registers)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((vector_size(16)));

// The wgrad kernel.  Block = WM x WN waves, wave = TM x TN MFMA tiles of 32x32; LDS holds both
// operands K(pixel)-contiguous with a 20-float pitch; each K-step is 16 pixels (128 MFMAs on the
// 256 x 256 tile).  What the first version's profile (matrix pipe idle ~45 %) led to:
//  * MFMA operand fragments are double-buffered in registers (reads for the next 64 MFMAs are
//    issued before the current 64),
//  * the next K-step's staging (pixel coordinates tracked incrementally instead of two integer
//    divisions per step, 32 buffer loads, 8 LDS stores) is spread over the eight 16-MFMA slots
//    of a K-step instead of running as one serial block in front of the MFMAs,
//  * rows past M / Nn are clamped instead of predicated (they only feed discarded outputs).
template <int TAPS, int TM, int TN, int WM, int WN>
// the 128 x 256 tile (128 accumulator registers) is built for two waves per SIMD, as in conv_igemm2.hip
__global__ void __launch_bounds__(64 * WM * WN, (TM * TN <= 8 && WM * WN == 4) ? 2 : 1)
wgrad2_kernel(const WgradParams p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    constexpr int ROWS = NT / 4;
    constexpr int PA = BM / ROWS, PB = BN / ROWS;
    static_assert(PA >= 1 && PB >= 1 && PA * ROWS == BM && PB * ROWS == BN, "loader shape");
    static_assert(PA <= 4 && PB <= 4, "slot schedule assumes <= 4 row passes per operand");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM][BKP]
    float* Bs = smem + 2 * BM * BKP;      // [2][BN][BKP]

    // block -> (split, tile): the tiles of one K-split share blockIdx % 8 (= one XCD under the
    // round-robin dispatch), so the dy / x pixels of that split are fetched into ONE L2 and the
    // other tiles of the split hit there; splits are dealt to the 8 XCD groups in turn
    const int tiles = p.tiles_m * p.tiles_n;
    int split, tile;
    {
        const int full = (p.splits / 8) * 8;              // splits covered by whole groups of 8
        const int gsz = 8 * tiles;
        const int b = blockIdx.x;
        if (b < (full / 8) * gsz) {
            const int g = b / gsz, r = b - g * gsz;
            tile = r >> 3;
            split = g * 8 + (r & 7);
        } else {                                         // tail: fewer than 8 splits left
            const int r = b - (full / 8) * gsz;
            const int rem = p.splits - full;
            tile = r / rem;
            split = full + (r - tile * rem);
        }
    }
    const int mt = tile / p.tiles_n, ntile = tile - mt * p.tiles_n;
    const int m0 = mt * BM, n0 = ntile * BN;
    const int kbeg = split * p.kchunk;
    int kend = kbeg + p.kchunk;
    if (kend > p.Kpix || kend < kbeg) kend = p.Kpix;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    // staging map: 16 consecutive lanes = 16 consecutive rows at one pixel quad, so the eight
    // lanes of a ds_write_b128 group hit 32 distinct banks with the 20-float row pitch
    const int kx = lane >> 4, rrow = wid * 16 + (lane & 15);

    const int img0 = kbeg / p.P;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy + (long long)img0 * p.dy_nstride), 0, kMaxRecords, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (long long)img0 * p.x_nstride), 0, kMaxRecords, 0x00020000);
    const int dyn = (int)p.dy_nstride, xn = (int)p.x_nstride;
    // row pitches (round 3: the narrow 3x3 convs read row-pitched operands too, since their forward / dgrad run on the
    // fused Winograd kernel, conv_winograd2.hip): rows xp / dyp floats apart, == W / Wo when dense
    const int xp = p.x_pitch, dyp = p.dy_pitch;
    const int dyP = p.Ho * dyp;                      // channel stride of dy

    // ---- per-thread rows (fixed over the K loop); out-of-range rows are clamped: they only
    // feed output rows / columns that are never stored
    int a_off[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        int m = m0 + rrow + ROWS * j;
        m = m < p.M ? m : p.M - 1;
        a_off[j] = m * dyP;
    }
    int b_coff[PB], b_dh[PB], b_dw[PB];
    const int HW = p.H * xp;                         // channel stride of x (xp == W when dense)
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        int nn = n0 + rrow + ROWS * j;
        nn = nn < p.Nn ? nn : p.Nn - 1;
        const int ci = nn / TAPS;
        const int t = nn - ci * TAPS;
        const int kh = (TAPS == 9) ? t / 3 : 0;
        const int kw = (TAPS == 9) ? t - kh * 3 : 0;
        b_dh[j] = kh * p.dil - p.pad;
        b_dw[j] = kw * p.dil - p.pad;
        b_coff[j] = ci * HW + b_dh[j] * xp + b_dw[j];    // channel + tap shift, relative to (ih, iw)
    }

    // ---- pixel coordinates of this thread's quad, advanced by BK pixels per K-step
    int c_im, c_oh, c_ow;            // image (relative to img0), output row / col of the quad
    {
        const int q0 = kbeg + 4 * kx;
        const int imabs = q0 / p.P;
        const int pq = q0 - imabs * p.P;
        c_im = imabs - img0;
        c_oh = pq / p.Wo;
        c_ow = pq - c_oh * p.Wo;
    }
    // quad path (Wo % 4 == 0): per-K-step scalars of the quad
    bool q_ok = false;               // quad inside [kbeg, kend)
    int q_a = 0, q_x = 0, q_ih = 0, q_iw = 0;   // dy offset, x offset of (ih, iw), ih, iw
    // generic path: per-element coordinates
    int img[4], pp[4], ih[4], iw[4];
    bool qv[4];
    float areg[PA][4], breg[PB][4];

    auto coords = [&](int kbase) {
        const int q0 = kbase + 4 * kx;
        if (p.quad_ok) {
            q_ok = q0 < kend;        // kchunk % 16 == 0 and Kpix % 4 == 0: a quad is all-or-nothing
            q_a = c_im * dyn + c_oh * dyp + c_ow;
            q_ih = c_oh * p.stride;
            q_iw = c_ow * p.stride;
            q_x = c_im * xn + q_ih * xp + q_iw;
            c_ow += BK;
            while (c_ow >= p.Wo) { c_ow -= p.Wo; ++c_oh; }
            while (c_oh >= p.Ho) { c_oh -= p.Ho; ++c_im; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = q0 + e;
                qv[e] = q < kend;
                const int imabs = q / p.P;
                const int pq = q - imabs * p.P;
                const int oh = pq / p.Wo, ow = pq - oh * p.Wo;
                img[e] = imabs - img0; pp[e] = oh * dyp + ow;
                ih[e] = oh * p.stride; iw[e] = ow * p.stride;
            }
        }
    };
    auto load_a = [&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (j < PA) {
            if (p.quad_ok) {         // 16-byte aligned quad of dy: one dwordx4
                const unsigned off = q_ok ? (unsigned)(q_a + a_off[j]) * 4u : kOob;
                // NOTE: bit-cast the whole vector first — indexing the builtin's integer vector
                // makes hipcc (ROCm 7.2) shrink the load to ONE dword (wrong results)
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0));
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    areg[j][e] = v[e];
                });
            } else {
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    const unsigned off = qv[e] ? (unsigned)(img[e] * dyn + a_off[j] + pp[e]) * 4u : kOob;
                    areg[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off, 0, 0));
                });
            }
        }
    };
    auto load_b = [&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (j < PB) {
            if (p.quad_ok && p.stride == 1) {
                // 4 consecutive input pixels of one row: one (possibly unaligned) dwordx4 when the
                // row is inside the image and the quad does not cross its left/right edge
                const int hh = q_ih + b_dh[j], ww = q_iw + b_dw[j];
                const bool rowok = q_ok && hh >= 0 && hh < p.H;
                const unsigned base = (unsigned)(q_x + b_coff[j]) * 4u;
                if (!rowok || (ww >= 0 && ww + 3 < p.W)) {
                    const f32x4 v = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, rowok ? base : kOob, 0, 0));
                    static_for<0, 4>([&](auto e_) {
                        constexpr int e = decltype(e_)::value;
                        breg[j][e] = v[e];
                    });
                } else {
                    static_for<0, 4>([&](auto e_) {
                        constexpr int e = decltype(e_)::value;
                        const bool ok = (ww + e) >= 0 && (ww + e) < p.W;
                        breg[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            b_rsrc, ok ? base + 4u * e : kOob, 0, 0));
                    });
                }
            } else if (p.quad_ok) {
                const int hh = q_ih + b_dh[j];
                const bool rowok = q_ok && hh >= 0 && hh < p.H;
                const unsigned base = (unsigned)(q_x + b_coff[j]) * 4u;
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    const int ww = q_iw + e * p.stride + b_dw[j];
                    const bool ok = rowok && ww >= 0 && ww < p.W;
                    breg[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        b_rsrc, ok ? base + (unsigned)(4 * e * p.stride) : kOob, 0, 0));
                });
            } else {
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    const int hh = ih[e] + b_dh[j], ww = iw[e] + b_dw[j];
                    const bool ok = qv[e] && hh >= 0 && ww >= 0 && hh < p.H && ww < p.W;
                    const unsigned off = ok ? (unsigned)(img[e] * xn + b_coff[j] + ih[e] * xp + iw[e]) * 4u : kOob;
                    breg[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, off, 0, 0));
                });
            }
        }
    };
    auto store_a = [&](int buf) {
        float* a = As + buf * (BM * BKP);
        static_for<0, PA>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const f32x4 v = {areg[j][0], areg[j][1], areg[j][2], areg[j][3]};
            *reinterpret_cast<f32x4*>(a + (rrow + ROWS * j) * BKP + 4 * kx) = v;
        });
    };
    auto store_b = [&](int buf) {
        float* b = Bs + buf * (BN * BKP);
        static_for<0, PB>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const f32x4 v = {breg[j][0], breg[j][1], breg[j][2], breg[j][3]};
            *reinterpret_cast<f32x4*>(b + (rrow + ROWS * j) * BKP + 4 * kx) = v;
        });
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        coords(kbeg);
        static_for<0, 4>([&](auto j_) { load_a(j_); load_b(j_); });
        store_a(0);
        store_b(0);
    }
    __syncthreads();

    const int a_row = wm * (TM * 32) + l31;
    const int b_row = wn * (TN * 32) + l31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        const float* a = As + cur * (BM * BKP) + a_row * BKP + 4 * lhi;
        const float* b = Bs + cur * (BN * BKP) + b_row * BKP + 4 * lhi;
        f32x4 af[2][TM], bf[2][TN];
        static_for<0, TM>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            af[0][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * BKP);
        });
        static_for<0, TN>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            bf[0][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * BKP);
        });
        static_for<0, 8>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int kq = s / 4, e = s % 4;
            if constexpr (s == 0) {   // fragments of the second half of this K-step
                static_for<0, TM>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    af[1][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * BKP + 8);
                });
                static_for<0, TN>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    bf[1][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * BKP + 8);
                });
            }
            if (more) {
                if constexpr (s == 0) { coords(kbeg + (kt + 1) * BK); load_a(std::integral_constant<int, 0>{}); load_a(std::integral_constant<int, 1>{}); }
                if constexpr (s == 1) { load_a(std::integral_constant<int, 2>{}); load_a(std::integral_constant<int, 3>{}); }
                if constexpr (s >= 2 && s <= 5) load_b(std::integral_constant<int, s - 2>{});
                if constexpr (s == 6) store_a(cur ^ 1);
                if constexpr (s == 7) store_b(cur ^ 1);
            }
            static_for<0, TM>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                static_for<0, TN>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kq][i][e], bf[kq][j][e], acc[i][j], 0, 0, 0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
    }

    float* o = p.out + (long long)split * p.M * p.Nn;
    int ncol = n0 + wn * (TN * 32) + l31;
    asm volatile("" : "+v"(ncol));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int m = m0 + wm * (TM * 32) + i * 32 + row;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nn = ncol + j * 32;
                if (nn < p.Nn) o[(long long)m * p.Nn + nn] = acc[i][j][r];
            }
        }
    }
}

// dw[i] = sum_s slab[s][i], s ascending (fixed order).
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long n, int splits) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (long long)gridDim.x * 256) {
        float s = ws[i];
        for (int k = 1; k < splits; ++k) s += ws[(long long)k * n + i];
        dw[i] = s;
    }
}

// the same sums (same order per element), four elements per thread and sixteen / eight slabs' loads in flight
__global__ void __launch_bounds__(256)
splitk_reduce_vec4_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long n4, long long n,
                          int splits) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f4* src = reinterpret_cast<const f4*>(ws) + i;
    const long long slab = n / 4;
    f4 s = src[0];
    int k = 1;
    for (; k + 16 <= splits; k += 16) {      // (16 slabs' loads in flight - the 1x1 weight gradients of layer3 have 64 slabs: 4 round trips)
        f4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(long long)(k + u) * slab];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k + 8 <= splits; k += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * slab];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += src[(long long)k * slab];
    reinterpret_cast<f4*>(dw)[i] = s;
}

// db[c] = sum_{n,p} dy[n,c,p].  One 1024-thread block per channel (the classifiers have 19: few blocks, so each must
// keep many loads in flight), 16-byte loads where the rows allow, fixed summation order.
__global__ void __launch_bounds__(1024)
bias_grad_kernel(const float* __restrict__ dy, long long dy_nstride, float* __restrict__ db, int N,
                 int P, int vec) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* r = dy + (long long)n * dy_nstride + (long long)c * P;
        if (vec) {
            for (int i = 4 * threadIdx.x; i < P; i += 4096) {
                const float4 v = *reinterpret_cast<const float4*>(r + i);
                s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
            }
        } else {
            for (int i = threadIdx.x; i < P; i += 1024) s0 += r[i];
        }
    }
    float s = wave_sum((s0 + s1) + (s2 + s3));
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k];
        db[c] = t;
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;   // MI355X; also the answer when no device is present (CPU-side queries)
    }
    return n;
}

// The 256 x 256 tile with both operands staged by LDS-DMA (`buffer_load ... lds`), for stride-1
// same-size convs whose rows are multiples of 16 pixels: a K-step is then 16 consecutive pixels
// of one image row, so a wave instruction copies the 16-pixel runs of 16 rows (dy channels, or
// (ci, tap) rows of x at that tap's shift) straight into the LDS image - no staging registers,
// no ds_write pass.  The image is [row][16 px] UNPADDED (LDS-DMA writes lane-linearly), made
// conflict-free for the 16-byte fragment reads by an XOR swizzle applied on the SOURCE side:
// 16-byte slot sl of row r holds pixel quad sl ^ ((r >> 2) & 3).  MIXED: some tap shifts the
// columns by a non-multiple of 4 (dilation 1, 2), so quads straddle the row ends: x is then
// copied pixel by pixel (dword copies, 4 rows per instruction); padding = out-of-range offsets.
// WIDE (round 2, pruned widths): the four waves sit side by side along the B rows (64 each) and every wave holds
// all 8 row blocks of dy, 8 x 2 MFMA tiles; row blocks at or past M = Cout are dead for the whole tile and their
// MFMAs are skipped - the cost follows ceil(Cout / 32) instead of ceil(Cout / 256) * 8 (or a register-staged
// 128- / 64-row tile).  Same staging, same K order per output element as the 2 x 2 layout.
// BATCH: p.batch independent problems in one launch (the Winograd weight gradient's 16 components) - an instance of
// its own so that profiles tell it from the 1x1 convs' weight gradients
// HALF (end of round 4, 1x1 convs): 128 rows of dy per tile - 128 accumulator registers, TWO workgroups per CU, each wave
// 64 x 128 - the arrangement that beat one 256-row workgroup per CU on every 1x1 forward / dgrad (section 3g); same splits, same
// K order per output element: the same bits as the 256-row tile.
template <int TAPS, bool MIXED, bool WIDE = false, bool BATCH = false, bool HALF = false>
__global__ void __launch_bounds__(256, HALF ? 2 : 1) wgrad_dma_kernel(const WgradParams p) {
    static_assert(!HALF || (TAPS == 1 && !MIXED && !WIDE), "the 128-row tile is built for the 1x1 burst loader only");
    constexpr int TM = WIDE ? 8 : (HALF ? 2 : 4), TN = WIDE ? 2 : 4, WN = WIDE ? 4 : 2, BM = HALF ? 128 : 256, BN = 256;
    constexpr int AW = BM / 4, AQ = BM / 64;      // dy rows a wave copies per K-step, in AQ instructions of 16 rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BM][BK]
    float* Bs = smem + 2 * BM * BK;     // [2][BN][BK]
    const int tiles = p.tiles_m * p.tiles_n;
    int split, tile, bi = 0;
    {
        const int full = (p.splits / 8) * 8;
        const int gsz = 8 * tiles;
        int b = blockIdx.x;
        if constexpr (BATCH) {         // problem-major: the workgroups of one problem share its operands in L2
            const int per = tiles * p.splits;
            bi = b / per;
            b -= bi * per;
        }
        if (b < (full / 8) * gsz) {
            const int g = b / gsz, r = b - g * gsz;
            tile = r >> 3;
            split = g * 8 + (r & 7);
        } else {
            const int r = b - (full / 8) * gsz;
            const int rem = p.splits - full;
            tile = r / rem;
            split = full + (r - tile * rem);
        }
    }
    const int mt = tile / p.tiles_n, ntile = tile - mt * p.tiles_n;
    const int m0 = mt * BM, n0 = ntile * BN;
    const int kbeg = split * p.kchunk;
    int kend = kbeg + p.kchunk;
    if (kend > p.Kpix || kend < kbeg) kend = p.Kpix;
    const int nk = (kend - kbeg) / BK;                  // kchunk, Kpix are multiples of 16 here

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int img0 = kbeg / p.P;
    const int HW = p.H * p.x_pitch;                  // channel stride of x (x_pitch == W when dense)
    const int dyP = p.Ho * p.dy_pitch;               // channel stride of dy
    const int padx = p.x_pitch - p.W;                // zero floats behind every row of x
    typedef unsigned u32x4 __attribute__((vector_size(16)));
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto make_desc = [](const void* base) {
        const unsigned long long a = (unsigned long long)base;
        u32x4 d = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, kMaxRecords, 0x00020000u};
        return d;
    };
    const u32x4 a_desc = make_desc(p.dy + (long long)img0 * p.dy_nstride + (long long)bi * p.dy_bstride);
    // pitched x: offsets are taken from padx floats before the image (readable zeros by contract), so that a quad
    // hanging over the left end of the very first row keeps a non-negative offset
    const u32x4 b_desc = make_desc(p.x + (long long)img0 * p.x_nstride + (long long)bi * p.x_bstride - padx);
    const unsigned lds_a0 = (unsigned)(size_t)(lds_ptr)As, lds_b0 = (unsigned)(size_t)(lds_ptr)Bs;

    // ---- A (dy): instruction q of this wave copies rows 64*wid + 16q + (lane >> 2); the lane's slot
    // lane & 3 receives pixel quad (lane & 3) ^ ((lane >> 4) & 3)
    const int gq = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned a_voff[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        int m = m0 + AW * wid + 16 * q + (lane >> 2);
        m = m < p.M ? m : p.M - 1;
        a_voff[q] = (unsigned)(m * dyP + 4 * gq) * 4u;
    }
    // ---- B (x at the row's tap shift): quad copies use the same lane -> (row, quad) map; dword copies
    // (MIXED) cover rows 64*wid + 16q + 4e + (lane >> 4), slot (lane >> 2) & 3 -> quad slot ^ e, pixel lane & 3
    int bq_c[4], bq_dh[4], bq_dw[4];           // per quad-instruction q: channel offset, tap shift
    int bd_c[16], bd_dh[16], bd_dw[16];        // per dword-instruction (q, e)
    auto row_tap = [&](int row, int& coff, int& dh, int& dw) {
        int nn = n0 + row;
        nn = nn < p.Nn ? nn : p.Nn - 1;
        const int ci = nn / TAPS, t = nn - ci * TAPS;
        const int kh = (TAPS == 9) ? t / 3 : 0, kw = (TAPS == 9) ? t - kh * 3 : 0;
        dh = kh * p.dil - p.pad; dw = kw * p.dil - p.pad;
        coff = ci * HW;
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) row_tap(64 * wid + 16 * q + (lane >> 2), bq_c[q], bq_dh[q], bq_dw[q]);
    if constexpr (MIXED) {
#pragma unroll
        for (int qe = 0; qe < 16; ++qe) row_tap(64 * wid + 4 * qe + (lane >> 4), bd_c[qe], bd_dh[qe], bd_dw[qe]);
    }
    const int gd = ((lane >> 2) & 3);          // dword copies: slot; quad = slot ^ e, element lane & 3

    // position of the K-step the loader copies next: image (relative to img0), row, first column
    int c_im = 0, c_oh, c_ow;
    {
        const int pq = kbeg - img0 * p.P;
        c_oh = pq / p.Wo; c_ow = pq - c_oh * p.Wo;
    }
    // The copies of K-step kt+1 are issued in 8 pieces (4 instructions for dy, 4 for x), each with its branch-free
    // address arithmetic, one piece behind every second MFMA of the step's first group, and the second half-step's
    // fragments are read behind the first MFMA.  (Round 2: with everything at the top of the step - ~100 VALU /
    // branch instructions and 16 LDS reads before the first MFMA - the kernel measured the same, 123..136 TF on
    // the R101 shapes; profiles/r02_wgrad_issue_ab.txt.  The steady state is not issue-bound.)
    auto issue_a = [&](int buf, auto q_) {
        constexpr int q = decltype(q_)::value;
        // (uniform values; readfirstlane makes the compiler keep them in SGPRs for the asm operands)
        const unsigned a_s = __builtin_amdgcn_readfirstlane((unsigned)(c_im * (int)p.dy_nstride + c_oh * p.dy_pitch + c_ow) * 4u);
        const unsigned la = __builtin_amdgcn_readfirstlane(lds_a0 + (unsigned)((buf * BM + AW * wid + 16 * q) * BK) * 4u);
        const unsigned av = a_voff[q], as_ = a_s;
        const u32x4 ad = a_desc;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(la), "v"(av), "s"(ad), "s"(as_) : "memory", "m0");
    };
    auto issue_b = [&](int buf, auto q_) {
        constexpr int q = decltype(q_)::value;
        const unsigned b_img = __builtin_amdgcn_readfirstlane((unsigned)(c_im * (int)p.x_nstride) * 4u);
        const int oh = c_oh, ow = c_ow;
        const unsigned lb = __builtin_amdgcn_readfirstlane(lds_b0 + (unsigned)((buf * BN + 64 * wid + 16 * q) * BK) * 4u);
        const int hh = oh + bq_dh[q], ww = ow + 4 * gq + bq_dw[q];
        const bool row_ok = (unsigned)hh < (unsigned)p.H;
        // pitched x: the quad may hang over a row end into the zero tail (of this or the previous row)
        const bool ok = row_ok & ((unsigned)(ww + padx) <= (unsigned)(p.W + 2 * padx - 4));
        const unsigned bs_ = b_img;
        const u32x4 bd = b_desc;
        bool quads = true;
        if constexpr (MIXED)      // a quad that straddles the image border: only on a row's first / last K-step
            quads = __ballot(row_ok && !ok && ww > -4 && ww < p.W) == 0;
        if (quads) {             // 16-byte copies; with a shifted tap the source is only 4/8-byte aligned
            const unsigned off = (unsigned)(bq_c[q] + hh * p.x_pitch + ww + padx) * 4u;
            const unsigned bv = ok ? off : kOob;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(lb), "v"(bv), "s"(bd), "s"(bs_) : "memory", "m0");
        } else if constexpr (MIXED) {
            static_for<0, 4>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                constexpr int qe = 4 * q + e;
                const unsigned le = lb + (unsigned)(4 * e * BK) * 4u;
                const int h1 = oh + bd_dh[qe], w1 = ow + 4 * (gd ^ e) + (lane & 3) + bd_dw[qe];
                const bool ok1 = ((unsigned)h1 < (unsigned)p.H) & ((unsigned)w1 < (unsigned)p.W);
                const unsigned off1 = (unsigned)(bd_c[qe] + h1 * p.x_pitch + w1 + padx) * 4u;
                const unsigned bv = ok1 ? off1 : kOob;
                const unsigned bs2 = bs_;
                const u32x4 bd2 = bd;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                             :: "s"(le), "v"(bv), "s"(bd2), "s"(bs2) : "memory", "m0");
            });
        }
    };
    auto advance = [&]() {
        c_ow += BK;
        if (c_ow >= p.Wo) { c_ow = 0; if (++c_oh >= p.Ho) { c_oh = 0; ++c_im; } }
    };
    // ---- 1x1 convs (TAPS == 1: no tap shift, no padding - every copy of a K-step reads 16 consecutive pixels of rows that
    // exist): the whole address of a copy is a lane constant plus ONE running wave-uniform byte offset per operand, stepped
    // by scalar adds (row / image wrap included), and the 8 copies of a K-step are issued as one burst.  Round 4: with one
    // wave per SIMD every scalar / vector ALU instruction of the loop costs ~4.5 cycles of matrix-pipe time
    // (tools/micro/mfma_shadow.hip) and the general path below spends ~150 of them per K-step on coordinates it re-derives
    // per piece; same copies, same order, same bits.
    unsigned b1_voff[4] = {0u, 0u, 0u, 0u};
    unsigned a1_run = 0, b1_run = 0;
    const unsigned a1_row = (unsigned)(p.dy_pitch - p.Wo) * 4u, b1_row = (unsigned)(p.x_pitch - p.W) * 4u;
    const unsigned a1_img = (unsigned)((int)p.dy_nstride - p.Ho * p.dy_pitch) * 4u, b1_img = (unsigned)((int)p.x_nstride - p.H * p.x_pitch) * 4u;
    const unsigned lds_a_w = __builtin_amdgcn_readfirstlane(lds_a0 + (unsigned)(AW * wid * BK) * 4u);
    const unsigned lds_b_w = __builtin_amdgcn_readfirstlane(lds_b0 + (unsigned)(64 * wid * BK) * 4u);
    if constexpr (TAPS == 1 && !MIXED) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b1_voff[q] = (unsigned)(bq_c[q] + 4 * gq + padx) * 4u;
        a1_run = __builtin_amdgcn_readfirstlane((unsigned)(c_oh * p.dy_pitch + c_ow) * 4u);
        b1_run = __builtin_amdgcn_readfirstlane((unsigned)(c_oh * p.x_pitch + c_ow) * 4u);
    }
    auto issue1 = [&](int buf) {
        const unsigned abuf = lds_a_w + (buf ? (unsigned)(BM * BK) * 4u : 0u), bbuf = lds_b_w + (buf ? (unsigned)(BN * BK) * 4u : 0u);
        static_for<0, AQ>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            const unsigned av = a_voff[q], base = abuf, run = a1_run;
            const u32x4 ad = a_desc;
            asm volatile("s_add_i32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(base), "v"(av), "s"(ad), "s"(run), "i"(16 * q * BK * 4) : "memory", "m0", "scc");
        });
        static_for<0, 4>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            const unsigned bv = b1_voff[q], base = bbuf, run = b1_run;
            const u32x4 bd = b_desc;
            asm volatile("s_add_i32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(base), "v"(bv), "s"(bd), "s"(run), "i"(16 * q * BK * 4) : "memory", "m0", "scc");
        });
        // step to the next 16 pixels: same row, next row, or next image
        c_ow += BK; a1_run += BK * 4u; b1_run += BK * 4u;
        if (c_ow >= p.Wo) {
            c_ow = 0; a1_run += a1_row; b1_run += b1_row;
            if (++c_oh >= p.Ho) { c_oh = 0; a1_run += a1_img; b1_run += b1_img; }
        }
    };
    auto issue = [&](int buf) {
        if constexpr (TAPS == 1 && !MIXED) { issue1(buf); return; }
        static_for<0, 4>([&](auto q_) { issue_a(buf, q_); });
        static_for<0, 4>([&](auto q_) { issue_b(buf, q_); });
        advance();
    };
    auto retire = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int nb = (p.M - m0 + 31) >> 5;             // WIDE: live dy row blocks of this tile (block-uniform)
    nb = nb > TM ? TM : nb;

    if (nk > 0) issue(0);
    retire();
    // fragment reads: lane (l31, lhi) reads 16-byte slot ((2 kq + lhi) ^ ((l31 >> 2) & 3)) of its row
    const int sw = (l31 >> 2) & 3;
    const int a_row = (wm * (TM * 32) + l31) * BK, b_row = (wn * (TN * 32) + l31) * BK;
    const int f0 = 4 * ((0 + lhi) ^ sw), f1 = 4 * ((2 + lhi) ^ sw);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        const float* a = As + cur * (BM * BK) + a_row;
        const float* b = Bs + cur * (BN * BK) + b_row;
        f32x4 af[2][TM], bf[2][TN];
        // the first half-step's fragments now, the second half's behind the first MFMA (they are needed 64 MFMAs on)
        static_for<0, TM>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            af[0][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * BK + f0);
        });
        static_for<0, TN>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            bf[0][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * BK + f0);
        });
        static_for<0, 8>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int kq = s / 4, e = s % 4;
            static_for<0, TM>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                static_for<0, TN>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    constexpr int n = i * TN + j;
                    if (!WIDE || i < nb)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kq][i][e], bf[kq][j][e], acc[i][j], 0, 0, 0);
                    if constexpr (s == 0 && n == 0) {
                        static_for<0, TM>([&](auto i2_) {
                            constexpr int i2 = decltype(i2_)::value;
                            af[1][i2] = *reinterpret_cast<const f32x4*>(a + i2 * 32 * BK + f1);
                        });
                        static_for<0, TN>([&](auto j2_) {
                            constexpr int j2 = decltype(j2_)::value;
                            bf[1][j2] = *reinterpret_cast<const f32x4*>(b + j2 * 32 * BK + f1);
                        });
                    }
                    if constexpr (TAPS == 1 && !MIXED) {
                        if constexpr (s == 0 && n == 1) {           // all 8 copies of the next step: one burst, one branch
                            if (more) issue1(cur ^ 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else if constexpr (s == 0 && (n & 1) == 1) {      // piece n / 2 of the next step's copies
                        if (more) {
                            constexpr int piece = n / 2;
                            if constexpr (piece < 4) issue_a(cur ^ 1, std::integral_constant<int, piece>{});
                            else issue_b(cur ^ 1, std::integral_constant<int, piece - 4>{});
                            if constexpr (piece == 7) advance();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        retire();
    }

    float* o = p.out + ((long long)split * p.batch + bi) * p.M * p.Nn;
    int ncol = n0 + wn * (TN * 32) + l31;
    asm volatile("" : "+v"(ncol));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int m = m0 + wm * (TM * 32) + i * 32 + row;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nn = ncol + j * 32;
                if (nn < p.Nn) o[(long long)m * p.Nn + nn] = acc[i][j][r];
            }
        }
    }
}

struct Plan {
    int bm, bn, cfg;  // cfg: 0 = 256x256, 1 = 128x256, 2 = 64x64 (one wave), 3 = 256x128, 4 = 256x64
    int tiles_m, tiles_n, splits, kchunk;
    bool wide;        // 256 x 256 tile on wgrad_dma_kernel<.., WIDE>: dead dy row blocks skipped (ragged / small Cout)
    bool half;        // 128 x 256 tile on wgrad_dma_kernel<1, .., HALF>: two workgroups per CU (1x1 convs, Cout % 128 == 0)
};

static bool math_bf16x3();

// shapes the LDS-DMA kernels take apart from the tile choice: stride 1, output = input size, rows multiple of 16 px
bool dma_geometry(const DcfpConvDesc* d) {
    static const bool on = [] { const char* e = getenv("DCFP_WGRAD_DMA"); return !e || atoi(e) != 0; }();   // =0: off
    return on && d->stride == 1 && d->Hout == d->H && d->Wout == d->W && d->W % 16 == 0;
}

Plan make_plan(const DcfpConvDesc* d) {
    Plan pl;
    const int M = d->Cout, Nn = d->Cin * d->KH * d->KW;
    // cfg: 0 = 256x256, 1 = 128x256, 2 = 64x64 (one wave), and - round 2 - tiles for the lop-sided weight
    // shapes of layer1 / layer2 / the stem, which used to fall to the one-wave tile (55..81 TF):
    // 3 = 256x128, 4 = 256x64 (tall: many output channels, <= 128 / <= 64 input columns): +13..15 % on the
    // layer1 / layer2 conv3 shapes.  (A 64x256 tile for M <= 64 measured WORSE than the one-wave 64x64 tile -
    // 65 vs 91 TF on the 64-channel 3x3 convs, 61 vs 65 TF on 256->64 1x1 - and is not used.)
    static const bool lopsided = [] { const char* e = getenv("DCFP_WGRAD_LOPSIDED"); return !e || atoi(e) != 0; }();   // =0: off
    if (M > 128 && Nn > 128) { pl.cfg = 0; pl.bm = 256; pl.bn = 256; }
    else if (M > 64 && Nn > 128) {
        pl.cfg = 1; pl.bm = 128; pl.bn = 256;
        // 128 x 192 where that pads the (channel, tap) columns less: 64 -> 128 3x3 has 576 = 3 x 192 of them (768 on
        // 256-column tiles: a quarter of the MFMAs on padding), 128 -> 128 has 1152 = 6 x 192 (1280)
        static const bool t192 = [] { const char* e = getenv("DCFP_WGRAD_T192"); return !e || atoi(e) != 0; }();   // =0: off
        if (t192 && (Nn + 191) / 192 * 192 < (Nn + 255) / 256 * 256) { pl.cfg = 5; pl.bn = 192; }
    }
    else if (lopsided && M > 128 && Nn > 64) { pl.cfg = 3; pl.bm = 256; pl.bn = 128; }
    else if (lopsided && M > 128 && Nn > 32) { pl.cfg = 4; pl.bm = 256; pl.bn = 64; }
    else { pl.cfg = 2; pl.bm = 64; pl.bn = 64; }
    // WIDE: 32-row granularity in Cout on the LDS-DMA pipeline.  DCFP_WGRAD_WIDE: 0 off, 2 wherever eligible (A/B).
    pl.wide = false;
    {
        static const int mode = [] { const char* e = getenv("DCFP_WGRAD_WIDE"); return e ? atoi(e) : 1; }();
        const bool mixed = d->KH == 3 && ((d->pad | d->dil) & 3) != 0;
        const bool x_pitched = d->x_pitch && d->x_pitch != d->W;
        const bool eligible = mode != 0 && !math_bf16x3() && dma_geometry(d) && Nn > 128 && (!mixed || x_pitched) &&
                              !(d->KH == 3 && d->pad != d->dil);
        if (eligible) {
            const long long old_rows = (long long)((M + pl.bm - 1) / pl.bm) * pl.bm, new_rows = (long long)((M + 31) / 32) * 32;
            // (not the narrow 3x3 convs of the stem / layer1 / layer2, which are row-pitched since round 3 - their forward and
            //  dgrad run on the fused Winograd kernel - and whose lop-sided register-staged tiles measure 82...110 TF
            //  against 69 TF here)
            const bool narrow = d->KH == 3 && M <= 128 && Nn <= 1152;
            if (mode == 2 || (!narrow && (pl.cfg != 0 || (M % pl.bm != 0 && 10 * new_rows <= 9 * old_rows)))) {
                pl.wide = true; pl.cfg = 0; pl.bm = 256; pl.bn = 256;
            }
        }
    }
    pl.half = false;
    {
        static const bool on = [] { const char* e = getenv("DCFP_WGRAD_HALF"); return !e || atoi(e) != 0; }();   // =0: off
        if (on && pl.cfg == 0 && !pl.wide && d->KH == 1 && M % 128 == 0 && !math_bf16x3() && dma_geometry(d)) {
            pl.half = true; pl.bm = 128;
        }
    }
    pl.tiles_m = (M + pl.bm - 1) / pl.bm;
    pl.tiles_n = (Nn + pl.bn - 1) / pl.bn;
    const long long tiles = (long long)pl.tiles_m * pl.tiles_n;
    const long long Kpix = (long long)d->N * d->Hout * d->Wout;
    // Split K so that the grid fills whole "rounds" of the chip: the big tiles run one
    // workgroup per CU (512 registers/lane), so tiles*splits just above a multiple of the CU
    // count wastes most of a round (513 blocks on 256 CUs took 1.35x the time of 252).
    const int cus = num_cus();
    // resident workgroups per CU (accumulator registers per lane: 256 / 128 / 64 / 32)
    const long long per_cu = pl.cfg == 2 ? 8 : (pl.cfg == 0 && !pl.half) ? 1 : 2;      // (cfg 5: 96 accumulator registers, 51 KB LDS: 2)
    const long long slots = (long long)cus * per_cu;
    const long long max_splits = (Kpix + BK * 8 - 1) / (BK * 8);   // >= 8 K-steps per split
    long long splits = 1;
    if (const char* e = getenv("DCFP_DBG_WGRAD_BLOCKS")) {
        splits = (atoll(e) + tiles - 1) / tiles;
    } else {
        double best = -1.0;
        for (long long sp = 1; sp <= max_splits && tiles * sp <= 4 * slots; ++sp) {
            const long long blocks = tiles * sp;
            const long long rounds = (blocks + slots - 1) / slots;
            double eff = (double)blocks / (double)(rounds * slots);
            if (blocks < slots) eff *= 0.999;                // prefer filling the chip at equal eff
            if (eff > best + 0.02) { best = eff; splits = sp; }   // smallest split within 2 % of best
        }
    }
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long kchunk = (Kpix + splits - 1) / splits;
    kchunk = (kchunk + BK - 1) / BK * BK;
    splits = (Kpix + kchunk - 1) / kchunk;
    pl.splits = (int)splits;
    pl.kchunk = (int)kchunk;
    return pl;
}

int check_desc(const DcfpConvDesc* d) {
    if (!d) return DCFP_E_BADDESC;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0 || d->stride <= 0 ||
        d->dil <= 0 || d->pad < 0)
        return DCFP_E_BADDESC;
    if (d->KH != d->KW || (d->KH != 1 && d->KH != 3)) return DCFP_E_UNSUPPORTED;
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    if (ho != d->Hout || wo != d->Wout || ho <= 0 || wo <= 0) return DCFP_E_BADDESC;
    if ((d->x_pitch != 0 && (d->x_pitch < d->W || (d->x_pitch & 3))) ||
        (d->dy_pitch != 0 && (d->dy_pitch < d->Wout || (d->dy_pitch & 3))))
        return DCFP_E_BADDESC;
    if ((long long)d->Cin * d->H * (d->x_pitch ? d->x_pitch : d->W) >= (1LL << 29) ||
        (long long)d->Cout * d->Hout * (d->dy_pitch ? d->dy_pitch : d->Wout) >= (1LL << 29) ||
        (long long)d->Cout * d->Cin * d->KH * d->KW >= (1LL << 29))
        return DCFP_E_UNSUPPORTED;
    return DCFP_OK;
}

// EXPERIMENTAL opt-in: DCFP_CONV_MATH=bf16x3 routes qualifying wgrad shapes to conv_wgrad3.hip
static bool math_bf16x3() {
    static const bool v = [] { const char* e = getenv("DCFP_CONV_MATH"); return e && !strcmp(e, "bf16x3"); }();
    return v;
}
static bool wgrad3_ok(const DcfpConvDesc* d, int cfg) {
    return math_bf16x3() && cfg == 0 && d->stride == 1 && (d->Wout % 4 == 0);
}


template <int TAPS, int TM, int TN, int WM, int WN>
int launch_cfg(const WgradParams& p, long long blocks, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    const size_t lds = (size_t)2 * (BM + BN) * BKP * sizeof(float);
    auto kern = wgrad2_kernel<TAPS, TM, TN, WM, WN>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DCFP_OK : (int)e;
}

// shapes the LDS-DMA kernel takes: 256 x 256 tile, stride 1, output = input size, rows of 16 k pixels
static bool wgrad_dma_ok(const DcfpConvDesc* d, int cfg) {
    static const bool on = [] { const char* e = getenv("DCFP_WGRAD_DMA"); return !e || atoi(e) != 0; }();   // =0: off
    static const bool mixed_too = [] { const char* e = getenv("DCFP_WGRAD_DMA_MIXED"); return !e || atoi(e) != 0; }();   // =0: off
    const bool mixed = d->KH == 3 && ((d->pad | d->dil) & 3) != 0;
    // same-box A/B against the register-staged kernel: +17 % where every tap keeps quads aligned (1x1,
    // dilation 4/8/12/...); +7 % for dilation 1/2 (16-byte copies from the 4/8-byte aligned shifted source,
    // dword copies only on the K-steps whose quads straddle the image border; -1 % with dword copies throughout)
    return on && cfg == 0 && d->stride == 1 && d->Hout == d->H && d->Wout == d->W && d->W % 16 == 0 &&
           (!mixed || mixed_too);
}
static bool wgrad_dma_mixed(const DcfpConvDesc* d) { return d->KH == 3 && ((d->pad | d->dil) & 3) != 0; }

template <int TAPS, bool MIXED, bool WIDE = false, bool BATCH = false, bool HALF = false>
int launch_dma(const WgradParams& p, long long blocks, hipStream_t stream) {
    const size_t lds = (size_t)2 * (HALF ? 384 : 512) * BK * sizeof(float);
    hipLaunchKernelGGL((wgrad_dma_kernel<TAPS, MIXED, WIDE, BATCH, HALF>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DCFP_OK : (int)e;
}

template <int TAPS>
int launch_taps(const WgradParams& p, const Plan& pl, hipStream_t stream) {
    const long long blocks = (long long)pl.tiles_m * pl.tiles_n * pl.splits;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    switch (pl.cfg) {
        case 0: return launch_cfg<TAPS, 4, 4, 2, 2>(p, blocks, stream);
        case 1: return launch_cfg<TAPS, 2, 4, 2, 2>(p, blocks, stream);
        case 3: return launch_cfg<TAPS, 4, 2, 2, 2>(p, blocks, stream);
        case 4: return launch_cfg<TAPS, 4, 1, 2, 2>(p, blocks, stream);
        case 5: return launch_cfg<TAPS, 2, 3, 2, 2>(p, blocks, stream);
        default: return launch_cfg<TAPS, 2, 2, 1, 1>(p, blocks, stream);
    }
}

}  // namespace

bool dcfp_gemv_shape(const DcfpConvDesc* d);     // conv_gemv.hip
bool dcfp_stem_shape(const DcfpConvDesc* d);     // conv_stem.hip
size_t dcfp_stem_wgrad_workspace_bytes(const DcfpConvDesc* d);
int dcfp_stem_wgrad(const DcfpConvDesc* d, const float* dy, long long dy_nstride, const float* x, float* dw, void* workspace,
                    size_t workspace_bytes, hipStream_t stream);
int dcfp_gemv_wgrad(const DcfpConvDesc* d, const float* dy, long long dy_nstride, const float* x, float* dw, hipStream_t stream);

// pitched operands: the LDS-DMA kernels and the register-staged wgrad2_kernel (not the bf16x3 split kernel), 3x3 with
// pad = dil and a tail that covers the column shifts
bool dcfp_wgrad_pitch_ok(const DcfpConvDesc* d) {
    if (check_desc(d) != DCFP_OK) return false;
    const Plan pl = make_plan(d);
    if (wgrad3_ok(d, pl.cfg)) return false;
    const int xp = d->x_pitch ? d->x_pitch : d->W;
    if (xp != d->W && (d->KH != 3 || d->pad != d->dil || xp < d->W + d->pad)) return false;
    // 31-bit byte offsets relative to the first image of a split: checked in the launcher with the pitched strides
    return true;
}

// `batch` independent products  out[b][m][c] = sum_t a[b][m][t] * bmat[b][c][t]  (t < K, K % 16 == 0, rows K floats
// apart) in ONE launch of the LDS-DMA 1x1 weight-gradient kernel - the 16 transformed components of a Winograd
// weight gradient (conv_winograd.hip).  workspace: split-K slabs [splits][batch][M][C] when splits > 1.
size_t dcfp_wgrad_batched_workspace_bytes(int batch, int M, int C, long long K, int* splits_out) {
    const long long tiles = (long long)((M + 255) / 256) * ((C + 255) / 256) * batch;
    const long long slots = num_cus();
    long long max_splits = K / (BK * 8);
    if (max_splits < 1) max_splits = 1;
    long long splits = 1;
    double best = -1.0;
    for (long long sp = 1; sp <= max_splits && tiles * sp <= 4 * slots; ++sp) {
        const long long blocks = tiles * sp, rounds = (blocks + slots - 1) / slots;
        double eff = (double)blocks / (double)(rounds * slots);
        if (blocks < slots) eff *= 0.999;
        if (eff > best + 0.02) { best = eff; splits = sp; }
    }
    long long kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + BK - 1) / BK * BK;
    splits = (K + kchunk - 1) / kchunk;
    if (splits_out) *splits_out = (int)splits;
    return splits > 1 ? (size_t)splits * batch * M * C * sizeof(float) : 0;
}

int dcfp_wgrad_batched_run(const float* a, const float* bmat, float* out, int batch, int M, int C, long long K,
                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (K % BK != 0 || K <= 0 || K >= (1LL << 30) || (long long)M * K >= (1LL << 29) || (long long)C * K >= (1LL << 29))
        return DCFP_E_UNSUPPORTED;
    int splits = 1;
    const size_t need = dcfp_wgrad_batched_workspace_bytes(batch, M, C, K, &splits);
    if (need && (!workspace || workspace_bytes < need || !dcfp_aligned16(workspace))) return DCFP_E_WORKSPACE;
    WgradParams p;
    p.dy = a; p.x = bmat;
    p.out = splits > 1 ? static_cast<float*>(workspace) : out;
    p.x_pitch = (int)K; p.dy_pitch = (int)K;
    p.dy_nstride = (long long)M * K; p.x_nstride = (long long)C * K;
    p.N = 1; p.M = M; p.Cin = C; p.Nn = C;
    p.H = 1; p.W = (int)K; p.Ho = 1; p.Wo = (int)K; p.P = (int)K;
    p.stride = 1; p.pad = 0; p.dil = 1;
    p.Kpix = (int)K;
    long long kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + BK - 1) / BK * BK;
    p.kchunk = (int)kchunk; p.splits = splits;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (C + 255) / 256;
    p.quad_ok = 1;
    p.batch = batch; p.dy_bstride = (long long)M * K; p.x_bstride = (long long)C * K;
    const long long blocks = (long long)p.tiles_m * p.tiles_n * splits * batch;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    // ragged M (pruned widths): the 8 x 2 wave layout skips the dead 32-row blocks of the last M tile
    const long long mp = (long long)p.tiles_m * 256, live = (long long)((M + 31) / 32) * 32;
    int rc = (M % 256 != 0 && 10 * live <= 9 * mp) ? launch_dma<1, false, true, true>(p, blocks, stream)
                                                   : launch_dma<1, false, false, true>(p, blocks, stream);
    if (rc) return rc;
    if (splits > 1) {
        const long long wn = (long long)batch * M * C;
        if (wn % 4 == 0 && dcfp_aligned16(out)) {
            const long long n4 = wn / 4;
            hipLaunchKernelGGL(splitk_reduce_vec4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream,
                               static_cast<const float*>(workspace), out, n4, wn, splits);
        } else {
            long long b = (wn + 255) / 256;
            if (b > 4096) b = 4096;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)b), dim3(256), 0, stream,
                               static_cast<const float*>(workspace), out, wn, splits);
        }
    }
    DCFP_RETURN_LAUNCH();
}

static bool wino_wgrad_pass(const DcfpConvDesc* d);
static int wino_wgrad_kind(const DcfpConvDesc* d);

int dcfp_wgrad_kernel_name(const DcfpConvDesc* d, char* buf, int buf_len) {
    if (dcfp_stem_shape(d)) return snprintf(buf, buf_len, "stem_wgrad_kernel");
    if (const int wk = wino_wgrad_kind(d))
        return wk == 2 ? snprintf(buf, buf_len, "winograd_f2x2_3x3 wgrad fused (wino_wgrad_fused_kernel)")
                       : snprintf(buf, buf_len, "winograd_f2x2_3x3 wgrad (wgrad_dma_kernel<1,false,false,true>)");
    const Plan pl = make_plan(d);
    const char* args = pl.cfg == 0 ? "4,4,2,2" : pl.cfg == 1 ? "2,4,2,2" : pl.cfg == 3 ? "4,2,2,2" :
                       pl.cfg == 4 ? "4,1,2,2" : pl.cfg == 5 ? "2,3,2,2" : "2,2,1,1";
    if (wgrad3_ok(d, pl.cfg)) return snprintf(buf, buf_len, "wgrad3_kernel<%d>", d->KH * d->KW);
    if (pl.wide) return snprintf(buf, buf_len, "wgrad_dma_kernel<%d,false,true>", d->KH * d->KW);
    if (wgrad_dma_ok(d, pl.cfg))
        return snprintf(buf, buf_len, "wgrad_dma_kernel<%d,%s>", d->KH * d->KW,
                        (wgrad_dma_mixed(d) && !(d->x_pitch && d->x_pitch != d->W)) ? "true" : "false");
    return snprintf(buf, buf_len, "wgrad2_kernel<%d,%s>", d->KH * d->KW, args);
}

// conv_winograd.hip
bool dcfp_wino_wgrad_ok(int N, int H, int W, int d, int M, int C);
size_t dcfp_wino_wgrad_workspace_bytes(int N, int H, int W, int d, int M, int C);
double dcfp_wino_exec_fraction(int N, int H, int W, int d, int M, int Ck);
int dcfp_wino_wgrad_run(const float* dy, long long dy_nstride, int dy_pitch, const float* x, long long x_nstride,
                        int x_pitch, float* dw, int N, int M, int C, int H, int W, int d, void* workspace,
                        size_t workspace_bytes, hipStream_t stream, const float* xform_in = nullptr);
extern "C" size_t dcfp_conv2d_xform_bytes(const DcfpConvDesc* d);

// executed-MFMA rate the cost model prices the fused Winograd weight gradient at (DCFP_WINO_WGRAD_RATE, TF: A/B), and
// DCFP_WINO_WGRAD_FUSED=2: take it wherever it applies and beats the direct kernel, whatever the batched path's model says
static double wino_wgrad_fused_rate() {
    static const double v = [] { const char* e = getenv("DCFP_WINO_WGRAD_RATE"); return (e ? atof(e) : 118.0) * 1e12; }();
    return v;
}
static bool wino_wgrad_fused_forced() {
    static const bool v = [] { const char* e = getenv("DCFP_WINO_WGRAD_FUSED"); return e && atoi(e) == 2; }();
    return v;
}

// conv_winograd3.hip: the fused weight-gradient kernel (both operands transformed inside the GEMM, no kept V)
bool dcfp_wino_wgrad_fused_ok(int N, int H, int W, int d, int M, int C, long long x_nstride, int x_pitch,
                              long long dy_nstride, int dy_pitch);
size_t dcfp_wino_wgrad_fused_workspace_bytes(int N, int H, int W, int d, int M, int C);
int dcfp_wino_wgrad_fused_run(const float* dy, long long dy_nstride, int dy_pitch, const float* x, long long x_nstride,
                              int x_pitch, float* dw, int N, int M, int C, int H, int W, int d, void* workspace,
                              size_t workspace_bytes, hipStream_t stream);

// Winograd F(2x2, 3x3) weight gradient where the cost model (same-box measurements, profiles/r02_winograd_ab.txt,
// profiles/r04_wino_wgrad_fused_ab.txt) says it beats the direct LDS-DMA kernel: direct = nominal FLOPs at 130 TF (123 on
// dense dilation-1 operands); batched Winograd (kind 1) = 16/36 x tile padding of them at 125 TF plus the x / dy transform
// passes at 4.2 / 5 TB/s; fused Winograd (kind 2, conv_winograd3.hip) = the same products on 64 x 64-channel blocks at the
// rate of its K loop, no passes - the only Winograd weight gradient below 128 channels (stem, layer1, layer2, pruned widths).
// Returns 0 (direct kernels), 1 or 2.  DCFP_CONV_WINOGRAD: 0 off, 1 model (default), 2 wherever eligible.
static int wino_wgrad_kind(const DcfpConvDesc* d) {
    static const int mode = [] { const char* e = getenv("DCFP_CONV_WINOGRAD"); return e ? atoi(e) : 1; }();
    if (mode == 0) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != d->dil || d->Hout != d->H || d->Wout != d->W) return 0;
    if (math_bf16x3()) return 0;
    const int xp = d->x_pitch ? d->x_pitch : d->W, dyp = d->dy_pitch ? d->dy_pitch : d->Wout;
    const bool batched = dcfp_wino_wgrad_ok(d->N, d->H, d->W, d->dil, d->Cout, d->Cin);
    // (a dy that is a channel slice of a wider tensor - the ASPP branches - has its own image stride: the entry point
    //  checks the real one again)
    const bool fused = dcfp_wino_wgrad_fused_ok(d->N, d->H, d->W, d->dil, d->Cout, d->Cin, (long long)d->Cin * d->H * xp, xp,
                                                (long long)d->Cout * d->Hout * dyp, dyp);
    if (!batched && !fused) return 0;
    if (mode == 2) return fused ? 2 : 1;
    const double pix = (double)d->N * d->H * d->W;
    const double nominal = 2.0 * pix * d->Cout * (double)d->Cin * 9.0;
    const bool dense_d1 = wgrad_dma_mixed(d) && !(d->x_pitch && d->x_pitch != d->W);
    const bool ragged = d->Cout % 256 != 0 || (d->Cin * 9) % 256 != 0;      // the direct kernel pays for its tile padding too
    const double dpad = ragged ? (double)((d->Cout + 31) / 32 * 32) / d->Cout * (double)((d->Cin * 9 + 255) / 256 * 256) / (d->Cin * 9) * 1.15
                               : 1.0;     // (x 1.15: the 8 x 2 wave layout of the ragged-M kernel)
    // (below 128 output channels the direct path is the register-staged wgrad2_kernel: 82...110 TF measured)
    const double t_direct = nominal * dpad / (d->Cout < 128 ? 100e12 : dense_d1 ? 123e12 : 130e12);
    const double f = dcfp_wino_exec_fraction(d->N, d->H, d->W, d->dil, d->Cout, d->Cin);
    double t_b = 1e30, t_f = 1e30;
    if (batched) {
        const double tiles = f * 9.0 / 16.0 * pix;
        const double pad2 = (double)((d->Cout + 255) / 256 * 256) / d->Cout * (double)((d->Cin + 255) / 256 * 256) / d->Cin;
        t_b = nominal * f * pad2 / 125e12 + (4.0 * pix * d->Cin + 64.0 * tiles * d->Cin) / 4.2e12 +
              (4.0 * pix * d->Cout + 64.0 * tiles * d->Cout) / 5.0e12 + 30e-6;
    }
    if (fused) {
        const double pad64 = (double)((d->Cout + 63) / 64 * 64) / d->Cout * (double)((d->Cin + 63) / 64 * 64) / d->Cin;
        t_f = nominal * f * pad64 / wino_wgrad_fused_rate() + 40e-6;      // (+ table kernel, split-K reduce + transform kernel)
    }
    if (fused && (t_f <= t_b || wino_wgrad_fused_forced())) return t_f < 0.97 * t_direct ? 2 : 0;
    return t_b < 0.97 * t_direct ? 1 : 0;
}
static bool wino_wgrad_pass(const DcfpConvDesc* d) { return wino_wgrad_kind(d) != 0; }

bool dcfp_wgrad_is_winograd(const DcfpConvDesc* d) { return wino_wgrad_pass(d); }
// ... and needs the forward's transformed input V (the batched path; the fused kernel transforms x itself)
bool dcfp_wgrad_wants_xform(const DcfpConvDesc* d) { return wino_wgrad_kind(d) == 1; }

// share of the nominal multiply-adds issued (the direct kernels execute every K-step: a tile of dW mixes all nine taps)
double dcfp_wgrad_exec_fraction(const DcfpConvDesc* d) {
    return wino_wgrad_pass(d) ? dcfp_wino_exec_fraction(d->N, d->H, d->W, d->dil, d->Cout, d->Cin) : 1.0;
}

size_t dcfp_conv2d_fwd_dgrad_workspace_bytes_(const DcfpConvDesc* d, int pass);

extern "C" size_t dcfp_conv2d_workspace_bytes(const DcfpConvDesc* d, int pass) {
    if (pass == DCFP_CONV_FWD || pass == DCFP_CONV_DGRAD) return dcfp_conv2d_fwd_dgrad_workspace_bytes_(d, pass);
    if (pass != DCFP_CONV_WGRAD || check_desc(d) != DCFP_OK) return 0;
    if (dcfp_stem_shape(d)) return dcfp_stem_wgrad_workspace_bytes(d);
    if (const int wk = wino_wgrad_kind(d)) {
        // (kind 2 can fall back to the batched path at run time - a dy slice whose image stride breaks the 31-bit offsets -
        //  so the query covers both where both apply)
        const size_t b1 = dcfp_wino_wgrad_ok(d->N, d->H, d->W, d->dil, d->Cout, d->Cin)
                              ? dcfp_wino_wgrad_workspace_bytes(d->N, d->H, d->W, d->dil, d->Cout, d->Cin) : 0;
        if (wk == 1) return b1;
        // kind 2 falls back at run time when the REAL image stride of dy (a channel slice of a wider tensor) breaks the
        // kernel's 31-bit offsets: to the batched path where that applies, else to the direct kernels - the query covers all
        const size_t b2 = dcfp_wino_wgrad_fused_workspace_bytes(d->N, d->H, d->W, d->dil, d->Cout, d->Cin);
        const Plan pl = make_plan(d);
        const size_t b0 = pl.splits > 1 ? (size_t)pl.splits * d->Cout * d->Cin * d->KH * d->KW * sizeof(float) : 0;
        const size_t m = b1 > b0 ? b1 : b0;
        return b2 > m ? b2 : m;
    }
    const Plan pl = make_plan(d);
    if (pl.splits <= 1) return 0;
    return (size_t)pl.splits * d->Cout * d->Cin * d->KH * d->KW * sizeof(float);
}

extern "C" int dcfp_conv2d_wgrad_f32_nchw(const DcfpConvDesc* d, const float* dy,
                                          int64_t dy_nstride, const float* x, float* dw, float* db,
                                          void* workspace, size_t workspace_bytes,
                                          dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !x || !dw) return DCFP_E_BADDESC;
    if (dcfp_gemv_shape(d) && !db)      // a 1x1 conv on a 1 x 1 map (ASPP image pool): conv_gemv.hip
        return dcfp_gemv_wgrad(d, dy, dy_nstride, x, dw, dcfp_s(stream));
    if (dcfp_stem_shape(d) && !db)      // Cin = 3, stride 2 (backbone.conv1.0): conv_stem.hip
        return dcfp_stem_wgrad(d, dy, dy_nstride, x, dw, workspace, workspace_bytes, dcfp_s(stream));
    int wk = wino_wgrad_kind(d);
    if (wk == 2) {      // the descriptor-level decision assumed a dense batch stride for dy: check the real one
        const int dyp = d->dy_pitch ? d->dy_pitch : d->Wout, xp = d->x_pitch ? d->x_pitch : d->W;
        const long long dyn = dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * dyp;
        if (!dcfp_wino_wgrad_fused_ok(d->N, d->H, d->W, d->dil, d->Cout, d->Cin, (long long)d->Cin * d->H * xp, xp, dyn, dyp))
            wk = dcfp_wino_wgrad_ok(d->N, d->H, d->W, d->dil, d->Cout, d->Cin) ? 1 : 0;     // batched Winograd, or the direct kernels below
    }
    if (wk) {
        const int dyp = d->dy_pitch ? d->dy_pitch : d->Wout, xp = d->x_pitch ? d->x_pitch : d->W;
        const long long dyn = dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * dyp;
        if (db && dyp != d->Wout) return DCFP_E_UNSUPPORTED;      // (the bias-gradient kernel reads dense rows; nothing launched yet)
        if (wk == 2) {
            rc = dcfp_wino_wgrad_fused_run(dy, dyn, dyp, x, (long long)d->Cin * d->H * xp, xp, dw, d->N, d->Cout, d->Cin, d->H,
                                           d->W, d->dil, workspace, workspace_bytes, dcfp_s(stream));
        } else
            rc = dcfp_wino_wgrad_run(dy, dyn, dyp, x, (long long)d->Cin * d->H * xp, xp, dw, d->N, d->Cout, d->Cin, d->H, d->W,
                                     d->dil, workspace, workspace_bytes, dcfp_s(stream));
        if (rc) return rc;
        if (db)
            hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)d->Cout), dim3(1024), 0, dcfp_s(stream), dy, dyn, db, d->N,
                               d->Hout * d->Wout, (int)((d->Hout * d->Wout) % 4 == 0 && dyn % 4 == 0 && dcfp_aligned16(dy)));
        DCFP_RETURN_LAUNCH();
    }
    const Plan pl = make_plan(d);
    const int T = d->KH * d->KW;
    const long long wn = (long long)d->Cout * d->Cin * T;
    const size_t need = pl.splits > 1 ? (size_t)pl.splits * wn * sizeof(float) : 0;
    if (need && (!workspace || workspace_bytes < need)) return DCFP_E_WORKSPACE;
    WgradParams p;
    p.dy = dy; p.x = x;
    p.out = pl.splits > 1 ? static_cast<float*>(workspace) : dw;
    p.x_pitch = d->x_pitch ? d->x_pitch : d->W;
    p.dy_pitch = d->dy_pitch ? d->dy_pitch : d->Wout;
    const bool pitched = p.x_pitch != d->W || p.dy_pitch != d->Wout;
    if (pitched && !dcfp_wgrad_pitch_ok(d)) return DCFP_E_UNSUPPORTED;
    if (db && p.dy_pitch != d->Wout) return DCFP_E_UNSUPPORTED;   // (the bias-gradient kernel reads dense rows) - before anything is launched
    p.dy_nstride = dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * p.dy_pitch;
    p.x_nstride = (long long)d->Cin * d->H * p.x_pitch;
    p.N = d->N; p.M = d->Cout; p.Cin = d->Cin; p.Nn = d->Cin * T;
    p.H = d->H; p.W = d->W; p.Ho = d->Hout; p.Wo = d->Wout; p.P = d->Hout * d->Wout;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    if ((long long)d->N * p.P >= (1LL << 30)) return DCFP_E_UNSUPPORTED;
    p.Kpix = d->N * p.P;
    p.kchunk = pl.kchunk; p.splits = pl.splits; p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
    p.quad_ok = (d->Wout % 4 == 0) ? 1 : 0;
    p.batch = 1; p.dy_bstride = p.x_bstride = 0;
    {   // 31-bit byte offsets relative to the first image of a split
        const long long span = (long long)pl.kchunk / p.P + 2;
        const long long big = p.dy_nstride > p.x_nstride ? p.dy_nstride : p.x_nstride;
        if (span * big * 4 >= (1LL << 31)) return DCFP_E_UNSUPPORTED;
    }
    if (wgrad3_ok(d, pl.cfg))
        rc = dcfp_wgrad3_launch(dy, p.dy_nstride, x, p.x_nstride, p.out, d->N, d->Cout, d->Cin, T, d->H, d->W,
                                d->Hout, d->Wout, d->pad, d->dil, p.Kpix, pl.kchunk, pl.splits, pl.tiles_m,
                                pl.tiles_n, dcfp_s(stream));
    else if (pl.wide) {
        const long long blocks = (long long)pl.tiles_m * pl.tiles_n * pl.splits;
        if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
        rc = T == 1 ? launch_dma<1, false, true>(p, blocks, dcfp_s(stream)) : launch_dma<9, false, true>(p, blocks, dcfp_s(stream));
    } else if (wgrad_dma_ok(d, pl.cfg)) {
        const long long blocks = (long long)pl.tiles_m * pl.tiles_n * pl.splits;
        if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
        // (pitched x: every shifted quad reads data or the rows' zero tails - the un-mixed kernel does it all)
        rc = T == 1 ? (pl.half ? launch_dma<1, false, false, false, true>(p, blocks, dcfp_s(stream)) : launch_dma<1, false>(p, blocks, dcfp_s(stream)))
                    : (wgrad_dma_mixed(d) && p.x_pitch == d->W) ? launch_dma<9, true>(p, blocks, dcfp_s(stream))
                                                                 : launch_dma<9, false>(p, blocks, dcfp_s(stream));
    } else
        rc = T == 1 ? launch_taps<1>(p, pl, dcfp_s(stream)) : launch_taps<9>(p, pl, dcfp_s(stream));
    if (rc) return rc;
    if (pl.splits > 1) {
        if (wn % 4 == 0 && dcfp_aligned16(workspace) && dcfp_aligned16(dw)) {
            const long long n4 = wn / 4;
            hipLaunchKernelGGL(splitk_reduce_vec4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                               dcfp_s(stream), static_cast<const float*>(workspace), dw, n4, wn, pl.splits);
        } else {
            long long b = (wn + 255) / 256;
            if (b > 4096) b = 4096;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)b), dim3(256), 0, dcfp_s(stream),
                               static_cast<const float*>(workspace), dw, wn, pl.splits);
        }
    }
    if (db) {
        hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)d->Cout), dim3(1024), 0, dcfp_s(stream),
                           dy, p.dy_nstride, db, d->N, p.P,
                           (int)(p.P % 4 == 0 && p.dy_nstride % 4 == 0 && dcfp_aligned16(dy) && p.dy_pitch == d->Wout));
    }
    DCFP_RETURN_LAUNCH();
}

// Weight gradient of a conv whose forward call left its transformed input behind (dcfp_conv2d_fwd_keep_f32_nchw):
// `xform` replaces x - the x transform pass (a fifth to a quarter of a Winograd weight gradient) is not run again.
extern "C" int dcfp_conv2d_wgrad_kept_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride,
                                               const float* xform, size_t xform_bytes, float* dw, void* workspace,
                                               size_t workspace_bytes, dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !xform || !dw) return DCFP_E_BADDESC;
    const size_t need = dcfp_conv2d_xform_bytes(d);
    if (need == 0) return DCFP_E_UNSUPPORTED;
    if (xform_bytes < need || !dcfp_aligned16(xform)) return DCFP_E_WORKSPACE;
    const int dyp = d->dy_pitch ? d->dy_pitch : d->Wout;
    const long long dyn = dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * dyp;
    return dcfp_wino_wgrad_run(dy, dyn, dyp, nullptr, 0, 0, dw, d->N, d->Cout, d->Cin, d->H, d->W, d->dil, workspace,
                               workspace_bytes, dcfp_s(stream), xform);
}

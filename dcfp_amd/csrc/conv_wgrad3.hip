// conv_wgrad3.hip — EXPERIMENTAL (opt-in: DCFP_CONV_MATH=bf16x3): conv weight gradient with the
// fp32 operands split three ways into bf16 (x = hi + mid + lo, exact for normal numbers) and
// the product formed from the SIX leading bf16 x bf16 terms on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
//     a*b ~= ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm        (dropped terms <= ~3*2^-24 |a*b|)
// i.e. fp32-grade products at 6/16 of the cost of the exact-fp32 MFMA (1/16 of the bf16 rate):
// 2.67x the fp32-MFMA roofline.  The default path stays exact fp32 (conv_wgrad.hip); this kernel
// exists to measure what the split buys at equal parity tolerances.
//
// Both wgrad operands are pixel(K)-contiguous in NCHW, which is exactly the bf16 MFMA fragment
// shape (8 consecutive k per lane), so no transposition is needed: LDS holds three bf16 planes
// per operand as [row][16 k] (pitch 24 bf16 = 48 B: conflict-free 16-byte reads).  A K-step is
// only 96 MFMAs x 32 cycles, too short to hide a global load, so tiles are prefetched TWO
// K-steps ahead through two register sets (K loop unrolled by 2).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;      // pixels per K-step = one bf16 MFMA depth
constexpr int PITCH = 24;   // bf16 per LDS row (16 + 8 pad)
constexpr unsigned kOob = 0x80000000u;
constexpr unsigned kMaxRecords = 0x7ffffffcu;

struct Wgrad3Params {
    const float* dy;
    const float* x;
    float* out;
    long long dy_nstride, x_nstride;
    int N, M, Cin, Nn;
    int H, W, Ho, Wo, P, pad, dil;
    int Kpix, kchunk, splits, tiles_m, tiles_n;
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void split4(const float (&v)[4], bf16x4& hi, bf16x4& mid, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        const float r1 = v[e] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[e] = h; mid[e] = m; lo[e] = (__bf16)r2;
    }
}

// 256 x 256 tile, 2 x 2 waves, 4 x 4 MFMA tiles per wave; requires Wo % 4 == 0 and stride 1.
template <int TAPS>
__global__ void __launch_bounds__(256) wgrad3_kernel(const Wgrad3Params p) {
    constexpr int BM = 256, BN = 256, ROWS = 64, PA = 4, PB = 4;
    constexpr int PLANE = BM * PITCH;            // bf16 elements per plane
    extern __shared__ __attribute__((aligned(16))) __bf16 smem3[];
    __bf16* As = smem3;                          // [2][3][BM][PITCH]
    __bf16* Bs = smem3 + 2 * 3 * PLANE;          // [2][3][BN][PITCH]

    const int tiles = p.tiles_m * p.tiles_n;
    int split, tile;
    {
        const int full = (p.splits / 8) * 8;
        const int gsz = 8 * tiles;
        const int b = blockIdx.x;
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

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int kx = lane >> 4, rrow = wid * 16 + (lane & 15);

    const int img0 = kbeg / p.P;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy + (long long)img0 * p.dy_nstride), 0, kMaxRecords, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (long long)img0 * p.x_nstride), 0, kMaxRecords, 0x00020000);
    const int dyn = (int)p.dy_nstride, xn = (int)p.x_nstride;

    int a_off[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        int m = m0 + rrow + ROWS * j;
        m = m < p.M ? m : p.M - 1;
        a_off[j] = m * p.P;
    }
    int b_coff[PB], b_dh[PB], b_dw[PB];
    const int HW = p.H * p.W;
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
        b_coff[j] = ci * HW + b_dh[j] * p.W + b_dw[j];
    }

    int c_im, c_oh, c_ow;
    {
        const int q0 = kbeg + 4 * kx;
        const int imabs = q0 / p.P;
        const int pq = q0 - imabs * p.P;
        c_im = imabs - img0;
        c_oh = pq / p.Wo;
        c_ow = pq - c_oh * p.Wo;
    }
    float areg[1][PA][4], breg[1][PB][4];   // staging registers (next K-step)

    // issue the global loads of the K-step starting at pixel kbase into register set S
    auto load_tile = [&](int kbase, auto s_) {
        constexpr int S = decltype(s_)::value;
        const int q0 = kbase + 4 * kx;
        const bool q_ok = q0 < kend;
        const int q_a = c_im * dyn + c_oh * p.Wo + c_ow;
        const int q_x = c_im * xn + c_oh * p.W + c_ow;
        const int q_ih = c_oh, q_iw = c_ow;
        c_ow += BK;
        while (c_ow >= p.Wo) { c_ow -= p.Wo; ++c_oh; }
        while (c_oh >= p.Ho) { c_oh -= p.Ho; ++c_im; }
        static_for<0, PA>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const unsigned off = q_ok ? (unsigned)(q_a + a_off[j]) * 4u : kOob;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0));
            static_for<0, 4>([&](auto e_) { constexpr int e = decltype(e_)::value; areg[S][j][e] = v[e]; });
        });
        static_for<0, PB>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const int hh = q_ih + b_dh[j], ww = q_iw + b_dw[j];
            const bool rowok = q_ok && hh >= 0 && hh < p.H;
            const unsigned base = (unsigned)(q_x + b_coff[j]) * 4u;
            if (!rowok || (ww >= 0 && ww + 3 < p.W)) {
                const f32x4 v = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, rowok ? base : kOob, 0, 0));
                static_for<0, 4>([&](auto e_) { constexpr int e = decltype(e_)::value; breg[S][j][e] = v[e]; });
            } else {
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    const bool ok = (ww + e) >= 0 && (ww + e) < p.W;
                    breg[S][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        b_rsrc, ok ? base + 4u * e : kOob, 0, 0));
                });
            }
        });
    };
    // split register set S into bf16 planes and store into LDS buffer `buf`
    auto store_tile = [&](int buf, auto s_) {
        constexpr int S = decltype(s_)::value;
        // one opaque per-thread base per call: every LDS address below is base + immediate, instead
        // of dozens of loop-invariant addresses hoisted out of the K loop (they spilled to scratch)
        int so = rrow * PITCH + 4 * kx;
        asm volatile("" : "+v"(so));
        __bf16* a = As + buf * (3 * PLANE) + so;
        __bf16* b = Bs + buf * (3 * PLANE) + so;
        static_for<0, PA>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            bf16x4 h, m, l;
            split4(areg[S][j], h, m, l);
            constexpr int o = ROWS * j * PITCH;
            *reinterpret_cast<bf16x4*>(a + o) = h;
            *reinterpret_cast<bf16x4*>(a + PLANE + o) = m;
            *reinterpret_cast<bf16x4*>(a + 2 * PLANE + o) = l;
        });
        static_for<0, PB>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            bf16x4 h, m, l;
            split4(breg[S][j], h, m, l);
            constexpr int o = ROWS * j * PITCH;
            *reinterpret_cast<bf16x4*>(b + o) = h;
            *reinterpret_cast<bf16x4*>(b + PLANE + o) = m;
            *reinterpret_cast<bf16x4*>(b + 2 * PLANE + o) = l;
        });
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    using S0 = std::integral_constant<int, 0>;
    if (nk > 0) {
        load_tile(kbeg, S0{});
        store_tile(0, S0{});
    }
    __syncthreads();

    const int a_row = (wm * 128 + l31) * PITCH + 8 * lhi;
    const int b_row = (wn * 128 + l31) * PITCH + 8 * lhi;

    // one K-step: MFMAs on LDS buffer kt&1; the loads of tile kt+1 are issued at its start and
    // split + stored into the other buffer at its end
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        int ao = a_row + cur * (3 * PLANE), bo = b_row + cur * (3 * PLANE);
        asm volatile("" : "+v"(ao), "+v"(bo));          // keep fragment addresses base + immediate
        const __bf16* a = As + ao;
        const __bf16* b = Bs + bo;
        if (more) load_tile(kbeg + (kt + 1) * BK, S0{});
        static_for<0, 2>([&](auto ip_) {
            constexpr int ip = decltype(ip_)::value;
            bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
            static_for<0, 2>([&](auto ii_) {
                constexpr int ii = decltype(ii_)::value;
                ah[ii] = *reinterpret_cast<const bf16x8*>(a + (2 * ip + ii) * 32 * PITCH);
                am[ii] = *reinterpret_cast<const bf16x8*>(a + PLANE + (2 * ip + ii) * 32 * PITCH);
                al[ii] = *reinterpret_cast<const bf16x8*>(a + 2 * PLANE + (2 * ip + ii) * 32 * PITCH);
            });
            bh[0] = *reinterpret_cast<const bf16x8*>(b);
            bm[0] = *reinterpret_cast<const bf16x8*>(b + PLANE);
            bl[0] = *reinterpret_cast<const bf16x8*>(b + 2 * PLANE);
            static_for<0, 4>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                constexpr int c = j & 1;
                if constexpr (j < 3) {
                    bh[c ^ 1] = *reinterpret_cast<const bf16x8*>(b + (j + 1) * 32 * PITCH);
                    bm[c ^ 1] = *reinterpret_cast<const bf16x8*>(b + PLANE + (j + 1) * 32 * PITCH);
                    bl[c ^ 1] = *reinterpret_cast<const bf16x8*>(b + 2 * PLANE + (j + 1) * 32 * PITCH);
                }
                if constexpr (ip == 1 && j == 2) { if (more) store_tile(cur ^ 1, S0{}); }
                static_for<0, 2>([&](auto ii_) {
                    constexpr int ii = decltype(ii_)::value;
                    constexpr int i = 2 * ip + ii;
                    f32x16 cc = acc[i][j];
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ii], bh[c], cc, 0, 0, 0);   // small terms first
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ii], bl[c], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ii], bm[c], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ii], bh[c], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ii], bm[c], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ii], bh[c], cc, 0, 0, 0);
                    acc[i][j] = cc;
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        __syncthreads();
    }

    float* o = p.out + (long long)split * p.M * p.Nn;
    int ncol = n0 + wn * 128 + l31;
    asm volatile("" : "+v"(ncol));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int m = m0 + wm * 128 + i * 32 + row;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nn = ncol + j * 32;
                if (nn < p.Nn) o[(long long)m * p.Nn + nn] = acc[i][j][r];
            }
        }
    }
}

}  // namespace

// Called from conv_wgrad.hip when DCFP_CONV_MATH=bf16x3 and the shape qualifies.
int dcfp_wgrad3_launch(const float* dy, long long dy_nstride, const float* x, long long x_nstride,
                       float* out, int N, int M, int Cin, int T, int H, int W, int Ho, int Wo, int pad,
                       int dil, int Kpix, int kchunk, int splits, int tiles_m, int tiles_n,
                       hipStream_t stream) {
    Wgrad3Params p;
    p.dy = dy; p.x = x; p.out = out; p.dy_nstride = dy_nstride; p.x_nstride = x_nstride;
    p.N = N; p.M = M; p.Cin = Cin; p.Nn = Cin * T; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.P = Ho * Wo;
    p.pad = pad; p.dil = dil; p.Kpix = Kpix; p.kchunk = kchunk; p.splits = splits;
    p.tiles_m = tiles_m; p.tiles_n = tiles_n;
    const long long blocks = (long long)tiles_m * tiles_n * splits;
    const size_t lds = (size_t)2 * 2 * 3 * 256 * PITCH * sizeof(__bf16);
    auto k1 = wgrad3_kernel<1>;
    auto k9 = wgrad3_kernel<9>;
    const void* kern = T == 1 ? reinterpret_cast<const void*>(k1) : reinterpret_cast<const void*>(k9);
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    if (T == 1)
        hipLaunchKernelGGL(wgrad3_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, stream, p);
    else
        hipLaunchKernelGGL(wgrad3_kernel<9>, dim3((unsigned)blocks), dim3(256), lds, stream, p);
    e = hipGetLastError();
    return e == hipSuccess ? DCFP_OK : (int)e;
}

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
// only 96 MFMAs x 32 cycles, so each staged quad is re-loaded the moment it has been split and
// stored: every global load gets one full K-step to land.
// Diagnosis builds: -DW3_DBG_NOLOAD (global loads fetch nothing), -DW3_DBG_NOSTAGE (no split /
// LDS store / reload at all) - the same instruction stream minus one cost.
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

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 t;
    t[0] = (__bf16)a; t[1] = (__bf16)b;        // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, t);
}
// One step of the exact split of a pair: emit the bf16 pair nearest (a, b), leave the remainders.
__device__ __forceinline__ unsigned peel(float& a, float& b) {
    const unsigned h = pack_bf16(a, b);
    a -= __builtin_bit_cast(float, h << 16);
    b -= __builtin_bit_cast(float, h & 0xffff0000u);
    return h;
}
// The split of one staged quad v[4] into packed planes pk = {hi01, hi23, mid01, mid23, lo01, lo23},
// cut into four stages of ~5 VALU instructions so that each hides behind one MFMA.
template <int K>
__device__ __forceinline__ void split_stage(float (&v)[4], unsigned (&pk)[6]) {
    // the empty asm pins each stage's results where they are computed (IR-level sinking would
    // otherwise move the arithmetic across the scheduling barriers to its first use)
    if constexpr (K == 0) { pk[0] = peel(v[0], v[1]); asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(pk[0])); }
    if constexpr (K == 1) { pk[1] = peel(v[2], v[3]); asm volatile("" : "+v"(v[2]), "+v"(v[3]), "+v"(pk[1])); }
    if constexpr (K == 2) {
        pk[2] = peel(v[0], v[1]); pk[4] = pack_bf16(v[0], v[1]);
        asm volatile("" : "+v"(pk[2]), "+v"(pk[4]));
    }
    if constexpr (K == 3) {
        pk[3] = peel(v[2], v[3]); pk[5] = pack_bf16(v[2], v[3]);
        asm volatile("" : "+v"(pk[3]), "+v"(pk[5]));
    }
}
__device__ __forceinline__ void split4(float (&v)[4], unsigned (&pk)[6]) {
    split_stage<0>(v, pk); split_stage<1>(v, pk); split_stage<2>(v, pk); split_stage<3>(v, pk);
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
    float areg[2][PA][4], breg[2][PB][4];   // two staging sets: one quad per (operand, row group)
    int q_a = 0, q_x = 0, q_ih = 0, q_iw = 0;
    bool q_ok = false;

    // pixel coordinates of this lane's quad in the K-step starting at kbase (advances the walker)
    auto next_coords = [&](int kbase) {
        q_ok = (kbase + 4 * kx) < kend;
#if defined(W3_DBG_NOLOAD)
        q_ok = false;
#endif
        q_a = c_im * dyn + c_oh * p.Wo + c_ow;
        q_x = c_im * xn + c_oh * p.W + c_ow;
        q_ih = c_oh; q_iw = c_ow;
        c_ow += BK;
        while (c_ow >= p.Wo) { c_ow -= p.Wo; ++c_oh; }
        while (c_oh >= p.Ho) { c_oh -= p.Ho; ++c_im; }
    };
    auto load_a = [&](auto s_, auto j_) {
        constexpr int S = decltype(s_)::value;
        constexpr int j = decltype(j_)::value;
        const unsigned off = q_ok ? (unsigned)(q_a + a_off[j]) * 4u : kOob;
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0));
        static_for<0, 4>([&](auto e_) { constexpr int e = decltype(e_)::value; areg[S][j][e] = v[e]; });
    };
    auto load_b = [&](auto s_, auto j_) {
        constexpr int S = decltype(s_)::value;
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
    };
    // store the packed planes of one split quad into LDS buffer `buf`; every LDS address is one
    // opaque per-thread base + immediate (loop-invariant addresses hoisted out of the K loop
    // spilled to scratch)
    auto store_planes = [&](__bf16* base, int buf, auto j_, const unsigned (&pk)[6]) {
        constexpr int j = decltype(j_)::value;
        int so = rrow * PITCH + 4 * kx + buf * (3 * PLANE);
        asm volatile("" : "+v"(so));
        __bf16* d = base + so;
        constexpr int o = ROWS * j * PITCH;
        u32x2 h, m, l;
        h[0] = pk[0]; h[1] = pk[1]; m[0] = pk[2]; m[1] = pk[3]; l[0] = pk[4]; l[1] = pk[5];
        *reinterpret_cast<u32x2*>(d + o) = h;
        *reinterpret_cast<u32x2*>(d + PLANE + o) = m;
        *reinterpret_cast<u32x2*>(d + 2 * PLANE + o) = l;
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    // prologue: tile 0 into LDS buffer 0; the loads of tiles 1 and 2 in flight (sets 1 and 0)
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    next_coords(kbeg);
    static_for<0, 4>([&](auto j_) { load_a(S0{}, j_); load_b(S0{}, j_); });
    next_coords(kbeg + BK);
    static_for<0, 4>([&](auto j_) { load_a(S1{}, j_); load_b(S1{}, j_); });
    static_for<0, 4>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        unsigned pk[6];
        split4(areg[0][j], pk); store_planes(As, 0, j_, pk);
        split4(breg[0][j], pk); store_planes(Bs, 0, j_, pk);
    });
    next_coords(kbeg + 2 * BK);
    static_for<0, 4>([&](auto j_) { load_a(S0{}, j_); load_b(S0{}, j_); });
    __syncthreads();

    const int a_row = (wm * 128 + l31) * PITCH + 8 * lhi;
    const int b_row = (wn * 128 + l31) * PITCH + 8 * lhi;

    // One K-step: 96 MFMAs on LDS buffer kt&1 in 8 slots.  Slot s splits + stores staged quad s of
    // tile kt+1 (register set S) into the other buffer and at once re-issues that quad's load for
    // tile kt+3 into the same registers, so every global load has two full K-steps (~6000 cycles)
    // to land.  Tiles past the end load zeros (out-of-range buffer offsets fetch nothing), which
    // also lets the loop run an even number of steps.
    auto kstep = [&](int kt, auto s_) {
        constexpr int S = decltype(s_)::value;
        const int cur = kt & 1;
        int ao = a_row + cur * (3 * PLANE), bo = b_row + cur * (3 * PLANE);
        asm volatile("" : "+v"(ao), "+v"(bo));          // keep fragment addresses base + immediate
        const __bf16* a = As + ao;
        const __bf16* b = Bs + bo;
        // coordinates of tile kt+2 (consumed by the loads re-issued below)
        next_coords(kbeg + (kt + 3) * BK);
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
                // Program order is pinned with scheduling barriers: the matrix pipe is busy 32
                // cycles per MFMA, so one ~5-instruction stage of the staging work (split, LDS
                // store, re-issued load) sits behind each of the first MFMAs instead of behind all 12.
                unsigned pk[6];
                auto stage = [&](auto k_) {
                    constexpr int k = decltype(k_)::value;
#if !defined(W3_DBG_NOSTAGE)
                    if constexpr (k >= 1 && k <= 4) {
                        if constexpr (ip == 0) split_stage<k - 1>(areg[S][j], pk);
                        else                   split_stage<k - 1>(breg[S][j], pk);
                    }
                    if constexpr (k == 5) store_planes(ip == 0 ? As : Bs, cur ^ 1, j_, pk);
                    if constexpr (k == 6) { if constexpr (ip == 0) load_a(s_, j_); else load_b(s_, j_); }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
                f32x16 c0 = acc[2 * ip][j], c1 = acc[2 * ip + 1][j];
                // small terms first; the two accumulator chains interleaved
#define W3_MFMA(K0, A, B)                                                                  \
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[c], c0, 0, 0, 0);     \
                stage(std::integral_constant<int, K0>{});                                  \
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[c], c1, 0, 0, 0);     \
                stage(std::integral_constant<int, K0 + 1>{});
                W3_MFMA(0, al, bh)
                W3_MFMA(2, ah, bl)
                W3_MFMA(4, am, bm)
                W3_MFMA(6, am, bh)
                W3_MFMA(8, ah, bm)
                W3_MFMA(10, ah, bh)
#undef W3_MFMA
                acc[2 * ip][j] = c0; acc[2 * ip + 1][j] = c1;
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, S1{});
        kstep(kt + 1, S0{});
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

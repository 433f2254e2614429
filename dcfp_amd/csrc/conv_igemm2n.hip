// conv_igemm2n.hip — LDS-DMA implicit-GEMM forward / data-gradient kernel for RAGGED output-channel counts
// (pruned models: pruners/channel_pruner.py:29-74 leaves widths such as 236 / 154 / 83 / 57 / 40; the fine-tune
// stage scripts/cs/finetune.sh:7-32 trains them).
//
// The 256 x 256-tile kernels (conv_igemm2.hip) give every wave 4 x 4 MFMA tiles whose rows interleave by 4
// (a lane owns 4 consecutive channels), so channels past M are spread over all row blocks and a ragged tile
// costs as much as a full one: 512->154 ran 40 % of its MFMAs on zero rows, 2048->83 fell to the register-staged
// 128-row tile.  Here the four waves sit side by side along the PIXELS (64 each) and every wave holds all 8 row
// blocks of the tile, 8 x 2 MFMA tiles of 32 x 32:
//   * the weights are staged with the rows of a 256-row tile PERMUTED (Wp position 8 l + i <-> channel 32 i + l,
//     done once by the weight-permute kernel), so that MFMA row block i covers the 32 CONSECUTIVE channels
//     32 i .. 32 i + 31 while a lane still fetches its 8 A values with two 16-byte LDS reads;
//   * row blocks at or past M are dead for the whole tile (wave-uniform): their MFMAs are skipped, so the cost
//     follows ceil(M / 32) instead of ceil(M / 256) * 8 - 32-row granularity with ONE kernel for every M;
//   * everything on the pixel side is the 256-pixel LDS-DMA pipeline of igemm2_dma_kernel: 1 KB k-rows copied
//     global -> LDS by `buffer_load_dwordx4 ... lds`, one tap per K-step for 3x3 convs, shifted 16-byte quads
//     (column shifts that are multiples of 4, or a row-pitched source with a zero tail: DcfpConvDesc.x_pitch).
// A lane owns 2 consecutive pixels (8-byte stores); the dispatcher uses this kernel only where it saves >= 20 %
// of the MFMA work against the tile the shape would otherwise get.
#include "igemm2_common.h"
#include <stdlib.h>

namespace {

typedef unsigned u32x4 __attribute__((vector_size(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// raw buffer descriptor (stride 0, 32-bit data) from wave-uniform values; readfirstlane tells the compiler so
// (the inline-asm copies take the descriptor and the scalar offset as "s" operands)
__device__ __forceinline__ u32x4 make_desc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 d = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
               (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    return d;
}

template <int TAPS, bool ACC>
__global__ void __launch_bounds__(256) igemm2_dma8_kernel(const Igemm2Params p) {
    constexpr int BM = 256, BN = 256, NRB = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BK][BM]  (rows permuted: position 8 l + i holds channel m0 + 32 i + l)
    float* Bs = smem + 2 * BK * BM;    // [2][BK][BN]
    const int group = 8 * p.tiles_m;
    const int g = blockIdx.x / group, local = blockIdx.x - g * group;
    int nt = g * 8 + (local & 7);
    const int mt = local >> 3;
    if (nt >= p.tiles_n_total) return;
    int img = nt / p.tiles_per_img;
    int ti = nt - img * p.tiles_per_img;
    if (TAPS == 9 && p.tapskip == 2) {     // centre-out dispatch order (see igemm2_dma_kernel)
        const int k = nt / p.N, c = p.tiles_per_img >> 1;
        img = nt - k * p.N;
        ti = (k & 1) ? c - 1 - (k >> 1) : c + (k >> 1);
        nt = img * p.tiles_per_img + ti;
    }
    const int p0 = ti * BN;
    const int m0 = mt * BM;
    int nb = (p.M - m0 + 31) >> 5;                 // live row blocks of this tile (block-uniform)
    nb = nb > NRB ? NRB : nb;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HiWi = p.Hi * p.in_pitch;             // channel stride of the B source
    const int padw = p.in_pitch - p.Wi;
    // wave w copies k-rows 4w .. 4w+3 of both operands
    unsigned a_voff[4], b_row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a_voff[q] = (unsigned)((4 * wid + q) * p.Mpad) * 4u + lane * 16u;
        b_row[q] = (unsigned)((4 * wid + q) * HiWi) * 4u;
    }
    // this lane's pixel quad 4*lane .. 4*lane+3 of the tile
    int q_oh, q_ow;
    bool q_in;
    {
        const int pp = p0 + 4 * lane;
        q_in = pp < p.P;                                // P % 4 == 0: a quad is inside or outside as a whole
        q_oh = pp / p.Wo; q_ow = pp - q_oh * p.Wo;
    }
    const int kpt = p.CkP / BK;
    // kernel rows that lie wholly in the padding for this tile are skipped (see igemm2_dma_kernel)
    int t_beg = 0, t_end = TAPS;
    if (TAPS == 9 && p.tapskip) {
        const int oh_lo = p0 / p.Wo;
        int oh_hi = (p0 + BN - 1 < p.P ? p0 + BN - 1 : p.P - 1) / p.Wo;
        oh_hi = oh_hi < p.Ho ? oh_hi : p.Ho - 1;
        // (the row offsets run upwards for the forward pass and downwards for dgrad: the live rows are contiguous)
        auto live = [&](int kh) { const int o = p.off0 + kh * p.offstep; return oh_hi + o >= 0 && oh_lo + o < p.Hi; };
        int kh_lo = 0, kh_hi = 2;
        while (kh_lo < 2 && !live(kh_lo)) ++kh_lo;
        while (kh_hi > kh_lo && !live(kh_hi)) --kh_hi;
        t_beg = 3 * kh_lo; t_end = 3 * kh_hi + 3;
    }
    const int nk = (t_end - t_beg) * kpt;
    unsigned boff4 = 0;
    int ld_t = t_beg, ld_cb = 0;      // (tap, channel block) of the K-step the loader copies next
    auto set_tap = [&](int t) {
        const int kh = (TAPS == 9) ? t / 3 : 0;
        const int kw = (TAPS == 9) ? t - kh * 3 : 0;
        const int offh = p.off0 + kh * p.offstep, offw = p.off0 + kw * p.offstep;
        const int hh = q_oh + offh, ww = q_ow + offw;
        const bool ok = q_in && hh >= 0 && hh < p.Hi && ww >= -padw && ww + 3 < p.Wi + padw;
        boff4 = ok ? (unsigned)(hh * p.in_pitch + ww + padw) * 4u : kOob;
    };
    const u32x4 a_desc = make_desc(p.wp, 0x7ffffffcu);
    const u32x4 b_desc = make_desc(p.in + (long long)img * p.in_nstride - padw, (unsigned)(p.Ck * HiWi + padw) * 4u);
    const unsigned lds_a0 = (unsigned)(size_t)(lds_ptr)As, lds_b0 = (unsigned)(size_t)(lds_ptr)Bs;
    auto issue = [&](int buf) {
        const unsigned a_s = __builtin_amdgcn_readfirstlane((unsigned)((ld_t * p.CkP + ld_cb * BK) * p.Mpad + m0) * 4u);
        const unsigned b_cb = (unsigned)(ld_cb * BK * HiWi) * 4u;
        const unsigned b_s = __builtin_amdgcn_readfirstlane(0u);
        static_for<0, 4>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            const unsigned la = __builtin_amdgcn_readfirstlane(lds_a0 + (unsigned)((buf * BK + 4 * wid + q) * BM) * 4u);
            const unsigned lb = __builtin_amdgcn_readfirstlane(lds_b0 + (unsigned)((buf * BK + 4 * wid + q) * BN) * 4u);
            const unsigned av = a_voff[q], as_ = a_s, bs_ = b_s;
            const unsigned bv = boff4 + b_row[q] + b_cb;
            const u32x4 ad = a_desc, bd = b_desc;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(la), "v"(av), "s"(ad), "s"(as_) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(lb), "v"(bv), "s"(bd), "s"(bs_) : "memory", "m0");
        });
        if (++ld_cb == kpt) {
            ld_cb = 0;
            if (++ld_t < t_end) set_tap(ld_t);
        }
    };
    auto retire = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    f32x16 acc[NRB][2];
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    set_tap(t_beg);
    issue(0);
    retire();
    const int a_off = 8 * l31;                  // the lane's 8 A values: channels 32 i + l31, i = 0..7
    const int b_off = 64 * wid + 2 * l31;       // the lane's 2 pixels
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1);
        const float* a = As + cur * (BK * BM) + a_off + lhi * BM;
        const float* b = Bs + cur * (BK * BN) + b_off + lhi * BN;
        f32x4 af[2][2];
        f32x2 bf[2];
        af[0][0] = *reinterpret_cast<const f32x4*>(a);
        af[0][1] = *reinterpret_cast<const f32x4*>(a + 4);
        bf[0] = *reinterpret_cast<const f32x2*>(b);
        static_for<0, BK / 2>([&](auto kk_) {
            constexpr int kk = decltype(kk_)::value;
            constexpr int fc = kk & 1;
            if constexpr (kk + 1 < BK / 2) {
                af[fc ^ 1][0] = *reinterpret_cast<const f32x4*>(a + (2 * kk + 2) * BM);
                af[fc ^ 1][1] = *reinterpret_cast<const f32x4*>(a + (2 * kk + 2) * BM + 4);
                bf[fc ^ 1] = *reinterpret_cast<const f32x2*>(b + (2 * kk + 2) * BN);
            }
            static_for<0, NRB>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                if (i < nb) {                    // block-uniform: dead row blocks cost nothing
                    const float av = af[fc][i >> 2][i & 3];
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[fc][0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[fc][1], acc[i][1], 0, 0, 0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        retire();
    }
    // ---- epilogue: 8-byte stores of the lane's 2 pixels, row = m0 + 32 i + MFMA row
    float* o_img = p.out + (long long)img * p.out_nstride;
    const int pix = p0 + b_off;
    if (pix < p.P) {                                 // P % 4 == 0 and pix even: both pixels in or out
        static_for<0, NRB>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            if (i < nb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (m < p.M) {
                        float* dst = o_img + (long long)m * p.P + pix;
                        f32x2 v = {acc[i][0][r], acc[i][1][r]};
                        if (p.bias) { const float bsv = p.bias[m]; v[0] += bsv; v[1] += bsv; }
                        if constexpr (ACC) v += *reinterpret_cast<const f32x2*>(dst);
                        *reinterpret_cast<f32x2*>(dst) = v;
                    }
                }
            }
        });
    }
}

}  // namespace

int dcfp_igemm2n_launch(const Igemm2Params& p, int T, hipStream_t stream) {
    const long long groups = ((long long)p.tiles_n_total + 7) / 8;
    const long long blocks = groups * 8 * p.tiles_m;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * BK * 512 * sizeof(float);
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, p);
        DCFP_RETURN_LAUNCH();
    };
    if (T == 1) return p.accumulate ? launch(igemm2_dma8_kernel<1, true>) : launch(igemm2_dma8_kernel<1, false>);
    return p.accumulate ? launch(igemm2_dma8_kernel<9, true>) : launch(igemm2_dma8_kernel<9, false>);
}

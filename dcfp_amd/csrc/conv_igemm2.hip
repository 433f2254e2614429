// conv_igemm2.hip — second-generation implicit-GEMM conv forward / data-gradient kernel
// (fp32 MFMA, NCHW).  Same math and tiling as conv_igemm.hip (see its header); what changed is
// everything AROUND the MFMAs, after the round-1 profile showed the matrix pipe idle ~45 % of
// the time while one wave per SIMD issued staging code and MFMAs back to back:
//
//  * K is ordered TAP-MAJOR (k = tap*C + c): a 16-deep K-step touches ONE tap, so the
//    per-pixel source offsets / border predicates are computed once per tap, and a row of the
//    B tile is just `+ c*H*W`.  Staging costs ~1 VALU per load instead of ~10.
//  * the A operand comes from a pre-permuted, zero-padded copy of the weights
//    Wp[tap][c][m] (m contiguous, built by permute_weights_kernel in the same call: <= 19 MB,
//    microseconds): 16-byte coalesced loads, 16-byte LDS stores, no predication at all.
//  * MFMA operand fragments are double-buffered in registers (the ds_read for step kk+1 is
//    issued before the 16 MFMAs of step kk), and the next tile's global loads / LDS stores are
//    spread over the eight 16-MFMA slots of a K-step instead of forming one serial block.
#include "igemm2_common.h"
#include <queue>
#include <vector>
#include <stdio.h>
#include <stdlib.h>

namespace {

// perm8 (conv_igemm2n.hip): inside every 256-row tile, position 8 l + i holds channel 32 i + l
__device__ __forceinline__ int wp_row(int m, int perm8) {
    if (!perm8) return m;
    const int j = m & 255;
    return (m & ~255) + 32 * (j & 7) + (j >> 3);
}

// Wp[t][c][m] = W[m*sAm + c*sAc + t]  (zero for c >= Ck or m >= M)
__global__ void __launch_bounds__(256)
permute_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int T, int Ck, int CkP,
                       int M, int Mpad, int sAm, int sAc, int perm8) {
    const long long total = (long long)T * CkP * Mpad;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (long long)gridDim.x * 256) {
        const int m = wp_row((int)(i % Mpad), perm8);
        const long long r = i / Mpad;
        const int c = (int)(r % CkP);
        const int t = (int)(r / CkP);
        float v = 0.f;
        if (m < M && c < Ck) v = w[(long long)m * sAm + (long long)c * sAc + t];
        wp[i] = v;
    }
}

// The same permutation for MANY weight tensors in one launch (all convs of a model, forward and dgrad
// layouts, once per optimizer step): block -> entry by binary search over first_block, 2048 elements per block.
constexpr int kWpBlockElems = DCFP_WP_BLOCK_ELEMS;
__global__ void __launch_bounds__(256)
permute_weights_multi_kernel(const DcfpWpEntry* __restrict__ table, int n) {
    const long long blk = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const DcfpWpEntry e = table[lo];
    const long long base = (blk - e.first_block) * kWpBlockElems;
    if (e.perm8 >= 2) {       // transformed filters of a fused Winograd conv (conv_winograd2.hip): one (c, m) pair per element
        const long long pairs = (long long)e.CkP * e.Mpad;
#pragma unroll 1
        for (int u = 0; u < kWpBlockElems / 256; ++u) {
            const long long i = base + u * 256 + threadIdx.x;
            if (i >= pairs) return;
            wino_filter2_pair(e.w, e.sAm, e.sAc, e.perm8 == 3, e.M, e.Ck, e.Mpad, i, e.wp);
        }
        return;
    }
    const long long total = (long long)e.T * e.CkP * e.Mpad;
#pragma unroll
    for (int u = 0; u < kWpBlockElems / 256; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i >= total) return;
        const int m = wp_row((int)(i % e.Mpad), e.perm8);
        const long long r = i / e.Mpad;
        const int c = (int)(r % e.CkP);
        const int t = (int)(r / e.CkP);
        float v = 0.f;
        if (m < e.M && c < e.Ck) v = e.w[(long long)m * e.sAm + (long long)c * e.sAc + t];
        e.wp[i] = v;
    }
}

// value of lane (l ^ MASK) within each 32-lane half (ds_swizzle bit-mask mode: and 0x1f, xor MASK)
__device__ __forceinline__ float half_xor(float v, int mask) {
    switch (mask) {
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (1 << 10) | 0x1f));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (2 << 10) | 0x1f));
        case 4: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (4 << 10) | 0x1f));
        case 8: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (8 << 10) | 0x1f));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1f));
    }
}

// ACC: dx += result (gradient fan-in of a residual branch); only the epilogue differs.
template <int TAPS, int TM, int TN, int WM, int WN, bool SD, bool ACC, int KB = 16>
// Tiles with <= 128 accumulator registers per lane are built for TWO waves per SIMD (the second
// launch-bounds argument is waves per execution unit in HIP): the small-Cout layers have 4x the
// staging work per MFMA of the 256x256 tile, and a second resident block hides it (same-box A/B on
// the stem / layer1 / layer2 3x3 shapes: +9...+22 %).  The accumulate-epilogue variants are left
// alone (they would spill).
__global__ void __launch_bounds__(64 * WM * WN, (TM * TN <= 8 && WM * WN == 4 && !ACC) ? 2 : 1)
igemm2_kernel(const Igemm2Params p) {
    constexpr int BK = KB;     // shadows the file-level default: K-step depth of this instance
    constexpr int NPART = BK / 4;   // staging is issued in NPART parts over the first NPART MFMA slots
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    // A: thread -> (4 consecutive m, rows ay + AROWS*j)
    constexpr int AX = BM / 4;                       // threads across a row
    constexpr int AROWS = (NT / AX) < BK ? (NT / AX) : BK;   // rows per pass
    constexpr int APASS = BK / AROWS;
    // B: thread -> (4 consecutive pixels, rows ty + TY*q)
    constexpr int TX = BN / 4, TY = NT / TX, RPT = BK / TY;
    static_assert(APASS >= 1 && APASS * AROWS == BK, "A loader shape");
    static_assert(TY >= 1 && RPT >= 1 && TY * RPT == BK, "B loader shape");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;    // [2][BK][BN]

    const int group = 8 * p.tiles_m;
    const int g = blockIdx.x / group, local = blockIdx.x - g * group;
    const int nt = g * 8 + (local & 7);
    const int mt = local >> 3;
    if (nt >= p.tiles_n_total) return;
    const int img = nt / p.tiles_per_img;
    const int ti_img = nt - img * p.tiles_per_img;
    // Strided data gradient (SD): a pixel tile holds output pixels of ONE phase (h % sd, w % sd) - BN consecutive
    // positions of that phase's coarse grid - so only the taps that can reach the phase are run: no MFMA
    // work on structural zeros (with mixed-phase tiles 3 of 4 products of a stride-2 dgrad were zero).
    const int phase = SD ? (p.zfold ? p.zfold - 1 : ti_img / p.tiles_per_phase) : 0;
    const int ph_h = SD ? phase / p.sd : 0, ph_w = SD ? phase - ph_h * p.sd : 0;
    const int p0 = (SD && !p.zfold ? ti_img - phase * p.tiles_per_phase : ti_img) * BN;
    const int m0 = mt * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- A loader
    const int ax = tid % AX, ay = tid / AX;
    const bool a_on = ay < BK;   // BM == 32: only half of the threads stage A
    const float* a_src = p.wp + m0 + 4 * ax;

    // ---- B loader
    const int tx = tid % TX, ty = tid / TX;
    int bh[4], bw[4];
    bool pv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int pp = p0 + 4 * tx + e;
        if constexpr (SD) {
            const int a = pp / p.Wc, b = pp - a * p.Wc;
            const int oh = a * p.sd + ph_h, ow = b * p.sd + ph_w;
            pv[e] = a < p.Hc && oh < p.Ho && ow < p.Wo;
            bh[e] = oh * p.sn;
            bw[e] = ow * p.sn;
        } else {
            pv[e] = pp < p.P;
            const int oh = pp / p.Wo;
            const int ow = pp - oh * p.Wo;
            bh[e] = oh * p.sn;
            bw[e] = ow * p.sn;
        }
    }
    const int HiWi = p.Hi * p.Wi;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in + (long long)img * p.in_nstride), 0, p.Ck * HiWi * 4, 0x00020000);

    const int ksteps_per_tap = p.CkP / BK;
    // SD: the taps whose source row / column is a multiple of sd for this phase, packed 4 bits each
    unsigned long long tap_list = 0;
    int ntaps = TAPS;
    if constexpr (SD) {
        ntaps = 0;
        for (int t = 0; t < TAPS; ++t) {
            const int kh = (TAPS == 9) ? t / 3 : 0, kw = (TAPS == 9) ? t - kh * 3 : 0;
            const int sh = ph_h + p.off0 + kh * p.offstep, sw = ph_w + p.off0 + kw * p.offstep;
            if (sh % p.sd == 0 && sw % p.sd == 0) {
                tap_list |= (unsigned long long)t << (4 * ntaps);
                ++ntaps;
            }
        }
    }
    const int nk = ntaps * ksteps_per_tap;

    unsigned boff[4];            // per-pixel byte offsets of the current tap (kOob when padded)
    bool bvec = false;           // the 4 pixels are one contiguous in-image run (or all padding)
    f32x4 areg[APASS];
    float breg[RPT][4];

    auto set_tap = [&](int t) {
        const int kh = (TAPS == 9) ? t / 3 : 0;
        const int kw = (TAPS == 9) ? t - kh * 3 : 0;
        const int offh = p.off0 + kh * p.offstep;
        const int offw = p.off0 + kw * p.offstep;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int hh = bh[e] + offh, ww = bw[e] + offw;
            bool ok = pv[e] && hh >= 0 && ww >= 0;
            if (SD) {
                ok = ok && (hh % p.sd == 0) && (ww % p.sd == 0);
                hh /= p.sd;
                ww /= p.sd;
            }
            ok = ok && hh < p.Hi && ww < p.Wi;
            boff[e] = ok ? (unsigned)(hh * p.Wi + ww) * 4u : kOob;
        }
        const bool none = boff[0] == kOob && boff[1] == kOob && boff[2] == kOob && boff[3] == kOob;
        const bool run = boff[0] != kOob && boff[1] == boff[0] + 4u && boff[2] == boff[0] + 8u &&
                         boff[3] == boff[0] + 12u;
        bvec = none || run;
    };
    // quarter PART (0..3) of the staging loads of K-step kt
    auto load_part = [&](int kt, auto part_) {
        constexpr int PART = decltype(part_)::value;
        // K-step order: tap-major (all channel blocks of tap 0, then tap 1, ...).  Building with
        // -DDCFP_TAP_INNER runs the nine taps of a channel block in nine consecutive K-steps instead,
        // so they re-read the same input rows while those are still in the XCD's L2: measured on
        // layer3 conv2, 2.8x less traffic beyond L2 (PMC FETCH_SIZE 602 -> 213 MB, L2 hit rate
        // 49 -> 75 %) but 3-6 % MORE time in a same-device A/B — the re-reads are absorbed by the
        // Infinity Cache at no MFMA cost, so the faster order is the default.
#ifdef DCFP_TAP_INNER
        const int cb = kt / TAPS;
        const int t = kt - cb * TAPS;
        const int c0 = cb * BK;
        if (PART == 0 && (TAPS > 1 || kt == 0)) set_tap(t);
#else
        const int ti_ = kt / ksteps_per_tap;
        const int c0 = (kt - ti_ * ksteps_per_tap) * BK;
        const int t = SD ? (int)((tap_list >> (4 * ti_)) & 15ull) : ti_;
        if (PART == 0 && c0 == 0) set_tap(t);
#endif
        static_for<0, APASS>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr ((j % NPART) == PART) {
                if (a_on)
                    areg[j] = *reinterpret_cast<const f32x4*>(
                        a_src + (long long)(t * p.CkP + c0 + ay + AROWS * j) * p.Mpad);
            }
        });
        static_for<0, RPT>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            if constexpr ((q % NPART) == PART) {
                int c = c0 + ty + TY * q;
                c = c < p.Ck ? c : p.Ck - 1;        // rows past Ck meet zero rows of Wp
                const unsigned coff = (unsigned)(c * HiWi) * 4u;
                if (bvec) {   // one (possibly unaligned) dwordx4; bit-cast the WHOLE vector (see wgrad)
                    const f32x4 v = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, boff[0] + coff, 0, 0));
                    static_for<0, 4>([&](auto e_) {
                        constexpr int e = decltype(e_)::value;
                        breg[q][e] = v[e];
                    });
                } else {
                    static_for<0, 4>([&](auto e_) {
                        constexpr int e = decltype(e_)::value;
                        breg[q][e] = __builtin_bit_cast(
                            float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, boff[e] + coff, 0, 0));
                    });
                }
            }
        });
    };
    auto store_a = [&](int buf) {
        float* a = As + buf * (BK * BM);
        static_for<0, APASS>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if (a_on) *reinterpret_cast<f32x4*>(a + (ay + AROWS * j) * BM + 4 * ax) = areg[j];
        });
    };
    auto store_b = [&](int buf) {
        float* b = Bs + buf * (BK * BN);
        static_for<0, RPT>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            f32x4 v = {breg[q][0], breg[q][1], breg[q][2], breg[q][3]};
            *reinterpret_cast<f32x4*>(b + (ty + TY * q) * BN + 4 * tx) = v;
        });
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {     // (SD: a phase no tap reaches is all zeros - 1x1 stride-2 dgrads are 3/4 of that)
        static_for<0, NPART>([&](auto part_) { load_part(0, part_); });
        store_a(0);
        store_b(0);
    }
    __syncthreads();

    const int a_off = wm * (TM * 32) + TM * l31;
    const int b_off = wn * (TN * 32) + TN * l31;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        const float* a = As + cur * (BK * BM) + a_off + lhi * BM;
        const float* b = Bs + cur * (BK * BN) + b_off + lhi * BN;
        float af[2][TM], bf[2][TN];
        Frag<TM>::ld(a, af[0]);
        Frag<TN>::ld(b, bf[0]);
        static_for<0, BK / 2>([&](auto kk_) {
            constexpr int kk = decltype(kk_)::value;
            constexpr int fc = kk & 1;
            if constexpr (kk + 1 < BK / 2) {   // operands of the next 16 MFMAs
                Frag<TM>::ld(a + (2 * kk + 2) * BM, af[fc ^ 1]);
                Frag<TN>::ld(b + (2 * kk + 2) * BN, bf[fc ^ 1]);
            }
            if (more) {
                if constexpr (kk < NPART) load_part(kt + 1, std::integral_constant<int, kk>{});
                if constexpr (kk == BK / 2 - 2) store_a(cur ^ 1);
                if constexpr (kk == BK / 2 - 1) store_b(cur ^ 1);
            }
            static_for<0, TM>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                static_for<0, TN>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fc][i], bf[fc][j], acc[i][j], 0, 0, 0);
                });
            });
#ifdef DCFP_SGB
            // ask the scheduler for MFMA, few others, MFMA, ... instead of [all staging][16 MFMA]
            static_for<0, TM * TN>([&](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x096, DCFP_SGB, 0);
            });
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
    }

    // ---- epilogue
    float* o_img = p.out + (long long)img * p.out_nstride;
    if constexpr (TN == 4) {
        // Interior tile of a plain training conv (block-uniform test): 16-byte stores through a
        // buffer descriptor whose base is block-uniform, row offsets scalar (soffset) and the
        // lane's column a single VGPR - no per-row predicates or 64-bit address arithmetic.
        if (!SD && m0 + BM <= p.M && p0 + BN <= p.P && p.vec_store && !p.bias && !p.scale && !p.relu) {
            constexpr unsigned kMaxRec = 0x7ffffffcu;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                o_img + (long long)m0 * p.P + p0, 0, kMaxRec, 0x00020000);
            unsigned voff = (unsigned)((wm * (TM * 32) + TM * 4 * lhi) * p.P + wn * (TN * 32) + TN * l31) * 4u;
            asm volatile("" : "+v"(voff));
            const unsigned P4 = (unsigned)p.P * 4u;
            if constexpr (!ACC) {
                // BatchNorm batch statistics of the conv output, for free: each wave half holds 128
                // pixels of 16 x TM rows, so it emits (mean, sum of squared deviations) of those 128
                // values per row - Welford-style partials that dcfp_bn_stats_from_partials_f32
                // merges in a fixed order in fp64 (no pivot needed, no second pass over y).
                if (p.stat_part) {
                    float* sp = p.stat_part +
                                ((long long)(nt * WN + wn) * p.M + m0 + wm * (TM * 32) + TM * 4 * lhi) * 2;
                    static_for<0, TM>([&](auto i_) {
                        constexpr int i = decltype(i_)::value;
                        // the 16 rows of one MFMA row tile advance through the butterfly together:
                        // 16 independent cross-lane exchanges in flight per step instead of a
                        // serial chain of ten per row
                        float sm[16], q[16];
                        static_for<0, 16>([&](auto r_) {
                            constexpr int r = decltype(r_)::value;
                            sm[r] = (acc[i][0][r] + acc[i][1][r]) + (acc[i][2][r] + acc[i][3][r]);
                        });
                        static_for<0, 5>([&](auto st_) {
                            constexpr int mask = 1 << decltype(st_)::value;
                            static_for<0, 16>([&](auto r_) { constexpr int r = decltype(r_)::value; sm[r] += half_xor(sm[r], mask); });
                        });
                        static_for<0, 16>([&](auto r_) {
                            constexpr int r = decltype(r_)::value;
                            sm[r] *= (1.0f / 128.0f);
                            const float d0 = acc[i][0][r] - sm[r], d1 = acc[i][1][r] - sm[r];
                            const float d2 = acc[i][2][r] - sm[r], d3 = acc[i][3][r] - sm[r];
                            q[r] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                        });
                        static_for<0, 5>([&](auto st_) {
                            constexpr int mask = 1 << decltype(st_)::value;
                            static_for<0, 16>([&](auto r_) { constexpr int r = decltype(r_)::value; q[r] += half_xor(q[r], mask); });
                        });
                        if (l31 == 0) {
                            static_for<0, 16>([&](auto r_) {
                                constexpr int r = decltype(r_)::value;
                                constexpr int row = TM * ((r & 3) + 8 * (r >> 2)) + i;
                                sp[row * 2] = sm[r]; sp[row * 2 + 1] = q[r];
                            });
                        }
                    });
                }
            }
            static_for<0, TM>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                f32x4 old[16];
                if constexpr (ACC) {   // 16 independent loads in flight, then 16 add+stores
                    static_for<0, 16>([&](auto r_) {
                        constexpr int r = decltype(r_)::value;
                        constexpr int row = TM * ((r & 3) + 8 * (r >> 2)) + i;
                        old[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            o_rsrc, voff, (unsigned)row * P4, 0));
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
                static_for<0, 16>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    constexpr int row = TM * ((r & 3) + 8 * (r >> 2)) + i;
                    f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                    if constexpr (ACC) v += old[r];
                    // The row offset goes into the VGPR offset, NOT the SGPR soffset field: a 16-byte
                    // buffer store with a register soffset whose data registers are rewritten by
                    // the next VALU instructions stored the NEW upper half in lanes 12-15 / 28-31
                    // of each wave half (tools/micro/acc_debug.py pins it; hipcc 7.2 only pads
                    // the immediate-soffset form of this store-data hazard).
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                        __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                        o_rsrc, voff + (unsigned)row * P4, 0, 0);
                });
                if constexpr (ACC) __builtin_amdgcn_sched_barrier(0);
            });
            return;
        }
    }
    // general path (edges, bias, fused inference epilogue); the opaque asm keeps its 64 row
    // pointers from being hoisted above the K loop
    int pix = p0 + wn * (TN * 32) + TN * l31;
    asm volatile("" : "+v"(pix));
    if constexpr (SD) {
        // the lane's TN coarse positions -> fine offsets (h * Wo + w), -1 where the phase has no such pixel;
        // 4-byte stores 2 floats apart (the other phases' tiles fill the gaps)
        int foff[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pp = pix + j;
            const int a = pp / p.Wc, b = pp - a * p.Wc;
            const int oh = a * p.sd + ph_h, ow = b * p.sd + ph_w;
            foff[j] = (a < p.Hc && oh < p.Ho && ow < p.Wo) ? oh * p.Wo + ow : -1;
        }
        if constexpr (!ACC && TN == 4) {
            // 1x1 stride-2: this phase's tiles are the only ones; a lane's 4 coarse pixels become the 8
            // consecutive fine pixels [v0 0 v1 0 v2 0 v3 0] of row 2a (+ph_h) and 8 zeros of the other row phase:
            // four 16-byte stores instead of 32 scattered 4-byte ones from four different tiles
            if (p.zfold && p.sd == 2 && (p.Wo & 1) == 0 && (p.Wc & 3) == 0 && p.vec_store) {
                const int a = pix / p.Wc, b = pix - a * p.Wc;
                if (a < p.Hc) {
                    const int oh = 2 * a + ph_h, oz = 2 * a + (1 - ph_h);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                            const int m = m0 + wm * (TM * 32) + TM * row + i;
                            if (m >= p.M) continue;
                            float* dst = o_img + (long long)m * p.P + 2 * b;
                            const float v0 = acc[i][0][r], v1 = acc[i][1][r], v2 = acc[i][2][r], v3 = acc[i][3][r];
                            const float4 lo = ph_w ? make_float4(0.f, v0, 0.f, v1) : make_float4(v0, 0.f, v1, 0.f);
                            const float4 hi = ph_w ? make_float4(0.f, v2, 0.f, v3) : make_float4(v2, 0.f, v3, 0.f);
                            if (oh < p.Ho) {
                                *reinterpret_cast<float4*>(dst + (long long)oh * p.Wo) = lo;
                                *reinterpret_cast<float4*>(dst + (long long)oh * p.Wo + 4) = hi;
                            }
                            if (oz < p.Ho) {
                                *reinterpret_cast<float4*>(dst + (long long)oz * p.Wo) = make_float4(0.f, 0.f, 0.f, 0.f);
                                *reinterpret_cast<float4*>(dst + (long long)oz * p.Wo + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                        }
                    }
                }
                return;
            }
        }
        if (p.zfold && !ACC) {   // generic form of the same: every phase's position of the lane's coarse pixels
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const int m = m0 + wm * (TM * 32) + TM * row + i;
                    if (m >= p.M) continue;
                    float* dst = o_img + (long long)m * p.P;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int pp = pix + j;
                        const int a = pp / p.Wc, b = pp - a * p.Wc;
                        if (a >= p.Hc) continue;
                        for (int qh = 0; qh < p.sd; ++qh)
                            for (int qw = 0; qw < p.sd; ++qw) {
                                const int oh = a * p.sd + qh, ow = b * p.sd + qw;
                                if (oh < p.Ho && ow < p.Wo)
                                    dst[oh * p.Wo + ow] = (qh == ph_h && qw == ph_w) ? acc[i][j][r] : 0.f;
                            }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int m = m0 + wm * (TM * 32) + TM * row + i;
                if (m >= p.M) continue;
                float* dst = o_img + (long long)m * p.P;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (foff[j] < 0) continue;
                    float v = acc[i][j][r];
                    if (ACC) v += dst[foff[j]];
                    dst[foff[j]] = v;
                }
            }
        }
        return;
    }
    const bool vec = (TN == 4) && p.vec_store && (pix + 3 < p.P);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float4 old[16];
        if (ACC) {   // 16 independent loads in flight per batch, not a serialized read-modify-write
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int m = m0 + wm * (TM * 32) + TM * row + i;
                const float* src = o_img + (long long)m * p.P + pix;
                old[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < p.M) {
                    if (vec) {
                        old[r] = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (pix + 0 < p.P) old[r].x = src[0];
                        if (TN > 1 && pix + 1 < p.P) old[r].y = src[1];
                        if (TN > 2 && pix + 2 < p.P) old[r].z = src[2];
                        if (TN > 3 && pix + 3 < p.P) old[r].w = src[3];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int m = m0 + wm * (TM * 32) + TM * row + i;
            if (m >= p.M) continue;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
            if (p.bias) {
                const float bsv = p.bias[m];
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] += bsv;
            }
            if (ACC) { v[0] += old[r].x; v[1] += old[r].y; v[2] += old[r].z; v[3] += old[r].w; }
            float* dst = o_img + (long long)m * p.P + pix;
            if (p.scale) {   // folded eval-mode BatchNorm (+residual) (+ReLU)
                const float sc = p.scale[m], sf = p.shift[m];
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = fmaf(v[j], sc, sf);
                if (p.residual) {
                    const float* rsrc = p.residual + (dst - p.out);
                    if (vec) {
                        const float4 rv = *reinterpret_cast<const float4*>(rsrc);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            if (pix + j < p.P) v[j] += rsrc[j];
                    }
                }
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
            }
            if (vec) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (pix + j < p.P) dst[j] = v[j];
                }
            }
        }
        if (ACC) __builtin_amdgcn_sched_barrier(0);
    }
}

// The 256 x 256 tile for 1x1 stride-1 convs with both operands staged by LDS-DMA (`buffer_load_dwordx4 ... lds`): a wave instruction copies one 1 KB k-row (256 output
// channels of Wp, or 256 pixels of one input channel) straight into the [k][256] LDS image the
// fragment reads already use - no staging VGPRs, no ds_write pass, no address arithmetic in the
// loop; the copy of tile kt+1 is issued before the 128 MFMAs of tile kt and the step's barrier
// (which waits vmcnt(0)) retires it.  Ragged shapes (pruned widths) work too: rows past M meet the zero
// rows of Wp, channel rows past Ck and pixels past P are out of the descriptor's range (zeros), and
// edge tiles take a predicated store path.  Same-box A/B against the register-staged kernel: +6...+11 % (layer4 1x1 dgrad 130 -> 142 TF = 90 % of peak).
// DCFP_IGEMM_DMA=0 switches it off.
// TAPS = 9: every K-step copies the activation rows of ONE tap, shifted by that tap's (dh, dw);
// lanes whose pixels fall into the padding use an out-of-range offset (the copy writes zeros).  With
// all column shifts multiples of 4 pixels (dilation 4, 8, 12, ...) a lane copies an aligned quad
// (MIXED = false); otherwise (dilation 1, 2) the shifted taps are copied pixel by pixel with four
// dword instructions per k-row (MIXED = true, chosen per tap).
template <int TAPS, bool MIXED, bool ACC>
__global__ void __launch_bounds__(256) igemm2_dma_kernel(const Igemm2Params p) {
    constexpr int TM = 4, TN = 4, WN = 2, BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;    // [2][BK][BN]
    const int group = 8 * p.tiles_m;
    const int g = blockIdx.x / group, local = blockIdx.x - g * group;
    int nt = g * 8 + (local & 7);
    const int mt = local >> 3;
    if (nt >= p.tiles_n_total) return;
    int img = nt / p.tiles_per_img;
    int ti = nt - img * p.tiles_per_img;
    if (TAPS == 9 && p.tapskip == 2) {
        // centre-out order over all images: the tiles that keep all three kernel rows are dispatched first, the
        // cheaper edge tiles fill the tail of the launch (longest-first scheduling; `nt` stays the tile's identity)
        const int k = nt / p.N, c = p.tiles_per_img >> 1;
        img = nt - k * p.N;
        ti = (k & 1) ? c - 1 - (k >> 1) : c + (k >> 1);
        nt = img * p.tiles_per_img + ti;
    }
    // Pixel tile: 256 consecutive pixels, or (tile2d, shifted taps only) 8 rows x 32 columns, so that only the
    // tiles on the image's left / right edge see quads that straddle the border and all others can copy 16
    // bytes per lane from the shifted source.  loc -> offset of tile pixel `loc` from the tile's first pixel.
    const bool t2d = MIXED && p.tile2d != 0;
    const int tsh = p.tile2d, tcols = 1 << tsh;        // tile2d = log2(columns per tile row)
    int p0 = ti * BN, t_ow0 = 0;
    if (t2d) {
        const int wt = p.Wi >> tsh;
        const int th = ti / wt;
        t_ow0 = (ti - th * wt) << tsh;
        p0 = th * (BN >> tsh) * p.Wi + t_ow0;
    }
    auto pixoff = [&](int loc) { return t2d ? (loc >> tsh) * p.Wi + (loc & (tcols - 1)) : loc; };
    const int m0 = mt * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HiWi = p.Hi * p.in_pitch;             // channel stride of the B source (in_pitch == Wi when dense)
    const int padw = p.in_pitch - p.Wi;              // zero floats behind every source row
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // wave w copies k-rows 4w .. 4w+3 of both operands
    unsigned a_voff[4], b_row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a_voff[q] = (unsigned)((4 * wid + q) * p.Mpad) * 4u + lane * 16u;
        b_row[q] = (unsigned)((4 * wid + q) * HiWi) * 4u;
    }
    // this lane's pixels: the quad 4*lane.. (16-byte copies) and pixels 64 s + lane (dword copies)
    int q_oh, q_ow, s_oh[4], s_ow[4];
    bool q_in, s_in[4] = {true, true, true, true};     // pixels past P (last tile of an image) copy zeros
    {
        const int pp = p0 + pixoff(4 * lane);
        q_in = pp < p.P;                                // P % 4 == 0: a quad is inside or outside as a whole
        q_oh = pp / p.Wo; q_ow = pp - q_oh * p.Wo;
        if constexpr (MIXED) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ps = p0 + pixoff(64 * e + lane);
                s_in[e] = ps < p.P;
                s_oh[e] = ps / p.Wo; s_ow[e] = ps - s_oh[e] * p.Wo;
            }
        }
    }
    const int kpt = p.CkP / BK;
    // Dead tap rows (round 2): with a dilation comparable to the image height (ASPP: 12 / 24 / 36 on 128 rows)
    // every source row of kernel row kh lies in the padding for the tiles near the top / bottom edge - all its
    // K-steps would copy zeros and multiply them.  Those K-steps are skipped (6 / 12 / 19 % of the layer's
    // MFMAs); the remaining ones keep their order, so the sums are the same bits (x + 0 * w == x for finite w).
    int t_beg = 0, t_end = TAPS;
    if (TAPS == 9 && p.tapskip) {
        const int rows = t2d ? (BN >> tsh) : 0;
        const int oh_lo = t2d ? (ti / (p.Wi >> tsh)) * rows : p0 / p.Wo;
        int oh_hi = t2d ? oh_lo + rows - 1 : (p0 + BN - 1 < p.P ? p0 + BN - 1 : p.P - 1) / p.Wo;
        oh_hi = oh_hi < p.Ho ? oh_hi : p.Ho - 1;
        // (the row offsets run upwards for the forward pass and downwards for dgrad: the live rows are contiguous)
        auto live = [&](int kh) { const int o = p.off0 + kh * p.offstep; return oh_hi + o >= 0 && oh_lo + o < p.Hi; };
        int kh_lo = 0, kh_hi = 2;
        while (kh_lo < 2 && !live(kh_lo)) ++kh_lo;
        while (kh_hi > kh_lo && !live(kh_hi)) --kh_hi;
        t_beg = 3 * kh_lo; t_end = 3 * kh_hi + 3;
    }
    const int nk = (t_end - t_beg) * kpt;
    unsigned boff4 = 0, boff1[4] = {0, 0, 0, 0};
    bool tap_quads = true;            // block-uniform: this tap's column shift keeps quads aligned
    int ld_t = t_beg, ld_cb = 0;      // (tap, channel block) of the tile the loader copies next
    auto set_tap = [&](int t) {
        const int kh = (TAPS == 9) ? t / 3 : 0;
        const int kw = (TAPS == 9) ? t - kh * 3 : 0;
        const int offh = p.off0 + kh * p.offstep, offw = p.off0 + kw * p.offstep;
        tap_quads = !MIXED || (offw & 3) == 0 || (t2d && t_ow0 + offw >= 0 && t_ow0 + tcols - 1 + offw < p.Wi);
        {
            const int hh = q_oh + offh, ww = q_ow + offw;
            // pitched source: a quad may start up to padw floats left of the row (the previous row's zero tail,
            // or before the buffer: out of range -> zeros) and end up to padw floats behind it
            // (offsets are relative to padw floats BEFORE the image - the descriptor below starts there - so that
            // a quad hanging over the left end of row 0 of channel 0 has a non-negative offset: a dwordx4 whose
            // first dword's offset wrapped below zero is dropped as a whole, the valid half included)
            const bool ok = q_in && hh >= 0 && hh < p.Hi && ww >= -padw && ww + 3 < p.Wi + padw;
            boff4 = ok ? (unsigned)(hh * p.in_pitch + ww + padw) * 4u : kOob;
        }
        if constexpr (MIXED) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int hh = s_oh[e] + offh, ww = s_ow[e] + offw;
                const bool ok = s_in[e] && hh >= 0 && hh < p.Hi && ww >= 0 && ww < p.Wi;
                boff1[e] = ok ? (unsigned)(hh * p.in_pitch + ww + padw) * 4u : kOob;
            }
        }
    };
    // The copies are issued from inline asm: through the builtin, hipcc (7.2) treats every ds_read as a
    // possible alias of the in-flight LDS-DMA and waits vmcnt(0) before the first fragment read, which
    // serialises copy and compute.  Asm loads are outside its bookkeeping, so the wait is placed by
    // hand: vmcnt(0) just before the step's (raw) barrier.
    typedef unsigned u32x4 __attribute__((vector_size(16)));
    auto make_desc = [](const void* base, unsigned bytes) {   // raw buffer descriptor: stride 0, 32-bit data
        const unsigned long long a = (unsigned long long)base;
        u32x4 d = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
        return d;
    };
    const u32x4 a_desc = make_desc(p.wp, 0x7ffffffcu);
    // pitched source: the padw floats in front of an image are readable zeros by contract (the previous image's
    // last row tail, or the buffer's leading margin)
    const u32x4 b_desc = make_desc(p.in + (long long)img * p.in_nstride - padw, (unsigned)(p.Ck * HiWi + padw) * 4u);
    const unsigned lds_a0 = (unsigned)(size_t)(lds_ptr)As, lds_b0 = (unsigned)(size_t)(lds_ptr)Bs;
    auto issue = [&](int buf) {      // copy tile (ld_t, ld_cb) into LDS buffer `buf`, then step the loader
        const unsigned a_s = (unsigned)((ld_t * p.CkP + ld_cb * BK) * p.Mpad + m0) * 4u;
        const unsigned b_cb = (unsigned)(ld_cb * BK * HiWi) * 4u;   // in the VGPR offset: the descriptor's
        const unsigned b_s = 0;                                      // bound must see it (rows past Ck -> zeros)
        static_for<0, 4>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            const unsigned la = __builtin_amdgcn_readfirstlane(lds_a0 + (unsigned)((buf * BK + 4 * wid + q) * BM) * 4u);
            const unsigned lb = __builtin_amdgcn_readfirstlane(lds_b0 + (unsigned)((buf * BK + 4 * wid + q) * BN) * 4u);
            const unsigned av = a_voff[q], as_ = a_s, bs_ = b_s;
            const u32x4 ad = a_desc, bd = b_desc;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(la), "v"(av), "s"(ad), "s"(as_) : "memory", "m0");
            if (tap_quads) {
                const unsigned bv = boff4 + b_row[q] + b_cb;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(lb), "v"(bv), "s"(bd), "s"(bs_) : "memory", "m0");
            } else if constexpr (MIXED) {
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    const unsigned bv = boff1[e] + b_row[q] + b_cb;
                    const unsigned le = lb + 256u * e, bs2 = bs_;
                    const u32x4 bd2 = bd;      // (asm operands must be locals of the innermost lambda)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                                 :: "s"(le), "v"(bv), "s"(bd2), "s"(bs2) : "memory", "m0");
                });
            }
        });
        if (++ld_cb == kpt) {
            ld_cb = 0;
            if (++ld_t < t_end) set_tap(ld_t);
        }
    };
    auto retire = [&]() {   // every copy landed and every fragment read done, then the barrier
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
    set_tap(t_beg);
    issue(0);
    retire();
    const int a_off = wm * (TM * 32) + TM * l31;
    const int b_off = wn * (TN * 32) + TN * l31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1);
        const float* a = As + cur * (BK * BM) + a_off + lhi * BM;
        const float* b = Bs + cur * (BK * BN) + b_off + lhi * BN;
        float af[2][TM], bf[2][TN];
        Frag<TM>::ld(a, af[0]);
        Frag<TN>::ld(b, bf[0]);
        static_for<0, BK / 2>([&](auto kk_) {
            constexpr int kk = decltype(kk_)::value;
            constexpr int fc = kk & 1;
            if constexpr (kk + 1 < BK / 2) {
                Frag<TM>::ld(a + (2 * kk + 2) * BM, af[fc ^ 1]);
                Frag<TN>::ld(b + (2 * kk + 2) * BN, bf[fc ^ 1]);
            }
            static_for<0, TM>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                static_for<0, TN>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fc][i], bf[fc][j], acc[i][j], 0, 0, 0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        retire();
    }
    float* o_img = p.out + (long long)img * p.out_nstride;
    if (m0 + BM > p.M || (!t2d && p0 + BN > p.P)) {     // edge tile (block-uniform): predicated stores
        int pix = p0 + pixoff(wn * (TN * 32) + TN * l31);
        asm volatile("" : "+v"(pix));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int m = m0 + wm * (TM * 32) + TM * row + i;
                if (m >= p.M || pix >= p.P) continue;      // P % 4 == 0: the lane's 4 pixels are in or out together
                float* dst = o_img + (long long)m * p.P + pix;
                f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                if constexpr (ACC) v += *reinterpret_cast<const f32x4*>(dst);
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        return;
    }
    if constexpr (!ACC) {
        if (p.stat_part) {
            // BatchNorm statistics of this tile's rows (mean, M2 over the wave's 128 pixels, Chan-merged by
            // dcfp_bn_stats_from_partials_f32).  The cross-lane part goes through the now idle LDS: every lane
            // leaves (sum, M2) of its 4 pixels per row, then lane L merges the 32 partials of one row - 128
            // LDS operations per wave instead of the 640 swizzles of a butterfly per row.
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2* st = reinterpret_cast<f32x2*>(smem) + wid * (64 * 33);
            float* sp = p.stat_part + ((long long)(nt * WN + wn) * p.M + m0 + wm * (TM * 32)) * 2;
            static_for<0, 2>([&](auto h_) {
                constexpr int half = decltype(h_)::value;
                static_for<0, 2>([&](auto ii_) {
                    constexpr int ii = decltype(ii_)::value;
                    constexpr int i = 2 * half + ii;
                    static_for<0, 16>([&](auto r_) {
                        constexpr int r = decltype(r_)::value;
                        const float a0 = acc[i][0][r], a1 = acc[i][1][r], a2 = acc[i][2][r], a3 = acc[i][3][r];
                        const float s4 = (a0 + a1) + (a2 + a3);
                        const float mu = 0.25f * s4;
                        const float d0 = a0 - mu, d1 = a1 - mu, d2 = a2 - mu, d3 = a3 - mu;
                        f32x2 v = {s4, (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)};
                        st[(lhi * 32 + ii * 16 + r) * 33 + l31] = v;
                    });
                });
                float sv[32], S = 0.f, M2 = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const f32x2 v = st[lane * 33 + k];
                    sv[k] = v.x; S += v.x; M2 += v.y;
                }
                const float mean = S * (1.0f / 128.0f);
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float d = 0.25f * sv[k] - mean;
                    M2 += 4.0f * d * d;
                }
                const int rr = lane & 15;
                const int row = 16 * (lane >> 5) + 4 * (rr & 3) + 32 * (rr >> 2) + 2 * half + ((lane >> 4) & 1);
                sp[row * 2] = mean;
                sp[row * 2 + 1] = M2;
            });
        }
    }
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        o_img + (long long)m0 * p.P + p0, 0, 0x7ffffffcu, 0x00020000);
    unsigned voff = (unsigned)((wm * (TM * 32) + TM * 4 * lhi) * p.P + pixoff(wn * (TN * 32) + TN * l31)) * 4u;
    asm volatile("" : "+v"(voff));
    const unsigned P4 = (unsigned)p.P * 4u;
    static_for<0, TM>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        f32x4 old[16];
        if constexpr (ACC) {
            static_for<0, 16>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                constexpr int row = TM * ((r & 3) + 8 * (r >> 2)) + i;
                old[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    o_rsrc, voff, (unsigned)row * P4, 0));
            });
            __builtin_amdgcn_sched_barrier(0);
        }
        static_for<0, 16>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            constexpr int row = TM * ((r & 3) + 8 * (r >> 2)) + i;
            f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            if constexpr (ACC) v += old[r];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                o_rsrc, voff + (unsigned)row * P4, 0, 0);   // row offset in the VGPR: see igemm2_kernel
        });
        if constexpr (ACC) __builtin_amdgcn_sched_barrier(0);
    });
}

template <int TAPS, int TM, int TN, int WM, int WN, bool SD = false, bool ACC = false, int KB = 16>
int launch_cfg(Igemm2Params& p, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    constexpr int BK = KB;
    const long long groups = ((long long)p.tiles_n_total + 7) / 8;
    const long long blocks = groups * 8 * p.tiles_m;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * BK * (BM + BN) * sizeof(float);
    auto kern = igemm2_kernel<TAPS, TM, TN, WM, WN, SD, ACC, KB>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, stream, p);
    DCFP_RETURN_LAUNCH();
}

// DCFP_IGEMM_BK32=1: 32-deep K-steps for the 256x256 tile (half the barriers; 128 KB of LDS)
static bool deep_k() {
    static const bool v = getenv("DCFP_IGEMM_BK32") != nullptr;
    return v;
}
static int ck_pad() { return deep_k() ? 32 : BK; }

struct TileCfg { int bm, bn, id; };
// 0: 32x512  1: 64x512  2: 128x256  3: 128x128  4: 256x256  5: 128x256 strided-dgrad
TileCfg pick_cfg(int M, long long px, int sd) {
    if (sd > 1) return {128, 256, 5};
    auto blocks = [&](int bm, int bn) { return ((long long)(M + bm - 1) / bm) * ((px + bn - 1) / bn); };
    if (M <= 32) return {32, 512, 0};
    if (M <= 64) return {64, 512, 1};
    if (M <= 128) return blocks(128, 256) >= 192 ? TileCfg{128, 256, 2} : TileCfg{128, 128, 3};
    if (blocks(256, 256) >= 192) return {256, 256, 4};
    if (blocks(128, 256) >= 192) return {128, 256, 2};
    return {128, 128, 3};
}


template <int TAPS>
int launch_taps(Igemm2Params& p, int cfg, hipStream_t stream) {
    if (p.accumulate) {   // residual-gradient fan-in (block inputs: >= 128 channels unless pruned)
        switch (cfg) {
            case 0: return launch_cfg<TAPS, 1, 4, 1, 4, false, true>(p, stream);
            case 1: return launch_cfg<TAPS, 2, 4, 1, 4, false, true>(p, stream);
            case 2: return launch_cfg<TAPS, 2, 4, 2, 2, false, true>(p, stream);
            case 3: return launch_cfg<TAPS, 2, 2, 2, 2, false, true>(p, stream);
            case 4: return launch_cfg<TAPS, 4, 4, 2, 2, false, true>(p, stream);
            default: return launch_cfg<TAPS, 2, 4, 2, 2, true, true>(p, stream);
        }
    }
    switch (cfg) {
        case 0: return launch_cfg<TAPS, 1, 4, 1, 4>(p, stream);
        case 1: return launch_cfg<TAPS, 2, 4, 1, 4>(p, stream);
        case 2: return launch_cfg<TAPS, 2, 4, 2, 2>(p, stream);
        case 3: return launch_cfg<TAPS, 2, 2, 2, 2>(p, stream);
        case 4:
            if (deep_k()) return launch_cfg<TAPS, 4, 4, 2, 2, false, false, 32>(p, stream);
            return launch_cfg<TAPS, 4, 4, 2, 2>(p, stream);
        default: return launch_cfg<TAPS, 2, 4, 2, 2, true>(p, stream);
    }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

// implemented in conv_igemm2n.hip: 8 x 2 wave tiles with dead row blocks skipped (ragged M)
int dcfp_igemm2n_launch(const Igemm2Params& p, int T, hipStream_t stream);

// implemented in conv_igemm2p.hip: the persistent 1x1 LDS-DMA kernel
int dcfp_igemm2p_launch(const Igemm2Params& p, hipStream_t stream);
bool dcfp_igemm2_persist();

// ---- entry points used by conv_igemm.hip's C-ABI functions
size_t dcfp_igemm2_workspace_bytes(int T, int M, int Ck, long long px, int sd) {
    // rows rounded to 256: covers every tile height and the permuted 256-row layout of conv_igemm2n.hip
    return (size_t)T * round_up(Ck, ck_pad()) * round_up(M, 256) * sizeof(float);
}

int dcfp_igemm2_cfg_id(int M, long long px, int sd) { return pick_cfg(M, px, sd).id; }

// Number of (mean, M2) partial rows per channel the forward kernel can emit for this shape (each
// over 128 pixels), or 0 when some tile is not interior / the tile shape has no fused statistics.
long long dcfp_igemm2_stat_slots(int M, int P, int N, long long out_nstride, const float* out) {
    const TileCfg c = pick_cfg(M, (long long)N * P, 1);
    if (c.id != 0 && c.id != 1 && c.id != 2 && c.id != 4) return 0;          // TN == 4 tiles only
    if (M % c.bm != 0 || P % c.bn != 0 || P % 4 != 0 || out_nstride % 4 != 0 || !dcfp_aligned16(out)) return 0;
    return (long long)N * (P / c.bn) * (c.bn / 128);
}

const char* dcfp_igemm2_cfg_args(int M, long long px, int sd) {
    switch (pick_cfg(M, px, sd).id) {
        case 0: return "1,4,1,4,0";
        case 1: return "2,4,1,4,0";
        case 2: return "2,4,2,2,0";
        case 3: return "2,2,2,2,0";
        case 4: return "4,4,2,2,0";
        default: return "2,4,2,2,1";
    }
}

// Ragged output-channel counts on the 8 x 2 kernel of conv_igemm2n.hip: same shape conditions as the other
// LDS-DMA kernels (shifted quads must not need border handling: column shifts multiples of 4, or a row-pitched
// source), enough pixel tiles to fill the chip, and >= 20 % less MFMA work than the tile the shape gets otherwise.
// DCFP_IGEMM_DMA8=0: off.
bool dcfp_igemm2_use_dma8(int T, int M, int P, long long px, int sn, int sd, int off0, int offstep, int HiWi,
                          int Wo, bool pitched) {
    static const int mode = [] { const char* e = getenv("DCFP_IGEMM_DMA8"); return e ? atoi(e) : 1; }();   // 0 off, 2 = wherever eligible (A/B)
    static const bool dma = [] { const char* e = getenv("DCFP_IGEMM_DMA"); return !e || atoi(e) != 0; }();
    const bool on = mode != 0;
    if (!on || !dma || deep_k() || sn != 1 || sd != 1 || HiWi != P || Wo % 4 != 0) return false;
    if (T == 1 ? off0 != 0 : ((((off0 | offstep) & 3) != 0 && !pitched) || off0 + 2 * offstep != -off0)) return false;
    const long long tiles = ((long long)P + 255) / 256 * (px / P) * ((M + 255) / 256);
    if (tiles < 192) return false;
    if (mode == 2) return true;
    // Measured (tools/conv_bench.py, profiles/r02_dma8_ab.txt): at 8 live row blocks this kernel is 10..17 % slower
    // than the 4 x 4-tile kernels (8-byte stores, 2-way LDS conflicts on the A fragments, no fused BatchNorm
    // statistics), at 7 of 8 it is 10 % faster, and on ragged M <= 64 it beats the register-staged 64 x 512
    // tile by 32..35 % at the same MFMA count.
    const TileCfg c = pick_cfg(M, px, sd);
    if (M % c.bm == 0) return false;           // exact fit: the tile kernels (and their statistics epilogue) stay
    if (c.bm <= 64) return true;
    const long long old_rows = (long long)((M + c.bm - 1) / c.bm) * c.bm, new_rows = (long long)((M + 31) / 32) * 32;
    return 10 * new_rows <= 9 * old_rows;
}

// Dead kernel rows (9-tap LDS-DMA kernels): 0 = no skipping, 1 = skip in tile order, 2 = skip with the tiles dispatched
// centre-out (full tiles first).  Skipping shortens SOME tiles; whether that shortens the launch depends on how the
// tiles pack onto the CUs: ASPP forward has 512 tiles = exactly two per CU, and with 320 of them still full
// (dilation 24) every order ends in two full rounds - skipping then only takes the workgroups out of lockstep
// (they share the weight slices through L2) and costs 1..5 %.  The three options are therefore list-scheduled on
// the host (workgroups go to the first free CU in blockIdx order, one per CU) and skipping is used where it
// predicts >= 3 % (measured: dilation 36 forward -17 %, layer4 dilation 16 -8 %, ASPP dgrad -4 / -10 / -17 %).
// DCFP_IGEMM_TAPSKIP = 0 / 1 / 2 forces a mode (A/B).
bool dcfp_igemm2_dma_shape(int T, int M, int Ck, int P, long long px, int sn, int sd, int off0, int HiWi, int Wo);

// kernel rows (of 3) that tile `ti` of an image executes - the same arithmetic as the kernels' t_beg / t_end
static int tile_live_rows(const Igemm2Params& p, int ti) {
    int oh_lo, oh_hi;
    if (p.tile2d) {
        const int rows = 256 >> p.tile2d;
        oh_lo = (ti / (p.Wi >> p.tile2d)) * rows;
        oh_hi = oh_lo + rows - 1;
    } else {
        const int p0 = ti * 256;
        oh_lo = p0 / p.Wo;
        oh_hi = (p0 + 255 < p.P ? p0 + 255 : p.P - 1) / p.Wo;
    }
    oh_hi = oh_hi < p.Ho ? oh_hi : p.Ho - 1;
    int lo = 0, hi = 2;
    auto live = [&](int kh) { const int o = p.off0 + kh * p.offstep; return oh_hi + o >= 0 && oh_lo + o < p.Hi; };
    while (lo < 2 && !live(lo)) ++lo;
    while (hi > lo && !live(hi)) --hi;
    return hi - lo + 1;
}

static int pick_tapskip(const Igemm2Params& p, int bn) {
    static const int forced = [] { const char* e = getenv("DCFP_IGEMM_TAPSKIP"); return e ? atoi(e) : -1; }();
    if (bn != 256) return 0;                      // only the 256-pixel-tile LDS-DMA kernels skip
    const bool t2d = p.tile2d != 0;
    if (forced >= 0) return (forced == 2 && t2d) ? 1 : forced;
    if (t2d) return 1;                            // (8-row tiles of the dilation 1 / 2 convs: nothing to order)
    struct Key { int Ho, Wo, Hi, N, tm, off0, offstep, mode; };
    thread_local Key cache[16];
    thread_local int used = 0;
    for (int i = 0; i < used; ++i) {
        const Key& k = cache[i];
        if (k.Ho == p.Ho && k.Wo == p.Wo && k.Hi == p.Hi && k.N == p.N && k.tm == p.tiles_m && k.off0 == p.off0 &&
            k.offstep == p.offstep)
            return k.mode;
    }
    const int T = p.tiles_per_img, cus = dcfp_num_cus();
    std::vector<float> work((size_t)T);
    bool any = false;
    for (int ti = 0; ti < T; ++ti) {
        const int live = tile_live_rows(p, ti);
        any |= live < 3;
        work[ti] = (float)live + 0.1f;            // + prologue / epilogue
    }
    int mode = 0;
    if (any) {
        auto makespan = [&](int m) {
            std::priority_queue<float, std::vector<float>, std::greater<float>> free_at;
            for (int i = 0; i < cus; ++i) free_at.push(0.f);
            float end = 0.f;
            const long long groups = ((long long)p.tiles_n_total + 7) / 8;
            for (long long g = 0; g < groups; ++g)
                for (int mt = 0; mt < p.tiles_m; ++mt)
                    for (int l = 0; l < 8; ++l) {
                        const long long nt = g * 8 + l;
                        if (nt >= p.tiles_n_total) continue;
                        int ti = (int)(nt % T);
                        if (m == 2) {
                            const int k = (int)(nt / p.N), c = T >> 1;
                            ti = (k & 1) ? c - 1 - (k >> 1) : c + (k >> 1);
                        }
                        const float t0 = free_at.top();
                        free_at.pop();
                        const float t1 = t0 + (m == 0 ? 3.1f : work[ti]);
                        free_at.push(t1);
                        end = t1 > end ? t1 : end;
                    }
            return end;
        };
        const float m0 = makespan(0), m1 = makespan(1), m2 = makespan(2);
        const float best = m2 <= m1 ? m2 : m1;
        if (best <= 0.97f * m0) mode = m2 <= m1 ? 2 : 1;
    }
    if (used < 16) {
        cache[used++] = Key{p.Ho, p.Wo, p.Hi, p.N, p.tiles_m, p.off0, p.offstep, mode};
    }
    return mode;
}

// Fraction of the nominal K-steps (9 taps x channels, padded taps included - the usual FLOP convention) that the
// kernel chosen for this problem executes: < 1 where dead kernel rows are skipped.  For honest accounting only.
double dcfp_igemm2_exec_fraction(int T, int M, int Ck, int N, int Hi, int Wi, int Ho, int Wo, int sn, int sd,
                                 int off0, int offstep, bool pitched) {
    if (T != 9 || sn != 1 || sd != 1) return 1.0;
    const long long px = (long long)N * Ho * Wo;
    const bool d8 = dcfp_igemm2_use_dma8(T, M, Ho * Wo, px, sn, sd, off0, offstep, Hi * Wi, Wo, pitched);
    if (!d8 && !dcfp_igemm2_dma_shape(T, M, Ck, Ho * Wo, px, sn, sd, off0, Hi * Wi, Wo)) return 1.0;
    Igemm2Params p = {};
    p.N = N; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.P = Ho * Wo; p.off0 = off0; p.offstep = offstep;
    p.tiles_per_img = (p.P + 255) / 256; p.tiles_n_total = p.tiles_per_img * N; p.tiles_m = (M + 255) / 256;
    {
        static const int t2d = [] { const char* e = getenv("DCFP_IGEMM_2D"); return e ? atoi(e) : 5; }();
        const bool fits = (t2d == 4 || t2d == 5) && Hi % (256 >> t2d) == 0 && Wi % (1 << t2d) == 0;
        p.tile2d = (!d8 && fits && !pitched && ((off0 | offstep) & 3) != 0 && Ho == Hi && Wo == Wi) ? t2d : 0;
    }
    if (pick_tapskip(p, 256) == 0) return 1.0;
    long long live = 0;
    for (int ti = 0; ti < p.tiles_per_img; ++ti) live += tile_live_rows(p, ti);
    return (double)live / (3.0 * p.tiles_per_img);
}

// Layout of the Wp copy dcfp_igemm2_run would build for this problem (everything but the pointers)
void dcfp_igemm2_wp_layout(int T, int M, int Ck, int P, long long px, int sn, int sd, int off0, int offstep, int HiWi,
                           int Wo, bool pitched, int sAm, int sAc, DcfpWpEntry* e) {
    const TileCfg c = pick_cfg(M, px, sd);
    const bool d8 = dcfp_igemm2_use_dma8(T, M, P, px, sn, sd, off0, offstep, HiWi, Wo, pitched);
    e->T = T; e->Ck = Ck; e->CkP = round_up(Ck, ck_pad()); e->M = M; e->Mpad = round_up(M, d8 ? 256 : c.bm);
    e->sAm = sAm; e->sAc = sAc; e->perm8 = d8 ? 1 : 0;
}

int dcfp_igemm2_permute_multi(const DcfpWpEntry* table, int n, long long total_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(permute_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, table, n);
    DCFP_RETURN_LAUNCH();
}

// DCFP_IGEMM_PERSIST=0: 1x1 convs on the one-tile-per-workgroup kernel instead of conv_igemm2p.hip
bool dcfp_igemm2_persist() {
    static const bool persist = [] { const char* e = getenv("DCFP_IGEMM_PERSIST"); return !e || atoi(e) != 0; }();
    return persist;
}

// shapes the LDS-DMA kernel takes (also used for dcfp_conv2d_kernel_name)
bool dcfp_igemm2_dma_shape(int T, int M, int Ck, int P, long long px, int sn, int sd, int off0, int HiWi, int Wo) {
    static const bool dma = [] { const char* e = getenv("DCFP_IGEMM_DMA"); return !e || atoi(e) != 0; }();   // =0: off
    static const bool dma9 = [] { const char* e = getenv("DCFP_IGEMM_DMA9"); return !e || atoi(e) != 0; }();
    if (!dma || pick_cfg(M, px, sd).id != 4 || sn != 1 || sd != 1 || HiWi != P || Wo % 4 != 0) return false;
    return T == 1 ? off0 == 0 : dma9;
}

// in: B-source tensor; w: reference-layout weights; (sAm, sAc): A strides in w
int dcfp_igemm2_run(const float* in, long long in_nstride, const float* w, int sAm, int sAc,
                    const float* bias, float* out, long long out_nstride, int N, int M, int Ck, int T,
                    int Hi, int Wi, int Ho, int Wo, int sn, int sd, int off0, int offstep,
                    int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream,
                    const float* scale, const float* shift, const float* residual, int relu,
                    float* stat_part, int wp_valid, int in_pitch, long long wp_nstride, const float* fan_src,
                    const unsigned long long* fan_mask, const Igemm2Red* red) {
    const long long px = (long long)N * Ho * Wo;
    const TileCfg c = pick_cfg(M, px, sd);
    Igemm2Params p;
    p.stat_part = stat_part;
    p.wp_nstride = wp_nstride;
    p.fan_src = fan_src; p.fan_mask = fan_mask;
    p.red_x = red ? red->x : nullptr; p.red_mask = red ? red->mask : nullptr;
    p.red_mean = red ? red->mean : nullptr; p.red_part = red ? red->part : nullptr;
    if (red && !fan_src) return DCFP_E_BADDESC;
    p.in = in; p.bias = bias; p.out = out;
    p.scale = scale; p.shift = shift; p.residual = residual; p.relu = relu;
    p.in_nstride = in_nstride; p.out_nstride = out_nstride;
    p.N = N; p.M = M; p.Ck = Ck; p.CkP = round_up(Ck, ck_pad()); p.Mpad = round_up(M, c.bm);
    const bool plain = !scale && !relu && !stat_part;     // (bias is handled by the ragged-M kernel's epilogue)
    const bool d8 = plain && !wp_nstride && !fan_src && dcfp_igemm2_use_dma8(T, M, Ho * Wo, px, sn, sd, off0, offstep, Hi * Wi, Wo,
                                                                 in_pitch > 0 && in_pitch != Wi);
    if (d8) p.Mpad = round_up(M, 256);
    p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.P = Ho * Wo;
    p.in_pitch = in_pitch > 0 ? in_pitch : Wi;
    const bool pitched = p.in_pitch != Wi;
    if (pitched) {   // only the 9-tap LDS-DMA kernels read pitched sources; every tap's column shift must fit the tail
        const int reach = off0 < 0 ? -off0 : off0;      // |first tap shift|; the last is off0 + 2*offstep = -off0 (pad = dil)
        if (T != 9 || p.in_pitch < Wi + reach || off0 + 2 * offstep != -off0 || (p.in_pitch & 3) ||
            (!d8 && (bias || !dcfp_igemm2_dma_shape(T, M, Ck, Ho * Wo, px, sn, sd, off0, Hi * Wi, Wo))) || scale || relu)
            return DCFP_E_UNSUPPORTED;
        if (!d8 && stat_part && dcfp_igemm2_use_dma8(T, M, Ho * Wo, px, sn, sd, off0, offstep, Hi * Wi, Wo, true))
            return DCFP_E_UNSUPPORTED;                  // (cannot happen: such shapes report no statistics slots)
    }
    {
        static const int t2d = [] { const char* e = getenv("DCFP_IGEMM_2D"); return e ? atoi(e) : 5; }();   // 0: off; 4 / 5: 16 / 32 columns
        const bool fits = (t2d == 4 || t2d == 5) && Hi % (256 >> t2d) == 0 && Wi % (1 << t2d) == 0;
        p.tile2d = (fits && !pitched && T == 9 && ((off0 | offstep) & 3) != 0 && Ho == Hi && Wo == Wi) ? t2d : 0;
    }
    p.tapskip = 0;      // decided below, once the tiling is known
    {
        static const int nt = [] { const char* e = getenv("DCFP_IGEMM_NT"); return e ? atoi(e) : 0; }();
        p.nt_store = nt;
    }
    p.tiles_per_img = (p.P + (d8 ? 256 : c.bn) - 1) / (d8 ? 256 : c.bn);
    p.Hc = p.Wc = p.tiles_per_phase = p.zfold = 0;
    if (sd > 1) {     // strided dgrad: sd*sd phases, each tiled over its own coarse grid
        p.Hc = (Ho + sd - 1) / sd; p.Wc = (Wo + sd - 1) / sd;
        p.tiles_per_phase = (p.Hc * p.Wc + c.bn - 1) / c.bn;
        p.tiles_per_img = sd * sd * p.tiles_per_phase;
        if (T == 1) {   // a 1x1 conv reaches ONE phase: only that phase gets tiles (they also write the zeros)
            const int ph = ((-off0) % sd + sd) % sd;
            p.zfold = ph * sd + ph + 1;
            p.tiles_per_img = p.tiles_per_phase;
        }
    }
    p.tiles_n_total = p.tiles_per_img * N;
    p.tiles_m = p.Mpad / (d8 ? 256 : c.bm);
    p.sn = sn; p.sd = sd; p.off0 = off0; p.offstep = offstep;
    p.accumulate = accumulate;
    p.vec_store = (p.P % 4 == 0) && (out_nstride % 4 == 0) && dcfp_aligned16(out);
    if (T == 9 && sd == 1 && sn == 1) p.tapskip = pick_tapskip(p, d8 ? 256 : c.bn);
    const size_t need = (size_t)T * p.CkP * p.Mpad * sizeof(float);
    if (!workspace || workspace_bytes < need || !dcfp_aligned16(workspace)) return DCFP_E_WORKSPACE;
    float* wp = static_cast<float*>(workspace);
    p.wp = wp;
    if (!wp_valid) {   // (the caller's buffer already holds Wp of these weights for this pass and shape otherwise)
        const long long total = (long long)T * p.CkP * p.Mpad;
        long long b = (total + 255) / 256;
        if (b > 2048) b = 2048;
        hipLaunchKernelGGL(permute_weights_kernel, dim3((unsigned)b), dim3(256), 0, stream, w, wp, T,
                           Ck, p.CkP, M, p.Mpad, sAm, sAc, d8 ? 1 : 0);
    }
    if (d8) {
        if (!p.vec_store || wp_nstride || fan_src) return DCFP_E_UNSUPPORTED;
        return dcfp_igemm2n_launch(p, T, stream);
    }
    if (dcfp_igemm2_dma_shape(T, M, Ck, p.P, px, sn, sd, off0, Hi * Wi, Wo) && p.vec_store && !bias && !scale &&
        !relu && !(stat_part && accumulate)) {
        const long long groups = ((long long)p.tiles_n_total + 7) / 8;
        const long long blocks = groups * 8 * p.tiles_m;
        // 64 KB of operand buffers; the statistics epilogue re-uses them as 4 x [64 rows][33] (sum, M2) pairs
        const size_t lds = stat_part ? (size_t)4 * 64 * 33 * 8 : (size_t)2 * BK * 512 * sizeof(float);
        auto launch = [&](auto kern) -> int {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, p);
            DCFP_RETURN_LAUNCH();
        };
        if (T == 1) {
            if (fan_src && (M % 256 != 0 || p.P % 256 != 0 || !dcfp_igemm2_persist())) return DCFP_E_UNSUPPORTED;
            if (dcfp_igemm2_persist()) return dcfp_igemm2p_launch(p, stream);
            if (wp_nstride) return DCFP_E_UNSUPPORTED;
            return accumulate ? launch(igemm2_dma_kernel<1, false, true>) : launch(igemm2_dma_kernel<1, false, false>);
        }
        if (((off0 | offstep) & 3) == 0 || pitched)    // pitched rows: shifted quads need no border handling
            return accumulate ? launch(igemm2_dma_kernel<9, false, true>) : launch(igemm2_dma_kernel<9, false, false>);
        return accumulate ? launch(igemm2_dma_kernel<9, true, true>) : launch(igemm2_dma_kernel<9, true, false>);
    }
    if (wp_nstride || fan_src) return DCFP_E_UNSUPPORTED;     // per-image weights / masked fan-in: the persistent 1x1 kernel only
    return T == 1 ? launch_taps<1>(p, c.id, stream) : launch_taps<9>(p, c.id, stream);
}

int dcfp_igemm2_ck_pad() { return ck_pad(); }

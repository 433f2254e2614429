// conv_igemm2p.hip — PERSISTENT variant of the 256 x 256-tile LDS-DMA kernel for 1x1 stride-1 convs
// (forward of Bottleneck conv1/conv3 and their data gradients: networks/backbone/resnet.py:25-30).
//
// The one-tile-per-workgroup kernel (igemm2_dma_kernel<1,...> in conv_igemm2.hip) pays, per 256 x 256
// tile, a cold first copy (global -> LDS latency with nothing to overlap) and an epilogue during which the
// matrix pipe idles: 8.3 us per tile by a fit over the K = 256 and K = 1024 layers, i.e. 15 % of a
// K = 256 tile (55 us).  Here a workgroup stays on its CU and walks tiles t = blockIdx.x, + gridDim.x, ...:
//   * the first K-step of tile t+1 is copied into the free LDS buffer during the LAST K-step of tile t,
//     so it has landed before the epilogue of t starts (its latency is covered by 128 MFMAs);
//   * the epilogue's 64 stores per lane are issued and NOT waited for: they drain while tile t+1's first
//     K-step computes (the store data registers are read at issue; the accumulators are re-zeroed
//     afterwards), and that step's end-of-step vmcnt(0) retires them;
//   * the BatchNorm-statistics merge has its own 66 KB of LDS (a workgroup owns the CU's 160 KB anyway:
//     one wave per SIMD), so it no longer has to wait for the operand buffers to be idle.
// Tile order, operand images, fragment reads, MFMA order and the statistics arithmetic are those of the
// non-persistent kernel: results are bit-identical (tests/test_conv_large_gpu.py compares the two through
// DCFP_IGEMM_PERSIST).
//
// Measured on MI355X (tools/ab_p.sh, same box, 20 launches each; profiles/r02_persist_ab.txt):
//   256->1024 @4x128x256 (K = 256) forward 116.1 -> 121.5 TF, its M = 1024 dgrad 120.8 -> 128.7 TF;
//   K = 1024 shapes 136 -> 137.5 TF; 512->2048 134.7 -> 137.3 TF.
// What the per-tile remainder is (debug build, -DDCFP_P_DEBUG): without the epilogue stores a K = 256 tile
// takes 64.6 us, with them 71.7 us on 256 CUs but 65.1 / 64.2 us when only 128 / 64 workgroups run - the
// stores are free until the WHOLE chip bursts 64 MB at once (equal tiles keep the CUs in lockstep).
// Staggering the workgroups' start by 1..6 us per phase group (8 groups) gave back exactly what the idle
// time cost (0.563 -> 0.570..0.579 ms); issuing the 8 copies of a K-step one per 16-MFMA sub-step instead
// of back to back: no gain (the runtime "is there a next step" test becomes a branch per piece).
// Register note: every loop-carried lane value must stay in registers - a spill reload is a VMEM operation
// the compiler tracks, and its conservative vmcnt(N) waits in front of the next copies then also wait for
// the (untracked, inline-asm) LDS-DMA copies in flight: the first version of this kernel lost 12 % to that.
#include "igemm2_common.h"
#include <stdlib.h>

namespace {

typedef unsigned u32x4 __attribute__((vector_size(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ u32x4 make_desc(const void* base, unsigned bytes) {   // raw buffer: stride 0, 32-bit data
    const unsigned long long a = (unsigned long long)base;
    u32x4 d = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return d;
}

constexpr int kStatFloats = 4 * 64 * 33 * 2;   // 4 waves x [64 rows][33] (sum, M2) pairs
constexpr int kStatFloats2 = 4 * 32 * 32 * 2;  // 128-row tiles: 4 waves x [32 rows][32] pairs, columns rotated by the row (no pad:
                                               // 48 KB of operand stages + these 32 KB are exactly half of the CU's 160 KB)

// WPI: every image has its own weight copy (p.wp_nstride floats apart) - the batched GEMM of conv_winograd.hip,
// a kernel instance of its own so that profiles tell it from the 1x1 convs
// TM_ = 2: 128 x 256 tiles, 128 accumulator registers per lane - TWO workgroups per CU, so that one's epilogue
// stores and barrier waits overlap the other's MFMAs (the K = 256 layers, whose 256 x 256 tiles are store-bound).
// FAN: out = result + (mask bit ? fan_src : 0) - the block-input gradient of a Bottleneck without materialising the
// residual branch's gradient (dy * ReLU mask): the BatchNorm backward then skips that 4 B/element write
// RED (with FAN): the epilogue also emits the BatchNorm-backward sums of the residual block whose output gradient this
// fan-in produces (Igemm2Params::red_*): per output row and 128-pixel slot, reduced over the 32 lanes of a row by DPP
template <bool ACC, bool WPI = false, int TM_ = 4, bool FAN = false, bool RED = false>
__global__ void __launch_bounds__(256, TM_ == 2 ? 2 : 1) igemm2_dma1p_kernel(const Igemm2Params p, int total_tiles, int dbg) {
    constexpr int TM = TM_, TN = 4, WN = 2, BM = 64 * TM_, BN = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;    // [2][BK][BN]
    float* Ss = smem + 2 * BK * (BM + BN);   // statistics scratch (only touched when p.stat_part)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HiWi = p.Hi * p.Wi;
    const int group = 8 * p.tiles_m;
    const int nk = p.CkP / BK;

    // wave w copies k-rows 4w .. 4w+3 of both operands: row offsets are wave-uniform (kept out of the VGPR
    // file: nothing loop-carried may spill), the lane part is lane * 16 for A and the pixel quad for B
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const unsigned lane16 = lane * 16u;
    const unsigned a2_voff = (unsigned)((lane >> 5) * p.Mpad) * 4u + (unsigned)(lane & 31) * 16u;   // TM == 2: two k-rows per copy
    const u32x4 a_desc = make_desc(p.wp, 0x7ffffffcu);
    const unsigned lds_a0 = (unsigned)(size_t)(lds_ptr)As, lds_b0 = (unsigned)(size_t)(lds_ptr)Bs;

    // decode of a linear tile id: the same (pixel tile, M tile) order as the one-tile kernel, so that the
    // workgroups of one XCD (equal blockIdx % 8 under round-robin placement) share pixel tiles in their L2
    auto decode = [&](int t, int& nt, int& mt) {
        const int g = t / group, local = t - g * group;
        nt = g * 8 + (local & 7);
        mt = local >> 3;
    };
    auto next_valid = [&](int t) {          // tiles past tiles_n_total in the last group are padding
        while (t < total_tiles) {
            int nt, mt;
            decode(t, nt, mt);
            if (nt < p.tiles_n_total) break;
            t += gridDim.x;
        }
        return t;
    };

    // ---- loader state: the tile whose K-steps are being copied, and the next K-step of it.
    // Everything a copy needs beyond its lane offset is a RUNNING wave-uniform byte offset, stepped by one scalar add per
    // K-step (round 4: with one wave per SIMD every scalar / vector ALU instruction of the loop costs ~4.5 cycles of
    // matrix-pipe time - tools/micro/mfma_shadow.hip - and recomputing (k-row) x (row pitch) per piece was ~70 of them per
    // K-step): a_run / b_run = offset of k-row 4w of the K-step in the weight copy / the activation image.
    int ld_tile = next_valid(blockIdx.x), ld_cb = 0;
    unsigned ld_boff4 = 0;
    unsigned a_run = 0, b_run = 0;
    u32x4 ld_bdesc = a_desc;
    const unsigned a_row = (unsigned)p.Mpad * 4u, b_row = (unsigned)HiWi * 4u;           // one k-row
    const unsigned a_kstep = (unsigned)(BK * p.Mpad) * 4u, b_kstep = (unsigned)(BK * HiWi) * 4u;
    const unsigned lds_a_w = __builtin_amdgcn_readfirstlane(lds_a0 + (unsigned)(4 * wid_s * BM) * 4u);
    const unsigned lds_b_w = __builtin_amdgcn_readfirstlane(lds_b0 + (unsigned)(4 * wid_s * BN) * 4u);
    auto ld_set_tile = [&]() {
        int nt, mt;
        decode(ld_tile, nt, mt);
        const int img = nt / p.tiles_per_img;
        const int p0 = (nt - img * p.tiles_per_img) * BN;
        a_run = __builtin_amdgcn_readfirstlane((unsigned)(mt * BM + (WPI ? img * (int)p.wp_nstride : 0)) * 4u +
                                               (unsigned)(4 * wid_s) * a_row);
        b_run = __builtin_amdgcn_readfirstlane((unsigned)(4 * wid_s) * b_row);
        ld_bdesc = make_desc(p.in + (long long)img * p.in_nstride, (unsigned)(p.Ck * HiWi) * 4u);
        const int pp = p0 + 4 * lane;                 // P % 4 == 0: a quad is inside or outside as a whole
        ld_boff4 = pp < p.P ? (unsigned)pp * 4u : kOob;
    };
    // one of the 8 copies of a K-step: piece 2q = k-row 4w+q of A, piece 2q+1 = the same row of B
    auto issue_piece = [&](unsigned abuf, unsigned bbuf, auto piece_) {      // abuf / bbuf: LDS byte address of the wave's rows in the stage
        constexpr int piece = decltype(piece_)::value;
        constexpr int q = piece >> 1;
        if constexpr ((piece & 1) == 0) {
            if constexpr (TM == 4) {
                const unsigned a_s = a_run + (unsigned)q * a_row;
                const unsigned av = lane16;
                const u32x4 ad = a_desc;
                // (M0 = LDS address of the row, written by the add itself: stage base + a literal)
                asm volatile("s_add_i32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(abuf), "v"(av), "s"(ad), "s"(a_s), "i"(q * BM * 4) : "memory", "m0", "scc");
            } else if constexpr (q < 2) {
                // a 128-float k-row is 512 bytes: one instruction copies rows 4w + 2q (lanes 0..31) and 4w + 2q + 1
                const unsigned a_s = a_run + (unsigned)(2 * q) * a_row;
                const unsigned av = a2_voff;
                const u32x4 ad = a_desc;
                asm volatile("s_add_i32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(abuf), "v"(av), "s"(ad), "s"(a_s), "i"(2 * q * BM * 4) : "memory", "m0", "scc");
            }
        } else {
            // channel-row offset in the VGPR offset: the descriptor's bound must see it (rows past Ck -> zeros)
            const unsigned b_cb = b_run + (unsigned)q * b_row;
            const unsigned b_s = 0;
            const unsigned bv = ld_boff4 + b_cb;
            const u32x4 bd = ld_bdesc;
            asm volatile("s_add_i32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(bbuf), "v"(bv), "s"(bd), "s"(b_s), "i"(q * BN * 4) : "memory", "m0", "scc");
        }
    };
    auto ld_advance = [&]() {        // after the 8 pieces of a K-step: step the loader (and cross the tile boundary)
        a_run += a_kstep;
        b_run += b_kstep;
        if (++ld_cb == nk) {
            ld_cb = 0;
            ld_tile = next_valid(ld_tile + gridDim.x);
            if (ld_tile < total_tiles) ld_set_tile();
        }
    };
    auto issue = [&](int buf) {      // copy K-step ld_cb of tile ld_tile into LDS buffer `buf`, then step the loader
        // LDS byte address of this wave's first row in the stage (the pieces add their row as a literal)
        const unsigned abuf = lds_a_w + (buf ? (unsigned)(BK * BM) * 4u : 0u), bbuf = lds_b_w + (buf ? (unsigned)(BK * BN) * 4u : 0u);
        static_for<0, 8>([&](auto piece_) { issue_piece(abuf, bbuf, piece_); });
        ld_advance();
    };
    auto retire = [&]() {   // every copy (and every older store) done, every fragment read done, then the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    int tile = ld_tile;
    if (tile >= total_tiles) return;          // block-uniform
#ifdef DCFP_P_DEBUG
    if (dbg & 0xff00) {                        // stagger: phase group (blockIdx / 8) % 8 waits g * dbg[15:8] * 64 x 64 clocks
        const int g = (blockIdx.x >> 3) & 7;
        for (int i = 0; i < g * ((dbg >> 8) & 0xff); ++i) __builtin_amdgcn_s_sleep(31);   // ~1 us each
    }
#endif
    ld_set_tile();
    int cur = 0;
    issue(cur);
    retire();
    const int a_off = wm * (TM * 32) + TM * l31;
    const int b_off = wn * (TN * 32) + TN * l31;

    while (tile < total_tiles) {
        int nt, mt;
        decode(tile, nt, mt);
        const int img = nt / p.tiles_per_img;
        const int p0 = (nt - img * p.tiles_per_img) * BN;
        const int m0 = mt * BM;
        const int next_tile = next_valid(tile + gridDim.x);

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
            // the loader runs one K-step ahead - across the tile boundary too
            const bool more = kt + 1 < nk || next_tile < total_tiles;
            if (more) issue(cur ^ 1);
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
            cur ^= 1;
        }

        // ---- epilogue of this tile (the next tile's first K-step is already in LDS buffer `cur`)
        float* o_img = p.out + (long long)img * p.out_nstride;
#ifdef DCFP_P_DEBUG
        if (dbg & 1) {                          // no stores at all (invalid results: timing decomposition only)
            if (acc[0][0][0] == 123.456f) o_img[0] = 1.f;
            tile = next_tile;
            continue;
        }
#endif
        if (m0 + BM > p.M || p0 + BN > p.P) {     // edge tile (block-uniform): predicated stores
            int pix = p0 + wn * (TN * 32) + TN * l31;
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
        } else {
            if constexpr (!ACC && TM == 4) {
                if (p.stat_part) {
                    // BatchNorm statistics of this tile's rows: see igemm2_dma_kernel (same arithmetic, same order)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2* st = reinterpret_cast<f32x2*>(Ss) + wid * (64 * 33);
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
                        // (two sweeps over the row's 32 partials instead of holding them in 32 registers: the
                        // persistent loop keeps more state live across the epilogue and must not spill - a spill
                        // reload is a VMEM op the compiler then waits for in front of the next copies)
                        float S = 0.f, M2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 32; ++k) {
                            const f32x2 v = st[lane * 33 + k];
                            S += v.x; M2 += v.y;
                        }
                        const float mean = S * (1.0f / 128.0f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < 32; ++k) {
                            const float d = 0.25f * st[lane * 33 + k].x - mean;
                            M2 += 4.0f * d * d;
                        }
                        const int rr = lane & 15;
                        const int row = 16 * (lane >> 5) + 4 * (rr & 3) + 32 * (rr >> 2) + 2 * half + ((lane >> 4) & 1);
                        sp[row * 2] = mean;
                        sp[row * 2 + 1] = M2;
                    });
                }
            }
            if constexpr (!ACC && TM == 2 && !FAN) {
                if (p.stat_part) {
                    // the same statistics from a 128-row tile (round 4): a wave owns 64 rows - row 2 * q + i for its 32 row
                    // slots q = (r & 3) + 8 * (r >> 2) + 4 * lhi - and merges them in two passes (i = 0, 1) of 32 rows through
                    // a [32][32] image whose columns are rotated by the row (conflict-free writes along a row and reads
                    // down a column without the pad the 64-row image has: LDS is full); per row the arithmetic and its order
                    // are those of the 256-row tile, lanes 0..31 merge one row each
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2* st = reinterpret_cast<f32x2*>(Ss) + wid * (32 * 32);
                    float* sp = p.stat_part + ((long long)(nt * WN + wn) * p.M + m0 + wm * (TM * 32)) * 2;
                    static_for<0, 2>([&](auto i_) {
                        constexpr int i = decltype(i_)::value;
                        static_for<0, 16>([&](auto r_) {
                            constexpr int r = decltype(r_)::value;
                            const float a0 = acc[i][0][r], a1 = acc[i][1][r], a2 = acc[i][2][r], a3 = acc[i][3][r];
                            const float s4 = (a0 + a1) + (a2 + a3);
                            const float mu = 0.25f * s4;
                            const float d0 = a0 - mu, d1 = a1 - mu, d2 = a2 - mu, d3 = a3 - mu;
                            f32x2 v = {s4, (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)};
                            const int row = lhi * 16 + r;
                            st[row * 32 + ((l31 + row) & 31)] = v;
                        });
                        const int rw = lane & 31;
                        float S = 0.f, M2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 32; ++k) {
                            const f32x2 v = st[rw * 32 + ((k + rw) & 31)];
                            S += v.x; M2 += v.y;
                        }
                        const float mean = S * (1.0f / 128.0f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < 32; ++k) {
                            const float d = 0.25f * st[rw * 32 + ((k + rw) & 31)].x - mean;
                            M2 += 4.0f * d * d;
                        }
                        if (lane < 32) {
                            const int rr = lane & 15;
                            const int row = 2 * ((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 4)) + i;
                            sp[row * 2] = mean;
                            sp[row * 2 + 1] = M2;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            }
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                o_img + (long long)m0 * p.P + p0, 0, 0x7ffffffcu, 0x00020000);
            unsigned voff = (unsigned)((wm * (TM * 32) + TM * 4 * lhi) * p.P + wn * (TN * 32) + TN * l31) * 4u;
            asm volatile("" : "+v"(voff));
            const unsigned P4 = (unsigned)p.P * 4u;
            if constexpr (FAN) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.fan_src) + (long long)img * p.out_nstride + (long long)m0 * p.P + p0, 0, 0x7ffffffcu,
                    0x00020000);
                const int words = p.P >> 6;
                // mask words of this lane's first row; element e of a row sits in word 4 * (e >> 8) + (e & 3), bit (e & 255) >> 2
                const unsigned long long* mrow = p.fan_mask +
                    ((long long)img * p.M + m0 + wm * (TM * 32) + TM * 4 * lhi) * words + 4 * (p0 >> 8);
                const int bit = wn * 32 + l31;
                // RED operands: same tile geometry as the output
                const long long row0 = (long long)m0 + wm * (TM * 32) + TM * 4 * lhi;      // this lane's first output row
                __amdgpu_buffer_rsrc_t x_rsrc = o_rsrc;
                const unsigned long long* xrow = nullptr;
                const float* mu_row = nullptr;
                float* rp = nullptr;
                if constexpr (RED) {
                    x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(p.red_x) + (long long)img * p.out_nstride + (long long)m0 * p.P + p0, 0, 0x7ffffffcu,
                        0x00020000);
                    xrow = p.red_mask + ((long long)img * p.M + row0) * words + 4 * (p0 >> 8);
                    mu_row = p.red_mean + row0;
                    rp = p.red_part + ((long long)(nt * WN + wn) * p.M + row0) * 2;
                }
                // sum over the 32 lanes that share an output row (lanes 0..31 / 32..63): three DPP steps inside the rows of
                // 16, row_bcast15 into the odd rows, readlane 31 / 63
                auto red32 = [&](float v) -> float {
                    auto dpp = [](float x, auto ctrl_) {
                        constexpr int ctrl = decltype(ctrl_)::value;
                        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true));
                    };
                    v += dpp(v, std::integral_constant<int, 0xB1>{});        // quad_perm [1,0,3,2]
                    v += dpp(v, std::integral_constant<int, 0x4E>{});        // quad_perm [2,3,0,1]
                    v += dpp(v, std::integral_constant<int, 0x141>{});       // row_half_mirror
                    v += dpp(v, std::integral_constant<int, 0x140>{});       // row_mirror
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
                    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
                    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
                    return lhi ? hi : lo;
                };
                static_for<0, TM>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    static_for<0, 4>([&](auto g_) {          // four rows at a time (registers)
                        constexpr int g4 = decltype(g_)::value;
                        f32x4 src[4];
                        u64x2 ma[4], mb[4];
                        f32x4 xs[4];
                        u64x2 xa[4], xb[4];
                        float mu[4];
                        static_for<0, 4>([&](auto q_) {
                            constexpr int r = 4 * g4 + decltype(q_)::value;
                            constexpr int row = TM * ((r & 3) + 8 * (r >> 2)) + i;
                            src[r & 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                f_rsrc, voff, (unsigned)row * P4, 0));
                            const u64x2* mq = reinterpret_cast<const u64x2*>(mrow + (long long)row * words);
                            ma[r & 3] = mq[0]; mb[r & 3] = mq[1];
                            if constexpr (RED) {
                                xs[r & 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    x_rsrc, voff, (unsigned)row * P4, 0));
                                const u64x2* xq = reinterpret_cast<const u64x2*>(xrow + (long long)row * words);
                                xa[r & 3] = xq[0]; xb[r & 3] = xq[1];
                                mu[r & 3] = mu_row[row];
                            }
                        });
                        static_for<0, 4>([&](auto q_) {
                            constexpr int r = 4 * g4 + decltype(q_)::value;
                            constexpr int row = TM * ((r & 3) + 8 * (r >> 2)) + i;
                            f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                            const f32x4 s4 = src[r & 3];
                            v[0] += ((ma[r & 3][0] >> bit) & 1ull) ? s4[0] : 0.f;
                            v[1] += ((ma[r & 3][1] >> bit) & 1ull) ? s4[1] : 0.f;
                            v[2] += ((mb[r & 3][0] >> bit) & 1ull) ? s4[2] : 0.f;
                            v[3] += ((mb[r & 3][1] >> bit) & 1ull) ? s4[3] : 0.f;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                                __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                o_rsrc, voff + (unsigned)row * P4, 0, 0);
                            if constexpr (RED) {
                                const float g0 = ((xa[r & 3][0] >> bit) & 1ull) ? v[0] : 0.f;
                                const float g1 = ((xa[r & 3][1] >> bit) & 1ull) ? v[1] : 0.f;
                                const float g2 = ((xb[r & 3][0] >> bit) & 1ull) ? v[2] : 0.f;
                                const float g3 = ((xb[r & 3][1] >> bit) & 1ull) ? v[3] : 0.f;
                                const f32x4 x4 = xs[r & 3];
                                const float m_ = mu[r & 3];
                                const float s1 = red32((g0 + g1) + (g2 + g3));
                                const float s2 = red32((g0 * (x4[0] - m_) + g1 * (x4[1] - m_)) +
                                                       (g2 * (x4[2] - m_) + g3 * (x4[3] - m_)));
                                if (l31 == 0) {
                                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                                    const f32x2 pr = {s1, s2};
                                    *reinterpret_cast<f32x2*>(rp + row * 2) = pr;
                                }
                            }
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            } else
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
                    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u4;
                    if (ACC || p.nt_store == 0)       // (block-uniform; the cache-policy bits are an immediate)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v),
                            o_rsrc, voff + (unsigned)row * P4, 0, 0);   // row offset in the VGPR: see igemm2_kernel
                    else if (p.nt_store == 1)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), o_rsrc, voff + (unsigned)row * P4, 0, 2);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), o_rsrc, voff + (unsigned)row * P4, 0, 18);
                });
                if constexpr (ACC) __builtin_amdgcn_sched_barrier(0);
            });
        }
        tile = next_tile;
    }
}

}  // namespace

// 128-row tiles, two workgroups per CU: DCFP_IGEMM_P128 = 0 off, 1 the K <= 256 problems (the default of rounds 2-3, when the
// long-K layers measured 5...9 % slower on them), 2 all (default since the end of round 4: with the leaner loader of round 4
// every 1x1 forward / dgrad entry of the step is faster or equal on them - the M = 256, K = 1024 dgrads 11.74 -> 11.29 ms, the
// K = 512 fan-in dgrads of layer4 4.63 -> 4.19 ms, 1.6 ms per step in all: profiles/r04_p128_all_ab.txt)
static int p128_mode() {
    static const int p128 = [] { const char* e = getenv("DCFP_IGEMM_P128"); return e ? atoi(e) : 2; }();
    return p128;
}
// the fan-in launch of (Mpad, CkP) takes the 128-row tiles - the only ones with the BatchNorm-sums side output
bool dcfp_igemm2p_fan_red_ok(int Mpad, int CkP) {
    return Mpad % 256 == 0 && (p128_mode() == 2 || (p128_mode() == 1 && CkP <= 256));
}

// Launch over all tiles of the problem described by `p` (filled by dcfp_igemm2_run).  One workgroup per CU.
int dcfp_igemm2p_launch(const Igemm2Params& p, hipStream_t stream) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    const int p128 = p128_mode();
    // launches WITH the statistics epilogue take the 128-row tiles at EVERY K (DCFP_IGEMM_P128_STATS: 0 never, 1 K <= 256 only,
    // 2 always - the default): with one workgroup per CU that epilogue (2...3 us of vector ALU per 256 x 256 tile) is exposed, with
    // two it runs under the other workgroup's MFMAs - 512 -> 2048 forward 2.150 -> 2.027 ms, 1024 -> 256 0.526 -> 0.504 ms
    // (profiles/r04_p128_stats_ab.txt), although the same GEMMs WITHOUT the epilogue are 5...9 % slower on 128-row tiles at K >= 512
    static const int half_stats = [] { const char* e = getenv("DCFP_IGEMM_P128_STATS"); return e ? atoi(e) : 2; }();
    const bool k_ok = p128 == 2 || (p128 == 1 && p.CkP <= 256);
    const bool half = p.Mpad % 256 == 0 &&
                      (p.stat_part ? (!p.accumulate && !p.fan_src && !p.wp_nstride && (half_stats == 2 || (half_stats == 1 && k_ok)))
                                   : k_ok);
    Igemm2Params q = p;
    if (half) q.tiles_m = p.tiles_m * 2;
    const long long groups = ((long long)p.tiles_n_total + 7) / 8;
    const long long total = groups * 8 * q.tiles_m;
    if (total > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    const long long slots = half ? 2LL * cus : cus;
    long long blocks = total < slots ? total : slots;
    int dbg = 0;
#ifdef DCFP_P_DEBUG
    if (const char* e = getenv("DCFP_DBG_P_BLOCKS")) blocks = atoll(e) < blocks ? atoll(e) : blocks;
    if (const char* e = getenv("DCFP_DBG_P")) dbg = atoi(e);
#endif
    blocks = blocks / 8 * 8;                       // grid % 8 == 0 keeps a workgroup's tiles on one pixel-tile residue
    if (blocks < 8) blocks = total < 8 ? total : 8;
    const size_t lds = (size_t)(2 * BK * (half ? 384 : 512) + (p.stat_part ? (half ? kStatFloats2 : kStatFloats) : 0)) * sizeof(float);
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, q, (int)total, dbg);
        DCFP_RETURN_LAUNCH();
    };
    if (p.wp_nstride) {
        if (p.accumulate) return DCFP_E_UNSUPPORTED;
        return half ? launch(igemm2_dma1p_kernel<false, true, 2>) : launch(igemm2_dma1p_kernel<false, true>);
    }
    if (p.fan_src) {      // interior tiles only (checked by the caller: M % 256 == 0, P % 256 == 0)
        if (p.accumulate || p.stat_part || !p.fan_mask) return DCFP_E_UNSUPPORTED;
        if (p.red_part) {
            if (!p.red_x || !p.red_mask || !p.red_mean) return DCFP_E_BADDESC;
            if (!half) return DCFP_E_UNSUPPORTED;      // (the 256-row instance would spill: dcfp_igemm2p_fan_red_ok)
            return launch(igemm2_dma1p_kernel<false, false, 2, true, true>);
        }
        return half ? launch(igemm2_dma1p_kernel<false, false, 2, true>) : launch(igemm2_dma1p_kernel<false, false, 4, true>);
    }
    if (half) return p.accumulate ? launch(igemm2_dma1p_kernel<true, false, 2>) : launch(igemm2_dma1p_kernel<false, false, 2>);
    return p.accumulate ? launch(igemm2_dma1p_kernel<true>) : launch(igemm2_dma1p_kernel<false>);
}

// conv_winograd2.hip — FUSED Winograd F(2x2, 3x3): input transform, the sixteen component GEMMs and the output
// transform in ONE kernel (forward and dgrad of the wide 3x3 stride-1 convs: networks/backbone/resnet.py:27-28,
// networks/tools/aspp.py:37-39, networks/deeplabv3.py:25-41).
//
// conv_winograd.hip runs the same algebra as three memory-bound passes around a batched GEMM and moves 9.3x the
// algorithmic bytes (the 16-plane V written + read, the 16-plane M written + read).  Here a workgroup owns ALL 16
// components xi of a (64 output channels x 64 tiles) block: 16 x 64 x 64 / 256 lanes = 256 accumulator registers per
// lane (one wave per SIMD, like the 256 x 256 tile of the direct kernels), so
//   * the output transform  y = A^T m A  is register arithmetic in the epilogue (each lane holds all 16 components of
//     its 16 (channel, tile) elements) and only y is written - M never exists;
//   * the B operand V = B^T d B is built on the fly: per K-step (8 input channels) every thread loads the 4x4 patches
//     of TWO neighbouring tiles of ONE channel straight from x (16-byte / 8-byte buffer loads; out-of-image rows and
//     columns read zeros through an out-of-range offset or the zero tail of a row-pitched x), transforms them
//     (64 additions) and writes 16 x 8 bytes into the LDS image the MFMA fragments are read from - V never exists either
//     (it is written on the side, each block 1/mblocks of the channels, when the weight gradient wants to take it over:
//     dcfp_conv2d_fwd_keep_f32_nchw);
//   * the A operand (the transformed filters U, <= 34 MB, L2-resident) is copied global -> LDS by LDS-DMA from a layout
//     that is already MFMA-fragment order: Ug[c/8][m/64][xi][m%64/32][lane][kk] = U[xi][c = 8 cb + 2 kk + lane/32][m],
//     so a lane's four K-pairs of one component are ONE ds_read_b128.
// Operand traffic into LDS is 16 flop/B for A + B from V (tools/micro/wino_fused_probe.hip: 87 TF at C = 256, copies
// alone 11 TB/s = the L2 -> LDS ceiling) - which is why B is transformed in the kernel instead of copied: the A copies
// alone are free (133 TF with and without them).
//
// Tile numbering, super-blocks for dilation d, and the transforms' arithmetic (operation order included) are those of
// conv_winograd.hip: V written on the side is bit-identical to wino_input_kernel's.
#include "wino_patch.h"
#include <stdlib.h>

namespace {

constexpr int FBK = 8;                     // input channels per K-step
constexpr int FSTAGE = 16 * FBK * 64;      // floats of one operand of one LDS stage (32 KB)

struct WinoFusedParams {
    const float* in;        // x (forward) or dy (dgrad): [N][Ck][H][pitch], images in_nstride floats apart
    const float* in_base;   // descriptor base: `in` minus `lead` floats (readable zeros in front of a pitched tensor)
    unsigned in_bytes;      // descriptor bound
    int lead;
    long long in_nstride;
    int pitch;
    const float* ug;        // transformed filters, fragment order (see above)
    float* out;             // y (forward) or dx (dgrad): [N][M][H][W] dense rows, images out_nstride floats apart
    long long out_nstride;
    float* xform_out;       // nullable: V[16][Ck][T16]
    float* stat_part;       // nullable: BatchNorm partials of y, [slot][M][2] = (mean, M2) over 128 outputs
    const float* scale;     // nullable: inference epilogue y = act(conv * scale[m] + shift[m] (+ residual)) - the eval-mode
    const float* shift;     // BatchNorm folded into the conv (dcfp_conv2d_fwd_fused_f32_nchw)
    const float* residual;  // laid out as out
    int relu;
    int N, M, Ck, H, W, d, TH, TW;
    long long T, T16;
    int mblocks, tblocks, nk;
    int accumulate;
    int ragged_c;           // Ck % 16 != 0
    int vec_epi;            // 16-byte output stores (W % 4 == 0, 16-byte aligned images, 32-bit byte offsets)
    unsigned out_bytes;
};

// SIDE: V is written on the side (p.xform_out).  RAGGED: Ck % 16 != 0, the channels past Ck in the last K-steps read zeros.
// The K loop is ONE basic block (no branch around the MFMAs: with the accumulators live across a diamond hipcc 7.2 moves
// all 256 of them through VGPRs / scratch every iteration): conditional work is expressed through buffer offsets - an
// out-of-range offset makes a load return zeros and drops a store.
template <int DM, bool SIDE, bool RAGGED>
__global__ void __launch_bounds__(256, 1) wino_fused_kernel(const WinoFusedParams p) {
    constexpr int NV = PatchCfg<DM>::NV;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 stages][A 8192 | B 8192]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    // blocks b, b + 8, ... share an XCD (round-robin placement): the mblocks blocks of one tile block sit next to each
    // other there, so the patches they all read are served by that XCD's L2
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int mb = jb % p.mblocks, tb = (jb / p.mblocks) * 8 + xcd;
    if (tb >= p.tblocks) return;                                   // block-uniform
    const long long t0 = (long long)tb * 64;
    const int nk = p.nk, d = p.d, H = p.H, W = p.W, TW = p.TW, tpi = p.TH * p.TW, pitch = p.pitch;

    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const u32x4 a_desc = make_desc(p.ug, 0x7ffffffcu);
    const unsigned lane16 = lane * 16u;

    // ---- producer role: tile pair (tp, tp + 1), channel 2 wid + lhi of the K-step
    const int ch = 2 * wid + lhi;
    const long long tp = t0 + 2 * l31;
    unsigned voff[NV];
    {
        const bool tvalid = tp < p.T;
        const long long tq = tvalid ? tp : 0;
        const int n = (int)(tq / tpi), tt = (int)(tq - (long long)n * tpi);
        const int trow = tt / TW, tcol = tt - trow * TW;
        const int h0 = trow + d * (trow / d) - d, w0 = tcol + d * (tcol / d) - d;
        const long long base = (long long)n * p.in_nstride + (long long)lhi * H * pitch + p.lead;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = h0 + r * d;
            const bool rok = tvalid && (unsigned)h < (unsigned)H;
            const long long row = base + (long long)h * pitch;
            if constexpr (DM == 1 || DM == 2) {
                // a vector that starts inside the row may hang over its end (and the first one over its start): the
                // pitched layout keeps >= 4 zeros there
                voff[2 * r] = (rok && w0 < W) ? (unsigned)((row + w0) * 4) : kOob;
                voff[2 * r + 1] = (rok && w0 + 4 < W) ? (unsigned)((row + w0 + 4) * 4) : kOob;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int ww = w0 + s * d;                  // even, W even: the pair is inside or outside as a whole
                    voff[4 * r + s] = (rok && (unsigned)ww < (unsigned)W) ? (unsigned)((row + ww) * 4) : kOob;
                }
            }
        }
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_base), 0,
                                                                             p.in_bytes, 0x00020000);
    const unsigned chan_bytes = (unsigned)(H * pitch) * 4u;
    // patches of K-step j live in R[j & 1]: loaded during step j - 3 (slots 8..15), consumed during step j - 1 (slots 0..3)
    float R[2][4][8], q[4][8];
    auto load_step = [&](int kt, auto b_) {          // patches of K-step kt -> R[b]
        constexpr int b = decltype(b_)::value;
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(kt * FBK + 2 * wid_s) * chan_bytes);
        if constexpr (RAGGED) {
            unsigned vm[NV];
            const bool cok = kt * FBK + ch < p.Ck;
#pragma unroll
            for (int i = 0; i < NV; ++i) vm[i] = cok ? voff[i] : kOob;
            load_patches<DM>(in_rsrc, vm, soff, R[b]);
        } else {
            load_patches<DM>(in_rsrc, voff, soff, R[b]);
        }
    };
    auto issue_a = [&](int kt, int buf) {   // the 32 KB A image of K-step kt: 8 of its 32 one-KB pieces per wave
        const unsigned abase = (unsigned)((kt * p.mblocks + mb) * FSTAGE) * 4u;
        static_for<0, 8>([&](auto q_) {
            constexpr int qq = decltype(q_)::value;
            const unsigned piece = (unsigned)(wid_s * 8 + qq);
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * 2 * FSTAGE) * 4u + piece * 1024u);
            const unsigned a_s = __builtin_amdgcn_readfirstlane(abase + piece * 1024u);
            const unsigned av = lane16;
            const u32x4 ad = a_desc;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(la), "v"(av), "s"(ad), "s"(a_s) : "memory", "m0");
        });
    };
    auto retire = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // end of a K-step of the loop: everything but the NV youngest vector-memory operations has landed - those are the patch
    // loads issued in slots 8..15 (behind the last LDS-DMA piece), which have another whole step to arrive
    // (SIDE: a step that stores V issues its 16 stores in slots 8..15 too, behind the last LDS-DMA piece: 16 more
    //  operations may stay in flight.  The branches around the stores and between the two waits are scalar and hold no MFMA.)
    auto retire_keep_loads = [&](bool stored) {
        if (SIDE && stored) {
            if constexpr (NV == 8) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
        } else {
            if constexpr (NV == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    // component XI of the K-step whose row transform sits in q: into the B image of stage `buf` (and V on the side)
    // V[xi][c][tp .. tp + 1]: one descriptor per component (a plane is Ck * T16 * 4 bytes < 2 GB), lane offset of the
    // pair and the odd channel, the K-step's channel offset in the scalar offset
    const long long vplane = (long long)p.Ck * p.T16;
    const unsigned vs_lane = (SIDE && tp < p.T16) ? (unsigned)(((long long)lhi * p.T16 + tp) * 4) : kOob;
    auto side_on = [&](int kt) -> bool { return SIDE && kt % p.mblocks == mb; };   // this block stores these K-steps
    auto side_off = [&](int kt) -> unsigned {
        bool on = true;
        if constexpr (RAGGED) on = kt * FBK + ch < p.Ck;
        return on ? vs_lane : kOob;
    };
    auto side_store = [&](auto xi_, f32x2 o, int kt, unsigned vs) {
        constexpr int XI = decltype(xi_)::value;
        const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.xform_out + XI * vplane, 0,
                                                                                (unsigned)(vplane * 4), 0x00020000);
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((long long)(kt * FBK + 2 * wid_s) * p.T16 * 4));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), v_rsrc, vs, soff, 0);
    };
    auto produce = [&](auto xi_, int buf) -> f32x2 {
        constexpr int XI = decltype(xi_)::value;
        const f32x2 o = col_transform<DM, XI>(q);
        float* bs = smem + buf * 2 * FSTAGE + FSTAGE + (XI * FBK + ch) * 64 + 2 * l31;
        *reinterpret_cast<f32x2*>(bs) = o;
        return o;
    };

    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- prologue: stage 0 = K-step 0
    load_step(0, std::integral_constant<int, 0>{});
    row_transform<DM>(R[0], q);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(0, 0);
    {
        const unsigned vs = side_off(0);
        const bool st = side_on(0);
        static_for<0, 16>([&](auto xi_) {
            const f32x2 o = produce(xi_, 0);
            if constexpr (SIDE) {
                if (st) side_store(xi_, o, 0, vs);
            }
        });
    }
    load_step(1, std::integral_constant<int, 1>{});            // nk >= 2, even
    __builtin_amdgcn_sched_barrier(0);                         // (issue order = the loop's: set 1 wholly before set 0)
    load_step(nk > 2 ? 2 : 1, std::integral_constant<int, 0>{});
    retire();

    const int a_lane = wm * 256 + lane * 4;
    const int b_lane = lhi * 64 + wn * 32 + l31;
    // ---- K loop.  A K-step is 16 slots, one per component g: slot g reads the fragments of component g (consumed one
    // slot later) and issues the 4 MFMAs of component g - 1; slot 0 issues those of the PREVIOUS step's component 15, whose
    // fragments stay in registers across the barrier - the matrix pipe has work queued while the waves meet and while the
    // first fragments of the new stage arrive.  Beside the MFMAs a slot carries its share of building K-step kt + 1:
    // slots 0..3 one row of B^T d each, every slot the component's column transform + LDS write (+ the V store), slots
    // 0..7 one of the wave's 8 LDS-DMA pieces of A, slots 8..15 the patch loads of K-step kt + 3 into the register set the
    // row transform has just released (two sets: a load has more than a whole step to land).  The step parity is static
    // (the loop body is two steps): LDS stage and register set are compile-time.
    f32x4 af[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float bf[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto step = [&](auto prod_, auto par_, int kt) {
        constexpr bool PROD = decltype(prod_)::value;
        constexpr int cur = decltype(par_)::value, rb = cur ^ 1;
        unsigned vs = kOob;
        bool st = false;
        if constexpr (PROD) { vs = side_off(kt + 1); st = side_on(kt + 1); }
        const float* As = smem + cur * 2 * FSTAGE + a_lane;
        const float* Bs = smem + cur * 2 * FSTAGE + FSTAGE + b_lane;
        const int ktl = kt + 3 < nk ? kt + 3 : nk - 1;     // (the last loads are harmless repeats: no branch in the loop)
        const unsigned l_soff = __builtin_amdgcn_readfirstlane((unsigned)(ktl * FBK + 2 * wid_s) * chan_bytes);
        unsigned vm[NV];
        if constexpr (PROD) {
#pragma unroll
            for (int i = 0; i < NV; ++i) vm[i] = voff[i];
            if constexpr (RAGGED) {
                const bool cok = ktl * FBK + ch < p.Ck;
#pragma unroll
                for (int i = 0; i < NV; ++i) vm[i] = cok ? voff[i] : kOob;
            }
        }
        const unsigned abase = (unsigned)(((kt + 1) * p.mblocks + mb) * FSTAGE) * 4u;
        f32x2 okeep[8];
        static_for<0, 16>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            constexpr int fb = g & 1, pg = (g + 15) & 15;
            af[fb] = *reinterpret_cast<const f32x4*>(As + g * 512);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) bf[fb][kk] = Bs[(g * FBK + 2 * kk) * 64];
            // ONE burst of vector / scalar ALU work per slot, in front of its MFMAs (round 4, tools/micro/mfma_shadow.hip: with one
            // wave per SIMD every VALU / SALU instruction costs ~4.5 cycles of matrix-pipe time, every switch MFMA -> ALU -> MFMA
            // ~8 more, and an LDS write right behind the instruction that produced its data ~20; LDS reads / writes by themselves
            // are free): the row transform (slots 0..3), this component's column transform, the LDS-DMA piece's M0 set-up.  The
            // LDS write of the component follows two MFMAs later.
            f32x2 o = {0.f, 0.f};
            if constexpr (PROD) {
                if constexpr (g < 4) {
                    row_transform_one<DM, g>(R[rb], q);
                    // pin the row here: left alone, hipcc sinks these subtractions to the column transforms of slots 4..15,
                    // which keeps R alive under the loads of slots 8..15 (register copies behind fresh loads = stalls)
#pragma unroll
                    for (int c = 0; c < PatchCfg<DM>::NCOL; ++c) asm volatile("" : "+v"(q[g][c]));
                }
                o = col_transform<DM, g>(q);
                asm volatile("" : "+v"(o));
                if constexpr (g < 8) {        // piece g of this wave's 8 LDS-DMA pieces of the next A image
                    const unsigned piece = (unsigned)(wid_s * 8 + g);
                    const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((cur ^ 1) * 2 * FSTAGE) * 4u + piece * 1024u);
                    const unsigned a_s = __builtin_amdgcn_readfirstlane(abase + piece * 1024u);
                    const unsigned av = lane16;
                    const u32x4 ad = a_desc;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                 :: "s"(la), "v"(av), "s"(ad), "s"(a_s) : "memory", "m0");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][0], bf[fb ^ 1][0], acc[pg], 0, 0, 0);
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][1], bf[fb ^ 1][1], acc[pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROD) {
                float* bs = smem + (cur ^ 1) * 2 * FSTAGE + FSTAGE + (g * FBK + ch) * 64 + 2 * l31;
                *reinterpret_cast<f32x2*>(bs) = o;
                if constexpr (SIDE) {
                    if constexpr (g < 8) okeep[g] = o;
                    else if (st) {
                        side_store(std::integral_constant<int, g - 8>{}, okeep[g - 8], kt + 1, vs);
                        side_store(g_, o, kt + 1, vs);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][2], bf[fb ^ 1][2], acc[pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROD && g >= 8) {
                constexpr int per = NV / 8;
                static_for<0, per>([&](auto j_) { load_one<DM, (g - 8) * per + decltype(j_)::value>(in_rsrc, vm, l_soff, R[rb]); });
            }
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][3], bf[fb ^ 1][3], acc[pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    constexpr std::integral_constant<int, 0> even{};
    constexpr std::integral_constant<int, 1> odd{};
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {          // nk is even: the producing steps 0 .. nk - 2 are pairs + one
        step(std::true_type{}, even, kt);
        retire_keep_loads(side_on(kt + 1));
        step(std::true_type{}, odd, kt + 1);
        retire_keep_loads(side_on(kt + 2));
    }
    step(std::true_type{}, even, nk - 2);
    retire();
    step(std::false_type{}, odd, nk - 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)        // component 15 of the last step
        acc[15] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][kk], bf[1][kk], acc[15], 0, 0, 0);

    // ---- epilogue: y = A^T m A per (channel, tile), this lane's tile and 16 channels
    const long long te = t0 + wn * 32 + l31;
    const bool tv = te < p.T;
    const long long tq = tv ? te : 0;
    const int n = (int)(tq / tpi), tt = (int)(tq - (long long)n * tpi);
    const int trow = tt / TW, tcol = tt - trow * TW;
    const int ho = trow + d * (trow / d), wo = tcol + d * (tcol / d);
    const bool okr0 = tv && ho < H, okr1 = tv && ho + d < H;
    const bool okc0 = wo < W, okc1 = wo + d < W;
    const long long HW = (long long)H * W;
    const int m_lane = mb * 64 + wm * 32 + 4 * lhi;
    const int acc_out = p.accumulate;
    float* const sp0 = p.stat_part;          // BatchNorm partials: (mean, M2) over the 128 outputs of this wave's 32 tiles
    const long long slot = (long long)tb * 2 + wn;
    auto out_transform = [&](int r, float (&o)[2][2]) {
        float u[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u[0][s] = (acc[0 + s][r] + acc[4 + s][r]) + acc[8 + s][r];
            u[1][s] = (acc[4 + s][r] - acc[8 + s][r]) - acc[12 + s][r];
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            o[rr][0] = (u[rr][0] + u[rr][1]) + u[rr][2];
            o[rr][1] = (u[rr][1] - u[rr][2]) - u[rr][3];
        }
    };
    // sum over the 32 lanes of a wave half, valid in every lane: DPP adds inside the 16-lane rows (quad exchanges, then the
    // mirrored halves), row_bcast15 into the odd rows, two v_readlane (a ds_bpermute butterfly cost 0.27 ms of a 0.84 ms
    // launch on the 64-channel stem conv, whose blocks are only 8 K-steps long)
    auto red32 = [&](float v) -> float {
        auto dpp = [](float x, auto ctrl_) {
            constexpr int ctrl = decltype(ctrl_)::value;
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true));
        };
        v += dpp(v, std::integral_constant<int, 0xB1>{});        // quad_perm [1,0,3,2]
        v += dpp(v, std::integral_constant<int, 0x4E>{});        // quad_perm [2,3,0,1]
        v += dpp(v, std::integral_constant<int, 0x141>{});       // row_half_mirror
        v += dpp(v, std::integral_constant<int, 0x140>{});       // row_mirror: every lane holds its row's sum
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
        return lhi ? hi : lo;
    };
    auto stats = [&](int m, const float (&o)[2][2]) {
        // block-uniform; only launched where every tile lies wholly inside the image (dcfp_wino_stat_slots)
        const float sum = red32((o[0][0] + o[0][1]) + (o[1][0] + o[1][1]));
        const float mean = sum * (1.0f / 128.0f);
        const float d0 = o[0][0] - mean, d1 = o[0][1] - mean, d2 = o[1][0] - mean, d3 = o[1][1] - mean;
        const float m2 = red32((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        if (l31 == 0 && m < p.M && t0 + wn * 32 < p.T) {      // (T is a multiple of 32 here, not necessarily of 64)
            float* sp = sp0 + (slot * p.M + m) * 2;
            sp[0] = mean; sp[1] = m2;
        }
    };
    if (p.vec_epi) {
        // 16-byte stores: the 2x2 outputs of neighbouring tiles are regrouped among the lanes of a quad so that every lane
        // holds 4 consecutive pixels of one output row.  Offsets are 32-bit (checked by the host); a lane without
        // anything to store carries an out-of-range offset (the store is dropped): no branches.
        auto xor1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); };
        auto xor2 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); };
        const bool odd = l31 & 1, hi2 = l31 & 2;
        int row, col;       // the row / first column this lane stores
        bool ok;
        if constexpr (DM == 4) {
            const int i = l31 & 3;
            row = ho + (i >> 1) * d;
            col = wo - i + (i & 1) * d;
            ok = tv && row < H && col < W;
        } else if constexpr (DM == 2) {
            row = ho + (odd ? d : 0);
            col = wo - (odd ? 1 : 0);
            ok = tv && row < H && col < W;
        } else {
            row = ho + (odd ? 1 : 0);
            col = wo - (odd ? 2 : 0);
            ok = tv && row < H && col < W;
        }
        const unsigned evoff = ok ? (unsigned)(((long long)n * p.out_nstride + (long long)m_lane * HW + (long long)row * W + col) * 4) : kOob;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
        const unsigned HW4 = (unsigned)HW * 4u;
        static_for<0, 16>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            constexpr int mr = (r & 3) + 8 * (r >> 2);
            float o[2][2];
            out_transform(r, o);
            if (sp0) stats(m_lane + mr, o);
            f32x4 v;
            if constexpr (DM == 4) {
                const float x0 = odd ? o[0][0] : o[0][1], x1 = odd ? o[1][0] : o[1][1];
                const float y0 = xor1(x0), y1 = xor1(x1);
                const float b0 = odd ? y0 : o[0][0], b1 = odd ? o[0][1] : y0, b2 = odd ? y1 : o[1][0], b3 = odd ? o[1][1] : y1;
                const float z0 = hi2 ? b0 : b2, z1 = hi2 ? b1 : b3;
                const float w0 = xor2(z0), w1 = xor2(z1);
                v[0] = hi2 ? w0 : b0; v[1] = hi2 ? w1 : b1; v[2] = hi2 ? b2 : w0; v[3] = hi2 ? b3 : w1;
            } else {
                const float s0 = odd ? o[0][0] : o[1][0], s1 = odd ? o[0][1] : o[1][1];
                const float r0 = xor1(s0), r1 = xor1(s1);
                if constexpr (DM == 2) {
                    v[0] = odd ? r0 : o[0][0]; v[1] = odd ? o[1][0] : r0; v[2] = odd ? r1 : o[0][1]; v[3] = odd ? o[1][1] : r1;
                } else {
                    v[0] = odd ? r0 : o[0][0]; v[1] = odd ? r1 : o[0][1]; v[2] = odd ? o[1][0] : r0; v[3] = odd ? o[1][1] : r1;
                }
            }
            const unsigned vo = (m_lane + mr < p.M) ? evoff : kOob;
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)mr * HW4);
            if (p.scale) {      // block-uniform
                const int mm = m_lane + mr < p.M ? m_lane + mr : p.M - 1;
                const float sc = p.scale[mm], sf = p.shift[mm];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc, sf);
                if (p.residual) {
                    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0,
                                                                                            p.out_bytes, 0x00020000);
                    v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, vo, so, 0));
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (acc_out) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, vo, so, 0));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, vo, so, 0);
        });
        return;
    }
    float* obase = p.out + (long long)n * p.out_nstride + (long long)ho * W + wo;
    const long long dW = (long long)d * W;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m_lane + (r & 3) + 8 * (r >> 2);
        float o[2][2];
        out_transform(r, o);
        if (m < p.M && p.scale) {
            const float sc = p.scale[m], sf = p.shift[m];
            const float* rs = p.residual ? p.residual + (obase - p.out) + (long long)m * HW : nullptr;
            const long long offs[2][2] = {{0, d}, {dW, dW + d}};
            const bool oks[2][2] = {{okr0 && okc0, okr0 && okc1}, {okr1 && okc0, okr1 && okc1}};
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    float t = fmaf(o[rr][cc], sc, sf);
                    if (rs && oks[rr][cc]) t += rs[offs[rr][cc]];
                    o[rr][cc] = p.relu ? (t > 0.f ? t : 0.f) : t;
                }
        }
        if (m < p.M) {
            float* e = obase + (long long)m * HW;
            if (acc_out) {
                if (okr0 && okc0) e[0] += o[0][0];
                if (okr0 && okc1) e[d] += o[0][1];
                if (okr1 && okc0) e[dW] += o[1][0];
                if (okr1 && okc1) e[dW + d] += o[1][1];
            } else {
                if (okr0 && okc0) e[0] = o[0][0];
                if (okr0 && okc1) e[d] = o[0][1];
                if (okr1 && okc0) e[dW] = o[1][0];
                if (okr1 && okc1) e[dW + d] = o[1][1];
            }
        }
        if (sp0) stats(m, o);
    }
}

// Ug[cb][mb][xi][wm][lane][kk] = (G g G^T)[xi] of g = w[m * sAm + c * sAc + tap] (flip: 8 - tap), c = 8 cb + 2 kk + lane / 32,
// m = 64 mb + 32 wm + lane % 32; zeros past M / Ck
__global__ void __launch_bounds__(256) wino_filter2_kernel(const float* __restrict__ w, int sAm, int sAc, int flip, int M,
                                                           int Ck, int CkP, int Mpad, float* __restrict__ Ug) {
    const long long total = (long long)CkP * Mpad;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        wino_filter2_pair(w, sAm, sAc, flip, M, Ck, Mpad, idx, Ug);      // (igemm2_common.h)
}

struct FusedPlan {
    int TH, TW, mblocks, tblocks, nk, CkP, Mpad;
    long long T, T16, ug_floats;
};
FusedPlan fused_plan(int N, int H, int W, int d, int M, int Ck) {
    FusedPlan pl;
    pl.TH = d * ((H + 2 * d - 1) / (2 * d));
    pl.TW = d * ((W + 2 * d - 1) / (2 * d));
    pl.TW = (pl.TW + 3) / 4 * 4;
    pl.T = (long long)N * pl.TH * pl.TW;
    pl.T16 = (pl.T + 15) / 16 * 16;
    pl.CkP = (Ck + 15) / 16 * 16;                 // an even number of K-steps (the K loop's body is two steps)
    pl.Mpad = (M + 63) / 64 * 64;
    pl.mblocks = pl.Mpad / 64;
    pl.tblocks = (int)((pl.T + 63) / 64);
    pl.nk = pl.CkP / 8;
    pl.ug_floats = ((long long)16 * pl.CkP * pl.Mpad + 63) / 64 * 64;
    return pl;
}

}  // namespace

// DCFP_WINO_FUSED: 0 off (the three-pass path of conv_winograd.hip), 1 on (default)
bool dcfp_wino_fused_enabled() {
    static const int v = [] { const char* e = getenv("DCFP_WINO_FUSED"); return e ? atoi(e) : 1; }();
    return v != 0;
}

// patch-load mode for this input, or -1 where the kernel does not apply
static int fused_mode(int N, int H, int W, int d, int Ck, long long in_nstride, int pitch) {
    if (pitch <= 0) pitch = W;
    const long long span = (long long)(N - 1) * in_nstride + (long long)((Ck + 15) / 16 * 16) * H * pitch + 64;
    if (span * 4 >= 0x7fffff00LL) return -1;                 // 32-bit byte offsets with bit 31 as the out-of-range mark
    if (d == 1 && pitch >= W + 4) return 1;
    if (d == 2 && pitch >= W + 4) return 2;
    if (d >= 4 && d % 2 == 0 && W % 2 == 0) return 4;
    return -1;
}

bool dcfp_wino_fused_ok(int N, int H, int W, int d, int M, int Ck, long long in_nstride, int pitch) {
    if (!dcfp_wino_fused_enabled()) return false;
    if (M < 48 || Ck < 48) return false;       // (the cost model of conv_igemm.hip decides above that)
    const FusedPlan pl = fused_plan(N, H, W, d, M, Ck);
    if (pl.T16 >= (1LL << 30) || (long long)pl.tblocks * pl.mblocks + 8 * pl.mblocks >= (1LL << 31)) return false;
    if ((long long)pl.CkP * pl.Mpad * 16 * 4 >= 0x7fffff00LL) return false;
    return fused_mode(N, H, W, d, Ck, in_nstride, pitch) >= 0;
}

// padded channel counts of the transformed-filter image (for dcfp_conv2d_wp_layout)
void dcfp_wino_fused_pads(int N, int H, int W, int d, int M, int Ck, int* CkP, int* Mpad) {
    const FusedPlan pl = fused_plan(N, H, W, d, M, Ck);
    *CkP = pl.CkP; *Mpad = pl.Mpad;
}

size_t dcfp_wino_fused_workspace_bytes(int N, int H, int W, int d, int M, int Ck) {
    return (size_t)fused_plan(N, H, W, d, M, Ck).ug_floats * sizeof(float);
}

int dcfp_wino_fused_run(const float* in, long long in_nstride, int in_pitch, const float* w, int sAm, int sAc, int flip,
                        float* out, long long out_nstride, int N, int M, int Ck, int H, int W, int d, int accumulate,
                        void* workspace, size_t workspace_bytes, hipStream_t stream, float* xform_out, float* stat_part,
                        const float* scale, const float* shift, const float* residual, int relu, int wp_valid) {
    const FusedPlan pl = fused_plan(N, H, W, d, M, Ck);
    if (!workspace || !dcfp_aligned16(workspace) || workspace_bytes < (size_t)pl.ug_floats * sizeof(float))
        return DCFP_E_WORKSPACE;
    const int pitch = in_pitch > 0 ? in_pitch : W;
    const int mode = fused_mode(N, H, W, d, Ck, in_nstride, pitch);
    if (mode < 0) return DCFP_E_UNSUPPORTED;
    float* Ug = static_cast<float*>(workspace);
    if (!wp_valid) {      // (valid: the caller kept U from an earlier call / the multi-tensor refresh and the weights are unchanged)
        long long b = ((long long)pl.CkP * pl.Mpad + 255) / 256;
        if (b > 4096) b = 4096;
        hipLaunchKernelGGL(wino_filter2_kernel, dim3((unsigned)b), dim3(256), 0, stream, w, sAm, sAc, flip, M, Ck, pl.CkP,
                           pl.Mpad, Ug);
    }
    WinoFusedParams p;
    p.in = in;
    p.lead = (mode == 1 || mode == 2) ? 4 : 0;          // the pitched layout keeps pitch - W >= 4 readable zeros in front
    p.in_base = in - p.lead;
    p.in_bytes = (unsigned)(((long long)(N - 1) * in_nstride + (long long)Ck * H * pitch + p.lead) * 4);
    p.in_nstride = in_nstride;
    p.pitch = pitch;
    p.ug = Ug;
    p.out = out;
    p.out_nstride = out_nstride;
    p.xform_out = xform_out;
    p.stat_part = stat_part;
    p.scale = scale; p.shift = shift; p.residual = residual; p.relu = relu;
    p.N = N; p.M = M; p.Ck = Ck; p.H = H; p.W = W; p.d = d; p.TH = pl.TH; p.TW = pl.TW;
    p.T = pl.T; p.T16 = pl.T16;
    p.mblocks = pl.mblocks; p.tblocks = pl.tblocks; p.nk = pl.nk;
    p.accumulate = accumulate;
    p.ragged_c = (Ck % 16) != 0;
    {
        const long long ospan = ((long long)(N - 1) * out_nstride + (long long)M * H * W) * 4;
        p.vec_epi = W % 4 == 0 && out_nstride % 4 == 0 && dcfp_aligned16(out) && ospan < 0x7fffff00LL &&
                    (mode != 4 || d % 4 == 0);
        p.out_bytes = p.vec_epi ? (unsigned)ospan : 0u;
        static const int no_vec = [] { const char* e = getenv("DCFP_WF_SCALAR_EPI"); return e ? atoi(e) : 0; }();
        if (no_vec) p.vec_epi = 0;
    }
    const long long grid = (long long)((pl.tblocks + 7) / 8) * 8 * pl.mblocks;
    const size_t lds = (size_t)4 * FSTAGE * sizeof(float);
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
        DCFP_RETURN_LAUNCH();
    };
    if (xform_out && (long long)Ck * pl.T16 * 4 >= 0x7fffff00LL) return DCFP_E_UNSUPPORTED;
    const bool ragged = (Ck % 16) != 0;
#define DCFP_WF(DM_) (xform_out ? (ragged ? launch(wino_fused_kernel<DM_, true, true>) : launch(wino_fused_kernel<DM_, true, false>)) \
                                : (ragged ? launch(wino_fused_kernel<DM_, false, true>) : launch(wino_fused_kernel<DM_, false, false>)))
    switch (mode) {
        case 1: return DCFP_WF(1);
        case 2: return DCFP_WF(2);
        case 4: return DCFP_WF(4);
        default: return DCFP_E_UNSUPPORTED;
    }
#undef DCFP_WF
}

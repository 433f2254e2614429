// conv_winograd3.hip — FUSED Winograd F(2x2, 3x3) WEIGHT GRADIENT: both operands transformed inside the GEMM
// (networks/backbone/resnet.py:27-28 conv2 of every Bottleneck, :88-96 the stem's 3x3 convs, networks/tools/aspp.py:37-39,
// networks/deeplabv3.py:25-41 - the autograd wgrad of those nn.Conv2d).
//
//   dU[xi][m][c] = sum_t (A dy_t A^T)[xi] * (B^T x_t B)[xi],   dw = G^T dU G          (t = 2x2 output tiles)
//
// conv_winograd.hip computes this from TWO materialised 16-plane tensors - V = B^T x B (kept from the forward pass: 4x the
// activation, 22 GB per step for the model's 32 wide convs) and Y = A dy A^T (a wino_dy_kernel pass) - through a batched
// GEMM that fetches 2x its algorithmic bytes (half-line K-steps).  Here a workgroup owns ALL 16 components of a
// (64 output channels x 64 input channels) block of dU - 256 accumulator registers per lane, the budget of the fused
// forward kernel (conv_winograd2.hip) - and walks K = tiles, 8 per K-step:
//   * every thread loads the 4x4 x-patches of ONE tile pair of ONE input channel (as conv_winograd2.hip does) AND the 2x2
//     dy-tiles of the same pair of ONE output channel, transforms both in registers (64 + 24 additions) and writes
//     16 + 16 eight-byte values into the two LDS images the MFMA fragments are read from: V and Y never exist;
//   * both images keep a channel's tiles contiguous: a lane's four K-pairs of a component are ONE ds_read_b128 for A and
//     one for B (the forward kernel needs four ds_read_b32 for its B operand);
//   * the sign of a dy component, (-1)^[i == 3] (-1)^[j == 3] (A = [1 0; 1 1; 1 -1; 0 -1]), is applied to the accumulator
//     in the epilogue instead of to 16 values per K-step;
//   * split-K over the tiles in a fixed order: slabs [split][xi][M][C], summed and transformed G^T dU G by ONE small
//     kernel (wino_dw_reduce_kernel) - deterministic, no atomics.
// Where a tile pair sits (image, rows, columns, which of them lie outside the image) comes from a table the launcher's
// wino_wg_table_kernel writes per call (32 bytes per pair, L2-resident): the K loop has no divisions and no branches.
// No instruction in the loop is inline-asm memory traffic, so the compiler's own vmcnt bookkeeping is exact: a step ends
// in `s_waitcnt lgkmcnt(0)` + barrier and the patch loads issued three K-steps ahead are waited for where they are used.
#include "wino_patch.h"
#include <stdlib.h>

namespace {

constexpr int GBK = 8;                      // tiles per K-step
// LDS image of one operand of one stage, per component: [half (2: wave row / column)][k half (2)][channel 32][4 tiles], the
// k halves 144 floats apart (128 + 16: half a bank row) -
//   * a lane's fragment (channel l31, tiles 4 lhi .. 4 lhi + 3) is 16 bytes and consecutive lanes read consecutive 16-byte
//     pieces: conflict-free ds_read_b128 for both halves of the wave;
//   * a producer thread (channel, tile pair j) writes 8 bytes at k half j >> 1: the 16 lanes of a ds_write_b64 group cover 4
//     channels x 4 pairs = banks 0..15 (j < 2) and 16..31 (j >= 2) - conflict-free because of the 16-float skew.
// (A plain [channel 64][tile 8] image made both accesses 2-way conflicts: profiles/r04_wino_wgrad_fused_sq_pmc.txt.)
constexpr int GKH = 144;                    // floats between the k halves
constexpr int GHALF = 2 * GKH;              // floats of one 32-channel half of a component
constexpr int GCOMP = 2 * GHALF;            // floats of one component (576)
constexpr int GSTAGE = 16 * GCOMP;          // floats of one operand of one LDS stage (36 KB)
constexpr unsigned kSat = 0xffffffffu;

struct WinoWgParams {
    const float* x_base;     // x minus `lead` floats (DM 1 / 2: readable zeros in front of a pitched tensor)
    unsigned x_bytes;
    const float* dy_base;
    unsigned dy_bytes;
    const unsigned* table;   // [T16 / 2][8], see wino_wg_table_kernel
    unsigned table_bytes;
    float* out;              // [splits][16][M][C]
    int M, C, mblocks, cblocks, nblk;
    unsigned x_chan_bytes, dy_chan_bytes;
    int kchunk;              // tiles per split: a multiple of 16 (the K loop's body is two steps)
    long long T16;
    int d;
};

// Table entry of the tile pair (2 pi, 2 pi + 1) - tiles numbered as in conv_winograd.hip (n, tile row, tile column;
// super-blocks of 2d x 2d pixels):
//   e[0..3]  byte offset of patch row r of x (image, row; DM 1 / 2: + first column + lead), kOob where the row lies
//            outside the image or the pair has no output inside it
//   e[4..5]  byte offset of output row r of dy (DM 1 / 2: + first column), kOob likewise
//   e[6]     DM 4: bits 0..3 patch column s inside the image, bits 4..5 output column s inside the image
//   e[7]     DM 4: 4 * w0 (first patch column, may be negative)
// DM 1 / 2 need no column flags: x is row-pitched (>= 4 zeros behind each row, `lead` zeros in front of the first), W % 4 == 0.
__global__ void __launch_bounds__(256) wino_wg_table_kernel(unsigned* __restrict__ table, long long npairs, long long T, int N,
                                                            int H, int W, int d, int TH, int TW, long long x_nstride,
                                                            int x_pitch, int lead, long long dy_nstride, int dy_pitch, int dm) {
    const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pi >= npairs) return;
    unsigned e[8] = {kOob, kOob, kOob, kOob, kOob, kOob, 0u, 0u};
    const long long t = 2 * pi;
    if (t < T) {
        const int tpi = TH * TW;
        const int n = (int)(t / tpi), tt = (int)(t - (long long)n * tpi);
        const int trow = tt / TW, tcol = tt - trow * TW;
        const int ho = trow + d * (trow / d), wo = tcol + d * (tcol / d);
        if (ho < H && wo < W) {
            const int h0 = ho - d, w0 = wo - d;
            for (int r = 0; r < 4; ++r) {
                const int h = h0 + r * d;
                if (h >= 0 && h < H)
                    e[r] = dm == 4 ? (unsigned)(((long long)n * x_nstride + (long long)h * x_pitch) * 4)
                                   : (unsigned)(((long long)n * x_nstride + (long long)h * x_pitch + w0 + lead) * 4);
            }
            for (int r = 0; r < 2; ++r) {
                const int h = ho + r * d;
                if (h < H)
                    e[4 + r] = dm == 4 ? (unsigned)(((long long)n * dy_nstride + (long long)h * dy_pitch) * 4)
                                       : (unsigned)(((long long)n * dy_nstride + (long long)h * dy_pitch + wo) * 4);
            }
            if (dm == 4) {
                unsigned m = 0;
                for (int s = 0; s < 4; ++s)
                    if (w0 + s * d >= 0 && w0 + s * d < W) m |= 1u << s;
                for (int s = 0; s < 2; ++s)
                    if (wo + s * d < W) m |= 16u << s;
                e[6] = m;
                e[7] = (unsigned)(w0 * 4);
            }
        }
    }
    u32x4* dst = reinterpret_cast<u32x4*>(table + pi * 8);
    dst[0] = u32x4{e[0], e[1], e[2], e[3]};
    dst[1] = u32x4{e[4], e[5], e[6], e[7]};
}

// dw[m][c][3][3] = G^T (sum_s slab[s]) G, slabs [split][xi][mc]; s ascending (fixed order), then wino_dw_kernel's expressions
__global__ void __launch_bounds__(256) wino_dw_reduce_kernel(const float* __restrict__ slabs, int splits, long long mc,
                                                             float* __restrict__ dw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mc) return;
    float u[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = slabs[k * mc + i];
    // (the loads of FOUR slabs in flight together - 64 per thread - and added in slab order: the same sums as one slab at a
    //  time, a quarter of the dependent round trips: 22 -> ~10 us per launch, 42 launches per step)
    int s = 1;
    for (; s + 4 <= splits; s += 4) {
        float v[4][16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* sl = slabs + (long long)(s + q) * 16 * mc + i;
#pragma unroll
            for (int k = 0; k < 16; ++k) v[q][k] = sl[k * mc];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 16; ++k) u[k] += v[q][k];
    }
    for (; s < splits; ++s) {
        const float* sl = slabs + (long long)s * 16 * mc + i;
#pragma unroll
        for (int k = 0; k < 16; ++k) u[k] += sl[k * mc];
    }
    float t[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float u0 = u[0 + s], u1 = u[4 + s], u2 = u[8 + s], u3 = u[12 + s];
        t[0][s] = u0 + 0.5f * (u1 + u2);
        t[1][s] = 0.5f * (u1 - u2);
        t[2][s] = 0.5f * (u1 + u2) + u3;
    }
    float* o = dw + i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[3 * r + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
        o[3 * r + 1] = 0.5f * (t[r][1] - t[r][2]);
        o[3 * r + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    }
}

__device__ __forceinline__ unsigned sat_add(unsigned a, unsigned b) {      // min(a + b, 2^32 - 1): an out-of-range mark survives
    unsigned r;
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// dy registers of a pair: D[r][..] output row r.  DM 1 / 2: four consecutive floats (one 16-byte load); DM 4: D[r][2s + t] =
// column s of tile t (two 8-byte loads).  Column of tile A / B for output column s:
template <int DM> __device__ __forceinline__ constexpr int dcolA(int s) { return DM == 1 ? s : 2 * s; }
template <int DM> __device__ __forceinline__ constexpr int dcolB(int s) { return DM == 1 ? 2 + s : 2 * s + 1; }
template <int DM> struct DyCfg { static constexpr int NY = DM == 4 ? 4 : 2; };

template <int DM, int I>
__device__ __forceinline__ void load_dy_one(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&voff)[DyCfg<DM>::NY],
                                            float (&D)[2][4]) {
    if constexpr (DM == 4) {
        constexpr int r = I >> 1, s = I & 1;
        const f32x2 a = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[I], 0, 0));
        D[r][2 * s] = a[0]; D[r][2 * s + 1] = a[1];
    } else {
        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[I], 0, 0));
        D[I][0] = a[0]; D[I][1] = a[1]; D[I][2] = a[2]; D[I][3] = a[3];
    }
}

// component xi = 4 i + j of (A g A^T) WITHOUT its sign, tiles A and B; U[i] = row i of A g per register column
template <int DM, int XI>
__device__ __forceinline__ f32x2 dy_component(const float (&U)[4][4]) {
    constexpr int i = XI >> 2, j = XI & 3;
    const float a0 = U[i][dcolA<DM>(0)], a1 = U[i][dcolA<DM>(1)], b0 = U[i][dcolB<DM>(0)], b1 = U[i][dcolB<DM>(1)];
    f32x2 o;
    if constexpr (j == 0) { o[0] = a0; o[1] = b0; }
    if constexpr (j == 1) { o[0] = a0 + a1; o[1] = b0 + b1; }
    if constexpr (j == 2) { o[0] = a0 - a1; o[1] = b0 - b1; }
    if constexpr (j == 3) { o[0] = a1; o[1] = b1; }
    return o;
}

// RAGGED: M % 64 != 0 or C % 64 != 0 (pruned widths): lanes past the channel count read zeros and store nothing.
// The K loop is ONE basic block per step (conv_winograd2.hip: a branch around the MFMAs makes hipcc 7.2 move the 256
// accumulators through VGPRs every iteration); conditional work is expressed through buffer offsets.
template <int DM, bool RAGGED>
__global__ void __launch_bounds__(256, 1) wino_wgrad_fused_kernel(const WinoWgParams p) {
    constexpr int NX = PatchCfg<DM>::NV, NY = DyCfg<DM>::NY, NT = 2 + NX + NY;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 stages][A GSTAGE | B GSTAGE]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    // linear block L = (split * cblocks + cb) * mblocks + mb, cut into 8 contiguous runs, one per XCD (blocks b, b + 8, ...
    // share an XCD): the blocks of a split read the same tiles of x and dy and find them in that XCD's L2
    const int per = (p.nblk + 7) >> 3;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int L = xcd * per + jb;
    if (L >= p.nblk) return;                                      // block-uniform
    const int mb = L % p.mblocks, cb = (L / p.mblocks) % p.cblocks, split = L / (p.mblocks * p.cblocks);
    const long long tile0 = (long long)split * p.kchunk;
    const long long left = p.T16 - tile0;
    const int nk = (int)((left < p.kchunk ? left : (long long)p.kchunk) / GBK);      // even, >= 2

    // ---- producer role: tile pair j of the K-step, x channel cb * 64 + cl and dy channel mb * 64 + cl
    const int j = tid & 3, cl = tid >> 2;
    const int cg = cb * 64 + cl, mg = mb * 64 + cl;
    const bool c_ok = !RAGGED || cg < p.C, m_ok = !RAGGED || mg < p.M;
    const unsigned xc = (unsigned)cg * p.x_chan_bytes, dyc = (unsigned)mg * p.dy_chan_bytes;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x_base), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy_base), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.table), 0, p.table_bytes, 0x00020000);
    const unsigned ev = (unsigned)j * 32u;
    const unsigned e_s0 = (unsigned)(tile0 >> 1) * 32u;           // (table_bytes < 2^31: checked by the launcher)
    const unsigned d4 = (unsigned)p.d * 4u;

    unsigned E[8];
    auto load_entry = [&](int k) {
        const unsigned soff = __builtin_amdgcn_readfirstlane(e_s0 + (unsigned)k * 128u);
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(t_rsrc, ev, soff, 0);
        E[0] = a[0]; E[1] = a[1]; E[2] = a[2]; E[3] = a[3];
        if constexpr (DM == 4) {
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(t_rsrc, ev + 16u, soff, 0);
            E[4] = b[0]; E[5] = b[1]; E[6] = b[2]; E[7] = b[3];
        } else {
            const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(t_rsrc, ev + 16u, soff, 0);
            E[4] = b[0]; E[5] = b[1];
        }
    };
    unsigned vx[NX], vy[NY];
    auto make_offsets = [&]() {
        if constexpr (DM == 4) {
            const unsigned w0b = E[7];
            unsigned colk[4], dcol[2];
#pragma unroll
            for (int s = 0; s < 4; ++s) colk[s] = ((E[6] >> s) & 1u) && c_ok ? xc + w0b + (unsigned)s * d4 : kSat;
#pragma unroll
            for (int s = 0; s < 2; ++s) dcol[s] = ((E[6] >> (4 + s)) & 1u) && m_ok ? dyc + w0b + (unsigned)(1 + s) * d4 : kSat;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) vx[4 * r + s] = sat_add(E[r], colk[s]);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int s = 0; s < 2; ++s) vy[2 * r + s] = sat_add(E[4 + r], dcol[s]);
        } else {
            // (kOob + a channel offset below 2^31 stays out of range)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned a = c_ok ? E[r] + xc : kOob;
                vx[2 * r] = a; vx[2 * r + 1] = a + 16u;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) vy[r] = m_ok ? E[4 + r] + dyc : kOob;
        }
    };
    // patches of K-step k live in R[k & 1] / D[k & 1]: loaded during step k - 3 (slots 8..15), consumed during step k - 1
    float R[2][4][8], D[2][2][4], q[4][8], U[4][4];
    auto load_all = [&](auto b_) {
        constexpr int b = decltype(b_)::value;
        static_for<0, NX>([&](auto i_) { load_one<DM, decltype(i_)::value>(x_rsrc, vx, 0u, R[b]); });
        static_for<0, NY>([&](auto i_) { load_dy_one<DM, decltype(i_)::value>(y_rsrc, vy, D[b]); });
    };
    // vector-memory operation I of a step's NT: the next entry first (it must have landed when the NX + NY patch loads
    // behind it are still in flight), then x, then dy
    auto vmem_op = [&](auto i_, auto b_, int k_entry) {
        constexpr int I = decltype(i_)::value, b = decltype(b_)::value;
        if constexpr (I == 0) load_entry(k_entry);
        else if constexpr (I == 1) { /* second half of the entry: issued with the first */ }
        else if constexpr (I < 2 + NX) load_one<DM, I - 2>(x_rsrc, vx, 0u, R[b]);
        else load_dy_one<DM, I - 2 - NX>(y_rsrc, vy, D[b]);
    };
    const int pw = (cl >> 5) * GHALF + (j >> 1) * GKH + (cl & 31) * 4 + 2 * (j & 1);     // this thread's slot in a component's image
    auto produce_x = [&](auto xi_, int buf) {
        constexpr int XI = decltype(xi_)::value;
        const f32x2 o = col_transform<DM, XI>(q);
        *reinterpret_cast<f32x2*>(smem + buf * 2 * GSTAGE + GSTAGE + XI * GCOMP + pw) = o;
    };
    auto produce_y = [&](auto xi_, int buf) {
        constexpr int XI = decltype(xi_)::value;
        const f32x2 o = dy_component<DM, XI>(U);
        *reinterpret_cast<f32x2*>(smem + buf * 2 * GSTAGE + XI * GCOMP + pw) = o;
    };
    auto dy_rows = [&](auto g_, const float (&Dd)[2][4]) {        // row g of A g (unsigned), pinned into its slot
        constexpr int g = decltype(g_)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (g == 0) { U[0][c] = Dd[0][c]; U[3][c] = Dd[1][c]; }      // (plain copies: the register allocator's business)
            if constexpr (g == 1) { U[1][c] = Dd[0][c] + Dd[1][c]; asm volatile("" : "+v"(U[1][c])); }
            if constexpr (g == 2) { U[2][c] = Dd[0][c] - Dd[1][c]; asm volatile("" : "+v"(U[2][c])); }
        }
    };
    auto retire = [&]() {
        // lgkmcnt(0) only (0xC07F: vmcnt / expcnt left alone) - through the builtin, which hipcc's own wait-count pass
        // understands: behind an opaque asm wait it would wait for the NEXT step's first fragment reads again in front
        // of the first MFMA of slot 0
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    };

    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- prologue: K-steps 0, 1, 2 loaded, stage 0 = K-step 0 built, entry of K-step 3 in E
    constexpr std::integral_constant<int, 0> even{};
    constexpr std::integral_constant<int, 1> odd{};
    load_entry(0); make_offsets(); load_all(even);
    load_entry(1); make_offsets(); load_all(odd);
    row_transform<DM>(R[0], q);
    static_for<0, 4>([&](auto g_) { dy_rows(g_, D[0]); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 16>([&](auto xi_) { produce_y(xi_, 0); produce_x(xi_, 0); });
    load_entry(nk > 2 ? 2 : 1); make_offsets(); load_all(even);
    load_entry(nk > 3 ? 3 : nk - 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int a_lane = wm * GHALF + lhi * GKH + l31 * 4;
    const int b_lane = wn * GHALF + lhi * GKH + l31 * 4;
    // ---- K loop.  A K-step is 16 slots, one per component g: slot g reads the fragments of component g (consumed one
    // slot later) and issues the 4 MFMAs of component g - 1; slot 0 issues those of the previous step's component 15.
    // Beside the MFMAs a slot carries its share of building K-step kt + 1 into the other LDS stage: slots 0..3 one row of
    // B^T x and of A g each, every slot its component's two column transforms + LDS writes, slot 7 the buffer offsets of
    // K-step kt + 3 from the table entry loaded a step earlier, slots 8..15 the next entry and the patch loads of K-step
    // kt + 3 into the register set slots 0..3 have just released.
    f32x4 af[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 bf[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto step = [&](auto prod_, auto par_, int kt) {
        constexpr bool PROD = decltype(prod_)::value;
        constexpr int cur = decltype(par_)::value, rb = cur ^ 1;
        const float* As = smem + cur * 2 * GSTAGE + a_lane;
        const float* Bs = smem + cur * 2 * GSTAGE + GSTAGE + b_lane;
        const int k_entry = kt + 4 < nk ? kt + 4 : nk - 1;         // (the last loads are harmless repeats: no branch in the loop)
        static_for<0, 16>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            constexpr int fb = g & 1, pg = (g + 15) & 15;
            af[fb] = *reinterpret_cast<const f32x4*>(As + g * GCOMP);
            bf[fb] = *reinterpret_cast<const f32x4*>(Bs + g * GCOMP);
            // ONE burst of vector-ALU work per slot, in front of its MFMAs (tools/micro/mfma_shadow.hip: with one wave per SIMD
            // every VALU / SALU instruction costs ~4.5 cycles of matrix-pipe time and every switch MFMA -> VALU -> MFMA ~8 more;
            // LDS reads and writes cost nothing - unless a write sits right behind the instruction that produced its data,
            // ~20 cycles): the row transforms (slots 0..3), this component's two column transforms, and in slot 7 the buffer
            // offsets of K-step kt + 3.  The LDS writes follow two MFMAs later, the loads one MFMA after them.
            f32x2 ox = {0.f, 0.f}, oy = {0.f, 0.f};
            if constexpr (PROD) {
                if constexpr (g < 4) {
                    row_transform_one<DM, g>(R[rb], q);
                    // pin the row here: left alone, hipcc sinks these subtractions to the column transforms of later slots,
                    // which keeps R alive under the loads of slots 8..15 (conv_winograd2.hip)
#pragma unroll
                    for (int c = 0; c < PatchCfg<DM>::NCOL; ++c) asm volatile("" : "+v"(q[g][c]));
                    dy_rows(g_, D[rb]);
                }
                oy = dy_component<DM, g>(U);
                ox = col_transform<DM, g>(q);
                asm volatile("" : "+v"(ox), "+v"(oy));      // (computed here, not sunk to the stores)
                if constexpr (g == 7) make_offsets();
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][0], bf[fb ^ 1][0], acc[pg], 0, 0, 0);
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][1], bf[fb ^ 1][1], acc[pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROD) {
                *reinterpret_cast<f32x2*>(smem + rb * 2 * GSTAGE + g * GCOMP + pw) = oy;
                *reinterpret_cast<f32x2*>(smem + rb * 2 * GSTAGE + GSTAGE + g * GCOMP + pw) = ox;
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][2], bf[fb ^ 1][2], acc[pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROD && g >= 8) {
                constexpr int lo = (g - 8) * NT / 8, hi = (g - 7) * NT / 8;
                static_for<lo, hi>([&](auto i_) { vmem_op(i_, std::integral_constant<int, rb>{}, k_entry); });
            }
            acc[pg] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fb ^ 1][3], bf[fb ^ 1][3], acc[pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {          // nk is even: the producing steps 0 .. nk - 2 are pairs + one
        step(std::true_type{}, even, kt);
        retire();
        step(std::true_type{}, odd, kt + 1);
        retire();
    }
    step(std::true_type{}, even, nk - 2);
    retire();
    step(std::false_type{}, odd, nk - 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)        // component 15 of the last step
        acc[15] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][kk], bf[1][kk], acc[15], 0, 0, 0);

    // ---- epilogue: the slab of this split, [xi][m][c]; a lane holds rows 4 lhi + (r & 3) + 8 (r >> 2), column l31 of its wave's
    // 32 x 32 block of every component
    const long long mc = (long long)p.M * p.C;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (long long)split * 16 * mc, 0,
                                                                            (unsigned)(16 * mc * 4), 0x00020000);
    const int m0 = mb * 64 + wm * 32 + 4 * lhi, c0 = cb * 64 + wn * 32 + l31;
    const unsigned ov = (!RAGGED || c0 < p.C) ? (unsigned)(((long long)m0 * p.C + c0) * 4) : kOob;
    const unsigned C4 = (unsigned)p.C * 4u, mc4 = (unsigned)(mc * 4);
    static_for<0, 16>([&](auto xi_) {
        constexpr int XI = decltype(xi_)::value;
        constexpr bool neg = ((XI >> 2) == 3) != ((XI & 3) == 3);
        static_for<0, 16>([&](auto r_) {
            constexpr int r = decltype(r_)::value, mr = (r & 3) + 8 * (r >> 2);
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)XI * mc4 + (unsigned)mr * C4);
            const unsigned vo = (!RAGGED || m0 + mr < p.M) ? ov : kOob;
            const float v = neg ? -acc[XI][r] : acc[XI][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, vo, so, 0);
        });
    });
}

struct WgPlan {
    int TH, TW, mblocks, cblocks, splits, kchunk, mode, lead;
    long long T, T16, table_floats, slab_floats;
};

// patch-load mode of the fused kernels for these operands, or -1 (conv_winograd2.hip's fused_mode, plus the dy side)
int wg_mode(int N, int H, int W, int d, int M, int C, long long x_nstride, int x_pitch, long long dy_nstride, int dy_pitch) {
    const long long xspan = ((long long)(N - 1) * x_nstride + (long long)((C + 63) / 64 * 64) * H * x_pitch + 64) * 4;
    const long long yspan = ((long long)(N - 1) * dy_nstride + (long long)((M + 63) / 64 * 64) * H * dy_pitch + 64) * 4;
    if (xspan >= 0x7fffff00LL || yspan >= 0x7fffff00LL) return -1;      // 32-bit byte offsets, bit 31 = out of range
    if ((d == 1 || d == 2) && x_pitch >= W + 4 && W % 4 == 0) return d;
    if (d >= 4 && d % 2 == 0 && W % 2 == 0) return 4;
    return -1;
}

WgPlan wg_plan(int N, int H, int W, int d, int M, int C) {
    WgPlan pl;
    pl.TH = d * ((H + 2 * d - 1) / (2 * d));
    pl.TW = d * ((W + 2 * d - 1) / (2 * d));
    pl.TW = (pl.TW + 3) / 4 * 4;               // (conv_winograd.hip's tile grid: even, so a pair never straddles a tile row)
    pl.T = (long long)N * pl.TH * pl.TW;
    pl.T16 = (pl.T + 15) / 16 * 16;
    pl.mblocks = (M + 63) / 64;
    pl.cblocks = (C + 63) / 64;
    // split K so that blocks fill whole rounds of the CUs (one workgroup per CU), each at least 32 K-steps long
    const long long tiles = (long long)pl.mblocks * pl.cblocks, slots = dcfp_num_cus();
    long long max_splits = pl.T16 / (32 * GBK);
    if (max_splits < 1) max_splits = 1;
    long long splits = 1;
    double best = -1.0;
    for (long long sp = 1; sp <= max_splits && tiles * sp <= 2 * slots; ++sp) {
        const long long blocks = tiles * sp, rounds = (blocks + slots - 1) / slots;
        const double eff = (double)blocks / (double)(rounds * slots);
        if (eff > best + 0.02) { best = eff; splits = sp; }
    }
    long long kchunk = (pl.T16 + splits - 1) / splits;
    kchunk = (kchunk + 15) / 16 * 16;
    splits = (pl.T16 + kchunk - 1) / kchunk;
    pl.splits = (int)splits;
    pl.kchunk = (int)kchunk;
    pl.table_floats = (pl.T16 / 2 * 8 + 63) / 64 * 64;
    pl.slab_floats = (long long)pl.splits * 16 * M * C;
    return pl;
}

}  // namespace

// DCFP_WINO_WGRAD_FUSED: 0 off (the batched path of conv_winograd.hip with the kept transform), 1 on (default)
bool dcfp_wino_wgrad_fused_enabled() {
    static const int v = [] { const char* e = getenv("DCFP_WINO_WGRAD_FUSED"); return e ? atoi(e) : 1; }();
    return v != 0;
}

bool dcfp_wino_wgrad_fused_ok(int N, int H, int W, int d, int M, int C, long long x_nstride, int x_pitch,
                              long long dy_nstride, int dy_pitch) {
    if (!dcfp_wino_wgrad_fused_enabled()) return false;
    if (M < 48 || C < 48) return false;        // (the cost model of conv_wgrad.hip decides above that)
    if (x_pitch <= 0) x_pitch = W;
    if (dy_pitch <= 0) dy_pitch = W;
    if (wg_mode(N, H, W, d, M, C, x_nstride, x_pitch, dy_nstride, dy_pitch) < 0) return false;
    const WgPlan pl = wg_plan(N, H, W, d, M, C);
    if (pl.T16 >= (1LL << 30) || pl.T16 / 2 * 32 >= 0x7fffff00LL) return false;
    if (16LL * M * C * 4 >= 0x7fffff00LL) return false;
    return (long long)pl.mblocks * pl.cblocks * pl.splits < (1LL << 28);
}

size_t dcfp_wino_wgrad_fused_workspace_bytes(int N, int H, int W, int d, int M, int C) {
    const WgPlan pl = wg_plan(N, H, W, d, M, C);
    return (size_t)(pl.table_floats + pl.slab_floats) * sizeof(float);
}

int dcfp_wino_wgrad_fused_run(const float* dy, long long dy_nstride, int dy_pitch, const float* x, long long x_nstride,
                              int x_pitch, float* dw, int N, int M, int C, int H, int W, int d, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
    if (x_pitch <= 0) x_pitch = W;
    if (dy_pitch <= 0) dy_pitch = W;
    const int mode = wg_mode(N, H, W, d, M, C, x_nstride, x_pitch, dy_nstride, dy_pitch);
    if (mode < 0) return DCFP_E_UNSUPPORTED;
    const WgPlan pl = wg_plan(N, H, W, d, M, C);
    if (!workspace || !dcfp_aligned16(workspace) || workspace_bytes < dcfp_wino_wgrad_fused_workspace_bytes(N, H, W, d, M, C))
        return DCFP_E_WORKSPACE;
    unsigned* table = static_cast<unsigned*>(workspace);
    float* slabs = static_cast<float*>(workspace) + pl.table_floats;
    const int lead = mode == 4 ? 0 : 4;          // the pitched layout keeps x_pitch - W >= 4 readable zeros in front
    const long long npairs = pl.T16 / 2;
    hipLaunchKernelGGL(wino_wg_table_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, stream, table, npairs, pl.T,
                       N, H, W, d, pl.TH, pl.TW, x_nstride, x_pitch, lead, dy_nstride, dy_pitch, mode);
    WinoWgParams p;
    p.x_base = x - lead;
    p.x_bytes = (unsigned)(((long long)(N - 1) * x_nstride + (long long)C * H * x_pitch + lead) * 4);
    p.dy_base = dy;
    p.dy_bytes = (unsigned)(((long long)(N - 1) * dy_nstride + (long long)M * H * dy_pitch) * 4);
    p.table = table;
    p.table_bytes = (unsigned)(npairs * 32);
    p.out = slabs;
    p.M = M; p.C = C; p.mblocks = pl.mblocks; p.cblocks = pl.cblocks;
    p.nblk = pl.mblocks * pl.cblocks * pl.splits;
    p.x_chan_bytes = (unsigned)((long long)H * x_pitch * 4);
    p.dy_chan_bytes = (unsigned)((long long)H * dy_pitch * 4);
    p.kchunk = pl.kchunk;
    p.T16 = pl.T16;
    p.d = d;
    const unsigned grid = (unsigned)(((p.nblk + 7) / 8) * 8);
    const size_t lds = (size_t)4 * GSTAGE * sizeof(float);
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
        hipError_t e2 = hipGetLastError();
        return e2 == hipSuccess ? DCFP_OK : (int)e2;
    };
    const bool ragged = (M % 64) != 0 || (C % 64) != 0;
    int rc;
#define DCFP_WG(DM_) (ragged ? launch(wino_wgrad_fused_kernel<DM_, true>) : launch(wino_wgrad_fused_kernel<DM_, false>))
    switch (mode) {
        case 1: rc = DCFP_WG(1); break;
        case 2: rc = DCFP_WG(2); break;
        default: rc = DCFP_WG(4); break;
    }
#undef DCFP_WG
    if (rc) return rc;
    const long long mc = (long long)M * C;
    hipLaunchKernelGGL(wino_dw_reduce_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, stream, slabs, pl.splits, mc, dw);
    DCFP_RETURN_LAUNCH();
}

// conv_igemm.hip — conv2d forward and data-gradient as an implicit GEMM on the fp32
// matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 157.3 TFLOP/s).
// Replaces nn.Conv2d fwd/dgrad reached from networks/backbone/resnet.py:25-30,88-96,
// 110-114, networks/tools/aspp.py:13-14,57,63 and networks/deeplabv3.py:25-33,37-41.
//
// Per image:   Out[m][p] = sum_k A[m][k] * B[k][p]     k = (c, tap) flattened c-major
//   forward :  m = co, c = ci,  A = w[co][ci][tap],  B = x [ci][ (oh*s - pad + kh*d, ow*s - pad + kw*d) ]
//   dgrad   :  m = ci, c = co,  A = w[co][ci][tap],  B = dy[co][ ((h + pad - kh*d)/s, (w + pad - kw*d)/s) ]
// i.e. a 3x3 (dilated) conv is nine shifted 1x1 GEMM slabs accumulated in one MFMA
// accumulator; large dilations (12/24/36) make halo reuse worthless, so every slab is
// fetched with border predication (L2/MALL absorb the 9x re-read of the same rows).
//
// Tiling (NCHW, pixels contiguous): block = WM x WN waves, wave = TM x TN MFMA tiles of
// 32x32.  LDS holds A as [k][m] and B as [k][pixel]; lane j of a wave owns TM consecutive
// m and TN consecutive pixels, so ONE ds_read_b128 feeds four MFMA tiles and the
// epilogue stores 16 contiguous bytes per lane.  Global->LDS staging goes through
// registers, issued one K-step ahead of the MFMAs (double-buffered LDS, one barrier per
// K-step).  With the 4x4 wave tile a K-step is 128 MFMAs (8192 cycles/SIMD) against ~40
// staging instructions per lane, which is why plain predicated dword loads suffice.
#include "common.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmParams {
    const float* in;
    const float* wgt;
    const float* bias;
    float* out;
    long long in_nstride, out_nstride;
    int N, M, Ck, Ktot;
    int Hi, Wi, Ho, Wo, P, tiles_per_img, tiles_n_total, tiles_m;
    int sAm, sAc;
    int sn, sd, off0, offstep;
    int accumulate, vec_store;
    int wgt_bytes;
};

constexpr int BK = 16;
constexpr unsigned kOob = 0x80000000u;  // buffer offset beyond any record count: loads return 0

template <int T>
struct Frag;
template <>
struct Frag<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
};
template <>
struct Frag<2> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[2]) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        f[0] = v.x; f[1] = v.y;
    }
};
template <>
struct Frag<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[1]) { f[0] = p[0]; }
};

// SD: dgrad of a strided conv (source index = (o + off) / stride when divisible).
template <int TAPS, int TM, int TN, int WM, int WN, bool SD>
__global__ void __launch_bounds__(64 * WM * WN) igemm_kernel(const IgemmParams p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    constexpr int AG = NT / BM, KPA = BK / AG;      // A: thread -> (m, KPA consecutive k)
    constexpr int TX = BN / 4, TY = NT / TX, RPT = BK / TY;  // B: thread -> (4 pixels, RPT rows)
    static_assert(AG >= 1 && KPA >= 1 && AG * KPA == BK, "A loader shape");
    static_assert(TY >= 1 && RPT >= 1 && TY * RPT == BK, "B loader shape");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;    // [2][BK][BN]

    // ---- block -> (m tile, pixel tile); the tiles_m blocks of one pixel tile share blockIdx%8
    const int group = 8 * p.tiles_m;
    const int g = blockIdx.x / group, local = blockIdx.x - g * group;
    const int nt = g * 8 + (local & 7);
    const int mt = local >> 3;
    if (nt >= p.tiles_n_total) return;
    const int img = nt / p.tiles_per_img;
    const int p0 = (nt - img * p.tiles_per_img) * BN;
    const int m0 = mt * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- A loader state (buffer loads: an out-of-range offset returns 0, no branches)
    const int am = tid % BM, ak0 = (tid / BM) * KPA;
    const int a_m = m0 + am;
    const bool a_mok = a_m < p.M;
    const int a_moff = a_m * p.sAm;
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- B loader state: 4 consecutive output pixels
    const int tx = tid % TX, ty = tid / TX;
    int bh[4], bw[4];
    bool pv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int pp = p0 + 4 * tx + e;
        pv[e] = pp < p.P;
        const int oh = pp / p.Wo;
        const int ow = pp - oh * p.Wo;
        bh[e] = oh * p.sn;
        bw[e] = ow * p.sn;
    }
    const int HiWi = p.Hi * p.Wi;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in + (long long)img * p.in_nstride), 0, p.Ck * HiWi * 4, 0x00020000);

    float areg[KPA];
    float breg[RPT][4];

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int q = 0; q < KPA; ++q) {
            const int k = k0 + ak0 + q;
            const int c = k / TAPS;
            const int t = k - c * TAPS;
            const bool ok = a_mok && (k < p.Ktot);
            const unsigned off = ok ? (unsigned)(a_moff + c * p.sAc + t) * 4u : kOob;
            areg[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int k = k0 + ty + TY * q;
            const int c = k / TAPS;
            const int t = k - c * TAPS;
            const int kh = (TAPS == 9) ? t / 3 : 0;
            const int kw = (TAPS == 9) ? t - kh * 3 : 0;
            const int offh = p.off0 + kh * p.offstep;
            const int offw = p.off0 + kw * p.offstep;
            const bool kok = k < p.Ktot;
            const int coff = c * HiWi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int hh = bh[e] + offh, ww = bw[e] + offw;
                bool ok = kok && pv[e] && hh >= 0 && ww >= 0;
                if (SD) {  // only stride 2 exists on this path; generic divisor kept for safety
                    ok = ok && (hh % p.sd == 0) && (ww % p.sd == 0);
                    hh /= p.sd;
                    ww /= p.sd;
                }
                ok = ok && hh < p.Hi && ww < p.Wi;
                const unsigned off = ok ? (unsigned)(coff + hh * p.Wi + ww) * 4u : kOob;
                breg[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, off, 0, 0));
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * (BK * BM);
#pragma unroll
        for (int q = 0; q < KPA; ++q) a[(ak0 + q) * BM + am] = areg[q];
        float* b = Bs + buf * (BK * BN);
#pragma unroll
        for (int q = 0; q < RPT; ++q)
            *reinterpret_cast<float4*>(b + (ty + TY * q) * BN + 4 * tx) =
                make_float4(breg[q][0], breg[q][1], breg[q][2], breg[q][3]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.Ktot + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int a_off = wm * (TM * 32) + TM * l31;
    const int b_off = wn * (TN * 32) + TN * l31;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) load_tile((kt + 1) * BK);
        const float* a = As + cur * (BK * BM) + a_off;
        const float* b = Bs + cur * (BK * BN) + b_off;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = 2 * kk + lhi;
            float af[TM], bf[TN];
            Frag<TM>::ld(a + krow * BM, af);
            Frag<TN>::ld(b + krow * BN, bf);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds, per (tm, r), TN consecutive pixels of output row m
    float* o_img = p.out + (long long)img * p.out_nstride;
    const int pix = p0 + wn * (TN * 32) + TN * l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int m = m0 + wm * (TM * 32) + TM * row + i;
            if (m >= p.M) continue;
            float v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
            if (p.bias) {
                const float bsv = p.bias[m];
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] += bsv;
            }
            float* dst = o_img + (long long)m * p.P + pix;
            if (TN == 4 && p.vec_store && pix + 3 < p.P) {
                float4 o = make_float4(v[0], v[1], v[TN > 2 ? 2 : 0], v[TN > 3 ? 3 : 0]);
                if (p.accumulate) {
                    const float4 old = *reinterpret_cast<const float4*>(dst);
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *reinterpret_cast<float4*>(dst) = o;
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (pix + j < p.P) dst[j] = p.accumulate ? dst[j] + v[j] : v[j];
                }
            }
        }
    }
}

template <int TAPS, int TM, int TN, int WM, int WN, bool SD = false>
int launch_cfg(IgemmParams& p, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    p.tiles_per_img = (p.P + BN - 1) / BN;
    p.tiles_n_total = p.tiles_per_img * p.N;
    p.tiles_m = (p.M + BM - 1) / BM;
    const long long groups = ((long long)p.tiles_n_total + 7) / 8;
    const long long blocks = groups * 8 * p.tiles_m;
    if (blocks > 0x7fffffffLL) return DCFP_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * BK * (BM + BN) * sizeof(float);
    auto kern = igemm_kernel<TAPS, TM, TN, WM, WN, SD>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, stream, p);
    DCFP_RETURN_LAUNCH();
}

// Tile choice: the largest tile that still yields >= ~1 block per CU.
// 0: 32x512  1: 64x512  2: 128x256  3: 128x128  4: 256x256  5: 128x256 strided-dgrad
int pick_cfg(const IgemmParams& p) {
    if (p.sd > 1) return 5;
    const long long px = (long long)p.N * p.P;
    auto blocks = [&](int bm, int bn) {
        return ((long long)(p.M + bm - 1) / bm) * ((px + bn - 1) / bn);
    };
    if (p.M <= 32) return 0;
    if (p.M <= 64) return 1;
    if (p.M <= 128) return blocks(128, 256) >= 192 ? 2 : 3;
    if (blocks(256, 256) >= 192) return 4;
    if (blocks(128, 256) >= 192) return 2;
    return 3;
}

const char* cfg_args(int cfg) {
    switch (cfg) {
        case 0: return "1,4,1,4,0";
        case 1: return "2,4,1,4,0";
        case 2: return "2,4,2,2,0";
        case 3: return "2,2,2,2,0";
        case 4: return "4,4,2,2,0";
        default: return "2,4,2,2,1";
    }
}

template <int TAPS>
int launch_taps(IgemmParams& p, hipStream_t stream) {
    switch (pick_cfg(p)) {
        case 0: return launch_cfg<TAPS, 1, 4, 1, 4>(p, stream);
        case 1: return launch_cfg<TAPS, 2, 4, 1, 4>(p, stream);
        case 2: return launch_cfg<TAPS, 2, 4, 2, 2>(p, stream);
        case 3: return launch_cfg<TAPS, 2, 2, 2, 2>(p, stream);
        case 4: return launch_cfg<TAPS, 4, 4, 2, 2>(p, stream);
        default: return launch_cfg<TAPS, 2, 4, 2, 2, true>(p, stream);
    }
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
    // buffer descriptors address one image / the weight tensor with 31-bit byte offsets
    if ((long long)d->Cin * d->H * d->W >= (1LL << 29) ||
        (long long)d->Cout * d->Hout * d->Wout >= (1LL << 29) ||
        (long long)d->Cout * d->Cin * d->KH * d->KW >= (1LL << 29))
        return DCFP_E_UNSUPPORTED;
    return DCFP_OK;
}

}  // namespace

extern "C" int dcfp_abi_version(void) { return 1; }

// implemented in conv_wgrad.hip / conv_igemm2.hip
int dcfp_wgrad_kernel_name(const DcfpConvDesc* d, char* buf, int buf_len);
size_t dcfp_igemm2_workspace_bytes(int T, int M, int Ck, long long px, int sd);
const char* dcfp_igemm2_cfg_args(int M, long long px, int sd);
int dcfp_igemm2_run(const float* in, long long in_nstride, const float* w, int sAm, int sAc,
                    const float* bias, float* out, long long out_nstride, int N, int M, int Ck, int T,
                    int Hi, int Wi, int Ho, int Wo, int sn, int sd, int off0, int offstep,
                    int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream,
                    const float* scale = nullptr, const float* shift = nullptr,
                    const float* residual = nullptr, int relu = 0, float* stat_part = nullptr);
long long dcfp_igemm2_stat_slots(int M, int P, int N, long long out_nstride, const float* out);

// DCFP_IGEMM_V1=1 keeps the first-generation kernel (A/B comparisons in one process).
static bool use_v1() {
    static const bool v = getenv("DCFP_IGEMM_V1") != nullptr;
    return v;
}

// implemented in conv_igemm3.hip — EXPERIMENTAL opt-in (DCFP_CONV_MATH=bf16x3): 3-way bf16 split
size_t dcfp_igemm3_workspace_bytes(int T, int M, int Ck);
int dcfp_igemm3_run(const float* in, long long in_nstride, const float* w, int sAm, int sAc,
                    const float* bias, float* out, long long out_nstride, int N, int M, int Ck, int T,
                    int Hi, int Wi, int Ho, int Wo, int off0, int offstep, int accumulate,
                    void* workspace, size_t workspace_bytes, hipStream_t stream,
                    const float* scale = nullptr, const float* shift = nullptr,
                    const float* residual = nullptr, int relu = 0);
int dcfp_igemm2_cfg_id(int M, long long px, int sd);
static bool math_bf16x3() {
    static const bool v = [] { const char* e = getenv("DCFP_CONV_MATH"); return e && !strcmp(e, "bf16x3"); }();
    return v;
}
// the shapes the split kernel takes: unit sampling strides and the 256 x 256 tile of the fp32 path
static bool igemm3_ok(int M, long long px, int sn, int sd) {
    return math_bf16x3() && !use_v1() && sn == 1 && sd == 1 && dcfp_igemm2_cfg_id(M, px, sd) == 4;
}

extern "C" size_t dcfp_conv2d_fwd_dgrad_workspace_bytes_(const DcfpConvDesc* d, int pass) {
    if (check_desc(d) != DCFP_OK || use_v1()) return 0;
    const int T = d->KH * d->KW;
    size_t b2, b3 = 0;
    if (pass == DCFP_CONV_FWD) {
        const long long px = (long long)d->N * d->Hout * d->Wout;
        b2 = dcfp_igemm2_workspace_bytes(T, d->Cout, d->Cin, px, 1);
        if (igemm3_ok(d->Cout, px, d->stride, 1)) b3 = dcfp_igemm3_workspace_bytes(T, d->Cout, d->Cin);
    } else {
        const long long px = (long long)d->N * d->H * d->W;
        b2 = dcfp_igemm2_workspace_bytes(T, d->Cin, d->Cout, px, d->stride);
        if (igemm3_ok(d->Cin, px, 1, d->stride)) b3 = dcfp_igemm3_workspace_bytes(T, d->Cin, d->Cout);
    }
    return b2 > b3 ? b2 : b3;
}

extern "C" int dcfp_conv2d_kernel_name(const DcfpConvDesc* d, int pass, char* buf, int buf_len) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!buf || buf_len <= 0) return DCFP_E_BADDESC;
    if (pass == DCFP_CONV_WGRAD) return dcfp_wgrad_kernel_name(d, buf, buf_len);
    IgemmParams p;
    p.N = d->N;
    if (pass == DCFP_CONV_FWD) { p.M = d->Cout; p.P = d->Hout * d->Wout; p.sd = 1; }
    else { p.M = d->Cin; p.P = d->H * d->W; p.sd = d->stride; }
    if (igemm3_ok(p.M, (long long)p.N * p.P, pass == DCFP_CONV_FWD ? d->stride : 1, p.sd))
        return snprintf(buf, buf_len, "igemm3_kernel<%d>", d->KH * d->KW);
    if (!use_v1())
        return snprintf(buf, buf_len, "igemm2_kernel<%d,%s>", d->KH * d->KW,
                        dcfp_igemm2_cfg_args(p.M, (long long)p.N * p.P, p.sd));
    return snprintf(buf, buf_len, "igemm_kernel<%d,%s>", d->KH * d->KW, cfg_args(pick_cfg(p)));
}

extern "C" int dcfp_conv2d_fwd_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                                        const float* bias, float* y, int64_t y_nstride,
                                        void* workspace, size_t workspace_bytes,
                                        dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y) return DCFP_E_BADDESC;
    const int T = d->KH * d->KW;
    if (!bias && igemm3_ok(d->Cout, (long long)d->N * d->Hout * d->Wout, d->stride, 1))
        return dcfp_igemm3_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, bias, y,
                               y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N,
                               d->Cout, d->Cin, T, d->H, d->W, d->Hout, d->Wout, -d->pad, d->dil, 0,
                               workspace, workspace_bytes, dcfp_s(stream));
    if (!use_v1())
        return dcfp_igemm2_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, bias, y,
                               y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N,
                               d->Cout, d->Cin, T, d->H, d->W, d->Hout, d->Wout, d->stride, 1, -d->pad,
                               d->dil, 0, workspace, workspace_bytes, dcfp_s(stream));
    IgemmParams p;
    p.in = x; p.wgt = w; p.bias = bias; p.out = y;
    p.in_nstride = (long long)d->Cin * d->H * d->W;
    p.out_nstride = y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout;
    p.N = d->N; p.M = d->Cout; p.Ck = d->Cin; p.Ktot = d->Cin * T;
    p.Hi = d->H; p.Wi = d->W; p.Ho = d->Hout; p.Wo = d->Wout; p.P = d->Hout * d->Wout;
    p.sAm = d->Cin * T; p.sAc = T;
    p.sn = d->stride; p.sd = 1; p.off0 = -d->pad; p.offstep = d->dil;
    p.accumulate = 0;
    p.wgt_bytes = d->Cout * d->Cin * T * 4;
    p.vec_store = (p.P % 4 == 0) && (p.out_nstride % 4 == 0) && dcfp_aligned16(y);
    return T == 1 ? launch_taps<1>(p, dcfp_s(stream)) : launch_taps<9>(p, dcfp_s(stream));
}

// Forward conv that also emits BatchNorm batch-statistics partials of its output (see
// dcfp_bn_stats_from_partials_f32).  slots == 0: this shape / math mode has no fused statistics.
extern "C" int64_t dcfp_conv2d_fwd_stat_slots(const DcfpConvDesc* d, const float* y, int64_t y_nstride) {
    if (check_desc(d) != DCFP_OK || use_v1()) return 0;
    if (igemm3_ok(d->Cout, (long long)d->N * d->Hout * d->Wout, d->stride, 1)) return 0;
    return dcfp_igemm2_stat_slots(d->Cout, d->Hout * d->Wout, d->N,
                                  y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, y);
}

extern "C" int dcfp_conv2d_fwd_stats_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                                              float* y, int64_t y_nstride, float* stat_partials,
                                              void* workspace, size_t workspace_bytes,
                                              dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !stat_partials) return DCFP_E_BADDESC;
    if (dcfp_conv2d_fwd_stat_slots(d, y, y_nstride) <= 0) return DCFP_E_UNSUPPORTED;
    const int T = d->KH * d->KW;
    return dcfp_igemm2_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, nullptr, y,
                           y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N,
                           d->Cout, d->Cin, T, d->H, d->W, d->Hout, d->Wout, d->stride, 1, -d->pad,
                           d->dil, 0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr,
                           nullptr, 0, stat_partials);
}

extern "C" int dcfp_conv2d_dgrad_f32_nchw(const DcfpConvDesc* d, const float* dy,
                                          int64_t dy_nstride, const float* w, float* dx,
                                          int accumulate, void* workspace, size_t workspace_bytes,
                                          dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !w || !dx) return DCFP_E_BADDESC;
    const int T = d->KH * d->KW;
    if (igemm3_ok(d->Cin, (long long)d->N * d->H * d->W, 1, d->stride))
        return dcfp_igemm3_run(dy, dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * d->Wout, w,
                               T, d->Cin * T, nullptr, dx, (long long)d->Cin * d->H * d->W, d->N, d->Cin,
                               d->Cout, T, d->Hout, d->Wout, d->H, d->W, d->pad, -d->dil,
                               accumulate ? 1 : 0, workspace, workspace_bytes, dcfp_s(stream));
    if (!use_v1())
        return dcfp_igemm2_run(dy, dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * d->Wout, w,
                               T, d->Cin * T, nullptr, dx, (long long)d->Cin * d->H * d->W, d->N, d->Cin,
                               d->Cout, T, d->Hout, d->Wout, d->H, d->W, 1, d->stride, d->pad, -d->dil,
                               accumulate ? 1 : 0, workspace, workspace_bytes, dcfp_s(stream));
    IgemmParams p;
    p.in = dy; p.wgt = w; p.bias = nullptr; p.out = dx;
    p.in_nstride = dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * d->Wout;
    p.out_nstride = (long long)d->Cin * d->H * d->W;
    p.N = d->N; p.M = d->Cin; p.Ck = d->Cout; p.Ktot = d->Cout * T;
    p.Hi = d->Hout; p.Wi = d->Wout; p.Ho = d->H; p.Wo = d->W; p.P = d->H * d->W;
    p.sAm = T; p.sAc = d->Cin * T;
    p.sn = 1; p.sd = d->stride; p.off0 = d->pad; p.offstep = -d->dil;
    p.accumulate = accumulate ? 1 : 0;
    p.wgt_bytes = d->Cout * d->Cin * T * 4;
    p.vec_store = (p.P % 4 == 0) && dcfp_aligned16(dx);
    return T == 1 ? launch_taps<1>(p, dcfp_s(stream)) : launch_taps<9>(p, dcfp_s(stream));
}

// Inference: conv + folded eval-mode BatchNorm (+residual) (+ReLU) in the conv epilogue.
extern "C" int dcfp_conv2d_fwd_fused_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                                              const float* scale, const float* shift,
                                              const float* residual, int relu, float* y,
                                              void* workspace, size_t workspace_bytes,
                                              dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !scale || !shift) return DCFP_E_BADDESC;
    if (use_v1()) return DCFP_E_UNSUPPORTED;
    const int T = d->KH * d->KW;
    if (igemm3_ok(d->Cout, (long long)d->N * d->Hout * d->Wout, d->stride, 1))   // opt-in bf16x3 inference
        return dcfp_igemm3_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, nullptr, y,
                               (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout, d->Cin, T, d->H, d->W,
                               d->Hout, d->Wout, -d->pad, d->dil, 0, workspace, workspace_bytes,
                               dcfp_s(stream), scale, shift, residual, relu ? 1 : 0);
    return dcfp_igemm2_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, nullptr, y,
                           (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout, d->Cin, T, d->H, d->W,
                           d->Hout, d->Wout, d->stride, 1, -d->pad, d->dil, 0, workspace, workspace_bytes,
                           dcfp_s(stream), scale, shift, residual, relu ? 1 : 0);
}

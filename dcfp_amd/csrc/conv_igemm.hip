// conv_igemm.hip — C-ABI entry points of conv2d forward / data gradient (include/dcfp_hip.h) and
// their dispatch to the implicit-GEMM kernels.  Replaces nn.Conv2d fwd/dgrad reached from
// networks/backbone/resnet.py:25-30,88-96,110-114, networks/tools/aspp.py:13-14,57,63 and
// networks/deeplabv3.py:25-33,37-41.
//
// Per image:   Out[m][p] = sum_k A[m][k] * B[k][p]     k = (tap, c)
//   forward :  m = co, c = ci,  A = w[co][ci][tap],  B = x [ci][ (oh*s - pad + kh*d, ow*s - pad + kw*d) ]
//   dgrad   :  m = ci, c = co,  A = w[co][ci][tap],  B = dy[co][ ((h + pad - kh*d)/s, (w + pad - kw*d)/s) ]
// i.e. a 3x3 (dilated) conv is nine shifted 1x1 GEMM slabs accumulated in one MFMA accumulator;
// large dilations (12/24/36) make halo reuse worthless, so every slab is fetched with border
// predication (L2/MALL absorb the 9x re-read of the same rows).
//   conv_igemm2.hip  exact fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s) - the default
//   conv_igemm3.hip  opt-in DCFP_CONV_MATH=bf16x3: fp32 operands as three bf16 planes
#include "common.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

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
    // buffer descriptors address one image / the weight tensor with 31-bit byte offsets
    if ((long long)d->Cin * d->H * (d->x_pitch ? d->x_pitch : d->W) >= (1LL << 29) ||
        (long long)d->Cout * d->Hout * (d->dy_pitch ? d->dy_pitch : d->Wout) >= (1LL << 29) ||
        (long long)d->Cout * d->Cin * d->KH * d->KW >= (1LL << 29))
        return DCFP_E_UNSUPPORTED;
    return DCFP_OK;
}

}  // namespace

extern "C" int dcfp_abi_version(void) { return 2; }

// implemented in conv_wgrad.hip / conv_igemm2.hip
int dcfp_wgrad_kernel_name(const DcfpConvDesc* d, char* buf, int buf_len);
double dcfp_wgrad_exec_fraction(const DcfpConvDesc* d);
double dcfp_igemm2_exec_fraction(int T, int M, int Ck, int N, int Hi, int Wi, int Ho, int Wo, int sn, int sd,
                                 int off0, int offstep, bool pitched);
size_t dcfp_igemm2_workspace_bytes(int T, int M, int Ck, long long px, int sd);
const char* dcfp_igemm2_cfg_args(int M, long long px, int sd);
int dcfp_igemm2_run(const float* in, long long in_nstride, const float* w, int sAm, int sAc,
                    const float* bias, float* out, long long out_nstride, int N, int M, int Ck, int T,
                    int Hi, int Wi, int Ho, int Wo, int sn, int sd, int off0, int offstep,
                    int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream,
                    const float* scale = nullptr, const float* shift = nullptr,
                    const float* residual = nullptr, int relu = 0, float* stat_part = nullptr, int wp_valid = 0,
                    int in_pitch = 0, long long wp_nstride = 0, const float* fan_src = nullptr,
                    const unsigned long long* fan_mask = nullptr, const Igemm2Red* red = nullptr);
// conv_winograd.hip
bool dcfp_wino_ok(int N, int H, int W, int d, int M, int Ck);
size_t dcfp_wino_workspace_bytes(int N, int H, int W, int d, int M, int Ck);
double dcfp_wino_exec_fraction(int N, int H, int W, int d, int M, int Ck);
int dcfp_wino_run(const float* in, long long in_nstride, int in_pitch, const float* w, int sAm, int sAc, int flip,
                  float* out, long long out_nstride, int N, int M, int Ck, int H, int W, int d, int accumulate,
                  void* workspace, size_t workspace_bytes, hipStream_t stream, float* xform_out = nullptr,
                  float* stat_part = nullptr, const float* scale = nullptr, const float* shift = nullptr,
                  const float* residual = nullptr, int relu = 0, int wp_valid = 0);
long long dcfp_wino_stat_slots(int N, int H, int W, int d);
bool dcfp_stem_shape(const DcfpConvDesc* d);                                                            // conv_stem.hip
int dcfp_stem_fwd(const DcfpConvDesc* d, const float* x, const float* w, const float* bias, float* y, long long y_nstride,
                  hipStream_t stream, float* stat_part = nullptr);
long long dcfp_stem_stat_slots(const DcfpConvDesc* d, const float* y, long long y_nstride);
bool dcfp_gemv_shape(const DcfpConvDesc* d);                                                            // conv_gemv.hip
int dcfp_gemv_fwd(const DcfpConvDesc* d, const float* x, const float* w, const float* bias, float* y, long long y_nstride,
                  hipStream_t stream);
int dcfp_gemv_dgrad(const DcfpConvDesc* d, const float* dy, long long dy_nstride, const float* w, float* dx, int accumulate,
                    hipStream_t stream);
bool dcfp_wino_fused_ok(int N, int H, int W, int d, int M, int Ck, long long in_nstride, int pitch);   // conv_winograd2.hip
size_t dcfp_wino_fused_workspace_bytes(int N, int H, int W, int d, int M, int Ck);
void dcfp_wino_fused_pads(int N, int H, int W, int d, int M, int Ck, int* CkP, int* Mpad);
size_t dcfp_wino_xform_bytes(int N, int H, int W, int d, int C);
bool dcfp_wgrad_is_winograd(const DcfpConvDesc* d);      // conv_wgrad.hip
bool dcfp_wgrad_wants_xform(const DcfpConvDesc* d);
long long dcfp_igemm2_stat_slots(int M, int P, int N, long long out_nstride, const float* out);
bool dcfp_igemm2_dma_shape(int T, int M, int Ck, int P, long long px, int sn, int sd, int off0, int HiWi, int Wo);
bool dcfp_igemm2_persist();
void dcfp_igemm2_wp_layout(int T, int M, int Ck, int P, long long px, int sn, int sd, int off0, int offstep, int HiWi,
                           int Wo, bool pitched, int sAm, int sAc, DcfpWpEntry* e);
bool dcfp_igemm2_use_dma8(int T, int M, int P, long long px, int sn, int sd, int off0, int offstep, int HiWi,
                          int Wo, bool pitched);
int dcfp_igemm2_permute_multi(const DcfpWpEntry* table, int n, long long total_blocks, hipStream_t stream);


// implemented in conv_igemm3.hip — EXPERIMENTAL opt-in (DCFP_CONV_MATH=bf16x3): 3-way bf16 split
size_t dcfp_igemm3_workspace_bytes(int T, int M, int Ck);
int dcfp_igemm3_run(const float* in, long long in_nstride, const float* w, int sAm, int sAc,
                    const float* bias, float* out, long long out_nstride, int N, int M, int Ck, int T,
                    int Hi, int Wi, int Ho, int Wo, int off0, int offstep, int accumulate,
                    void* workspace, size_t workspace_bytes, hipStream_t stream,
                    const float* scale = nullptr, const float* shift = nullptr,
                    const float* residual = nullptr, int relu = 0, int wp_valid = 0);
int dcfp_igemm2_cfg_id(int M, long long px, int sd);
static bool math_bf16x3() {
    static const bool v = [] { const char* e = getenv("DCFP_CONV_MATH"); return e && !strcmp(e, "bf16x3"); }();
    return v;
}
// the shapes the split kernel takes: unit sampling strides and the 256 x 256 tile of the fp32 path
static bool igemm3_ok(int M, long long px, int sn, int sd) {
    return math_bf16x3() && sn == 1 && sd == 1 && dcfp_igemm2_cfg_id(M, px, sd) == 4;
}

// Winograd F(2x2, 3x3) (conv_winograd.hip) takes a wide 3x3 stride-1 pad == dil conv, forward or dgrad, where a
// cost model calibrated on this chip says it is faster than the direct LDS-DMA kernel (same-box measurements,
// profiles/r02_winograd_ab.txt): direct = nominal FLOPs x executed share (dead kernel rows) at 143 TF; Winograd =
// 16/36 of the nominal FLOPs x tile padding at 118 ... 135 TF (by GEMM shape) plus the two transform passes
// at 4.2 / 5.4 TB/s.  DCFP_CONV_WINOGRAD: 0 off, 1 model (default), 2 wherever eligible (A/B).
// which Winograd path a pass would take: 0 none (direct kernels), 1 the three passes of conv_winograd.hip, 2 the fused kernel
// of conv_winograd2.hip (the only one for fewer than 129 / 128 channels; needs a row-pitched operand for dilation 1 / 2)
static int wino_kind(const DcfpConvDesc* d, int pass) {
    static const int mode = [] { const char* e = getenv("DCFP_CONV_WINOGRAD"); return e ? atoi(e) : 1; }();
    if (mode == 0) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != d->dil || d->Hout != d->H || d->Wout != d->W)
        return 0;
    if (math_bf16x3()) return 0;
    const bool fwd = pass == DCFP_CONV_FWD;
    const int M = fwd ? d->Cout : d->Cin, Ck = fwd ? d->Cin : d->Cout;
    const int sp = fwd ? d->x_pitch : d->dy_pitch;
    const bool pitched = sp && sp != d->W;
    const bool three = dcfp_wino_ok(d->N, d->H, d->W, d->dil, M, Ck);
    const bool fused = dcfp_wino_fused_ok(d->N, d->H, d->W, d->dil, M, Ck, (long long)Ck * d->H * (sp ? sp : d->W), sp);
    if (!three && !fused) return 0;
    if (mode == 2) return three ? 1 : 2;
    const double nominal = 2.0 * d->N * (double)d->H * d->W * (double)M * Ck * 9.0;
    const double f_direct = fwd ? dcfp_igemm2_exec_fraction(9, M, Ck, d->N, d->H, d->W, d->H, d->W, 1, 1, -d->pad, d->dil, pitched)
                                : dcfp_igemm2_exec_fraction(9, M, Ck, d->N, d->H, d->W, d->H, d->W, 1, 1, d->pad, -d->dil, pitched);
    // (channel counts off the 256 grid - pruned models, the narrow layers - run the ragged-M / register-staged direct
    //  kernels: 100...127 TF measured)
    const double t_direct = nominal * f_direct / (M % 256 != 0 ? 115e12 : 143e12);
    const double f_wino = dcfp_wino_exec_fraction(d->N, d->H, d->W, d->dil, M, Ck);
    double t_wino = 1e30;
    if (three) {
        const double tiles = f_wino * 9.0 / 16.0 * d->N * (double)d->H * d->W;          // T
        // batched GEMM rate by shape class (measured): K = 256 is store-bound (118 TF at M = 256, 130 at M >= 1024),
        // deeper K runs at 127 TF with one row of M tiles and 135 TF with several
        const double rate = Ck <= 256 ? (M >= 1024 ? 130e12 : 118e12) : (M <= 256 ? 127e12 : 135e12);
        const double pix = (double)d->N * d->H * d->W;
        const double t_in = (4.0 * pix * Ck + 64.0 * tiles * Ck) / 4.2e12;
        const double t_out = (64.0 * tiles * M + 4.0 * pix * M * (fwd ? 1.0 : 2.0)) / 5.4e12;
        const double mpad = (double)((M + 255) / 256 * 256) / M;      // the GEMM's tiles are 256 (128) rows: ragged M pays for the padding
        t_wino = nominal * f_wino * mpad / rate + t_in + t_out + 20e-6;
    }
    if (fused) {
        // the fused kernel (what dcfp_wino_run takes wherever it applies, except where the forward must leave V behind):
        // 125 TF of executed MFMA work in its K loop since round 4 (one ALU burst per slot), about four K-steps' worth of
        // prologue + epilogue per block (measured executed rates, profiles/r04_wino_fused_burst_ab.txt: 87 TF at 64 input
        // channels, 110 at 256, 122 at 512, 118 at 2048 with 16 % tile padding); blocks are 64 output channels - off the 256
        // grid (narrow layers, pruned widths) it pads M to 64 instead of 256
        const double nk = (double)((Ck + 15) / 16 * 2);
        const double mpad = (double)((M + 63) / 64 * 64) / M;
        const double t_f = nominal * f_wino * mpad / (125e12 * nk / (nk + 4.0)) + 10e-6;
        if (t_f < t_wino) t_wino = t_f;
    }
    if (!(t_wino < 0.97 * t_direct)) return 0;
    return three ? 1 : 2;
}
static bool wino_pass(const DcfpConvDesc* d, int pass) { return wino_kind(d, pass) != 0; }

// A Winograd pass that ALWAYS runs the fused kernel (conv_winograd2.hip) needs nothing in its workspace but the transformed
// filters U (16/9 of the weights): like the permuted copies of the direct kernels they can be kept per conv, rebuilt for the
// whole model by the multi-tensor refresh after an optimizer step and passed back with wp_valid = 1 - 86 filter-transform
// launches of ~10 us per step otherwise (round 4).  Not the forward that leaves its transformed INPUT behind for a batched
// weight gradient with more than 256 input channels (three passes, all scratch).  DCFP_WINO_KEEP_U=0: off.
static bool wino_keeps_u(const DcfpConvDesc* d, int pass) {
    static const bool on = [] { const char* e = getenv("DCFP_WINO_KEEP_U"); return !e || atoi(e) != 0; }();
    if (!on || (pass != DCFP_CONV_FWD && pass != DCFP_CONV_DGRAD) || !wino_pass(d, pass)) return false;
    const bool fwd = pass == DCFP_CONV_FWD;
    const int M = fwd ? d->Cout : d->Cin, Ck = fwd ? d->Cin : d->Cout;
    const int sp = fwd ? d->x_pitch : d->dy_pitch;
    if (!dcfp_wino_fused_ok(d->N, d->H, d->W, d->dil, M, Ck, (long long)Ck * d->H * (sp ? sp : d->W), sp)) return false;
    if (fwd && Ck > 256 && dcfp_wino_ok(d->N, d->H, d->W, d->dil, M, Ck) && dcfp_conv2d_xform_bytes(d) > 0) return false;
    return true;
}

extern "C" int dcfp_conv2d_workspace_is_scratch(const DcfpConvDesc* d, int pass) {
    if (check_desc(d) != DCFP_OK || pass == DCFP_CONV_WGRAD) return 0;
    return (wino_pass(d, pass) && !wino_keeps_u(d, pass)) ? 1 : 0;
}

size_t dcfp_conv2d_fwd_dgrad_workspace_bytes_(const DcfpConvDesc* d, int pass) {
    if (check_desc(d) != DCFP_OK) return 0;
    if (const int wk = wino_kind(d, pass)) {
        const int M = pass == DCFP_CONV_FWD ? d->Cout : d->Cin, Ck = pass == DCFP_CONV_FWD ? d->Cin : d->Cout;
        if (wino_keeps_u(d, pass)) {
            // (a forward WITH bias of the same descriptor runs a direct kernel - dcfp_conv2d_fwd_f32_nchw - and builds its
            //  permuted copy in whatever workspace it is handed: large enough for either)
            const size_t u = dcfp_wino_fused_workspace_bytes(d->N, d->H, d->W, d->dil, M, Ck);
            const size_t direct = dcfp_igemm2_workspace_bytes(9, M, Ck, (long long)d->N * d->H * d->W, 1);
            return u > direct ? u : direct;
        }
        return wk == 1 ? dcfp_wino_workspace_bytes(d->N, d->H, d->W, d->dil, M, Ck)
                       : dcfp_wino_fused_workspace_bytes(d->N, d->H, d->W, d->dil, M, Ck);
    }
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
    if (dcfp_gemv_shape(d)) return snprintf(buf, buf_len, "gemv_1x1_map_kernel");
    if (pass == DCFP_CONV_WGRAD) return dcfp_wgrad_kernel_name(d, buf, buf_len);
    if (pass == DCFP_CONV_FWD && dcfp_stem_shape(d)) return snprintf(buf, buf_len, "stem_fwd_kernel");
    if (const int wk = wino_kind(d, pass))
        return wk == 1 ? snprintf(buf, buf_len, "winograd_f2x2_3x3 (igemm2_dma1p_kernel<false,true>)")
                       : snprintf(buf, buf_len, "winograd_f2x2_3x3 fused (wino_fused_kernel)");
    const int M = pass == DCFP_CONV_FWD ? d->Cout : d->Cin;
    const long long px = pass == DCFP_CONV_FWD ? (long long)d->N * d->Hout * d->Wout : (long long)d->N * d->H * d->W;
    const int sd = pass == DCFP_CONV_FWD ? 1 : d->stride;
    if (igemm3_ok(M, px, pass == DCFP_CONV_FWD ? d->stride : 1, sd))
        return snprintf(buf, buf_len, "igemm3_kernel<%d>", d->KH * d->KW);
    const int P = pass == DCFP_CONV_FWD ? d->Hout * d->Wout : d->H * d->W;
    const int HiWi = pass == DCFP_CONV_FWD ? d->H * d->W : d->Hout * d->Wout;
    const int Ck = pass == DCFP_CONV_FWD ? d->Cin : d->Cout;
    {
        const int sp = pass == DCFP_CONV_FWD ? d->x_pitch : d->dy_pitch;
        const bool pitched = sp && sp != (pass == DCFP_CONV_FWD ? d->W : d->Wout);
        if (dcfp_igemm2_use_dma8(d->KH * d->KW, M, P, px, pass == DCFP_CONV_FWD ? d->stride : 1, sd,
                                 pass == DCFP_CONV_FWD ? -d->pad : d->pad, pass == DCFP_CONV_FWD ? d->dil : -d->dil,
                                 HiWi, pass == DCFP_CONV_FWD ? d->Wout : d->W, pitched))
            return snprintf(buf, buf_len, "igemm2_dma8_kernel<%d>", d->KH * d->KW);
    }
    if (dcfp_igemm2_dma_shape(d->KH * d->KW, M, Ck, P, px, pass == DCFP_CONV_FWD ? d->stride : 1, sd, d->pad, HiWi,
                              pass == DCFP_CONV_FWD ? d->Wout : d->W)) {
        if (d->KH == 1 && dcfp_igemm2_persist()) return snprintf(buf, buf_len, "igemm2_dma1p_kernel");   // <ACC>
        const int src_pitch = pass == DCFP_CONV_FWD ? d->x_pitch : d->dy_pitch;
        const bool pitched = src_pitch && src_pitch != (pass == DCFP_CONV_FWD ? d->W : d->Wout);
        return snprintf(buf, buf_len, "igemm2_dma_kernel<%d,%s>", d->KH * d->KW,
                        (d->KH == 3 && ((d->pad | d->dil) & 3) != 0 && !pitched) ? "true" : "false");   // <TAPS, MIXED>
    }
    return snprintf(buf, buf_len, "igemm2_kernel<%d,%s>", d->KH * d->KW, dcfp_igemm2_cfg_args(M, px, sd));
}

extern "C" double dcfp_conv2d_executed_fraction(const DcfpConvDesc* d, int pass) {
    if (check_desc(d) != DCFP_OK) return 1.0;
    if (pass == DCFP_CONV_WGRAD) return dcfp_wgrad_exec_fraction(d);
    const bool fwd = pass == DCFP_CONV_FWD;
    if (wino_pass(d, pass))
        return fwd ? dcfp_wino_exec_fraction(d->N, d->H, d->W, d->dil, d->Cout, d->Cin)
                   : dcfp_wino_exec_fraction(d->N, d->H, d->W, d->dil, d->Cin, d->Cout);
    const int M = fwd ? d->Cout : d->Cin;
    const long long px = fwd ? (long long)d->N * d->Hout * d->Wout : (long long)d->N * d->H * d->W;
    if (igemm3_ok(M, px, fwd ? d->stride : 1, fwd ? 1 : d->stride)) return 1.0;
    const int sp = fwd ? d->x_pitch : d->dy_pitch;
    const bool pitched = sp && sp != (fwd ? d->W : d->Wout);
    return fwd ? dcfp_igemm2_exec_fraction(d->KH * d->KW, d->Cout, d->Cin, d->N, d->H, d->W, d->Hout, d->Wout, d->stride, 1,
                                           -d->pad, d->dil, pitched)
               : dcfp_igemm2_exec_fraction(d->KH * d->KW, d->Cin, d->Cout, d->N, d->Hout, d->Wout, d->H, d->W, 1, d->stride,
                                           d->pad, -d->dil, pitched);
}

extern "C" int64_t dcfp_conv2d_fwd_stat_slots(const DcfpConvDesc* d, const float* y, int64_t y_nstride);

// Bytes of the transformed input (Winograd V, conv_winograd.hip) that the forward pass of this conv can write into a
// caller-owned buffer (dcfp_conv2d_fwd_keep_f32_nchw) for its weight gradient to take over
// (dcfp_conv2d_wgrad_kept_f32_nchw) instead of transforming x again; 0 where either pass is not Winograd.
extern "C" size_t dcfp_conv2d_xform_bytes(const DcfpConvDesc* d) {
    if (check_desc(d) != DCFP_OK) return 0;
    if (!wino_pass(d, DCFP_CONV_FWD) || !dcfp_wgrad_wants_xform(d)) return 0;
    return dcfp_wino_xform_bytes(d->N, d->H, d->W, d->dil, d->Cin);
}

extern "C" int dcfp_conv2d_fwd_keep_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w, float* y,
                                             int64_t y_nstride, float* xform_out, size_t xform_bytes,
                                             float* stat_partials, void* workspace, size_t workspace_bytes,
                                             dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !xform_out) return DCFP_E_BADDESC;
    const size_t need = dcfp_conv2d_xform_bytes(d);
    if (need == 0) return DCFP_E_UNSUPPORTED;
    if (xform_bytes < need) return DCFP_E_WORKSPACE;
    const int T = 9;
    return dcfp_wino_run(x, (long long)d->Cin * d->H * (d->x_pitch ? d->x_pitch : d->W), d->x_pitch, w, d->Cin * T, T, 0, y,
                         y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout, d->Cin, d->H, d->W,
                         d->dil, 0, workspace, workspace_bytes, dcfp_s(stream), xform_out,
                         (stat_partials && dcfp_conv2d_fwd_stat_slots(d, y, y_nstride) > 0) ? stat_partials : nullptr);
}

extern "C" int dcfp_conv2d_fwd_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                                        const float* bias, float* y, int64_t y_nstride,
                                        void* workspace, size_t workspace_bytes, int wp_valid,
                                        dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y) return DCFP_E_BADDESC;
    const int T = d->KH * d->KW;
    if (d->x_pitch && d->x_pitch != d->W && math_bf16x3()) return DCFP_E_UNSUPPORTED;
    if (dcfp_gemv_shape(d)) return dcfp_gemv_fwd(d, x, w, bias, y, y_nstride, dcfp_s(stream));
    if (dcfp_stem_shape(d)) return dcfp_stem_fwd(d, x, w, bias, y, y_nstride, dcfp_s(stream));     // Cin = 3, stride 2: conv_stem.hip
    if (!bias && wino_pass(d, DCFP_CONV_FWD))
        return dcfp_wino_run(x, (long long)d->Cin * d->H * (d->x_pitch ? d->x_pitch : d->W), d->x_pitch, w, d->Cin * T, T,
                             0, y, y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout,
                             d->Cin, d->H, d->W, d->dil, 0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr,
                             nullptr, nullptr, nullptr, 0, (wp_valid && wino_keeps_u(d, DCFP_CONV_FWD)) ? 1 : 0);
    if (!bias && igemm3_ok(d->Cout, (long long)d->N * d->Hout * d->Wout, d->stride, 1))
        return dcfp_igemm3_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, bias, y,
                               y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N,
                               d->Cout, d->Cin, T, d->H, d->W, d->Hout, d->Wout, -d->pad, d->dil, 0,
                               workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr, nullptr, 0, wp_valid);
    return dcfp_igemm2_run(x, (long long)d->Cin * d->H * (d->x_pitch ? d->x_pitch : d->W), w, d->Cin * T, T, bias, y,
                           y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N,
                           d->Cout, d->Cin, T, d->H, d->W, d->Hout, d->Wout, d->stride, 1, -d->pad,
                           d->dil, 0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr, nullptr, 0,
                           nullptr, wp_valid, d->x_pitch);
}

// Forward conv that also emits BatchNorm batch-statistics partials of its output (see
// dcfp_bn_stats_from_partials_f32).  slots == 0: this shape / math mode has no fused statistics.
extern "C" int64_t dcfp_conv2d_fwd_stat_slots(const DcfpConvDesc* d, const float* y, int64_t y_nstride) {
    if (check_desc(d) != DCFP_OK) return 0;
    if (dcfp_stem_shape(d)) return dcfp_stem_stat_slots(d, y, y_nstride);
    if (wino_pass(d, DCFP_CONV_FWD))                // (the output transform emits them where every tile is interior)
        return (y_nstride % 4 == 0 && dcfp_aligned16(y)) ? dcfp_wino_stat_slots(d->N, d->H, d->W, d->dil) : 0;
    if (igemm3_ok(d->Cout, (long long)d->N * d->Hout * d->Wout, d->stride, 1)) return 0;
    return dcfp_igemm2_stat_slots(d->Cout, d->Hout * d->Wout, d->N,
                                  y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, y);
}

extern "C" int dcfp_conv2d_fwd_stats_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                                              float* y, int64_t y_nstride, float* stat_partials,
                                              void* workspace, size_t workspace_bytes, int wp_valid,
                                              dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !stat_partials) return DCFP_E_BADDESC;
    if (dcfp_conv2d_fwd_stat_slots(d, y, y_nstride) <= 0) return DCFP_E_UNSUPPORTED;
    const int T = d->KH * d->KW;
    if (dcfp_stem_shape(d)) return dcfp_stem_fwd(d, x, w, nullptr, y, y_nstride, dcfp_s(stream), stat_partials);
    if (wino_pass(d, DCFP_CONV_FWD))
        return dcfp_wino_run(x, (long long)d->Cin * d->H * (d->x_pitch ? d->x_pitch : d->W), d->x_pitch, w, d->Cin * T, T, 0, y,
                             y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout, d->Cin, d->H, d->W,
                             d->dil, 0, workspace, workspace_bytes, dcfp_s(stream), nullptr, stat_partials, nullptr, nullptr,
                             nullptr, 0, (wp_valid && wino_keeps_u(d, DCFP_CONV_FWD)) ? 1 : 0);
    return dcfp_igemm2_run(x, (long long)d->Cin * d->H * (d->x_pitch ? d->x_pitch : d->W), w, d->Cin * T, T, nullptr, y,
                           y_nstride ? y_nstride : (long long)d->Cout * d->Hout * d->Wout, d->N,
                           d->Cout, d->Cin, T, d->H, d->W, d->Hout, d->Wout, d->stride, 1, -d->pad,
                           d->dil, 0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr,
                           nullptr, 0, stat_partials, wp_valid, d->x_pitch);
}

extern "C" int dcfp_conv2d_dgrad_f32_nchw(const DcfpConvDesc* d, const float* dy,
                                          int64_t dy_nstride, const float* w, float* dx,
                                          int accumulate, void* workspace, size_t workspace_bytes,
                                          int wp_valid, dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !w || !dx) return DCFP_E_BADDESC;
    const int T = d->KH * d->KW;
    if (d->dy_pitch && d->dy_pitch != d->Wout && math_bf16x3()) return DCFP_E_UNSUPPORTED;
    if (dcfp_gemv_shape(d)) return dcfp_gemv_dgrad(d, dy, dy_nstride, w, dx, accumulate ? 1 : 0, dcfp_s(stream));
    if (wino_pass(d, DCFP_CONV_DGRAD))      // dx = conv(dy, w') with w'[ci][co] = w[co][ci] rotated by 180 degrees
        return dcfp_wino_run(dy, dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * (d->dy_pitch ? d->dy_pitch : d->Wout),
                             d->dy_pitch, w, T, d->Cin * T, 1, dx, (long long)d->Cin * d->H * d->W, d->N, d->Cin, d->Cout,
                             d->H, d->W, d->dil, accumulate ? 1 : 0, workspace, workspace_bytes, dcfp_s(stream), nullptr,
                             nullptr, nullptr, nullptr, nullptr, 0, (wp_valid && wino_keeps_u(d, DCFP_CONV_DGRAD)) ? 1 : 0);
    if (igemm3_ok(d->Cin, (long long)d->N * d->H * d->W, 1, d->stride))
        return dcfp_igemm3_run(dy, dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * d->Wout, w,
                               T, d->Cin * T, nullptr, dx, (long long)d->Cin * d->H * d->W, d->N, d->Cin,
                               d->Cout, T, d->Hout, d->Wout, d->H, d->W, d->pad, -d->dil,
                               accumulate ? 1 : 0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr,
                               nullptr, 0, wp_valid);
    return dcfp_igemm2_run(dy, dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * (d->dy_pitch ? d->dy_pitch : d->Wout), w,
                           T, d->Cin * T, nullptr, dx, (long long)d->Cin * d->H * d->W, d->N, d->Cin,
                           d->Cout, T, d->Hout, d->Wout, d->H, d->W, 1, d->stride, d->pad, -d->dil,
                           accumulate ? 1 : 0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr, nullptr, 0,
                           nullptr, wp_valid, d->dy_pitch);
}

// Gradient fan-in of a residual block (resnet.py:52-56 backward): dx = dgrad(dy) + fan_src * mask, where mask is the 1-bit
// ReLU mask of dcfp_bn_apply_relu_mask_f32 and fan_src the gradient that arrived at the block's output - the residual
// branch's gradient is never written.  1x1 stride-1 convs on the persistent LDS-DMA kernel, Cin % 256 == 0,
// H * W % 256 == 0 (dcfp_conv2d_dgrad_fanin_supported); fan_src has dx's layout.
extern "C" int dcfp_conv2d_dgrad_fanin_supported(const DcfpConvDesc* d) {
    if (check_desc(d) != DCFP_OK || d->KH != 1 || d->stride != 1 || d->pad != 0 || math_bf16x3()) return 0;
    const int P = d->H * d->W;
    const long long px = (long long)d->N * P;
    if (d->Cin % 256 != 0 || P % 256 != 0 || !dcfp_igemm2_persist()) return 0;
    if (igemm3_ok(d->Cin, px, 1, 1)) return 0;
    return dcfp_igemm2_dma_shape(1, d->Cin, d->Cout, P, px, 1, 1, 0, P, d->W) ? 1 : 0;
}

extern "C" int dcfp_conv2d_dgrad_fanin_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride, const float* w,
                                                float* dx, const float* fan_src, const void* fan_mask, void* workspace,
                                                size_t workspace_bytes, int wp_valid, dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !w || !dx || !fan_src || !fan_mask) return DCFP_E_BADDESC;
    if (!dcfp_conv2d_dgrad_fanin_supported(d)) return DCFP_E_UNSUPPORTED;
    return dcfp_igemm2_run(dy, dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * d->Wout, w, 1, d->Cin, nullptr, dx,
                           (long long)d->Cin * d->H * d->W, d->N, d->Cin, d->Cout, 1, d->Hout, d->Wout, d->H, d->W, 1, 1, 0, -1,
                           0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr, nullptr, 0, nullptr, wp_valid, 0, 0,
                           fan_src, static_cast<const unsigned long long*>(fan_mask));
}

// The fan-in with the PREVIOUS residual block's BatchNorm-backward sums as a side output: dx (this call's result) is the
// gradient arriving at that block's output; with its bn3 input `red_x`, its ReLU bit mask `red_mask` and batch mean
// `red_mean` the epilogue emits red_part[slot][Cin][2] = (sum g, sum g*(x - mean)) per 128 pixels, g = dx * mask -
// dcfp_bn_bwd_sums_from_partials_f32 turns them into what dcfp_bn_bwd_reduce_f32 returns, and that block's backward skips
// its reduce kernel (which would re-read dx and x: 8 B/element).  dcfp_conv2d_dgrad_fanin_red_slots: the slot count
// (N*H*W/128) where this form exists (fan-in supported and the launch takes 128-row tiles), else 0.
bool dcfp_igemm2p_fan_red_ok(int Mpad, int CkP);
int dcfp_igemm2_ck_pad();
extern "C" int64_t dcfp_conv2d_dgrad_fanin_red_slots(const DcfpConvDesc* d) {
    if (!dcfp_conv2d_dgrad_fanin_supported(d)) return 0;
    const int pad = dcfp_igemm2_ck_pad();
    const int ckp = (d->Cout + pad - 1) / pad * pad;
    if (!dcfp_igemm2p_fan_red_ok(d->Cin, ckp)) return 0;
    return (int64_t)d->N * d->H * d->W / 128;
}

extern "C" int dcfp_conv2d_dgrad_fanin_red_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride,
                                                    const float* w, float* dx, const float* fan_src, const void* fan_mask,
                                                    const float* red_x, const void* red_mask, const float* red_mean,
                                                    float* red_part, void* workspace, size_t workspace_bytes, int wp_valid,
                                                    dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !w || !dx || !fan_src || !fan_mask || !red_x || !red_mask || !red_mean || !red_part) return DCFP_E_BADDESC;
    if (dcfp_conv2d_dgrad_fanin_red_slots(d) <= 0) return DCFP_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(red_part) & 7u) return DCFP_E_BADDESC;
    const Igemm2Red red = {red_x, static_cast<const unsigned long long*>(red_mask), red_mean, red_part};
    return dcfp_igemm2_run(dy, dy_nstride ? dy_nstride : (long long)d->Cout * d->Hout * d->Wout, w, 1, d->Cin, nullptr, dx,
                           (long long)d->Cin * d->H * d->W, d->N, d->Cin, d->Cout, 1, d->Hout, d->Wout, d->H, d->W, 1, 1, 0, -1,
                           0, workspace, workspace_bytes, dcfp_s(stream), nullptr, nullptr, nullptr, 0, nullptr, wp_valid, 0, 0,
                           fan_src, static_cast<const unsigned long long*>(fan_mask), &red);
}

// Inference: conv + folded eval-mode BatchNorm (+residual) (+ReLU) in the conv epilogue.
extern "C" int dcfp_conv2d_fwd_fused_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                                              const float* scale, const float* shift,
                                              const float* residual, int relu, float* y,
                                              void* workspace, size_t workspace_bytes, int wp_valid,
                                              dcfp_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !scale || !shift) return DCFP_E_BADDESC;
    const int T = d->KH * d->KW;
    if (wino_pass(d, DCFP_CONV_FWD))      // the folded BatchNorm (+residual) (+ReLU) rides on the Winograd output transform
        return dcfp_wino_run(x, (long long)d->Cin * d->H * d->W, 0, w, d->Cin * T, T, 0, y,
                             (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout, d->Cin, d->H, d->W, d->dil, 0, workspace,
                             workspace_bytes, dcfp_s(stream), nullptr, nullptr, scale, shift, residual, relu ? 1 : 0);
    if (igemm3_ok(d->Cout, (long long)d->N * d->Hout * d->Wout, d->stride, 1))   // opt-in bf16x3 inference
        return dcfp_igemm3_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, nullptr, y,
                               (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout, d->Cin, T, d->H, d->W,
                               d->Hout, d->Wout, -d->pad, d->dil, 0, workspace, workspace_bytes,
                               dcfp_s(stream), scale, shift, residual, relu ? 1 : 0, wp_valid);
    return dcfp_igemm2_run(x, (long long)d->Cin * d->H * d->W, w, d->Cin * T, T, nullptr, y,
                           (long long)d->Cout * d->Hout * d->Wout, d->N, d->Cout, d->Cin, T, d->H, d->W,
                           d->Hout, d->Wout, d->stride, 1, -d->pad, d->dil, 0, workspace, workspace_bytes,
                           dcfp_s(stream), scale, shift, residual, relu ? 1 : 0, nullptr, wp_valid);
}

// Wp layout of (descriptor, pass) for a caller that refreshes the permuted copies of many convs in one
// launch after an optimizer step (dcfp_conv2d_permute_weights_multi_f32) and then passes wp_valid = 1.
extern "C" int dcfp_conv2d_wp_layout(const DcfpConvDesc* d, int pass, DcfpWpEntry* e) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!e || (pass != DCFP_CONV_FWD && pass != DCFP_CONV_DGRAD)) return DCFP_E_BADDESC;
    const int T = d->KH * d->KW;
    if (wino_pass(d, pass)) {
        if (!wino_keeps_u(d, pass)) return DCFP_E_UNSUPPORTED;            // all scratch: nothing to keep
        // the fused Winograd kernel's transformed filters: perm8 = 2 (forward) / 3 (dgrad: taps rotated by 180 degrees, the
        // roles of the channel strides swapped) - one (c, m) pair per element of the refresh kernel
        const bool fwd = pass == DCFP_CONV_FWD;
        e->T = 16; e->M = fwd ? d->Cout : d->Cin; e->Ck = fwd ? d->Cin : d->Cout;
        dcfp_wino_fused_pads(d->N, d->H, d->W, d->dil, e->M, e->Ck, &e->CkP, &e->Mpad);
        e->sAm = fwd ? d->Cin * T : T; e->sAc = fwd ? T : d->Cin * T; e->perm8 = fwd ? 2 : 3;
        const long long pairs = (long long)e->CkP * e->Mpad;
        e->n_blocks = (pairs + DCFP_WP_BLOCK_ELEMS - 1) / DCFP_WP_BLOCK_ELEMS;
        return DCFP_OK;
    }
    if (pass == DCFP_CONV_FWD) {
        const long long px = (long long)d->N * d->Hout * d->Wout;
        if (igemm3_ok(d->Cout, px, d->stride, 1)) return DCFP_E_UNSUPPORTED;     // bf16x3 keeps its own split copy
        dcfp_igemm2_wp_layout(T, d->Cout, d->Cin, d->Hout * d->Wout, px, d->stride, 1, -d->pad, d->dil, d->H * d->W,
                              d->Wout, d->x_pitch && d->x_pitch != d->W, d->Cin * T, T, e);
    } else {
        const long long px = (long long)d->N * d->H * d->W;
        if (igemm3_ok(d->Cin, px, 1, d->stride)) return DCFP_E_UNSUPPORTED;
        dcfp_igemm2_wp_layout(T, d->Cin, d->Cout, d->H * d->W, px, 1, d->stride, d->pad, -d->dil, d->Hout * d->Wout,
                              d->W, d->dy_pitch && d->dy_pitch != d->Wout, T, d->Cin * T, e);
    }
    const long long total = (long long)e->T * e->CkP * e->Mpad;
    e->n_blocks = (total + DCFP_WP_BLOCK_ELEMS - 1) / DCFP_WP_BLOCK_ELEMS;
    return DCFP_OK;
}

extern "C" int dcfp_conv2d_permute_weights_multi_f32(const DcfpWpEntry* table, int n, int64_t total_blocks,
                                                     dcfp_stream_t stream) {
    if (n == 0 || total_blocks == 0) return DCFP_OK;
    if (!table || n < 0 || total_blocks < 0 || total_blocks > 0x7fffffffLL) return DCFP_E_BADDESC;
    return dcfp_igemm2_permute_multi(table, n, total_blocks, dcfp_s(stream));
}

bool dcfp_wgrad_pitch_ok(const DcfpConvDesc* d);   // conv_wgrad.hip

extern "C" int dcfp_conv2d_pitch_supported(const DcfpConvDesc* d) {
    if (check_desc(d) != DCFP_OK || math_bf16x3()) return 0;
    if (d->KH != 3 || d->stride != 1 || d->pad != d->dil || d->Hout != d->H || d->Wout != d->W) return 0;
    const int xp = d->x_pitch ? d->x_pitch : d->W, dp = d->dy_pitch ? d->dy_pitch : d->Wout;
    if (xp < d->W + d->pad && xp != d->W) return 0;
    if (dp < d->Wout + d->pad && dp != d->Wout) return 0;
    const long long pxo = (long long)d->N * d->Hout * d->Wout, pxi = (long long)d->N * d->H * d->W;
    // forward and dgrad on the fused Winograd kernel (it is what makes the narrow layers Winograd at all - it reads the
    // shifted operand through 16-byte loads that rely on the zero tail), weight gradient on a kernel that takes the pitch
    if (xp != d->W && dp != d->Wout && wino_kind(d, DCFP_CONV_FWD) == 2 && wino_kind(d, DCFP_CONV_DGRAD) == 2)
        return dcfp_wgrad_pitch_ok(d) ? 1 : 0;
    // each of forward / dgrad on the 256 x 256 LDS-DMA kernel or on the ragged-M one (conv_igemm2n.hip)
    if (!dcfp_igemm2_dma_shape(9, d->Cout, d->Cin, d->Hout * d->Wout, pxo, 1, 1, -d->pad, d->H * d->W, d->Wout) &&
        !dcfp_igemm2_use_dma8(9, d->Cout, d->Hout * d->Wout, pxo, 1, 1, -d->pad, d->dil, d->H * d->W, d->Wout, true))
        return 0;
    if (!dcfp_igemm2_dma_shape(9, d->Cin, d->Cout, d->H * d->W, pxi, 1, 1, d->pad, d->Hout * d->Wout, d->W) &&
        !dcfp_igemm2_use_dma8(9, d->Cin, d->H * d->W, pxi, 1, 1, d->pad, -d->dil, d->Hout * d->Wout, d->W, true))
        return 0;
    return dcfp_wgrad_pitch_ok(d) ? 1 : 0;
}

// conv_winograd.hip — Winograd F(2x2, 3x3) for the wide 3x3 stride-1 convs (pad == dilation), forward and dgrad
// (networks/backbone/resnet.py:27-28 layer3/layer4 conv2, networks/tools/aspp.py:37-39, networks/deeplabv3.py:25-41).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 4x4 input patch d -> 2x2 outputs, 16 products instead of 36
//
// fp32 throughout: the multiplicands are fp32 sums / half-sums of fp32 values and go through the same
// v_mfma_f32_32x32x2_f32 as the direct kernels (this is an algebraic restructuring, not a narrower number format;
// cuDNN / MIOpen pick the same algorithm for fp32 3x3 convs).  Three memory-bound passes around ONE batched GEMM:
//   1. wino_input_kernel   x[N][C][H][W]            -> V[16][C][T]      (B^T d B, adds only)
//   2. igemm2_dma1p_kernel  M[xi] = U[xi] * V[xi]   for xi = 0..15: the persistent 1x1 LDS-DMA kernel of
//                           conv_igemm2p.hip with "16 images" whose weights differ per image (wp_nstride)
//   3. wino_output_kernel  M[16][K][T]              -> y[N][K][H][W]    (A^T m A, adds only; += for the dgrad fan-in)
// and wino_filter_kernel   w -> U[16][CkP][Mpad]    (G g G^T; rebuilt per call: <= 34 MB, ~10 us).
//
// Dilation d: the image is cut into super-blocks of 2d x 2d pixels; tile (a, b) of a super-block owns the outputs
// at rows {a, a + d} x columns {b, b + d} and reads rows a - d, a, a + d, a + 2d (same for columns) - the 3x3 conv
// with dilation d restricted to the residue class (a, b) mod d is an undilated conv.  Tile column index
// tcol = j * d + b, so that consecutive tiles read / write consecutive pixels (runs of d).
#include "igemm2_common.h"

int dcfp_igemm2_run(const float* in, long long in_nstride, const float* w, int sAm, int sAc,
                    const float* bias, float* out, long long out_nstride, int N, int M, int Ck, int T,
                    int Hi, int Wi, int Ho, int Wo, int sn, int sd, int off0, int offstep,
                    int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream,
                    const float* scale, const float* shift, const float* residual, int relu, float* stat_part,
                    int wp_valid, int in_pitch, long long wp_nstride, const float* fan_src = nullptr,
                    const unsigned long long* fan_mask = nullptr, const Igemm2Red* red = nullptr);
bool dcfp_igemm2_dma_shape(int T, int M, int Ck, int P, long long px, int sn, int sd, int off0, int HiWi, int Wo);
bool dcfp_igemm2_use_dma8(int T, int M, int P, long long px, int sn, int sd, int off0, int offstep, int HiWi,
                          int Wo, bool pitched);
bool dcfp_igemm2_persist();
int dcfp_igemm2_ck_pad();
// conv_winograd2.hip: the fused kernel (transforms inside the GEMM)
bool dcfp_wino_fused_ok(int N, int H, int W, int d, int M, int Ck, long long in_nstride, int pitch);
size_t dcfp_wino_fused_workspace_bytes(int N, int H, int W, int d, int M, int Ck);
int dcfp_wino_fused_run(const float* in, long long in_nstride, int in_pitch, const float* w, int sAm, int sAc, int flip,
                        float* out, long long out_nstride, int N, int M, int Ck, int H, int W, int d, int accumulate,
                        void* workspace, size_t workspace_bytes, hipStream_t stream, float* xform_out, float* stat_part,
                        const float* scale, const float* shift, const float* residual, int relu, int wp_valid);

namespace {

inline long long align64(long long v) { return (v + 63) / 64 * 64; }   // floats: 256-byte sections

struct WinoPlan {
    int TH, TW;           // tiles per image (rows, columns)
    long long T;          // tiles over the batch
    long long T16;        // ... rounded up to 16: row length of the transformed tensors = GEMM pixels per component (the
                          // tail is zeros), the same in the forward pass and the weight gradient, which can take over its V
    int CkP, Mpad;
    long long u_floats, v_floats, m_floats;
};

WinoPlan wino_plan(int N, int H, int W, int d, int M, int Ck) {
    WinoPlan pl;
    pl.TH = d * ((H + 2 * d - 1) / (2 * d));
    pl.TW = d * ((W + 2 * d - 1) / (2 * d));
    pl.TW = (pl.TW + 3) / 4 * 4;                  // GEMM rows of 16-byte quads (the extra tiles have no outputs)
    pl.T = (long long)N * pl.TH * pl.TW;
    pl.T16 = (pl.T + 15) / 16 * 16;
    pl.CkP = (Ck + dcfp_igemm2_ck_pad() - 1) / dcfp_igemm2_ck_pad() * dcfp_igemm2_ck_pad();
    pl.Mpad = (M + 255) / 256 * 256;
    pl.u_floats = align64(16LL * pl.CkP * pl.Mpad);
    pl.v_floats = align64(16LL * Ck * pl.T16);
    pl.m_floats = align64(16LL * M * pl.T16);
    return pl;
}

// U[xi][c][m] = (G g G^T)[xi],  g[kh][kw] = w[m * sAm + c * sAc + tap],  tap = kh*3+kw (flip: 8 - tap: dgrad)
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ w, int sAm, int sAc, int flip,
                                                          int M, int Ck, int CkP, int Mpad, float* __restrict__ U) {
    const long long plane = (long long)CkP * Mpad;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < plane; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx / Mpad), m = (int)(idx - (long long)c * Mpad);
        float g[9];
        if (m < M && c < Ck) {
            const float* src = w + (long long)m * sAm + (long long)c * sAc;
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t] = src[flip ? 8 - t : t];
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t] = 0.f;
        }
        // rows: G g  (4 x 3)
        float r[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float g0 = g[k], g1 = g[3 + k], g2 = g[6 + k];
            r[0][k] = g0;
            r[1][k] = 0.5f * ((g0 + g2) + g1);
            r[2][k] = 0.5f * ((g0 + g2) - g1);
            r[3][k] = g2;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = r[i][0], b = r[i][1], cc = r[i][2];
            U[(4 * i + 0) * plane + idx] = a;
            U[(4 * i + 1) * plane + idx] = 0.5f * ((a + cc) + b);
            U[(4 * i + 2) * plane + idx] = 0.5f * ((a + cc) - b);
            U[(4 * i + 3) * plane + idx] = cc;
        }
    }
}

// ---- transform kernels.  A thread handles VEC tiles that are neighbours in a tile row (tcol0 .. tcol0 + VEC - 1,
// VEC | d or d == 1 with VEC == 1): their pixels are VEC consecutive floats in every row they touch, so loads and
// stores are 4 / 8 / 16-byte vectors (VEC = 4 for the dilations 4 ... 36, 2 for dilation 2, 1 for dilation 1 and for
// shapes whose rows are not VEC-aligned).  Rows of the transformed tensors are `rowlen` floats: N * TH * TW tiles,
// then (weight-gradient use) a zero tail up to a multiple of 16.
template <int VEC>
struct VecT;
template <>
struct VecT<1> { typedef float type; };
template <>
struct VecT<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <>
struct VecT<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 1) v[0] = *p;
    else {
        const typename VecT<VEC>::type t = *reinterpret_cast<const typename VecT<VEC>::type*>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = t[i];
    }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 1) *p = v[0];
    else {
        typename VecT<VEC>::type t;
#pragma unroll
        for (int i = 0; i < VEC; ++i) t[i] = v[i];
        *reinterpret_cast<typename VecT<VEC>::type*>(p) = t;
    }
}

// V[xi][c][n * TH * TW + t] = (B^T d B)[xi] of tile t of image n, channel c.  grid (ceil(rowlen / VEC / 256), 1, C).
// (A version that staged the strip's four input rows in LDS with coalesced loads measured 4...19 % slower per conv.)
template <int VEC>
__global__ void __launch_bounds__(256) wino_input_kernel(const float* __restrict__ x, long long x_nstride, int pitch,
                                                         int N, int C, int H, int W, int d, int TH, int TW,
                                                         float* __restrict__ V, long long rowlen) {
    const int tpi = TH * TW;
    const long long tg = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (tg >= rowlen) return;
    const int c = blockIdx.z;
    const int n = (int)(tg / tpi), t = (int)(tg - (long long)n * tpi);
    const long long plane = (long long)C * rowlen;
    float* dst = V + (long long)c * rowlen + tg;
    float q[4][4][VEC];
    if (n >= N) {
        float z[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) vstore<VEC>(dst + k * plane, z);
        return;
    }
    const int trow = t / TW, tcol = t - trow * TW;
    const int h0 = trow + d * (trow / d) - d, w0 = tcol + d * (tcol / d) - d;
    const float* src = x + (long long)n * x_nstride + (long long)c * H * pitch;
    float p[4][4][VEC];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = h0 + r * d;
        const bool hok = (unsigned)h < (unsigned)H;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ww = w0 + s * d;
            if (hok && ww >= 0 && ww + VEC <= W) vload<VEC>(src + (long long)h * pitch + ww, p[r][s]);
            else {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    p[r][s][v] = (hok && (unsigned)(ww + v) < (unsigned)W) ? src[(long long)h * pitch + ww + v] : 0.f;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {        // B^T p: rows
            q[0][s][v] = p[0][s][v] - p[2][s][v];
            q[1][s][v] = p[1][s][v] + p[2][s][v];
            q[2][s][v] = p[2][s][v] - p[1][s][v];
            q[3][s][v] = p[1][s][v] - p[3][s][v];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float o[4][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            o[0][v] = q[r][0][v] - q[r][2][v];
            o[1][v] = q[r][1][v] + q[r][2][v];
            o[2][v] = q[r][2][v] - q[r][1][v];
            o[3][v] = q[r][1][v] - q[r][3][v];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) vstore<VEC>(dst + (4 * r + s) * plane, o[s]);
    }
}

// Weight gradient, dy side: Y[xi][m][t] = (A dy_t A^T)[xi] of the 2x2 output-gradient tile t (zeros outside the image and
// in the rows' tail), A = [1 0; 1 1; 1 -1; 0 -1].  grid (ceil(rowlen / VEC / 256), 1, M)
template <int VEC>
__global__ void __launch_bounds__(256) wino_dy_kernel(const float* __restrict__ dy, long long dy_nstride, int pitch, int N,
                                                      int M, int H, int W, int d, int TH, int TW, float* __restrict__ Y,
                                                      long long rowlen) {
    const int tpi = TH * TW;
    const long long tg = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (tg >= rowlen) return;
    const int m = blockIdx.z;
    const int n = (int)(tg / tpi), t = (int)(tg - (long long)n * tpi);
    const long long plane = (long long)M * rowlen;
    float* dst = Y + (long long)m * rowlen + tg;
    float g[2][2][VEC];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int v = 0; v < VEC; ++v) g[r][s][v] = 0.f;
    if (n < N) {
        const int trow = t / TW, tcol = t - trow * TW;
        const int ho = trow + d * (trow / d), wo = tcol + d * (tcol / d);
        const float* src = dy + (long long)n * dy_nstride + (long long)m * H * pitch;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int h = ho + r * d, ww = wo + s * d;
                if (h < H && ww + VEC <= W) vload<VEC>(src + (long long)h * pitch + ww, g[r][s]);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (h < H && ww + v < W) g[r][s][v] = src[(long long)h * pitch + ww + v];
                }
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float o[4][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            // row r of A g (4 x 2), then its four columns of (A g) A^T
            const float u0 = r == 0 ? g[0][0][v] : r == 1 ? g[0][0][v] + g[1][0][v] : r == 2 ? g[0][0][v] - g[1][0][v] : -g[1][0][v];
            const float u1 = r == 0 ? g[0][1][v] : r == 1 ? g[0][1][v] + g[1][1][v] : r == 2 ? g[0][1][v] - g[1][1][v] : -g[1][1][v];
            o[0][v] = u0; o[1][v] = u0 + u1; o[2][v] = u0 - u1; o[3][v] = -u1;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) vstore<VEC>(dst + (4 * r + s) * plane, o[s]);
    }
}

// y[n][k][2x2 outputs of tile t] (+)= A^T m A,  m[xi] = Mb[xi][k][n * TH * TW + t].  grid (ceil(TH*TW / VEC / 256), N, K)
// STATS: also the BatchNorm statistics of y as partials (mean, M2) over 128 outputs each - the 32 tiles of 32 / VEC
// neighbouring lanes - in the layout dcfp_bn_stats_from_partials_f32 merges (part[slot][k], slot = tile / 32); only
// launched where every tile lies wholly inside the image (H, W multiples of 2 d) and tiles per image % 32 == 0.
template <int VEC, bool STATS = false>
__global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ Mb, long long T, int K,
                                                          float* __restrict__ y, long long y_nstride, int H, int W,
                                                          int d, int TH, int TW, int accumulate,
                                                          float* __restrict__ stat_part,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ residual, int relu) {
    const int tpi = TH * TW;
    const int t = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (t >= tpi) return;
    const int n = blockIdx.y, k = blockIdx.z;
    const int trow = t / TW, tcol = t - trow * TW;
    const int ho = trow + d * (trow / d), wo = tcol + d * (tcol / d);
    if (ho >= H || wo >= W) return;
    const long long plane = (long long)K * T;
    const float* src = Mb + (long long)k * T + (long long)n * tpi + t;
    float m[4][4][VEC];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) vload<VEC>(src + (4 * r + s) * plane, m[r][s]);
    float* dst = y + (long long)n * y_nstride + (long long)k * H * W;
    // inference: eval-mode BatchNorm folded into the epilogue, y = act(conv * scale[k] + shift[k] (+ residual))
    const float sc = scale ? scale[k] : 1.f, sf = scale ? shift[k] : 0.f;
    const float* rsd = residual ? residual + (long long)n * y_nstride + (long long)k * H * W : nullptr;
    float keep_o[2][2][VEC];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int h = ho + r * d;
        if (h >= H) continue;
        // A^T m: rows   A^T = [1 1 1 0; 0 1 -1 -1], then columns
        float u[4][VEC], o[2][VEC];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                u[s][v] = r == 0 ? (m[0][s][v] + m[1][s][v]) + m[2][s][v] : (m[1][s][v] - m[2][s][v]) - m[3][s][v];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            o[0][v] = (u[0][v] + u[1][v]) + u[2][v];
            o[1][v] = (u[1][v] - u[2][v]) - u[3][v];
            if constexpr (STATS) { keep_o[r][0][v] = o[0][v]; keep_o[r][1][v] = o[1][v]; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ww = wo + s * d;
            float* e = dst + (long long)h * W + ww;
            if (scale) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float t = fmaf(o[s][v], sc, sf);
                    if (rsd && ww + v < W) t += rsd[(long long)h * W + ww + v];
                    o[s][v] = relu ? (t > 0.f ? t : 0.f) : t;
                }
            }
            if (ww + VEC <= W) {
                if (accumulate) {
                    float old[VEC];
                    vload<VEC>(e, old);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[s][v] += old[v];
                }
                vstore<VEC>(e, o[s]);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (ww + v < W) e[v] = accumulate ? e[v] + o[s][v] : o[s][v];
            }
        }
    }
    if constexpr (STATS) {
        constexpr int G = 32 / VEC;                 // lanes per partial (32 tiles, 128 outputs)
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int v = 0; v < VEC; ++v) sum += keep_o[r][s2][v];
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum * (1.0f / 128.0f);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int v = 0; v < VEC; ++v) { const float dd = keep_o[r][s2][v] - mean; m2 += dd * dd; }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) m2 += __shfl_xor(m2, off, 64);
        if ((threadIdx.x & (G - 1)) == 0) {
            const long long slot = ((long long)n * tpi + t) >> 5;
            float* sp = stat_part + (slot * K + k) * 2;
            sp[0] = mean; sp[1] = m2;
        }
    }
}

// dw[m][c][3][3] = G^T dU G,  dU[xi][m][c]
__global__ void __launch_bounds__(256) wino_dw_kernel(const float* __restrict__ dU, long long mc, float* __restrict__ dw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mc) return;
    float t[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float u0 = dU[(0 + s) * mc + i], u1 = dU[(4 + s) * mc + i], u2 = dU[(8 + s) * mc + i], u3 = dU[(12 + s) * mc + i];
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

// y[n][k][2x2 outputs of tile t] (+)= A^T m A,  m[xi] = Mb[xi][k][n * TH * TW + t].  grid (ceil(TH*TW / 256), N, K)
__global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ Mb, long long T, int K,
                                                          float* __restrict__ y, long long y_nstride, int H, int W,
                                                          int d, int TH, int TW, int accumulate) {
    const int tpi = TH * TW;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tpi) return;
    const int n = blockIdx.y, k = blockIdx.z;
    const int trow = t / TW, tcol = t - trow * TW;
    const int ho = trow + d * (trow / d), wo = tcol + d * (tcol / d);
    if (ho >= H || wo >= W) return;
    const long long plane = (long long)K * T;
    const float* src = Mb + (long long)k * T + (long long)n * tpi + t;
    float m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) m[r][s] = src[(4 * r + s) * plane];
    // A^T m: rows   A^T = [1 1 1 0; 0 1 -1 -1]
    float u[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u[0][s] = (m[0][s] + m[1][s]) + m[2][s];
        u[1][s] = (m[1][s] - m[2][s]) - m[3][s];
    }
    float o[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        o[r][0] = (u[r][0] + u[r][1]) + u[r][2];
        o[r][1] = (u[r][1] - u[r][2]) - u[r][3];
    }
    float* dst = y + (long long)n * y_nstride + (long long)k * H * W;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int h = ho + r * d;
        if (h >= H) continue;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ww = wo + s * d;
            if (ww >= W) continue;
            float* e = dst + (long long)h * W + ww;
            *e = accumulate ? *e + o[r][s] : o[r][s];
        }
    }
}

}  // namespace

// tiles per thread of the transform kernels: the widest vector the dilation, the row length and the tensor's alignment allow
static int wino_vec(int d, int W, int pitch, const float* base, long long nstride, int TW) {
    static const int cap = [] { const char* e = getenv("DCFP_WINO_VEC"); return e ? atoi(e) : 4; }();   // 1 / 2 / 4 (A/B)
    for (int v = 4; v > 1; v >>= 1)
        if (v <= cap && d % v == 0 && W % v == 0 && pitch % v == 0 && nstride % v == 0 && TW % v == 0 &&
            (reinterpret_cast<uintptr_t>(base) & (4 * v - 1)) == 0)
            return v;
    return 1;
}

// 3x3, stride 1, pad == dil, same-size output; M = output channels of the pass, Ck = reduced channels
bool dcfp_wino_ok(int N, int H, int W, int d, int M, int Ck) {
    if (!dcfp_igemm2_persist()) return false;
    if (M < 129 || Ck < 128) return false;     // the transforms cost ~ 1/M + 1/Ck of the GEMM (the cost model decides)
    const WinoPlan pl = wino_plan(N, H, W, d, M, Ck);
    if (4 * pl.T > (long long)N * H * W * 27 / 20) return false;  // > 35 % padding of the 2d x 2d super-blocks
    if (pl.T16 >= (1LL << 26) || 16LL * pl.T16 >= (1LL << 30)) return false;
    if ((long long)Ck * pl.T16 >= (1LL << 29) || (long long)M * pl.T16 >= (1LL << 29)) return false;
    const int P = (int)pl.T16;
    // (ragged M - pruned models - runs on the persistent kernel's edge tiles: the batched GEMM never takes the
    //  ragged-M kernel, whose permuted weight layout has no per-image copies)
    return dcfp_igemm2_dma_shape(1, M, Ck, P, 16LL * P, 1, 1, 0, P, P);
}

size_t dcfp_wino_workspace_bytes(int N, int H, int W, int d, int M, int Ck) {
    const WinoPlan pl = wino_plan(N, H, W, d, M, Ck);
    return (size_t)(pl.u_floats + pl.v_floats + pl.m_floats) * sizeof(float);
}

// share of the nominal multiply-adds (2*N*M*H*W*Ck*9) that the batched GEMM issues
double dcfp_wino_exec_fraction(int N, int H, int W, int d, int M, int Ck) {
    const WinoPlan pl = wino_plan(N, H, W, d, M, Ck);
    return 16.0 * (double)pl.T / (9.0 * (double)N * H * W);
}

// in: x (forward) or dy (dgrad), rows at `in_pitch` floats; w with strides (sAm, sAc) as dcfp_igemm2_run takes them
// Partials (of 128 outputs each) per channel that the output transform can emit for the BatchNorm behind the conv; 0 where
// some tile is not wholly inside the image
long long dcfp_wino_stat_slots(int N, int H, int W, int d) {
    if (H % (2 * d) != 0 || W % (2 * d) != 0 || (W / 2) % 4 != 0) return 0;
    const long long tpi = (long long)(H / 2) * (W / 2);
    if (tpi % 32 != 0) return 0;
    return (long long)N * tpi / 32;
}

// xform_out (nullable): 16 * Ck * T16 floats of the caller's that receive the transformed input V instead of the scratch
// (the weight gradient of the same conv can take it over: dcfp_wino_wgrad_run's xform_in)
int dcfp_wino_run(const float* in, long long in_nstride, int in_pitch, const float* w, int sAm, int sAc, int flip,
                  float* out, long long out_nstride, int N, int M, int Ck, int H, int W, int d, int accumulate,
                  void* workspace, size_t workspace_bytes, hipStream_t stream, float* xform_out, float* stat_part,
                  const float* scale, const float* shift, const float* residual, int relu, int wp_valid) {
    const WinoPlan pl = wino_plan(N, H, W, d, M, Ck);
    if (!workspace || !dcfp_aligned16(workspace)) return DCFP_E_WORKSPACE;
    // The fused kernel (conv_winograd2.hip) where it measures faster than the three passes (same-box A/B,
    // profiles/r03_wino_fused_ab.txt, profiles/r04_wino_fused_burst_ab.txt): everywhere except a forward that must leave V
    // behind for a BATCHED weight gradient with more than 256 input channels - the three-pass path has V anyway, the fused
    // kernel writes it on the side at HBM speed (1...5 GB).  (Round 3 also kept accumulating dgrads with >= 1024 output
    // channels - the ASPP branches - on the three passes: a wash then, 0.3 ms per launch for the fused kernel since its
    // K loop got faster in round 4.)  Where the three passes do not apply at all (fewer than 129 / 128 channels: the narrow
    // layers of the stem, layer1, layer2 and of pruned models) the fused kernel is the Winograd path.
    // DCFP_WINO_FUSED=2 takes it wherever it applies (tests).
    static const int fused_mode = [] { const char* e = getenv("DCFP_WINO_FUSED"); return e ? atoi(e) : 1; }();
    const bool three_ok = dcfp_wino_ok(N, H, W, d, M, Ck);
    const bool fused_wins = fused_mode == 2 || !three_ok || !(xform_out && Ck > 256);
    if (fused_wins && dcfp_wino_fused_ok(N, H, W, d, M, Ck, in_nstride, in_pitch) &&
        workspace_bytes >= dcfp_wino_fused_workspace_bytes(N, H, W, d, M, Ck)) {
        if (stat_part && (!dcfp_wino_stat_slots(N, H, W, d) || accumulate)) return DCFP_E_UNSUPPORTED;
        return dcfp_wino_fused_run(in, in_nstride, in_pitch, w, sAm, sAc, flip, out, out_nstride, N, M, Ck, H, W, d,
                                   accumulate, workspace, workspace_bytes, stream, xform_out, stat_part, scale, shift,
                                   residual, relu, wp_valid);
    }
    if (!three_ok) return DCFP_E_UNSUPPORTED;
    if (workspace_bytes < dcfp_wino_workspace_bytes(N, H, W, d, M, Ck)) return DCFP_E_WORKSPACE;
    float* U = static_cast<float*>(workspace);
    float* V = xform_out ? xform_out : U + pl.u_floats;
    float* Mb = U + pl.u_floats + pl.v_floats;
    if (xform_out && !dcfp_aligned16(xform_out)) return DCFP_E_BADDESC;
    {
        long long b = ((long long)pl.CkP * pl.Mpad + 255) / 256;
        if (b > 4096) b = 4096;
        hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)b), dim3(256), 0, stream, w, sAm, sAc, flip, M, Ck,
                           pl.CkP, pl.Mpad, U);
    }
    const int tpi = pl.TH * pl.TW;
    if (Ck > 65535 || M > 65535 || N > 65535) return DCFP_E_UNSUPPORTED;
    const int vec = wino_vec(d, W, in_pitch > 0 ? in_pitch : W, in, in_nstride, pl.TW);
    const unsigned gin = (unsigned)((pl.T16 / vec + 255) / 256);
#define DCFP_WINO_IN(VEC_) hipLaunchKernelGGL(wino_input_kernel<VEC_>, dim3(gin, 1, (unsigned)Ck), dim3(256), 0, stream, in, \
                                              in_nstride, in_pitch > 0 ? in_pitch : W, N, Ck, H, W, d, pl.TH, pl.TW, V, pl.T16)
    if (vec == 4) DCFP_WINO_IN(4); else if (vec == 2) DCFP_WINO_IN(2); else DCFP_WINO_IN(1);
#undef DCFP_WINO_IN
    const int rc = dcfp_igemm2_run(V, (long long)Ck * pl.T16, nullptr, 0, 0, nullptr, Mb, (long long)M * pl.T16, 16, M, Ck, 1,
                                   1, (int)pl.T16, 1, (int)pl.T16, 1, 1, 0, 1, 0, U,
                                   (size_t)pl.u_floats * sizeof(float), stream, nullptr, nullptr, nullptr, 0, nullptr,
                                   /*wp_valid=*/1, 0, (long long)pl.CkP * pl.Mpad);
    if (rc) return rc;
    const int ovec = wino_vec(d, W, W, out, out_nstride, pl.TW);
    const unsigned gout = (unsigned)((tpi / ovec + 255) / 256);
#define DCFP_WINO_OUT(VEC_, ST_) hipLaunchKernelGGL((wino_output_kernel<VEC_, ST_>), dim3(gout, (unsigned)N, (unsigned)M), dim3(256), 0, \
                                               stream, Mb, pl.T16, M, out, out_nstride, H, W, d, pl.TH, pl.TW, accumulate, stat_part, scale, shift, residual, relu)
    if (stat_part) {
        if (!dcfp_wino_stat_slots(N, H, W, d) || accumulate) return DCFP_E_UNSUPPORTED;
        if (ovec == 4) DCFP_WINO_OUT(4, true); else if (ovec == 2) DCFP_WINO_OUT(2, true); else DCFP_WINO_OUT(1, true);
    } else if (ovec == 4) DCFP_WINO_OUT(4, false); else if (ovec == 2) DCFP_WINO_OUT(2, false); else DCFP_WINO_OUT(1, false);
#undef DCFP_WINO_OUT
    DCFP_RETURN_LAUNCH();
}

// ---------------------------------------------------------------- weight gradient
//   dU[xi][m][c] = sum_t (A dy_t A^T)[xi] * (B^T x_t B)[xi],   dw = G^T dU G
// 16 products over the tiles instead of 36 over the pixels: the x transform of the forward pass, a 2x2 -> 4x4 transform
// of dy, ONE launch of the LDS-DMA 1x1 weight-gradient kernel over 16 independent problems (conv_wgrad.hip), and a
// 16 -> 9 transform of the result.
size_t dcfp_wgrad_batched_workspace_bytes(int batch, int M, int C, long long K, int* splits_out);
int dcfp_wgrad_batched_run(const float* a, const float* bmat, float* out, int batch, int M, int C, long long K,
                           void* workspace, size_t workspace_bytes, hipStream_t stream);

namespace {
struct WinoWgradPlan { WinoPlan pl; long long T16, v, y, du, slabs; };
WinoWgradPlan wino_wgrad_plan(int N, int H, int W, int d, int M, int C) {
    WinoWgradPlan w;
    w.pl = wino_plan(N, H, W, d, M, C);
    w.T16 = w.pl.T16;
    w.v = align64(16LL * C * w.T16);
    w.y = align64(16LL * M * w.T16);
    w.du = align64(16LL * M * C);
    w.slabs = align64((long long)(dcfp_wgrad_batched_workspace_bytes(16, M, C, w.T16, nullptr) / sizeof(float)));
    return w;
}
}  // namespace

bool dcfp_wino_wgrad_ok(int N, int H, int W, int d, int M, int C) {
    if (M < 128 || C < 128) return false;
    const WinoWgradPlan w = wino_wgrad_plan(N, H, W, d, M, C);
    if (4 * w.pl.T > (long long)N * H * W * 27 / 20) return false;
    if (w.T16 >= (1LL << 30) || (long long)M * w.T16 >= (1LL << 29) || (long long)C * w.T16 >= (1LL << 29)) return false;
    return M <= 65535 && C <= 65535;
}

size_t dcfp_wino_wgrad_workspace_bytes(int N, int H, int W, int d, int M, int C) {
    const WinoWgradPlan w = wino_wgrad_plan(N, H, W, d, M, C);
    return (size_t)(w.v + w.y + w.du + w.slabs) * sizeof(float);
}

int dcfp_wino_wgrad_run(const float* dy, long long dy_nstride, int dy_pitch, const float* x, long long x_nstride,
                        int x_pitch, float* dw, int N, int M, int C, int H, int W, int d, void* workspace,
                        size_t workspace_bytes, hipStream_t stream, const float* xform_in) {
    const WinoWgradPlan w = wino_wgrad_plan(N, H, W, d, M, C);
    if (!workspace || !dcfp_aligned16(workspace) || workspace_bytes < dcfp_wino_wgrad_workspace_bytes(N, H, W, d, M, C))
        return DCFP_E_WORKSPACE;
    float* Vs = static_cast<float*>(workspace);
    const float* V = xform_in ? xform_in : Vs;
    float* Y = Vs + w.v;
    float* dU = Y + w.y;
    float* slabs = dU + w.du;
    const int xv = wino_vec(d, W, x_pitch > 0 ? x_pitch : W, x, x_nstride, w.pl.TW);
    const int yv = wino_vec(d, W, dy_pitch > 0 ? dy_pitch : W, dy, dy_nstride, w.pl.TW);
#define DCFP_WINO_IN(VEC_) hipLaunchKernelGGL(wino_input_kernel<VEC_>, dim3((unsigned)((w.T16 / VEC_ + 255) / 256), 1, (unsigned)C), \
                                              dim3(256), 0, stream, x, x_nstride, x_pitch > 0 ? x_pitch : W, N, C, H, W, d, \
                                              w.pl.TH, w.pl.TW, Vs, w.T16)
    if (xform_in) { /* the forward pass left V behind */ }
    else if (xv == 4) DCFP_WINO_IN(4); else if (xv == 2) DCFP_WINO_IN(2); else DCFP_WINO_IN(1);
#undef DCFP_WINO_IN
#define DCFP_WINO_DY(VEC_) hipLaunchKernelGGL(wino_dy_kernel<VEC_>, dim3((unsigned)((w.T16 / VEC_ + 255) / 256), 1, (unsigned)M), \
                                              dim3(256), 0, stream, dy, dy_nstride, dy_pitch > 0 ? dy_pitch : W, N, M, H, W, d, \
                                              w.pl.TH, w.pl.TW, Y, w.T16)
    if (yv == 4) DCFP_WINO_DY(4); else if (yv == 2) DCFP_WINO_DY(2); else DCFP_WINO_DY(1);
#undef DCFP_WINO_DY
    const int rc = dcfp_wgrad_batched_run(Y, V, dU, 16, M, C, w.T16, slabs, (size_t)w.slabs * sizeof(float), stream);
    if (rc) return rc;
    const long long mc = (long long)M * C;
    hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, stream, dU, mc, dw);
    DCFP_RETURN_LAUNCH();
}

// bytes of the transformed input V (16 x C x T16 floats) that a forward call can leave behind for the weight gradient
size_t dcfp_wino_xform_bytes(int N, int H, int W, int d, int C) {
    const WinoPlan pl = wino_plan(N, H, W, d, 256, C);
    return (size_t)16 * C * pl.T16 * sizeof(float);
}

/*
 * dcfp_hip.h — C-ABI of libdcfp_hip.so: the MI355X (gfx950) kernels under the DCFP
 * segmentation-training hot path.
 *
 * The reference (wzx99/DCFP) has no native ABI: every op below is reached through a
 * stock torch.nn module (cuDNN / ATen).  Each entry point cites the reference call
 * site it replaces (file:line into the reference tree) — SURVEY.md §8(a)/(b).
 *
 * Conventions
 *   - every buffer (including workspace) is owned by the caller; the library
 *     allocates nothing (one exception, on request: dcfp_p2p_alloc, because the SyncBN mailbox needs a fine-grained
 *     allocation and an IPC handle torch cannot give) and keeps no state between calls; all launches are
 *     stream-ordered on `stream` (a hipStream_t passed as void*; NULL = default).
 *     The only process-wide state is a set of tuning switches read ONCE from the
 *     environment into function-local statics on first use: DCFP_CONV_MATH (bf16x3 opt-in),
 *     DCFP_CONV_WINOGRAD (0 direct kernels only / 1 cost model, default / 2 wherever eligible),
 *     DCFP_IGEMM_{DMA,DMA9,DMA8,2D,BK32,PERSIST,P128,P128_STATS,TAPSKIP} (P128: 128-row tiles with two workgroups per CU for
 *     the 1x1 forward / dgrad - 0 off / 1 K <= 256 / 2 every K, default; P128_STATS: the same for launches with the statistics
 *     epilogue), DCFP_WGRAD_{DMA,DMA_MIXED,WIDE,LOPSIDED,T192,HALF} (HALF = 0: 1x1 weight gradients on 256-row tiles),
 *     DCFP_WINO_KEEP_U (0: the fused Winograd kernels' transformed filters are scratch again, rebuilt in every call),
 *     DCFP_WINO_VEC, DCFP_WINO_FUSED (0 three-pass Winograd only / 1 fused kernel where it wins, default / 2 wherever it
 *     applies), DCFP_WINO_WGRAD_FUSED (0 batched Winograd weight gradient on the kept transform only / 1 the fused
 *     weight-gradient kernel where the cost model prefers it, default / 2 wherever it applies and beats the direct
 *     kernel), DCFP_WINO_WGRAD_RATE (TF the cost model prices that kernel at), DCFP_CONV_STEM (0: the Cin = 3 stem conv
 *     through the general kernels), DCFP_IGEMM_NT (1 / 2: nt / sc1 output stores of the 1x1 kernel, A/B), DCFP_WF_SCALAR_EPI, DCFP_CONV_GEMV (0: 1x1 convs on a 1 x 1 map through the general kernels),
 *     DCFP_CE_BWD_CELLS (0: the per-output fused upsample + CE backward instead of the cell-organised one) -
 *     kernel / algorithm selection A/B knobs, results are identical up to the documented
 *     fp32 tolerances; changing them after the first call has no effect.  (A thread_local 16-entry cache
 *     of dispatch decisions for dilated convs is the only other state.)
 *   - tensors are fp32, NCHW, dense unless a *_nstride (batch stride, in elements)
 *     argument says otherwise.
 *   - return value: 0 ok; <0 bad descriptor / unsupported shape (DCFP_E_*);
 *     >0 a hipError_t from the launch.  Nothing throws across the ABI.
 */
#ifndef DCFP_HIP_H
#define DCFP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCFP_OK 0
#define DCFP_E_BADDESC (-1)     /* inconsistent sizes / null pointer            */
#define DCFP_E_UNSUPPORTED (-2) /* kernel size other than 1x1 / 3x3, groups...  */
#define DCFP_E_WORKSPACE (-3)   /* workspace too small                          */

typedef void* dcfp_stream_t; /* hipStream_t */

/* ABI version of this header (bumped on any signature change). */
int dcfp_abi_version(void);

/* ------------------------------------------------------------------ conv2d
 * Replaces nn.Conv2d fwd + autograd dgrad/wgrad:
 *   networks/backbone/resnet.py:25-30 (Bottleneck), :88-96 (deep stem),
 *   :110-114,127-131 (downsample), networks/tools/aspp.py:13-14,57,63,
 *   networks/deeplabv3.py:25-33,37-41 (heads).
 * Cross-correlation, zero padding, groups=1, square kernel 1x1 or 3x3.
 * Hout = (H + 2*pad - dil*(K-1) - 1)/stride + 1 must match the descriptor.
 */
typedef struct DcfpConvDesc {
    int32_t N, Cin, H, W;       /* input  [N,Cin,H,W]                  */
    int32_t Cout, KH, KW;       /* weight [Cout,Cin,KH,KW]             */
    int32_t stride, pad, dil;   /* same in h and w                     */
    int32_t Hout, Wout;         /* output [N,Cout,Hout,Wout]           */
    /* Row-pitched operands (0 = dense rows).  x_pitch: rows of x are x_pitch >= W floats apart and the
     * x_pitch - W floats behind each row are ZERO (images x_pitch*H*Cin apart); dy_pitch likewise for dy
     * (rows of Wout).  A 3x3 conv with pad = dil whose taps shift columns by a non-multiple of 4 (dilation
     * 1, 2) can then copy every shifted 16-byte quad as is - what hangs over a row end reads the zero tail -
     * instead of falling back to 4-byte copies at the image borders (dcfp_conv2d_pitch_supported tells
     * whether all three passes of a descriptor take pitched operands; the BatchNorm kernels that produce
     * x / dy write pitched outputs: dcfp_bn_apply_f32 / dcfp_bn_bwd_apply_f32 `*_pitch`).
     * fwd reads x pitched, dgrad reads dy pitched, wgrad reads both; outputs are always dense.
     * Contract for a pitched tensor: besides the zero tails, the (pitch - W) floats in FRONT of its first
     * element must be readable zeros too (a leading margin of the allocation): the quad that hangs over the
     * left end of the very first row reads them. */
    int32_t x_pitch, dy_pitch;
} DcfpConvDesc;

enum { DCFP_CONV_FWD = 0, DCFP_CONV_DGRAD = 1, DCFP_CONV_WGRAD = 2 };

/* Bytes of workspace a pass needs: fwd/dgrad — the permuted, zero-padded weight copy
 * Wp[tap][c][m] the kernel streams its A operand from; wgrad — the split-K slabs.
 * fwd/dgrad take `wp_valid`: 0 => the call (re)builds Wp in `workspace` first; != 0 => the caller
 * vouches that `workspace` still holds the Wp an earlier call with the SAME descriptor, pass and
 * weight values left there (weights change once per optimizer step, so a training loop keeps one
 * buffer per conv and pass and permutes at most once per step). */
size_t dcfp_conv2d_workspace_bytes(const DcfpConvDesc* d, int pass);

/* One launch for the Wp copies of MANY convs (a model's forward and dgrad layouts after an optimizer
 * step: optimizer.py:24-25 changes every weight once per iteration).  dcfp_conv2d_wp_layout fills the
 * layout fields and n_blocks of an entry for (descriptor, pass) (DCFP_E_UNSUPPORTED where the pass keeps
 * another kind of copy, e.g. DCFP_CONV_MATH=bf16x3); the caller sets w (reference-layout weights), wp
 * (its persistent buffer of dcfp_conv2d_workspace_bytes) and first_block = prefix sum of n_blocks, uploads
 * the table once, and after every weight update launches dcfp_conv2d_permute_weights_multi_f32 and passes
 * wp_valid = 1 to the conv calls. */
#define DCFP_WP_BLOCK_ELEMS 2048
typedef struct DcfpWpEntry {
    const float* w;
    float* wp;
    int64_t first_block;
    int64_t n_blocks;
    int32_t T, Ck, CkP, M, Mpad, sAm, sAc;
    int32_t perm8;   /* 1: rows of every 256-row tile permuted for the ragged-M kernel (position 8l+i <- channel 32i+l);
                        2 / 3: not a permutation but the transformed filters G g G^T of a fused Winograd conv (forward /
                        dgrad: T = 16, one (channel, filter) pair per element - dcfp_conv2d_workspace_is_scratch) */
} DcfpWpEntry;
int dcfp_conv2d_wp_layout(const DcfpConvDesc* d, int pass, DcfpWpEntry* entry);
int dcfp_conv2d_permute_weights_multi_f32(const DcfpWpEntry* table, int n_entries, int64_t total_blocks,
                                          dcfp_stream_t stream);

/* 1 when forward, dgrad and wgrad of this descriptor all accept the row-pitched operands it names
 * (x_pitch / dy_pitch > 0): 3x3, stride 1, pad = dil, the 256 x 256 LDS-DMA tiles; else 0. */
int dcfp_conv2d_pitch_supported(const DcfpConvDesc* d);

/* 1 when the pass's workspace is pure scratch (the Winograd path of conv_winograd.hip: transformed filters,
 * transformed input and the batched GEMM's output - up to a few GB for the ASPP branches): nothing in it survives
 * the call, `wp_valid` is ignored, and a caller should hand every such conv the SAME buffer instead of keeping one
 * per conv.  0: the workspace holds the permuted weight copy described under `wp_valid` - or, for a Winograd pass
 * that always runs the fused kernel (round 4), nothing but its transformed filters (16/9 of the weights), kept, refreshed
 * by dcfp_conv2d_permute_weights_multi_f32 and validated through `wp_valid` the same way. */
int dcfp_conv2d_workspace_is_scratch(const DcfpConvDesc* d, int pass);

/* Winograd convs (see dcfp_conv2d_workspace_is_scratch): forward and weight gradient transform the same input x
 * the same way.  dcfp_conv2d_xform_bytes > 0: the forward call may write that transform (16 x Cin x tiles floats, 4x
 * the size of x) into a caller-owned buffer, and the weight gradient takes it instead of x - a fifth to a quarter of
 * its time - at the price of keeping the buffer alive between the two (the way autograd keeps x).  Same results
 * either way (the same kernels produce the same values).  0: not available for this descriptor - or not wanted: since
 * round 4 the weight gradient of most Winograd convs is ONE kernel that transforms x and dy inside the GEMM
 * (conv_winograd3.hip; dilation 1 / 2 on row-pitched operands with W % 4 == 0, even dilations >= 4 with W even), needs no
 * kept transform and is what dcfp_conv2d_wgrad_f32_nchw dispatches to; the kept transform remains for the convs that
 * kernel does not take (dense dilation-1 / 2 operands: conv_deepsup.0 reading layer3's output). */
size_t dcfp_conv2d_xform_bytes(const DcfpConvDesc* d);
int dcfp_conv2d_fwd_keep_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w, float* y, int64_t y_nstride,
                                  float* xform_out, size_t xform_bytes,
                                  float* stat_partials /* nullable: as dcfp_conv2d_fwd_stats_f32_nchw, where
                                                          dcfp_conv2d_fwd_stat_slots > 0 */,
                                  void* workspace, size_t workspace_bytes, dcfp_stream_t stream);
int dcfp_conv2d_wgrad_kept_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride, const float* xform,
                                    size_t xform_bytes, float* dw, void* workspace, size_t workspace_bytes,
                                    dcfp_stream_t stream);

/* Gradient fan-in of a residual block (networks/backbone/resnet.py:52-56, backward): dx = dgrad(dy) + fan_src * mask,
 * mask = the 1-bit ReLU mask written by dcfp_bn_apply_relu_mask_f32, fan_src = the gradient that arrived at the block's
 * output (layout of dx).  The residual branch's gradient dy * mask is then never materialised (the BatchNorm backward
 * is called without its residual output).  Same workspace / wp_valid contract as dcfp_conv2d_dgrad_f32_nchw. */
int dcfp_conv2d_dgrad_fanin_supported(const DcfpConvDesc* d);
int dcfp_conv2d_dgrad_fanin_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride, const float* w, float* dx,
                                     const float* fan_src, const void* fan_mask, void* workspace, size_t workspace_bytes,
                                     int wp_valid, dcfp_stream_t stream);
/* The same fan-in that ALSO emits the BatchNorm-backward sums of the PREVIOUS residual block (resnet.py:41-56 backward):
 * dx, this call's result, is the gradient arriving at that block's output; with that block's bn3 input `red_x` (laid
 * out as dx), its ReLU bit mask `red_mask` (layout of dcfp_bn_apply_relu_mask_f32) and batch mean `red_mean[Cin]` the
 * epilogue writes red_part[slot][Cin][2] = (sum g, sum g*(x - mean)) over 128 pixels each, g = dx * mask bit;
 * dcfp_bn_bwd_sums_from_partials_f32 reduces the slots in a fixed order in fp64 to exactly the outputs of
 * dcfp_bn_bwd_reduce_f32 (sum_dy, sum_dy_xmu, dgamma, dbeta) - that block's backward then skips its reduce kernel and
 * never re-reads dx (8 B/element less).  dcfp_conv2d_dgrad_fanin_red_slots: the slot count N*H*W/128 where this form
 * exists (fan-in supported AND the launch takes the 128-row tiles, Cout <= 256), else 0. */
int64_t dcfp_conv2d_dgrad_fanin_red_slots(const DcfpConvDesc* d);
int dcfp_conv2d_dgrad_fanin_red_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride, const float* w,
                                         float* dx, const float* fan_src, const void* fan_mask, const float* red_x,
                                         const void* red_mask, const float* red_mean, float* red_part, void* workspace,
                                         size_t workspace_bytes, int wp_valid, dcfp_stream_t stream);
int dcfp_bn_bwd_sums_from_partials_f32(const float* partials, int64_t slots, int C, const float* var, float eps,
                                       float* sum_dy, float* sum_dy_xmu, float* dgamma /* nullable */,
                                       float* dbeta /* nullable */, dcfp_stream_t stream);

/* Name of the kernel instance a pass dispatches for this descriptor, e.g.
 * "igemm_kernel<9,4,4,2,2,0>" (template args: taps, TM, TN, WM, WN[, strided-dgrad]) — the
 * string rocprofv3 shows (demangled) for the launch; used by bench.py to label rooflines.
 * Returns the length written (excluding NUL), or <0 on a bad descriptor. */
int dcfp_conv2d_kernel_name(const DcfpConvDesc* d, int pass, char* buf, int buf_len);

/* Fraction (0 < f <= 1) of the pass's nominal multiply-adds - 2*N*Cout*Hout*Wout*Cin*KH*KW, padded taps included,
 * the usual convention and the one bench.py's `roofline.achieved` uses - that the dispatched kernel really issues.
 * < 1 for dilated 3x3 convs whose dilation is comparable to the image height (ASPP 12 / 24 / 36 on 128 rows,
 * networks/tools/aspp.py:37-39): the 9-tap LDS-DMA kernels skip, per pixel tile, the kernel rows that lie
 * wholly in the zero padding (x + 0*w == x: same bits for finite weights).  Accounting only (bench.py reports
 * `executed_frac` beside `frac`); DCFP_IGEMM_TAPSKIP=0 disables the skipping. */
double dcfp_conv2d_executed_fraction(const DcfpConvDesc* d, int pass);

/* y = conv(x, w) (+ bias[co] when bias != NULL).  y_nstride: batch stride of y in
 * elements (0 => Cout*Hout*Wout), lets a branch write into a channel slice of a
 * wider tensor (ASPP concat, aspp.py:77). */
int dcfp_conv2d_fwd_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                             const float* bias, float* y, int64_t y_nstride,
                             void* workspace, size_t workspace_bytes, int wp_valid,
                             dcfp_stream_t stream);
/* Inference path (evaluate.py:186-196 predict_whole): conv with the eval-mode BatchNorm folded
 * into the epilogue: y = act( conv(x,w)[co]*scale[co] + shift[co] (+ residual) ), act = ReLU if relu
 * (scale = gamma*rsqrt(running_var+eps), shift = beta - running_mean*scale).  Dense y / residual. */
int dcfp_conv2d_fwd_fused_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w,
                                   const float* scale, const float* shift, const float* residual,
                                   int relu, float* y, void* workspace, size_t workspace_bytes,
                                   int wp_valid, dcfp_stream_t stream);
/* dx = conv_transpose(dy, w); accumulate != 0 => dx += (fan-out gradients).
 * dy_nstride: batch stride of dy in elements (0 => dense). */
int dcfp_conv2d_dgrad_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride,
                               const float* w, float* dx, int accumulate,
                               void* workspace, size_t workspace_bytes, int wp_valid,
                               dcfp_stream_t stream);
/* dw[co,ci,kh,kw] = sum_{n,p} dy[n,co,p] * x[n,ci,src(p,kh,kw)]; deterministic
 * two-stage split-K (no float atomics).  db (nullable) = sum_{n,p} dy. */
int dcfp_conv2d_wgrad_f32_nchw(const DcfpConvDesc* d, const float* dy, int64_t dy_nstride,
                               const float* x, float* dw, float* db,
                               void* workspace, size_t workspace_bytes,
                               dcfp_stream_t stream);

/* ------------------------------------------------------------- batch norm
 * Replaces nn.BatchNorm2d(+ReLU inplace)(+residual add) train fwd/bwd:
 *   resnet.py:9,26-33,41-56; aspp.py:15-16,22-24; deeplabv3.py:26-31,38-39.
 * Statistics over (N,H,W) per channel; biased variance for normalisation.
 */
/* Module-side bookkeeping of a training-mode nn.BatchNorm2d forward, folded into the kernel that
 * finalises the statistics (resnet.py:9: momentum 0.1): when running_mean != NULL,
 *   running = (1-momentum)*running + momentum*stat   (variance unbiased by count/(count-1)),
 * and when num_batches_tracked != NULL (one int64) it is incremented by one. */
typedef struct DcfpBnRunning {
    float* running_mean;          /* nullable: no running-statistics update            */
    float* running_var;
    int64_t* num_batches_tracked; /* nullable                                          */
    float momentum;
    int32_t pad_;
} DcfpBnRunning;
/* Per-channel batch mean and biased variance of x[N,C,HW] (x_nstride elements
 * between images; 0 => C*HW).  workspace: dcfp_bn_workspace_bytes(N,C,HW).  run: nullable. */
size_t dcfp_bn_workspace_bytes(int N, int C, int HW);
int dcfp_bn_stats_f32(const float* x, int64_t x_nstride, int N, int C, int HW,
                      float* mean, float* var, const DcfpBnRunning* run,
                      void* workspace, size_t workspace_bytes, dcfp_stream_t stream);
/* y = act( (x-mean)*rsqrt(var+eps)*gamma + beta (+ residual) ), act = ReLU if relu.
 * y_nstride: batch stride of y (0 => dense).  y_pitch != 0 (with W, the row length): the rows of y are
 * written y_pitch >= W floats apart (DcfpConvDesc.x_pitch of the conv that reads y); the tail behind each
 * row is the caller's zero padding and is not written. */
int dcfp_bn_apply_f32(const float* x, const float* mean, const float* var,
                      const float* gamma, const float* beta, float eps,
                      const float* residual, int relu, float* y, int64_t y_nstride,
                      int N, int C, int HW, int W, int y_pitch, dcfp_stream_t stream);
/* Backward stage 1: with g = dy * mask:
 *   sum_dy[c] = sum g ;  sum_dy_xmu[c] = sum g*(x-mean[c])
 * (dbeta = sum_dy; dgamma = sum_dy_xmu * rsqrt(var+eps): the per-filter statistic
 * that feeds the EIC score, pruners/dcfp_pruner.py:18).
 * relu: 0 no ReLU (mask = 1); 1 mask = (y > 0) read from the saved output y; 3 `y` is the bit mask
 *       written by dcfp_bn_apply_relu_mask_f32;
 *       2 mask re-derived from x with the forward's own expression (only valid when the
 *         forward had no residual input; needs var/gamma/beta, y may be NULL). */
int dcfp_bn_bwd_reduce_f32(const float* dy, int64_t dy_nstride, const float* x,
                           const float* y, int64_t y_nstride, const float* mean,
                           const float* var, const float* gamma, const float* beta, float eps,
                           int relu, int N, int C, int HW,
                           float* sum_dy, float* sum_dy_xmu, float* dgamma /* nullable: sum_dy_xmu*istd */,
                           float* dbeta /* nullable: a second copy of sum_dy (the caller's gradient slot; sum_dy
                                           itself may then be all-reduced in place under SyncBN) */,
                           void* workspace, size_t workspace_bytes, dcfp_stream_t stream);
/* Conv forward that also emits the BatchNorm batch statistics of its OUTPUT as partials
 * (resnet.py:25-33: every conv is followed by a BatchNorm that needs mean/var over N,H,W):
 * stat_partials[slot][Cout][2] = (mean, sum of squared deviations) over 128 output pixels each.
 * dcfp_conv2d_fwd_stat_slots() gives the slot count for a shape (0: not available - tiles off
 * the channel/pixel grid, bias, or DCFP_CONV_MATH=bf16x3 - use dcfp_bn_stats_f32 then);
 * dcfp_bn_stats_from_partials_f32 merges them in a fixed order in fp64 (biased variance). */
int64_t dcfp_conv2d_fwd_stat_slots(const DcfpConvDesc* d, const float* y, int64_t y_nstride);
int dcfp_conv2d_fwd_stats_f32_nchw(const DcfpConvDesc* d, const float* x, const float* w, float* y,
                                   int64_t y_nstride, float* stat_partials, void* workspace,
                                   size_t workspace_bytes, int wp_valid, dcfp_stream_t stream);
int dcfp_bn_stats_from_partials_f32(const float* partials, int64_t slots, int slot_count, int C,
                                    float* mean, float* var, const DcfpBnRunning* run /* nullable; count =
                                    slots*slot_count */, dcfp_stream_t stream);
/* SyncBatchNorm (engine.py:65) forward exchange, device side: `gathered` = world rows of
 * (mean[C], var[C], count) as all-gathered from the ranks -> pooled mean, biased variance over all
 * ranks' pixels and the total count (one float, stays on the device). */
int dcfp_syncbn_combine_f32(const float* gathered, int world, int C, float* mean, float* var,
                            float* total_count, const DcfpBnRunning* run /* nullable; pooled statistics and
                            count */, dcfp_stream_t stream);
/* SyncBatchNorm exchange without a collective library call (engine.py:65; SURVEY C2: 115 all-gathers + 115
 * all-reduces of <= 4097 floats per step, all latency-bound): each rank owns a MAILBOX in its HBM that its peers map
 * (dcfp_p2p_export -> the 64-byte handle travels over the host-side process group -> dcfp_p2p_import), and ONE
 * single-workgroup kernel per BatchNorm layer and direction writes this rank's row into every rank's mailbox over
 * xGMI, waits for the others' rows in its own, and reduces them in rank order (bit-identical on all ranks).
 *   mailboxes[r]  rank r's mailbox as mapped in THIS process (the local allocation for r == rank), world <= 8;
 *   seq           1, 2, 3, ... - the same on all ranks for the same exchange (never 0);
 *   cap_floats    payload capacity the mailboxes were sized for (multiple of 32); n <= cap_floats;
 *   mode 0        out[world][n] = the rows (all-gather);
 *   mode 1        out[n] = sum over ranks, rank order (backward: [sum g, sum g*(x-mean)] adjacent, n = 2C);
 *   mode 2        rows = (mean[C], var[C], count), n = 2C+1: out = (pooled mean[C], pooled biased var[C], total count)
 *                 as dcfp_syncbn_combine_f32 computes them, and `run` (nullable) is updated from the pooled values;
 *   spin_limit    polling rounds (~0.55 us each) after which the kernel gives up: outputs = NaN, *status = seq
 *                 (a device int32 the caller zeroed once; it is never written otherwise).
 * dcfp_p2p_alloc kind: 0 fine-grained device memory (the default), 1 uncached, 2 plain hipMalloc; zero-filled.
 * All ranks must have imported every mailbox before the first exchange, and unmap before the owner frees. */
#define DCFP_P2P_HANDLE_BYTES 64
size_t dcfp_syncbn_p2p_mailbox_bytes(int world, int cap_floats);
int dcfp_p2p_alloc(size_t bytes, int kind, void** ptr);
int dcfp_p2p_free(void* ptr);
int dcfp_p2p_export(void* ptr, void* handle64);
int dcfp_p2p_import(const void* handle64, void** ptr);
int dcfp_p2p_unmap(void* ptr);
int dcfp_syncbn_p2p_exchange_f32(void* const* mailboxes, int world, int rank, uint32_t seq, int cap_floats,
                                 const float* local, int n, int mode, float* out, const DcfpBnRunning* run,
                                 uint32_t spin_limit, int32_t* status, dcfp_stream_t stream);
/* Running statistics of nn.BatchNorm2d in training mode (resnet.py:9, momentum 0.1):
 * running = (1-momentum)*running + momentum*stat, the variance unbiased by count/(count-1);
 * count_dev (nullable, one float) overrides `count` (SyncBN: global count on the device). */
int dcfp_bn_update_running_f32(const float* mean, const float* var, int C, float momentum,
                               float count, const float* count_dev, float* running_mean,
                               float* running_var, dcfp_stream_t stream);
/* Forward of a BatchNorm + ReLU with a residual input that ALSO writes the ReLU mask as one bit per
 * element (N*C*HW/8 bytes, 8-byte aligned; needs HW % 256 == 0).  The backward kernels take the mask
 * through their `y` argument with relu == 3 instead of re-reading the 4-byte output (resnet.py:52-56). */
int dcfp_bn_apply_relu_mask_f32(const float* x, const float* mean, const float* var,
                                const float* gamma, const float* beta, float eps,
                                const float* residual, float* y, void* relu_mask,
                                int N, int C, int HW, dcfp_stream_t stream);
/* Backward stage 2: dx = gamma*istd*( g - sum_dy/M - (x-mean)*istd^2*sum_dy_xmu/M ),
 * M = count (N*HW); under SyncBN the global count lives on the device: count_dev (nullable,
 * one float) then overrides `count` without a host round trip.  d_residual (nullable) = g. */
int dcfp_bn_bwd_apply_f32(const float* dy, int64_t dy_nstride, const float* x,
                          const float* y, int64_t y_nstride, const float* mean,
                          const float* var, const float* gamma, const float* beta, float eps,
                          const float* sum_dy, const float* sum_dy_xmu, float count,
                          const float* count_dev,
                          int relu, float* dx, float* d_residual,
                          int N, int C, int HW, int W, int dx_pitch /* as y_pitch above, for dx */,
                          dcfp_stream_t stream);

/* Backward stages 1 + 2 in ONE launch, for a BatchNorm whose sums need no exchange between ranks (plain
 * nn.BatchNorm2d, or SyncBatchNorm at world size 1): a block keeps its <= 8192 elements of dy and x in registers
 * between the two stages, so both tensors are read once (12 B/element instead of 20).  Outputs are the same bits as
 * dcfp_bn_bwd_reduce_f32 followed by dcfp_bn_bwd_apply_f32 (same plan, same summation order).  Arguments as there;
 * `count` = N*HW.  The blocks of a channel meet through `sync` (dcfp_bn_bwd_fused_sync_bytes(N,C,HW) bytes, 8-byte
 * aligned; 0 = shape not supported, the call then returns DCFP_E_UNSUPPORTED and the two-kernel path applies):
 *   - the CALLER zero-fills `sync` once (it may serve every later call on the same stream, any shape that fits) and
 *     passes a strictly increasing `epoch` >= 1 with every call on it (re-zero before wrapping to 1);
 *   - spin_limit: polls (~0.3 us each) after which a block gives up waiting for its channel's other blocks
 *     (0 = default 2^18, ~0.1 s): its dx and the channel's sums become NaN and *status (nullable device int32, zeroed once by
 *     the caller, never written otherwise) = epoch.  Needs 16-byte aligned tensors, HW % 4 == 0. */
size_t dcfp_bn_bwd_fused_sync_bytes(int N, int C, int HW);
int dcfp_bn_bwd_fused_f32(const float* dy, int64_t dy_nstride, const float* x, const float* y,
                          int64_t y_nstride, const float* mean, const float* var, const float* gamma,
                          const float* beta, float eps, float count, int relu, float* dx, float* d_residual,
                          int N, int C, int HW, int W, int dx_pitch, float* sum_dy, float* sum_dy_xmu,
                          float* dgamma, float* dbeta, void* sync, size_t sync_bytes, uint32_t epoch,
                          uint32_t spin_limit, int32_t* status, dcfp_stream_t stream);

/* ------------------------------------------------- element-wise / pooling
 * MaxPool2d(3,2,1) (resnet.py:100,149): -inf padding, first-max index. */
int dcfp_maxpool3x3s2_fwd_f32(const float* x, float* y, int32_t* argmax,
                              int N, int C, int H, int W, int Hout, int Wout,
                              dcfp_stream_t stream);
int dcfp_maxpool3x3s2_bwd_f32(const float* dy, const int32_t* argmax, float* dx,
                              int N, int C, int H, int W, int Hout, int Wout,
                              dcfp_stream_t stream);
/* AdaptiveAvgPool2d(1) (aspp.py:56): y[n,c] = scale * sum_hw x[n,c,:]. */
int dcfp_rowsum_f32(const float* x, int64_t x_nstride, float* y, float scale,
                    int N, int C, int HW, dcfp_stream_t stream);
/* bilinear 1x1 -> HxW (aspp.py:76) = broadcast: y[n,c,:] (+)= scale*v[n,c]. */
int dcfp_broadcast_hw_f32(const float* v, float scale, float* y, int64_t y_nstride,
                          int accumulate, int N, int C, int HW, dcfp_stream_t stream);
/* out = a + b (gradient fan-in); out may alias a. */
int dcfp_add_f32(const float* a, const float* b, float* out, int64_t n,
                 dcfp_stream_t stream);
/* Dropout2d (deeplabv3.py:40) with a host-drawn keep mask: y = x*mask[n,c]. */
int dcfp_channel_scale_f32(const float* x, const float* mask, float* y,
                           int N, int C, int HW, dcfp_stream_t stream);

/* --------------------------------------- bilinear upsample (+) cross-entropy
 * F.interpolate(bilinear, align_corners) (deeplabv3.py:47,50). */
int dcfp_upsample_bilinear_fwd_f32(const float* x, float* y, int N, int C,
                                   int h, int w, int H, int W, int align_corners,
                                   dcfp_stream_t stream);
int dcfp_upsample_bilinear_bwd_f32(const float* dy, float* dx, int N, int C,
                                   int h, int w, int H, int W, int align_corners,
                                   dcfp_stream_t stream);
/* Fused  F.interpolate -> nn.CrossEntropyLoss(ignore_index, 'mean')
 * (deeplabv3.py:47,50 + loss/criterion.py:60,65-67): never materialises the
 * full-resolution logits.  labels: int64 [N,H,W].
 *   fwd: out[0] = sum over valid pixels of -log_softmax(z)[label];
 *        out[1] = number of valid pixels (as float).
 *   bwd: dlogits[n,c,i,j] = grad_scale * sum_pixels weight*(softmax - onehot),
 *        gathered per low-res cell (deterministic, no atomics).
 * pixel_keep (nullable, uint8 [N,H,W]): 0 => pixel treated as ignored (OHEM). */
size_t dcfp_upsample_ce_workspace_bytes(int N, int H, int W);
/* lse (nullable on fwd): [N,H,W] log-sum-exp of the interpolated logits per pixel,
 * written by fwd and consumed by bwd (33.5 MB at 4x1024x2048 instead of 637 MB of
 * full-resolution logits).  gt_prob (nullable): softmax probability of the label
 * class per pixel (1.0 where the label is ignore_index) — feeds OHEM. */
int dcfp_upsample_ce_fwd_f32(const float* logits, const int64_t* labels,
                             const uint8_t* pixel_keep, int ignore_index,
                             int N, int C, int h, int w, int H, int W, int align_corners,
                             float* lse, float* gt_prob, float* out2,
                             void* workspace, size_t workspace_bytes,
                             dcfp_stream_t stream);
int dcfp_upsample_ce_bwd_f32(const float* logits, const int64_t* labels,
                             const uint8_t* pixel_keep, int ignore_index,
                             int N, int C, int h, int w, int H, int W, int align_corners,
                             const float* lse, const float* grad_scale /* device scalar */,
                             float* dlogits, dcfp_stream_t stream);

/* OHEM threshold input (loss/ohem.py:20-33): per position of the 1/factor-zoomed grid
 * (H8 x W8 = round(H/factor) x round(W/factor), scipy.ndimage.zoom coordinates) the nearest
 * label (lab8) and the order-1-zoomed softmax probability of that label (pred8), computed from
 * the low-resolution logits and the LSE map of dcfp_upsample_ce_fwd_f32. */
int dcfp_ohem_zoom_gt_prob_f32(const float* logits, const int64_t* labels, const float* lse,
                               int N, int C, int h, int w, int H, int W, int align_corners,
                               int H8, int W8, float* pred8, int32_t* lab8,
                               dcfp_stream_t stream);

/* OHEM threshold (loss/ohem.py:34-48, np.partition on the host in the reference) on the device:
 *   num_valid = #(lab8 != ignore_label) over the n zoomed positions;
 *   min_kept >= num_valid -> 1.0;  min_kept <= 0 -> thresh;  otherwise
 *   kth = the (min(num_valid, min_kept)-1)-th smallest pred8 among valid positions (exact radix
 *   select), *threshold = kth > thresh ? kth : thresh.  min_kept is the reference's
 *   `self.min_kept // (factor*factor)`.  threshold: ONE float in device memory, consumed by
 *   dcfp_ohem_keep_mask_u8 without a host round trip. */
int dcfp_ohem_threshold_f32(const float* pred8, const int32_t* lab8, int64_t n, int ignore_label,
                            float thresh, int64_t min_kept, float* threshold /* device scalar */,
                            dcfp_stream_t stream);
/* keep[i] = gt_prob[i] <= *threshold (loss/ohem.py:69 kept_flag) over the n full-resolution pixels:
 * the pixel_keep mask of dcfp_upsample_ce_{fwd,bwd}_f32.  gt_prob 16-byte, keep 4-byte aligned. */
int dcfp_ohem_keep_mask_u8(const float* gt_prob, const float* threshold /* device scalar */, int64_t n,
                           uint8_t* keep, dcfp_stream_t stream);

/* GSRL fine-tune loss (loss/criterion.py:77-101), fused the same way as the CE above:
 *   margin[pix]  = p1 - p2, the two largest softmax probabilities of the interpolated logits
 *                  (replaces softmax + sort over the full-resolution tensor, :89-91);
 *   maxfilter    = F.max_pool2d(weight, k, stride=1, padding=k/2) of the balance weights (:88);
 *   wce fwd/bwd  = CE(reduction='none') * pix_weight, summed per image:
 *                  out_per_image[n] = (sum w*ce, sum w); bwd scales image n by
 *                  grad_scale_per_image[n] (:94-99: per-image normalisation, mean over images). */
int dcfp_upsample_margin_f32(const float* logits, int N, int C, int h, int w, int H, int W,
                             int align_corners, float* margin, dcfp_stream_t stream);
int dcfp_maxfilter2d_s1_f32(const float* x, float* y, int planes, int H, int W, int k,
                            dcfp_stream_t stream);
size_t dcfp_upsample_wce_workspace_bytes(int N, int H, int W);
int dcfp_upsample_wce_fwd_f32(const float* logits, const int64_t* labels, const float* pix_weight,
                              int ignore_index, int N, int C, int h, int w, int H, int W,
                              int align_corners, float* lse, float* out_per_image,
                              void* workspace, size_t workspace_bytes, dcfp_stream_t stream);
int dcfp_upsample_wce_bwd_f32(const float* logits, const int64_t* labels, const float* pix_weight,
                              int ignore_index, int N, int C, int h, int w, int H, int W,
                              int align_corners, const float* lse,
                              const float* grad_scale_per_image, float* dlogits,
                              dcfp_stream_t stream);

/* Evaluation (evaluate.py:229-247,340-372): pred = argmax_c F.interpolate(logits)[c] per pixel
 * (never materialising the full-resolution logits) and the class confusion matrix
 * conf[gt*C + pred] += 1 over pixels with gt != ignore_index (int64, integer atomics: exact). */
int dcfp_upsample_argmax_f32(const float* logits, int N, int C, int h, int w, int H, int W,
                             int align_corners, int32_t* pred, dcfp_stream_t stream);
int dcfp_confusion_matrix_i64(const int32_t* pred, const int64_t* gt, int ignore_index,
                              int64_t n_pixels, int C, int64_t* conf /* [C*C], accumulated */,
                              dcfp_stream_t stream);

/* ------------------------------------------------------------- EIC score
 * dcfp_pruning.step (pruners/dcfp_pruner.py:15-20), all scored BN layers in one
 * launch.  table: device array of n_layers records; eic is updated in place:
 *   flag = (g*gamma > 0); t = flag ? |g| : eic; eic = eic*r + t*(1-r)
 * evaluated exactly in the reference's operation order (two roundings). */
typedef struct DcfpEicEntry {
    const float* gamma;
    const float* grad;
    float* eic;
    int32_t n;
    int32_t pad_;
} DcfpEicEntry;
int dcfp_eic_update_f32(const DcfpEicEntry* table, int n_layers, float r,
                        float one_minus_r /* float(1.0 - double(r)), as torch casts it */,
                        dcfp_stream_t stream);

/* --------------------------------------------------------- SGD (momentum)
 * torch.optim.SGD step as built by optimizer.py:24-25 over a list of tensors:
 *   g' = g + wd*p ; buf = first ? g' : mom*buf + g' ; p -= lr*buf. */
typedef struct DcfpSgdEntry {
    float* param;
    const float* grad;
    float* momentum_buf;
    int64_t n;
    int64_t first_chunk; /* prefix sum of ceil(n / DCFP_SGD_CHUNK) over earlier entries */
    float weight_decay;
    int32_t pad_;
} DcfpSgdEntry;
#define DCFP_SGD_CHUNK 16384
int dcfp_sgd_momentum_f32(const DcfpSgdEntry* table, int n_tensors, int64_t total_chunks,
                          float lr, float momentum, int first_step, dcfp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DCFP_HIP_H */

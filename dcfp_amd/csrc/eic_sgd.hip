// eic_sgd.hip — multi-tensor kernels for the two per-parameter updates of a step:
//   dcfp_pruning.step       pruners/dcfp_pruner.py:15-20   (EIC importance score)
//   torch.optim.SGD.step    optimizer.py:24-25             (momentum, weight decay)
// The reference issues ~10 launches per scored BN layer (113 layers) and ~4 foreach
// launches per SGD step; here each is ONE launch over a device-resident pointer table.
#include "common.h"

namespace {

// One block per scored BN layer (C <= 2048 floats each: latency-bound, not HBM-bound).
// __fmul_rn/__fadd_rn pin the reference's operation order (no FMA contraction), so the
// score is bit-identical to the CPU path for identical (gamma, grad) inputs, and
// exact zeros stay exact zeros.
__global__ void __launch_bounds__(256)
eic_update_kernel(const DcfpEicEntry* __restrict__ table, float r, float one_minus_r) {
    const DcfpEicEntry e = table[blockIdx.x];
    for (int i = threadIdx.x; i < e.n; i += 256) {
        const float g = e.grad[i];
        const float w = e.gamma[i];
        const float prev = e.eic[i];
        const bool flag = __fmul_rn(g, w) > 0.f;
        // flag*|g| + (!flag)*eic  — products with {0,1}, then an add with a zero term
        const float t = __fadd_rn(__fmul_rn(flag ? 1.f : 0.f, fabsf(g)),
                                  __fmul_rn(flag ? 0.f : 1.f, prev));
        e.eic[i] = __fadd_rn(__fmul_rn(prev, r), __fmul_rn(t, one_minus_r));
    }
}

__global__ void __launch_bounds__(256)
sgd_momentum_kernel(const DcfpSgdEntry* __restrict__ table, int n_tensors, float lr,
                    float momentum, int first_step) {
    // binary search: last entry with first_chunk <= blockIdx.x
    const long long chunk = blockIdx.x;
    int lo = 0, hi = n_tensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
    }
    const DcfpSgdEntry e = table[lo];
    const long long base = (chunk - e.first_chunk) * (long long)DCFP_SGD_CHUNK;
    long long end = base + DCFP_SGD_CHUNK;
    if (end > e.n) end = e.n;
    const float wd = e.weight_decay;
    for (long long i = base + threadIdx.x; i < end; i += 256) {
        const float p = e.param[i];
        float g = e.grad[i];
        if (wd != 0.f) g = fmaf(wd, p, g);          // grad.add(param, alpha=wd)
        float b;
        if (first_step) b = g;                        // buf = clone(grad)
        else b = __fadd_rn(__fmul_rn(e.momentum_buf[i], momentum), g);  // buf.mul_(m).add_(g)
        e.momentum_buf[i] = b;
        e.param[i] = fmaf(-lr, b, p);                 // param.add_(buf, alpha=-lr)
    }
}

}  // namespace

extern "C" int dcfp_eic_update_f32(const DcfpEicEntry* table, int n_layers, float r,
                                   float one_minus_r, dcfp_stream_t stream) {
    if (n_layers == 0) return DCFP_OK;
    if (!table || n_layers < 0) return DCFP_E_BADDESC;
    hipLaunchKernelGGL(eic_update_kernel, dim3((unsigned)n_layers), dim3(256), 0, dcfp_s(stream),
                       table, r, one_minus_r);
    DCFP_RETURN_LAUNCH();
}

extern "C" int dcfp_sgd_momentum_f32(const DcfpSgdEntry* table, int n_tensors,
                                     int64_t total_chunks, float lr, float momentum,
                                     int first_step, dcfp_stream_t stream) {
    if (n_tensors == 0 || total_chunks == 0) return DCFP_OK;
    if (!table || n_tensors < 0 || total_chunks < 0 || total_chunks > 0x7fffffffLL)
        return DCFP_E_BADDESC;
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3((unsigned)total_chunks), dim3(256), 0,
                       dcfp_s(stream), table, n_tensors, lr, momentum, first_step);
    DCFP_RETURN_LAUNCH();
}

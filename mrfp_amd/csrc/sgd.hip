// sgd.hip -- fused SGD(momentum, weight decay) step over a flat fp32 arena: one launch for all
// 192 trainable tensors (40.35 M elements), 16-byte loads, HBM-bound (reads p, g, m; writes p, m).
//
// Replaces (reference): torch.optim.SGD(lr=1e-2, momentum=0.9, weight_decay=5e-4).step() with the
// LambdaLR poly factor folded into `lr` (main.py:826-839, 863-864).  Update rule of torch.optim.SGD:
//   g' = g*gscale + wd*p ;  m = g' (first step) | mu*m + g' ;  p -= lr*m
// gscale = 1/world_size folds the gradient averaging of the data-parallel all-reduce in.
#include "common.hpp"

namespace mrfp {

__global__ __launch_bounds__(256) void sgd_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                  float4* __restrict__ m, int64_t n4, float lr, float mu, float wd,
                                                  float gscale, int first) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pv = p[i], gv = g[i], mv;
        gv.x = gv.x * gscale + wd * pv.x; gv.y = gv.y * gscale + wd * pv.y;
        gv.z = gv.z * gscale + wd * pv.z; gv.w = gv.w * gscale + wd * pv.w;
        if (first) {
            mv = gv;
        } else {
            mv = m[i];
            mv.x = mu * mv.x + gv.x; mv.y = mu * mv.y + gv.y; mv.z = mu * mv.z + gv.z; mv.w = mu * mv.w + gv.w;
        }
        pv.x -= lr * mv.x; pv.y -= lr * mv.y; pv.z -= lr * mv.z; pv.w -= lr * mv.w;
        m[i] = mv;
        p[i] = pv;
    }
}

}  // namespace mrfp

extern "C" int mrfp_sgd_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float weight_decay,
                             float gscale, int first, void* stream) {
    MRFP_CHECK(p && g && m && n > 0 && n % 4 == 0, "sgd_step: bad arguments (n must be a multiple of 4)");
    MRFP_CHECK(mrfp::aligned16(p) && mrfp::aligned16(g) && mrfp::aligned16(m), "sgd_step: arenas must be 16-byte aligned");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mrfp::sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                       (const float4*)g, (float4*)m, n / 4, lr, momentum, weight_decay, gscale, first);
    MRFP_LAUNCH_CHECK();
    return 0;
}

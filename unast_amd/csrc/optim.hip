// Fused gradient-clip + AdamW over flat parameter/gradient buffers (src/train.py:358-363, 929-932).
// Two launches per optimizer phase: (1) sum of squares of the active gradient ranges -> double scalar,
// (2) one pass that applies clip_grad_norm_'s coefficient and the torch.optim.AdamW update
//     p *= 1-lr*wd ; m = lerp(m,g,1-b1) ; v = b2*v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
// HBM-bound: 4 streams read (p,g,m,v) + 3 written (p,m,v), 16 B per lane.
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n4, size_t n, double* __restrict__ out) {
    float a = 0.f;
    double acc = 0.0;
    int cnt = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(g)[i];
        a += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        if (++cnt == 64) { acc += (double)a; a = 0.f; cnt = 0; }
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) a += g[i] * g[i];
    acc += (double)a;
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    size_t n, const double* __restrict__ sumsq, float max_norm, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt) {
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float total = (float)sqrt(sumsq[0]);
        coef = fminf(max_norm / (total + 1e-6f), 1.f);
    }
    const float decay = 1.f - lr * wd;
    const float step = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gg = g[i] * coef;
        float pp = p[i] * decay;
        float mm = m[i];
        mm = mm + (gg - mm) * (1.f - b1);
        const float vv = v[i] * b2 + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = pp - step * (mm / denom);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

extern "C" int unast_sumsq(const float* g, int64_t n, double* out, hipStream_t stream) {
    UNAST_REQUIRE(g && out && n > 0, "unast_sumsq: bad arguments");
    UNAST_REQUIRE((((uintptr_t)g) & 15) == 0, "unast_sumsq: buffer must be 16-byte aligned");
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, (size_t)n / 4, (size_t)n, out);
    return unast_check_launch("unast_sumsq");
}

extern "C" int unast_adamw(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int step, hipStream_t stream) {
    UNAST_REQUIRE(p && g && m && v && n > 0 && step >= 1, "unast_adamw: bad arguments");
    UNAST_REQUIRE(!(max_norm > 0.f) || sumsq, "unast_adamw: clipping needs the sum-of-squares scalar");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    size_t blocks = ((size_t)n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, (size_t)n, sumsq, max_norm, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2));
    return unast_check_launch("unast_adamw");
}

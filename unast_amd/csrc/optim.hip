// Fused gradient-clip + Adam / AdamW over flat parameter/gradient buffers (src/train.py:358-363, 929-932).
// Two launches per optimizer phase: (1) sum of squares of the active gradient ranges -> double scalar,
// (2) one pass that applies clip_grad_norm_'s coefficient and the torch.optim.AdamW update
//     p *= 1-lr*wd ; m = lerp(m,g,1-b1) ; v = b2*v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
//     (optim_type 'adam', src/train.py:929-930: no decay factor, g += wd * p instead -- torch.optim.Adam's L2 form).
// The step-dependent scalars (lr, 1-b1^t, sqrt(1-b2^t)) come by value or from three floats in device memory, so that a
// captured HIP graph of the train step replays with the current learning rate and bias corrections.
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

// n4 float4 groups (+ a scalar tail).  split_out (optional): the updated parameters again in the GEMM's pre-split operand
// format -- per 4 consecutive elements one 16-B chunk [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] of bf16 (same byte offsets as fp32).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    size_t n4, size_t n, const double* __restrict__ sumsq, float max_norm, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float* __restrict__ split_out,
                                                    int decoupled, const float* __restrict__ dev_hyper, int zero_grad) {
    if (dev_hyper) { lr = dev_hyper[0]; bc1 = dev_hyper[1]; bc2_sqrt = dev_hyper[2]; }
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float total = (float)sqrt(sumsq[0]);
        coef = fminf(max_norm / (total + 1e-6f), 1.f);
    }
    const float decay = decoupled ? 1.f - lr * wd : 1.f;
    const float l2 = decoupled ? 0.f : wd;
    const float step = lr / bc1;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg = gg * coef + l2 * pp;
        pp *= decay;
        mm = mm + (gg - mm) * (1.f - b1);
        vv = vv * b2 + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = pp - step * (mm / denom);
    };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 P = reinterpret_cast<float4*>(p)[i], M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        reinterpret_cast<float4*>(p)[i] = P; reinterpret_cast<float4*>(m)[i] = M; reinterpret_cast<float4*>(v)[i] = V;
        if (split_out) reinterpret_cast<uint4*>(split_out)[i] = split_chunk(P);
        if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);        // optimizer.zero_grad() of the same range, in the same pass
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            upd(pp, g[i], mm, vv);
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (zero_grad) g[i] = 0.f;
        }
}

// dst = src in the pre-split operand format (see adamw_kernel); n4 chunks of 4 elements.
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        reinterpret_cast<uint4*>(dst)[i] = split_chunk(reinterpret_cast<const float4*>(src)[i]);
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

extern "C" int unast_adamw(float* p, float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int step, float* split_out, int decoupled,
                           const float* dev_hyper, int zero_grad, hipStream_t stream) {
    UNAST_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || dev_hyper), "unast_adamw: bad arguments");
    if (step < 1) step = 1;
    UNAST_REQUIRE(!(max_norm > 0.f) || sumsq, "unast_adamw: clipping needs the sum-of-squares scalar");
    UNAST_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "unast_adamw: buffers must be 16-byte aligned");
    UNAST_REQUIRE(!split_out || ((((uintptr_t)split_out) & 15) == 0 && (n & 3) == 0), "unast_adamw: split_out needs 16-byte alignment and n %% 4 == 0");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, (size_t)n / 4, (size_t)n, sumsq, max_norm, lr, beta1,
                       beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), split_out, decoupled, dev_hyper, zero_grad);
    return unast_check_launch("unast_adamw");
}

// Up to 16 32-bit words, passed BY VALUE in the kernel arguments, written to device memory: how the host refreshes the
// step-state block (RNG epoch, learning rate, bias corrections) that a captured train step reads -- no host buffer has to
// stay untouched until an asynchronous copy has run.
struct StepWords { unsigned int w[16]; };
__global__ __launch_bounds__(64) void set_words_kernel(unsigned int* __restrict__ dst, StepWords v, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = v.w[i];
}

extern "C" int unast_set_words(unsigned int* dst, const unsigned int* host_words, int n, hipStream_t stream) {
    UNAST_REQUIRE(dst && host_words && n > 0 && n <= 16, "unast_set_words: need 1..16 words");
    StepWords v;
    for (int i = 0; i < 16; ++i) v.w[i] = i < n ? host_words[i] : 0u;
    hipLaunchKernelGGL(set_words_kernel, dim3(1), dim3(64), 0, stream, dst, v, n);
    return unast_check_launch("unast_set_words");
}

// Transposed pre-split copies of the 2-D weights: dst[k][n] (chunks of 4 consecutive n: [hi x4 | lo x4]) = src[n][k], for the input-
// gradient GEMMs dX = dY W, which then read W^T K-contiguously exactly like the forward GEMMs read W (one ds_read_b128 per
// fragment instead of two transposed reads, and no re-splitting).  One launch covers every matrix of a region: `tiles` holds one
// descriptor per 64 x 64 tile {src offset, dst offset, rows N, cols K, row0, col0} (offsets in floats from the two bases);
// columns n >= N of a padded destination row are written as zeros.
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                              const int* __restrict__ tiles) {
    __shared__ float tile[64][65];
    const int* d = tiles + (size_t)blockIdx.x * 6;
    const int N = d[2], K = d[3], r0 = d[4], c0 = d[5];
    const float* src = src_base + d[0];
    float* dst = dst_base + d[1];
    const int ldT = (N + 3) & ~3;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i, r = idx >> 4, c4 = (idx & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < N && c0 + c4 < K) v = *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * K + c0 + c4);      // K % 4 == 0
        tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i, k = idx >> 4, n4 = (idx & 15) * 4;
        if (c0 + k < K && r0 + n4 < ldT)
            *reinterpret_cast<uint4*>(dst + (size_t)(c0 + k) * ldT + r0 + n4) = split_chunk(make_float4(tile[n4][k], tile[n4 + 1][k], tile[n4 + 2][k], tile[n4 + 3][k]));
    }
}

extern "C" int unast_transpose_split(const float* src_base, float* dst_base, const int* tiles_dev, int ntiles, hipStream_t stream) {
    UNAST_REQUIRE(src_base && dst_base && tiles_dev && ntiles > 0, "unast_transpose_split: bad arguments");
    UNAST_REQUIRE(((((uintptr_t)src_base) | ((uintptr_t)dst_base)) & 15) == 0, "unast_transpose_split: bases must be 16-byte aligned");
    hipLaunchKernelGGL(transpose_split_kernel, dim3(ntiles), dim3(256), 0, stream, src_base, dst_base, tiles_dev);
    return unast_check_launch("unast_transpose_split");
}

extern "C" int unast_split_f32(const float* src, float* dst, int64_t n, hipStream_t stream) {
    UNAST_REQUIRE(src && dst && n > 0 && (n & 3) == 0, "unast_split_f32: need n %% 4 == 0 (n=%lld)", (long long)n);
    UNAST_REQUIRE(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "unast_split_f32: buffers must be 16-byte aligned");
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, (size_t)n / 4);
    return unast_check_launch("unast_split_f32");
}

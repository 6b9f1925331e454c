// Fused output-head losses, forward (scalar) and backward (dlogits in one pass), for gfx950.
//   speech_loss   src/train.py:100-103,113-122  masked MSE (pre + post) + BCE-with-logits(pos_weight) on stop tokens
//   text_loss     src/train.py:105-111          weighted cross-entropy, ignore_index = PAD
//   disc loss     src/train.py:147-164,296-335  BCE-with-logits against label-smoothed (0.9/0.1) targets
// Reductions are accumulated in double (wave shuffle -> one atomic per wave); the final scalar is fp32.
#include "common.h"

// ws layout (doubles): [0] sum mask*(gold-pre)^2  [1] sum mask*(gold-post)^2  [2] sum bce(stop)  [3] workgroups that have arrived.
// ONE launch: ws is zero on entry; every workgroup adds its partial sums, the last one to arrive forms the loss and leaves ws zero again
// (no memset node in front and no single-thread "final" launch behind it: both sat on the dependent chain between forward and backward).
__global__ __launch_bounds__(256) void speech_loss_fwd_kernel(const float* __restrict__ gold, const float* __restrict__ head, int ldh,
                                                              const float* __restrict__ post, const int* __restrict__ lens, int B, int T, int M,
                                                              float eos_weight, double* __restrict__ ws, float* __restrict__ loss) {
    const int rows = B * T;
    const int mq = M >> 2;
    const size_t total = (size_t)rows * mq;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / mq), c = (int)(i - (size_t)r * mq) * 4;
        const int b = r / T, t = r - b * T;
        const int len = lens[b];
        if (t < len) {
            float4 g = *reinterpret_cast<const float4*>(gold + (size_t)r * M + c);
            float4 p = *reinterpret_cast<const float4*>(head + (size_t)r * ldh + c);
            float4 q = *reinterpret_cast<const float4*>(post + (size_t)r * M + c);
            float d;
            d = g.x - p.x; a0 += d * d; d = g.y - p.y; a0 += d * d; d = g.z - p.z; a0 += d * d; d = g.w - p.w; a0 += d * d;
            d = g.x - q.x; a1 += d * d; d = g.y - q.y; a1 += d * d; d = g.z - q.z; a1 += d * d; d = g.w - q.w; a1 += d * d;
        }
        if (c == 0) {
            const float x = head[(size_t)r * ldh + M];
            const float y = (t == len - 1) ? 1.f : 0.f;
            const float lw = 1.f + (eos_weight - 1.f) * y;
            a2 += (1.f - y) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
        }
    }
    // one set of atomics per WORKGROUP (the three accumulators are same-address atomics: per wave they serialised, 154 us)
    __shared__ double red[4][3];
    double d0 = wave_sum_d((double)a0), d1 = wave_sum_d((double)a1), d2 = wave_sum_d((double)a2);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = d0; red[threadIdx.x >> 6][1] = d1; red[threadIdx.x >> 6][2] = d2; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(ws + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long*>(ws + 3), 1ull) + 1ull;
        if (arrived == gridDim.x) {
            __threadfence();
            const double s0 = atomicAdd(ws + 0, 0.0), s1 = atomicAdd(ws + 1, 0.0), s2 = atomicAdd(ws + 2, 0.0);
            double sl = 0.0;
            for (int b = 0; b < B; ++b) sl += (double)lens[b];
            const double denom = sl * (double)M;
            loss[0] = (float)(s0 / denom + s1 / denom + s2 / ((double)B * T));
            ws[0] = 0.0; ws[1] = 0.0; ws[2] = 0.0; reinterpret_cast<unsigned long long*>(ws)[3] = 0ull;
        }
    }
}

// d_head [rows, ldh]: cols [0,M) = g*2*(pre-gold)*mask/denom, col M = g*dBCE/(B*T), remaining pad cols = 0.
// d_post [rows, M]   = g*2*(post-gold)*mask/denom.   g = *gscale (device scalar: upstream dL/dloss).
__global__ __launch_bounds__(256) void speech_loss_bwd_kernel(const float* __restrict__ gold, const float* __restrict__ head, int ldh,
                                                              const float* __restrict__ post, const int* __restrict__ lens, int B, int T, int M,
                                                              float eos_weight, const float* __restrict__ gscale, float* __restrict__ d_head,
                                                              float* __restrict__ d_post) {
    __shared__ float s_denom;
    if (threadIdx.x == 0) {
        double sl = 0.0;
        for (int b = 0; b < B; ++b) sl += (double)lens[b];
        s_denom = (float)(sl * (double)M);
    }
    __syncthreads();
    const float g = gscale[0];
    const float km = 2.f * g / s_denom;
    const float ks = g / (float)(B * T);
    const int rows = B * T;
    const int hq = ldh >> 2;
    const size_t total = (size_t)rows * hq;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / hq), c = (int)(i - (size_t)r * hq) * 4;
        const int b = r / T, t = r - b * T;
        const int len = lens[b];
        float4 o = make_float4(0, 0, 0, 0);
        if (c < M) {
            float4 op = make_float4(0, 0, 0, 0);
            if (t < len) {
                float4 gd = *reinterpret_cast<const float4*>(gold + (size_t)r * M + c);
                float4 p = *reinterpret_cast<const float4*>(head + (size_t)r * ldh + c);
                float4 q = *reinterpret_cast<const float4*>(post + (size_t)r * M + c);
                o = make_float4(km * (p.x - gd.x), km * (p.y - gd.y), km * (p.z - gd.z), km * (p.w - gd.w));
                op = make_float4(km * (q.x - gd.x), km * (q.y - gd.y), km * (q.z - gd.z), km * (q.w - gd.w));
            }
            *reinterpret_cast<float4*>(d_post + (size_t)r * M + c) = op;
        } else if (c == M) {
            const float x = head[(size_t)r * ldh + M];
            const float y = (t == len - 1) ? 1.f : 0.f;
            const float lw = 1.f + (eos_weight - 1.f) * y;
            const float sig = 1.f / (1.f + expf(-x));
            o.x = ks * ((1.f - y) - lw * (1.f - sig));
        }
        *reinterpret_cast<float4*>(d_head + (size_t)r * ldh + c) = o;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Text loss: one thread per token (V = 46 logits).  ws doubles: [0] sum w*nll, [1] sum w, [3] workgroups arrived (the same word as in the
// speech loss: workspaces are handed out from one ring), [4] sum w of the LAST call (what the backward reads; need not be zero on entry).
// One launch, as speech_loss_fwd_kernel: [0..3] zero on entry and zero again on exit.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_loss_fwd_kernel(const float* __restrict__ logits, int ldl, const int64_t* __restrict__ gold, int rows, int V,
                                                            int eos_idx, float eos_weight, int pad_idx, double* __restrict__ ws, float* __restrict__ loss) {
    float a = 0.f, w = 0.f;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        const int y = (int)gold[r];
        if (y == pad_idx) continue;
        const float* lr = logits + (size_t)r * ldl;
        float mx = lr[0];
        for (int v = 1; v < V; ++v) mx = fmaxf(mx, lr[v]);
        float se = 0.f;
        for (int v = 0; v < V; ++v) se += expf(lr[v] - mx);
        const float wy = (y == eos_idx) ? eos_weight : 1.f;
        a += wy * (logf(se) + mx - lr[y]);
        w += wy;
    }
    double da = wave_sum_d((double)a), dw = wave_sum_d((double)w);
    if ((threadIdx.x & 63) == 0) { atomicAdd(ws + 0, da); atomicAdd(ws + 1, dw); }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long*>(ws + 3), 1ull) + 1ull;
        if (arrived == gridDim.x) {
            __threadfence();
            const double s0 = atomicAdd(ws + 0, 0.0), s1 = atomicAdd(ws + 1, 0.0);
            loss[0] = (float)(s0 / s1);
            ws[4] = s1;
            ws[0] = 0.0; ws[1] = 0.0; reinterpret_cast<unsigned long long*>(ws)[3] = 0ull;
        }
    }
}

__global__ __launch_bounds__(256) void text_loss_bwd_kernel(const float* __restrict__ logits, int ldl, const int64_t* __restrict__ gold, int rows, int V,
                                                            int eos_idx, float eos_weight, int pad_idx, const double* __restrict__ ws,
                                                            const float* __restrict__ gscale, float* __restrict__ dlogits) {
    const float k = gscale[0] / (float)ws[4];
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        const int y = (int)gold[r];
        float* dr = dlogits + (size_t)r * ldl;
        if (y == pad_idx) {
            for (int v = 0; v < ldl; ++v) dr[v] = 0.f;
            continue;
        }
        const float* lr = logits + (size_t)r * ldl;
        float mx = lr[0];
        for (int v = 1; v < V; ++v) mx = fmaxf(mx, lr[v]);
        float se = 0.f;
        for (int v = 0; v < V; ++v) se += expf(lr[v] - mx);
        const float wy = ((y == eos_idx) ? eos_weight : 1.f) * k;
        const float inv = 1.f / se;
        for (int v = 0; v < V; ++v) dr[v] = wy * (expf(lr[v] - mx) * inv - (v == y ? 1.f : 0.f));
        for (int v = V; v < ldl; ++v) dr[v] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Text head + loss in ONE launch (round 4): logits = X W^T + b (TextPostnet.fc1, src/module.py:243-246, V = 46 <= 48), the weighted
// cross-entropy of src/train.py:105-111 over them and d(loss)/d(logits), all from the head GEMM's accumulators.
// A workgroup (8 waves) owns 128 rows; wave w multiplies its 16 rows [16 x 256] by the whole weight matrix (48 x 256, split hi/lo into a
// swizzled LDS image once per workgroup) on the MFMA pipe, weights as the MFMA "A" operand so that a lane ends with logits of ONE row
// (m = lane & 15) and columns 16 c + 4 g + r: the row's softmax is 12 in-lane values and two permlane swaps across the lane groups g.
// The loss's normaliser sum(w) depends on the labels alone, so every workgroup recomputes it from `gold` first (rows x 8 bytes from
// L2) and the gradient leaves in its final form k * w_y * (softmax - onehot), k = gscale / sum(w) -- gscale is the host's 1 / accum_steps.
// ws as text_loss_fwd_kernel ([0] sum w * nll, [1] sum w, [3] arrivals: zero on entry, zero on exit; [4] sum w of this call).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int whl_off(int row, int col) { return row * 512 + ((((col >> 3) ^ (row & 15))) << 4) + ((col & 7) << 1); }      // [48][256] bf16, 16-B chunks XOR-ed by row

__global__ __launch_bounds__(512) void text_head_loss_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, const float* __restrict__ bias,
                                                             const int64_t* __restrict__ gold, int rows, int V, int eos_idx, float eos_weight, int pad_idx,
                                                             float gscale, float* __restrict__ logits, float* __restrict__ dlogits, int ldl,
                                                             double* __restrict__ ws, float* __restrict__ loss) {
    __shared__ __attribute__((aligned(16))) unsigned char sw[2][48 * 512];
    __shared__ float s_part[8];
    __shared__ float s_wsum;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    // weights -> hi / lo images (rows >= V are zero)
    for (int i = t; i < 48 * 64; i += 512) {
        const int n = i >> 6, c = (i & 63) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < V) v = *reinterpret_cast<const float4*>(W + (size_t)n * 256 + c);
        u32x2 hi, lo;
        split4<3>(v, hi, lo);
        *reinterpret_cast<u32x2*>(sw[0] + whl_off(n, c)) = hi;
        *reinterpret_cast<u32x2*>(sw[1] + whl_off(n, c)) = lo;
    }
    // sum of the label weights over ALL rows (every workgroup: it is part of every row's gradient)
    float wl = 0.f;
    for (int r = t; r < rows; r += 512) {
        const int y = (int)gold[r];
        wl += (y == pad_idx) ? 0.f : (y == eos_idx ? eos_weight : 1.f);
    }
    wl = wave_sum(wl);
    if (lane == 0) s_part[wave] = wl;
    __syncthreads();
    if (t == 0) s_wsum = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) + ((s_part[4] + s_part[5]) + (s_part[6] + s_part[7]));
    __syncthreads();
    const float kg = s_wsum > 0.f ? gscale / s_wsum : 0.f;

    const int m0 = blockIdx.x * 128 + wave * 16;
    const int m = m0 + l15;
    const bool mok = m < rows;
    f32x4 acc[3] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    if (m0 < rows) {
        const float* xr = X + (size_t)min(m, rows - 1) * ldx;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(xr + 32 * ks + 8 * g);
            const float4 c = *reinterpret_cast<const float4*>(xr + 32 * ks + 8 * g + 4);
            u32x2 h0, l0, h1, l1;
            split4<3>(a, h0, l0);
            split4<3>(c, h1, l1);
            const bf16x8_t xh = __builtin_bit_cast(bf16x8_t, (u32x4){h0[0], h0[1], h1[0], h1[1]});
            const bf16x8_t xl = __builtin_bit_cast(bf16x8_t, (u32x4){l0[0], l0[1], l1[0], l1[1]});
#pragma unroll
            for (int c2 = 0; c2 < 3; ++c2) {
                const int off = whl_off(16 * c2 + l15, 32 * ks + 8 * g);
                const bf16x8_t wh = *reinterpret_cast<const bf16x8_t*>(sw[0] + off);
                const bf16x8_t wlo = *reinterpret_cast<const bf16x8_t*>(sw[1] + off);
                acc[c2] = mfma16(wlo, xh, acc[c2]);
                acc[c2] = mfma16(wh, xl, acc[c2]);
                acc[c2] = mfma16(wh, xh, acc[c2]);
            }
        }
    }
    // lane: logits[m][16 c2 + 4 g + r]
    float a_nll = 0.f, a_w = 0.f;
    if (m0 < rows) {
        const int y = mok ? (int)gold[m] : pad_idx;
        float lg[3][4];
        float mx = -3.0e38f;
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * c2 + 4 * g + r;
                const float v = acc[c2][r] + (n < V ? bias[n] : 0.f);
                lg[c2][r] = v;
                if (n < V) mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float se = 0.f, ly = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * c2 + 4 * g + r;
                if (n < V) se += expf(lg[c2][r] - mx);
                if (n == y) ly = lg[c2][r];
            }
        se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
        ly += __shfl_xor(ly, 16, 64); ly += __shfl_xor(ly, 32, 64);
        const float wy = (y == pad_idx) ? 0.f : (y == eos_idx ? eos_weight : 1.f);
        const float inv = 1.f / se;
        if (mok) {
#pragma unroll
            for (int c2 = 0; c2 < 3; ++c2) {
                const int n0 = 16 * c2 + 4 * g;
                float4 lo4, dl4;
                float* lp = &lo4.x; float* dp = &dl4.x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    lp[r] = n < V ? lg[c2][r] : 0.f;
                    dp[r] = n < V ? kg * wy * (expf(lg[c2][r] - mx) * inv - (n == y ? 1.f : 0.f)) : 0.f;
                }
                if (n0 < ldl) {
                    *reinterpret_cast<float4*>(logits + (size_t)m * ldl + n0) = lo4;
                    *reinterpret_cast<float4*>(dlogits + (size_t)m * ldl + n0) = dl4;
                }
            }
            if (g == 0) { a_nll = wy * (logf(se) + mx - ly); a_w = wy; }
        }
    }
    double da = wave_sum_d((double)a_nll), dw = wave_sum_d((double)a_w);
    if (lane == 0 && m0 < rows) { atomicAdd(ws + 0, da); atomicAdd(ws + 1, dw); }
    __syncthreads();
    if (t == 0) {
        __threadfence();
        const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long*>(ws + 3), 1ull) + 1ull;
        if (arrived == gridDim.x) {
            __threadfence();
            const double s0 = atomicAdd(ws + 0, 0.0), s1 = atomicAdd(ws + 1, 0.0);
            loss[0] = (float)(s0 / s1);
            ws[4] = s1;
            ws[0] = 0.0; ws[1] = 0.0; reinterpret_cast<unsigned long long*>(ws)[3] = 0ull;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Speech heads + their loss terms in ONE launch (round 4): head[rows, 84] = X [linear_project | stop_linear]^T + b (80 mel columns and the
// stop logit, src/module.py:170-171), the pre-net masked MSE and the stop-token BCE of speech_loss (src/train.py:113-122) and
// d(loss)/d(head), from the head GEMM's accumulators; same fragment layout as text_head_loss_kernel with six 16-column tiles.
// The third term of the loss -- the post-net MSE -- needs the post-net's output, which does not exist yet: speech_post_loss_kernel
// below adds it, writes d(loss)/d(post) and finishes the scalar.  ws (doubles, zero on entry): [0] sum mask (gold - pre)^2, [2] sum bce;
// nothing is finalised here.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void speech_head_loss_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, const float* __restrict__ bias,
                                                               const float* __restrict__ gold, const int* __restrict__ lens, int B, int T, int M,
                                                               float eos_weight, float gscale, float* __restrict__ head, float* __restrict__ d_head, int ldh,
                                                               double* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sdyn[];          // [2][96 * 512]: hi / lo images of the 81 x 256 weights (rows >= 81 zero)
    unsigned char* sw0 = sdyn;
    unsigned char* sw1 = sdyn + 96 * 512;
    __shared__ float s_denom;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int rows = B * T, NO = M + 1;
    for (int i = t; i < 96 * 64; i += 512) {
        const int n = i >> 6, c = (i & 63) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < NO) v = *reinterpret_cast<const float4*>(W + (size_t)n * 256 + c);
        u32x2 hi, lo;
        split4<3>(v, hi, lo);
        *reinterpret_cast<u32x2*>(sw0 + whl_off(n, c)) = hi;
        *reinterpret_cast<u32x2*>(sw1 + whl_off(n, c)) = lo;
    }
    if (t == 0) {
        double sl = 0.0;
        for (int b = 0; b < B; ++b) sl += (double)lens[b];
        s_denom = (float)(sl * (double)M);
    }
    __syncthreads();
    const float km = 2.f * gscale / s_denom, ks = gscale / (float)rows;
    const int m0 = blockIdx.x * 128 + wave * 16;
    const int m = m0 + l15;
    const bool mok = m < rows;
    f32x4 acc[6];
#pragma unroll
    for (int c2 = 0; c2 < 6; ++c2) acc[c2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a0 = 0.f, a2 = 0.f;
    if (m0 < rows) {
        const float* xr = X + (size_t)min(m, rows - 1) * ldx;
#pragma unroll
        for (int ks8 = 0; ks8 < 8; ++ks8) {
            const float4 a = *reinterpret_cast<const float4*>(xr + 32 * ks8 + 8 * g);
            const float4 c = *reinterpret_cast<const float4*>(xr + 32 * ks8 + 8 * g + 4);
            u32x2 h0, l0, h1, l1;
            split4<3>(a, h0, l0);
            split4<3>(c, h1, l1);
            const bf16x8_t xh = __builtin_bit_cast(bf16x8_t, (u32x4){h0[0], h0[1], h1[0], h1[1]});
            const bf16x8_t xl = __builtin_bit_cast(bf16x8_t, (u32x4){l0[0], l0[1], l1[0], l1[1]});
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) {
                const int off = whl_off(16 * c2 + l15, 32 * ks8 + 8 * g);
                const bf16x8_t wh = *reinterpret_cast<const bf16x8_t*>(sw0 + off);
                const bf16x8_t wlo = *reinterpret_cast<const bf16x8_t*>(sw1 + off);
                acc[c2] = mfma16(wlo, xh, acc[c2]);
                acc[c2] = mfma16(wh, xl, acc[c2]);
                acc[c2] = mfma16(wh, xh, acc[c2]);
            }
        }
        if (mok) {
            const int b = m / T, tt = m - b * T;
            const int len = lens[b];
            const bool live = tt < len;
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) {
                const int n0 = 16 * c2 + 4 * g;
                if (n0 >= ldh) continue;
                float4 hv = make_float4(0.f, 0.f, 0.f, 0.f), dv = hv;
                float* hp = &hv.x; float* dp = &dv.x;
                if (n0 < M) {                                        // four mel columns (M % 4 == 0)
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + n0);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                    float gd[4] = {0.f, 0.f, 0.f, 0.f};
                    if (live) { const float4 g4 = *reinterpret_cast<const float4*>(gold + (size_t)m * M + n0); gd[0] = g4.x; gd[1] = g4.y; gd[2] = g4.z; gd[3] = g4.w; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = acc[c2][r] + bb[r];
                        hp[r] = p;
                        if (live) { const float d = p - gd[r]; a0 += d * d; dp[r] = km * d; }
                    }
                } else if (n0 == M) {                                // the stop logit, then zero padding
                    const float x = acc[c2][0] + bias[M];
                    const float y = (tt == len - 1) ? 1.f : 0.f;
                    const float lw = 1.f + (eos_weight - 1.f) * y;
                    hp[0] = x;
                    a2 += (1.f - y) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
                    dp[0] = ks * ((1.f - y) - lw * (1.f - 1.f / (1.f + expf(-x))));
                }
                *reinterpret_cast<float4*>(head + (size_t)m * ldh + n0) = hv;
                *reinterpret_cast<float4*>(d_head + (size_t)m * ldh + n0) = dv;
            }
        }
    }
    __shared__ double red[8][2];
    const double d0 = wave_sum_d((double)a0), d2 = wave_sum_d((double)a2);
    if (lane == 0) { red[wave][0] = d0; red[wave][1] = d2; }
    __syncthreads();
    if (t < 2) {
        double v = 0.0;
        for (int w = 0; w < 8; ++w) v += red[w][t];
        atomicAdd(ws + (t == 0 ? 0 : 2), v);
    }
}

// The post-net term: sum mask (gold - post)^2 into ws[1], d(loss)/d(post) = gscale * 2 (post - gold) mask / denom, and -- last workgroup to
// arrive -- the scalar: (ws[0] + ws[1]) / denom + ws[2] / (B T), with ws[0] and ws[2] left by speech_head_loss_kernel earlier on this stream;
// ws is zero again on exit ([3]: arrival counter).
__global__ __launch_bounds__(256) void speech_post_loss_kernel(const float* __restrict__ gold, const float* __restrict__ post, const int* __restrict__ lens,
                                                               int B, int T, int M, float gscale, float* __restrict__ d_post, double* __restrict__ ws,
                                                               float* __restrict__ loss) {
    __shared__ float s_denom;
    if (threadIdx.x == 0) {
        double sl = 0.0;
        for (int b = 0; b < B; ++b) sl += (double)lens[b];
        s_denom = (float)(sl * (double)M);
    }
    __syncthreads();
    const float km = 2.f * gscale / s_denom;
    const int rows = B * T, mq = M >> 2;
    const size_t total = (size_t)rows * mq;
    float a1 = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / mq), c = (int)(i - (size_t)r * mq) * 4;
        const int b = r / T, t = r - b * T;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < lens[b]) {
            const float4 g = *reinterpret_cast<const float4*>(gold + (size_t)r * M + c);
            const float4 q = *reinterpret_cast<const float4*>(post + (size_t)r * M + c);
            const float dx = q.x - g.x, dy = q.y - g.y, dz = q.z - g.z, dw = q.w - g.w;
            a1 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            o = make_float4(km * dx, km * dy, km * dz, km * dw);
        }
        *reinterpret_cast<float4*>(d_post + (size_t)r * M + c) = o;
    }
    __shared__ double red[4];
    const double d1 = wave_sum_d((double)a1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d1;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ws + 1, (red[0] + red[1]) + (red[2] + red[3]));
        __threadfence();
        const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long*>(ws + 3), 1ull) + 1ull;
        if (arrived == gridDim.x) {
            __threadfence();
            const double s0 = atomicAdd(ws + 0, 0.0), s1 = atomicAdd(ws + 1, 0.0), s2 = atomicAdd(ws + 2, 0.0);
            const double denom = (double)s_denom;
            loss[0] = (float)(s0 / denom + s1 / denom + s2 / ((double)B * T));
            ws[0] = 0.0; ws[1] = 0.0; ws[2] = 0.0; reinterpret_cast<unsigned long long*>(ws)[3] = 0ull;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Discriminator targets + BCE (single block; n = 2B is tiny).  target_i = (perm[i] < B ? 1-s : 1-(1-s)), flipped for
// the generator phase (src/train.py:150-164, 319-320).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void disc_targets_kernel(const int64_t* __restrict__ perm, int n, int B, int flip, float smoothing, float* __restrict__ out) {
    for (int i = threadIdx.x; i < n; i += 256) {
        float y = 1.f - smoothing;
        if (perm[i] >= B) y = 1.f - y;
        if (flip) y = 1.f - y;
        out[i] = y;
    }
}

__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ logits, int ldx, const float* __restrict__ targets, int n,
                                                         const float* __restrict__ gscale, float* __restrict__ loss, float* __restrict__ dlogits, int ldd) {
    __shared__ float red[4];
    float a = 0.f;
    const float g = gscale ? gscale[0] : 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float y = targets[i];
        const float x = logits[(size_t)i * ldx];
        a += (1.f - y) * x + log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f);
        if (dlogits) dlogits[(size_t)i * ldd] = g * (1.f / (1.f + expf(-x)) - y) / (float)n;
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0 && loss) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
static int ls_grid(size_t work, int cap = 1024) {
    size_t b = (work + 255) / 256;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

extern "C" int unast_speech_loss_fwd(const float* gold, const float* head, int ldh, const float* post, const int* lens, int B, int T, int M,
                                     float eos_weight, double* ws /* 4 doubles, zero on entry, zero on exit */, float* loss, hipStream_t stream) {
    UNAST_REQUIRE(gold && head && post && lens && ws && loss, "unast_speech_loss_fwd: null pointer");
    UNAST_REQUIRE(B > 0 && T > 0 && (M & 3) == 0 && ldh > M && (ldh & 3) == 0, "unast_speech_loss_fwd: need M%%4==0 and ldh>M, ldh%%4==0");
    hipLaunchKernelGGL(speech_loss_fwd_kernel, dim3(ls_grid((size_t)B * T * (M / 4), 512)), dim3(256), 0, stream, gold, head, ldh, post, lens, B, T, M,
                       eos_weight, ws, loss);
    return unast_check_launch("unast_speech_loss_fwd");
}

extern "C" int unast_speech_loss_bwd(const float* gold, const float* head, int ldh, const float* post, const int* lens, int B, int T, int M,
                                     float eos_weight, const float* gscale, float* d_head, float* d_post, hipStream_t stream) {
    UNAST_REQUIRE(gold && head && post && lens && gscale && d_head && d_post, "unast_speech_loss_bwd: null pointer");
    UNAST_REQUIRE(B > 0 && T > 0 && (M & 3) == 0 && ldh > M && (ldh & 3) == 0, "unast_speech_loss_bwd: need M%%4==0 and ldh>M, ldh%%4==0");
    hipLaunchKernelGGL(speech_loss_bwd_kernel, dim3(ls_grid((size_t)B * T * (ldh / 4))), dim3(256), 0, stream, gold, head, ldh, post, lens, B, T, M,
                       eos_weight, gscale, d_head, d_post);
    return unast_check_launch("unast_speech_loss_bwd");
}

extern "C" int unast_text_loss_fwd(const float* logits, int ldl, const int64_t* gold, int rows, int V, float eos_weight,
                                   double* ws /* 5 doubles: [0..3] zero on entry and on exit, [4] kept for bwd */, float* loss, hipStream_t stream) {
    UNAST_REQUIRE(logits && gold && ws && loss && rows > 0 && V > 0 && ldl >= V, "unast_text_loss_fwd: bad arguments");
    hipLaunchKernelGGL(text_loss_fwd_kernel, dim3(ls_grid(rows)), dim3(256), 0, stream, logits, ldl, gold, rows, V, 2, eos_weight, 0, ws, loss);
    return unast_check_launch("unast_text_loss_fwd");
}

extern "C" int unast_text_loss_bwd(const float* logits, int ldl, const int64_t* gold, int rows, int V, float eos_weight, const double* ws,
                                   const float* gscale, float* dlogits, hipStream_t stream) {
    UNAST_REQUIRE(logits && gold && ws && gscale && dlogits && rows > 0 && V > 0 && ldl >= V, "unast_text_loss_bwd: bad arguments");
    hipLaunchKernelGGL(text_loss_bwd_kernel, dim3(ls_grid(rows)), dim3(256), 0, stream, logits, ldl, gold, rows, V, 2, eos_weight, 0, ws, gscale, dlogits);
    return unast_check_launch("unast_text_loss_bwd");
}

// Uniform random permutation of 0..n-1 (torch.randperm at src/train.py:323): i.i.d. 32-bit keys from the counter RNG, ranked
// with the index as tie-break.  One workgroup; n <= 4096 (2 x batch rows).
__global__ __launch_bounds__(256) void randperm_kernel(long long* __restrict__ out, int n, uint32_t seed, uint32_t stream) {
    __shared__ uint32_t keys[4096];
    const uint32_t rk = rng_row_key(seed, stream, 0x9E3779B9u);
    for (int i = threadIdx.x; i < n; i += 256) keys[i] = rng_u32(rk, (uint32_t)i);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint32_t k = keys[i];
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const uint32_t kj = keys[j];
            r += (kj < k || (kj == k && j < i)) ? 1 : 0;
        }
        out[r] = i;
    }
}

extern "C" int unast_randperm(int64_t* out, int n, unsigned int seed, unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(out && n > 0 && n <= 4096, "unast_randperm: need 0 < n <= 4096 (got %d)", n);
    hipLaunchKernelGGL(randperm_kernel, dim3(1), dim3(256), 0, stream, (long long*)out, n, seed, stream_id);
    return unast_check_launch("unast_randperm");
}

extern "C" int unast_disc_targets(const int64_t* perm, int n, int B, int flip, float smoothing, float* out, hipStream_t stream) {
    UNAST_REQUIRE(perm && out && n > 0, "unast_disc_targets: bad arguments");
    hipLaunchKernelGGL(disc_targets_kernel, dim3(1), dim3(256), 0, stream, perm, n, B, flip, smoothing, out);
    return unast_check_launch("unast_disc_targets");
}

extern "C" int unast_bce_logits(const float* logits, int ldx, const float* targets, int n, const float* gscale, float* loss, float* dlogits,
                                int ldd, hipStream_t stream) {
    UNAST_REQUIRE(logits && targets && n > 0 && ldx >= 1, "unast_bce_logits: bad arguments");
    UNAST_REQUIRE((dlogits == nullptr) || (gscale && ldd >= 1), "unast_bce_logits: dlogits requires gscale and ldd");
    hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, stream, logits, ldx, targets, n, gscale, loss, dlogits, ldd);
    return unast_check_launch("unast_bce_logits");
}

// masked_mse (src/train.py:100-103) as its own entry: sum((gold - pred)^2 * mask) / sum(mask).  One launch: every workgroup adds its
// two fp64 partial sums into ws[0..1] and counts itself in ws[2]; the last one to arrive divides (ws must be zero on entry and is
// left zero again, so one workspace serves call after call).
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* __restrict__ gold, const float* __restrict__ pred, const float* __restrict__ mask,
                                                         size_t n, double* __restrict__ ws, float* __restrict__ out) {
    double num = 0.0, den = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = gold[i] - pred[i], m = mask[i];
        num += (double)(d * d * m);
        den += (double)m;
    }
    num = wave_sum_d(num);
    den = wave_sum_d(den);
    __shared__ double sn[4], sd[4];
    if ((threadIdx.x & 63) == 0) { sn[threadIdx.x >> 6] = num; sd[threadIdx.x >> 6] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ws + 0, (sn[0] + sn[1]) + (sn[2] + sn[3]));
        atomicAdd(ws + 1, (sd[0] + sd[1]) + (sd[2] + sd[3]));
        __threadfence();
        const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long*>(ws + 2), 1ull) + 1ull;
        if (arrived == gridDim.x) {
            __threadfence();
            const double a = atomicAdd(ws + 0, 0.0), b = atomicAdd(ws + 1, 0.0);
            out[0] = (float)(a / b);
            ws[0] = 0.0; ws[1] = 0.0; reinterpret_cast<unsigned long long*>(ws)[2] = 0ull;
        }
    }
}

extern "C" int unast_masked_mse(const float* gold, const float* pred, const float* mask, int64_t n, double* ws3, float* out, hipStream_t stream) {
    UNAST_REQUIRE(gold && pred && mask && ws3 && out && n > 0, "unast_masked_mse: bad arguments");
    size_t blocks = ((size_t)n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(masked_mse_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, gold, pred, mask, (size_t)n, ws3, out);
    return unast_check_launch("unast_masked_mse");
}

__global__ __launch_bounds__(64) void scalar_combine_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float div, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float s = a[0];
        if (b) s += b[0];
        if (c) s += c[0];
        out[0] = s / div;
    }
}

extern "C" int unast_scalar_combine(const float* a, const float* b, const float* c, float div, float* out, hipStream_t stream) {
    UNAST_REQUIRE(a && out && div != 0.f, "unast_scalar_combine: bad arguments");
    hipLaunchKernelGGL(scalar_combine_kernel, dim3(1), dim3(64), 0, stream, a, b, c, div, out);
    return unast_check_launch("unast_scalar_combine");
}

extern "C" int unast_text_head_loss(const float* X, int ldx, const float* W, const float* bias, const int64_t* gold, int rows, int K, int V,
                                    float eos_weight, float gscale, float* logits, float* dlogits, int ldl, double* ws, float* loss, hipStream_t stream) {
    const int eos_idx = 2, pad_idx = 0;
    UNAST_REQUIRE(X && W && bias && gold && logits && dlogits && ws && loss && rows > 0, "unast_text_head_loss: bad arguments");
    UNAST_REQUIRE(K == 256 && V > 0 && V <= 48 && ldl >= V && ldl <= 48 && (ldl & 3) == 0 && (ldx & 3) == 0, "unast_text_head_loss: built for K = 256, V <= 48, 16-byte row strides (K=%d V=%d ldl=%d)", K, V, ldl);
    UNAST_REQUIRE((((uintptr_t)X | (uintptr_t)W | (uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0, "unast_text_head_loss: operands must be 16-byte aligned");
    hipLaunchKernelGGL(text_head_loss_kernel, dim3((rows + 127) / 128), dim3(512), 0, stream, X, ldx, W, bias, gold, rows, V, eos_idx, eos_weight, pad_idx,
                       gscale, logits, dlogits, ldl, ws, loss);
    return unast_check_launch("unast_text_head_loss");
}

extern "C" int unast_speech_head_loss(const float* X, int ldx, const float* W, const float* bias, const float* gold, const int* lens, int B, int T, int K, int M,
                                      float eos_weight, float gscale, float* head, float* d_head, int ldh, double* ws, hipStream_t stream) {
    UNAST_REQUIRE(X && W && bias && gold && lens && head && d_head && ws && B > 0 && T > 0, "unast_speech_head_loss: bad arguments");
    UNAST_REQUIRE(K == 256 && M > 0 && (M & 3) == 0 && M + 1 <= 96 && ldh >= M + 1 && ldh <= 96 && (ldh & 3) == 0 && (ldx & 3) == 0,
                  "unast_speech_head_loss: built for K = 256, M %% 4 == 0, M + 1 <= 96 (K=%d M=%d ldh=%d)", K, M, ldh);
    UNAST_REQUIRE((((uintptr_t)X | (uintptr_t)W | (uintptr_t)bias | (uintptr_t)gold | (uintptr_t)head | (uintptr_t)d_head) & 15) == 0, "unast_speech_head_loss: operands must be 16-byte aligned");
    static bool attr = [] { return hipFuncSetAttribute((const void*)speech_head_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 96 * 512) == hipSuccess; }();
    (void)attr;
    const int rows = B * T;
    hipLaunchKernelGGL(speech_head_loss_kernel, dim3((rows + 127) / 128), dim3(512), 2 * 96 * 512, stream, X, ldx, W, bias, gold, lens, B, T, M, eos_weight, gscale,
                       head, d_head, ldh, ws);
    return unast_check_launch("unast_speech_head_loss");
}

extern "C" int unast_speech_post_loss(const float* gold, const float* post, const int* lens, int B, int T, int M, float gscale, float* d_post, double* ws,
                                      float* loss, hipStream_t stream) {
    UNAST_REQUIRE(gold && post && lens && d_post && ws && loss && B > 0 && T > 0 && M > 0 && (M & 3) == 0, "unast_speech_post_loss: bad arguments");
    hipLaunchKernelGGL(speech_post_loss_kernel, dim3(ls_grid((size_t)B * T * (M / 4))), dim3(256), 0, stream, gold, post, lens, B, T, M, gscale, d_post, ws, loss);
    return unast_check_launch("unast_speech_post_loss");
}

UNAST_DEFINE_RNG_EPOCH_SETTER(loss)

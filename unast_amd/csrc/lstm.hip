// Persistent recurrent kernels for the LSTM discriminator (src/network.py:172-186, src/module.py:297-336).
//
// The input projections X*W_ih^T of all timesteps are one MFMA GEMM (gemm.hip); what remains is the strictly
// sequential part: gates_t = xproj_t + b_ih + b_hh + h_{t-1} W_hh^T.  It is latency-bound (16 K MACs per step),
// so one 256-thread workgroup owns one (sequence, direction) pair for the WHOLE sequence: thread j keeps row j of
// W_hh (64 floats) in VGPRs, h lives in LDS, and the kernel loops over the valid timesteps on-chip — the packed-
// sequence semantics of pack_padded_sequence (steps t >= len[b] are never touched; the reverse direction starts at
// t = len[b]-1).  Gate order i,f,g,o as torch.nn.LSTM.  fp32 throughout (exact VALU FMAs).
#include "common.h"

#define LH 64             // hidden size
#define LG (4 * LH)       // gate rows

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// xproj [Bd,T,ndir*LG] (no bias), y [Bd,T,ndir*LH] (pre-zeroed), gates [Bd,T,ndir,LG], cs [Bd,T,ndir,LH],
// hprev [Bd,T,ndir,LH] (pre-zeroed), hfinal [Bd, ndir*LH]
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ whh, const float* __restrict__ b_ih,
                                                       const float* __restrict__ b_hh, const int* __restrict__ lens, float* __restrict__ y,
                                                       float* __restrict__ gates, float* __restrict__ cs, float* __restrict__ hprev,
                                                       float* __restrict__ hfinal, int T, int ndir, size_t whh_dir_stride, size_t bias_dir_stride) {
    __shared__ __attribute__((aligned(16))) float h_lds[LH];
    __shared__ float g_lds[LG];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int len = lens[b];
    float w[LH];
    const float* wr = whh + dir * whh_dir_stride + (size_t)j * LH;
#pragma unroll
    for (int k = 0; k < LH; k += 4) {
        float4 v = *reinterpret_cast<const float4*>(wr + k);
        w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
    }
    const float bias = b_ih[dir * bias_dir_stride + j] + b_hh[dir * bias_dir_stride + j];
    if (j < LH) h_lds[j] = 0.f;
    float c = 0.f, h = 0.f;
    const size_t xs = (size_t)ndir * LG;
    const float* xp = xproj + (size_t)b * T * xs + (size_t)dir * LG + j;
    __syncthreads();
    float xg = (len > 0) ? xp[(size_t)(dir ? len - 1 : 0) * xs] : 0.f;
    for (int step = 0; step < len; ++step) {
        const int t = dir ? (len - 1 - step) : step;
        float xn = 0.f;
        if (step + 1 < len) xn = xp[(size_t)(dir ? t - 1 : t + 1) * xs];     // prefetch next step's projection
        float a0 = xg + bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < LH; k += 4) {
            float4 hv = *reinterpret_cast<const float4*>(&h_lds[k]);
            a0 = fmaf(w[k], hv.x, a0); a1 = fmaf(w[k + 1], hv.y, a1); a2 = fmaf(w[k + 2], hv.z, a2); a3 = fmaf(w[k + 3], hv.w, a3);
        }
        const float pre = (a0 + a1) + (a2 + a3);
        const float act = ((j >> 6) == 2) ? tanhf(pre) : sigmoidf_(pre);      // wave-uniform: waves = i,f,g,o
        g_lds[j] = act;
        const size_t row = ((size_t)b * T + t) * ndir + dir;
        gates[row * LG + j] = act;
        __syncthreads();
        if (j < LH) {
            const float ig = g_lds[j], fg = g_lds[LH + j], gg = g_lds[2 * LH + j], og = g_lds[3 * LH + j];
            hprev[row * LH + j] = h;
            c = fg * c + ig * gg;
            h = og * tanhf(c);
            cs[row * LH + j] = c;
            y[((size_t)b * T + t) * (ndir * LH) + dir * LH + j] = h;
            h_lds[j] = h;
        }
        __syncthreads();
        xg = xn;
    }
    if (j < LH) hfinal[(size_t)b * (ndir * LH) + dir * LH + j] = h;
}

// Backward through time.  dy [Bd,T,ndir*LH] (may be null), dhfinal [Bd,ndir*LH] (may be null),
// dgates [Bd,T,ndir,LG] (pre-zeroed; receives d(pre-activation gates) for valid steps).
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dhfinal, const float* __restrict__ whh,
                                                       const float* __restrict__ gates, const float* __restrict__ cs, const int* __restrict__ lens,
                                                       float* __restrict__ dgates, int T, int ndir, size_t whh_dir_stride) {
    __shared__ __attribute__((aligned(16))) float dg_lds[LG];
    __shared__ float part_lds[4][LH];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int len = lens[b];
    const int k = j & 63, part = j >> 6;
    float wt[LH];                       // wt[i] = W_hh[64*part + i][k]
    const float* wr = whh + dir * whh_dir_stride + (size_t)(part * LH) * LH + k;
#pragma unroll
    for (int i = 0; i < LH; ++i) wt[i] = wr[(size_t)i * LH];
    float dh = 0.f, dc = 0.f;
    if (j < LH && dhfinal) dh = dhfinal[(size_t)b * (ndir * LH) + dir * LH + j];
    for (int step = len - 1; step >= 0; --step) {           // reverse of the forward processing order
        const int t = dir ? (len - 1 - step) : step;
        const size_t row = ((size_t)b * T + t) * ndir + dir;
        if (j < LH) {
            float dht = dh;
            if (dy) dht += dy[((size_t)b * T + t) * (ndir * LH) + dir * LH + j];
            const float ig = gates[row * LG + j], fg = gates[row * LG + LH + j], gg = gates[row * LG + 2 * LH + j], og = gates[row * LG + 3 * LH + j];
            const float ct = cs[row * LH + j];
            float cprev = 0.f;
            if (step > 0) {
                const int tp = dir ? t + 1 : t - 1;
                cprev = cs[(((size_t)b * T + tp) * ndir + dir) * LH + j];
            }
            const float tc = tanhf(ct);
            const float d_o = dht * tc * og * (1.f - og);
            const float dct = dc + dht * og * (1.f - tc * tc);
            const float d_i = dct * gg * ig * (1.f - ig);
            const float d_f = dct * cprev * fg * (1.f - fg);
            const float d_g = dct * ig * (1.f - gg * gg);
            dc = dct * fg;
            dg_lds[j] = d_i; dg_lds[LH + j] = d_f; dg_lds[2 * LH + j] = d_g; dg_lds[3 * LH + j] = d_o;
            float* dgr = dgates + row * LG;
            dgr[j] = d_i; dgr[LH + j] = d_f; dgr[2 * LH + j] = d_g; dgr[3 * LH + j] = d_o;
        }
        __syncthreads();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < LH; i += 4) {
            float4 dv = *reinterpret_cast<const float4*>(&dg_lds[part * LH + i]);
            a0 = fmaf(wt[i], dv.x, a0); a1 = fmaf(wt[i + 1], dv.y, a1); a2 = fmaf(wt[i + 2], dv.z, a2); a3 = fmaf(wt[i + 3], dv.w, a3);
        }
        part_lds[part][k] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (j < LH) dh = (part_lds[0][j] + part_lds[1][j]) + (part_lds[2][j] + part_lds[3][j]);
    }
}

// y = dropout(leaky_relu(x, slope));  backward: dx = dy*mask/(1-p) * (x > 0 ? 1 : slope).  slope = 1 gives plain dropout.
__global__ __launch_bounds__(256) void leaky_dropout_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int rows, int D,
                                                            float slope, uint32_t drop_thresh, float drop_scale, uint32_t seed, uint32_t stream) {
    const size_t total = (size_t)rows * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / D), c = (int)(i - (size_t)r * D);
        const float xv = x[i];
        float v = dy ? dy[i] * (xv > 0.f ? 1.f : slope) : (xv > 0.f ? xv : xv * slope);
        if (drop_thresh) v = rng_keep(rng_row_key(seed, stream, (uint32_t)r), c, drop_thresh) ? v * drop_scale : 0.f;
        out[i] = v;
    }
}

extern "C" int unast_lstm_fwd(const float* xproj, const float* whh, const float* b_ih, const float* b_hh, const int* lens, float* y, float* gates,
                              float* cs, float* hprev, float* hfinal, int Bd, int T, int ndir, int hidden, int64_t whh_dir_stride,
                              int64_t bias_dir_stride, hipStream_t stream) {
    UNAST_REQUIRE(xproj && whh && b_ih && b_hh && lens && y && gates && cs && hprev && hfinal, "unast_lstm_fwd: null pointer");
    UNAST_REQUIRE(hidden == LH, "unast_lstm_fwd: this build supports hidden=%d only (got %d)", LH, hidden);
    UNAST_REQUIRE(Bd > 0 && T > 0 && (ndir == 1 || ndir == 2), "unast_lstm_fwd: bad dims");
    UNAST_REQUIRE((((uintptr_t)whh) & 15) == 0 && (whh_dir_stride & 3) == 0, "unast_lstm_fwd: W_hh must be 16-byte aligned");
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(Bd, ndir), dim3(256), 0, stream, xproj, whh, b_ih, b_hh, lens, y, gates, cs, hprev, hfinal, T, ndir,
                       (size_t)whh_dir_stride, (size_t)bias_dir_stride);
    return unast_check_launch("unast_lstm_fwd");
}

extern "C" int unast_lstm_bwd(const float* dy, const float* dhfinal, const float* whh, const float* gates, const float* cs, const int* lens,
                              float* dgates, int Bd, int T, int ndir, int hidden, int64_t whh_dir_stride, hipStream_t stream) {
    UNAST_REQUIRE(whh && gates && cs && lens && dgates, "unast_lstm_bwd: null pointer");
    UNAST_REQUIRE(hidden == LH, "unast_lstm_bwd: this build supports hidden=%d only (got %d)", LH, hidden);
    UNAST_REQUIRE(Bd > 0 && T > 0 && (ndir == 1 || ndir == 2), "unast_lstm_bwd: bad dims");
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(Bd, ndir), dim3(256), 0, stream, dy, dhfinal, whh, gates, cs, lens, dgates, T, ndir, (size_t)whh_dir_stride);
    return unast_check_launch("unast_lstm_bwd");
}

extern "C" int unast_leaky_dropout(const float* x, const float* dy, float* out, int rows, int D, float slope, float drop_p, unsigned int seed,
                                   unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(x && out && rows > 0 && D > 0, "unast_leaky_dropout: bad arguments");
    size_t blocks = ((size_t)rows * D + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(leaky_dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, dy, out, rows, D, slope, drop_threshold(drop_p),
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id);
    return unast_check_launch("unast_leaky_dropout");
}

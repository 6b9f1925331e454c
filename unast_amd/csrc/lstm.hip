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

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

#define FCH 16            // forward: timesteps of input projections held in registers per chunk
#define BCH 8             // backward: timesteps of saved state held in registers per chunk

// Sums / exchanges over the four 16-lane rows of a wave by gfx950's permlane swaps (no LDS crossbar, no barrier).  Fed the same register
// twice, v_permlane16_swap returns (value of the pair's EVEN-row lane, value of its ODD-row lane) in every lane of a {l, l ^ 16} pair, and
// v_permlane32_swap (value of the lower-half lane, value of the upper-half lane) of a {l, l ^ 32} pair.
__device__ __forceinline__ void swap16(float v, float& even_row, float& odd_row) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    even_row = __uint_as_float(a[0]); odd_row = __uint_as_float(a[1]);
}
__device__ __forceinline__ void swap32(float v, float& lower, float& upper) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lower = __uint_as_float(a[0]); upper = __uint_as_float(a[1]);
}

// xproj [Bd,T,ndir*LG] (no bias), y [Bd,T,ndir*LH], gates [Bd,T,ndir,LG], cs [Bd,T,ndir,LH], hprev [Bd,T,ndir,LH], hfinal [Bd, ndir*LH].
// y and hprev of the padded steps t >= len are written as zeros here (they are GEMM operands of the next layer / of the weight
// gradients over all Bd*T rows): the caller allocates them uninitialised -- pre-zeroing them with a fill launch each was 26 MB of
// memset per tensor and call at config 3.
//
// Round 4 layout: the four gates of a hidden unit live in ONE wave.  Wave w owns units [16 w, 16 w + 16); lane l = 16 q + c computes gate
// q (i, f, g, o) of unit 16 w + c, i.e. row 64 q + 16 w + c of W_hh (64 floats in VGPRs).  After the activation three permlane swaps hand
// every lane the other three gates of its unit, the cell update happens in place (four times redundantly), and the only thing that
// crosses waves is h: 16 values per wave into a double-buffered LDS vector, ONE barrier per step, one LDS round trip on the step's
// dependent chain (the round-1..3 form published the activated gates through LDS, met at the barrier, read four gates back, and then
// paid a second write -> read hop for its wave-private copy of h: ~1 050 cycles per step, of which the two hops were about a third).
// Global latency stays off the chain: the projections of FCH steps are register-resident while the next chunk's loads are in flight.
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ whh, const float* __restrict__ b_ih,
                                                       const float* __restrict__ b_hh, const int* __restrict__ lens, float* __restrict__ y,
                                                       float* __restrict__ gates, float* __restrict__ cs, float* __restrict__ hprev,
                                                       float* __restrict__ hfinal, int T, int ndir, size_t whh_dir_stride, size_t bias_dir_stride) {
    __shared__ __attribute__((aligned(16))) float h_lds[2][LH];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int wave = j >> 6, lane = j & 63, q = lane >> 4;
    const int u = 16 * wave + (lane & 15);             // hidden unit of this lane
    const int row = q * LH + u;                        // its gate row
    const int len = lens[b];
    f32x2 w[LH / 2];                       // row `row` of W_hh as register pairs: the dot product below runs on v_pk_fma_f32
    const float* wr = whh + dir * whh_dir_stride + (size_t)row * LH;
#pragma unroll
    for (int k = 0; k < LH; k += 4) {
        float4 v = *reinterpret_cast<const float4*>(wr + k);
        w[k / 2] = (f32x2){v.x, v.y}; w[k / 2 + 1] = (f32x2){v.z, v.w};
    }
    const float bias = b_ih[dir * bias_dir_stride + row] + b_hh[dir * bias_dir_stride + row];
    if (j < LH) h_lds[0][j] = 0.f;
    for (int t = max(len, 0) + wave; t < T; t += 4) {                // padded steps: wave w clears every 4th one (lane = unit here)
        y[((size_t)b * T + t) * ((size_t)ndir * LH) + dir * LH + lane] = 0.f;
        hprev[(((size_t)b * T + t) * ndir + dir) * LH + lane] = 0.f;
    }
    if (len <= 0) {                                                  // (uniform) empty sequence: final state = initial state
        if (wave == 0) hfinal[(size_t)b * (ndir * LH) + dir * LH + lane] = 0.f;
        return;
    }
    __syncthreads();
    float c = 0.f, h = 0.f;
    const size_t xs = (size_t)ndir * LG;
    const float* xp = xproj + (size_t)b * T * xs + (size_t)dir * LG + row;
    const int tstep = dir ? -1 : 1;
    const int t0 = dir ? len - 1 : 0;
    // Output pointers of the first step, advanced by one time step per iteration (the 64-bit row arithmetic per step was
    // a fifth of the step's instructions, all of them on its dependent chain).
    const size_t row0 = ((size_t)b * T + t0) * ndir + dir;
    float* gp = gates + row0 * LG + row;
    float* hp = hprev + row0 * LH + u;
    float* cp = cs + row0 * LH + u;
    float* yp = y + ((size_t)b * T + t0) * ((size_t)ndir * LH) + dir * LH + u;
    const ptrdiff_t g_inc = (ptrdiff_t)tstep * ndir * LG, h_inc = (ptrdiff_t)tstep * ndir * LH, y_inc = (ptrdiff_t)tstep * ndir * LH;
    // sigmoid(x) = 1 / (1 + exp(-x)), tanh(x) = 2 / (1 + exp(-2x)) - 1: one branch-free form with per-lane constants (lane row 2 = g)
    const float act_m = (q == 2) ? -2.f : -1.f, act_s = (q == 2) ? 2.f : 1.f, act_o = (q == 2) ? -1.f : 0.f;
    const bool writer = q == 0;                                      // one lane row per wave stores the unit's state
    float xc[FCH], xn[FCH];
    auto load_chunk = [&](int s0, float (&x)[FCH]) {
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            const int st = min(s0 + i, len - 1);                   // clamped: a fixed number of loads per chunk
            x[i] = xp[(size_t)(t0 + st * tstep) * xs];
        }
    };
    load_chunk(0, xn);
    for (int s0 = 0; s0 < len; s0 += FCH) {
#pragma unroll
        for (int i = 0; i < FCH; ++i) xc[i] = xn[i];                // the only wait for global loads: once per chunk
        if (s0 + FCH < len) load_chunk(s0 + FCH, xn);
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            if (s0 + i < len) {                                      // (uniform; a guard, not a break, so that the chunk unrolls and xc[i] is a register)
                const float* hl = h_lds[i & 1];                      // (FCH is even: the step's parity is i & 1)
                f32x2 a0 = {xc[i] + bias, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};     // four independent chains of 8 packed FMAs
#pragma unroll
                for (int k = 0; k < LH; k += 8) {
                    const float4 hv0 = *reinterpret_cast<const float4*>(&hl[k]);
                    const float4 hv1 = *reinterpret_cast<const float4*>(&hl[k + 4]);
                    a0 = __builtin_elementwise_fma(w[k / 2], (f32x2){hv0.x, hv0.y}, a0);
                    a1 = __builtin_elementwise_fma(w[k / 2 + 1], (f32x2){hv0.z, hv0.w}, a1);
                    a2 = __builtin_elementwise_fma(w[k / 2 + 2], (f32x2){hv1.x, hv1.y}, a2);
                    a3 = __builtin_elementwise_fma(w[k / 2 + 3], (f32x2){hv1.z, hv1.w}, a3);
                }
                const float pre = ((a0[0] + a0[1]) + (a1[0] + a1[1])) + ((a2[0] + a2[1]) + (a3[0] + a3[1]));
                const float act = __builtin_fmaf(__builtin_amdgcn_rcpf(1.f + __expf(act_m * pre)), act_s, act_o);
                *gp = act;
                float e16, o16, ig, gg, fg, og;
                swap16(act, e16, o16);                               // rows (0, 1): (i, f); rows (2, 3): (g, o)
                swap32(e16, ig, gg);
                swap32(o16, fg, og);
                if (writer) *hp = h;
                c = fg * c + ig * gg;
                h = og * tanhf_(c);
                if (writer) { *cp = c; *yp = h; h_lds[(i + 1) & 1][u] = h; }
                gp += g_inc; hp += h_inc; cp += h_inc; yp += y_inc;
                __syncthreads();
            }
        }
    }
    if (writer) hfinal[(size_t)b * (ndir * LH) + dir * LH + u] = h;
}

// Backward through time.  dy [Bd,T,ndir*LH] (may be null), dhfinal [Bd,ndir*LH] (may be null),
// dgates [Bd,T,ndir,LG]: receives d(pre-activation gates) for the valid steps and zeros for the padded ones (t >= len).
// Same ownership as the forward (round 4): wave w owns units [16 w, 16 w + 16), lane l = 16 q + c.  Every lane forms the cell's gradients
// of ITS unit from registers (saved gates / cell states / dy of BCH steps are register-resident while the next chunk's loads are in
// flight), keeps the one of gate q, and publishes it in a double-buffered LDS vector of the 256 gate gradients -- ONE barrier --; then
// lane (q, c) multiplies the 64 gradients of gate type q by column 16 w + c of that gate's W_hh block and two permlane swaps sum the four
// gate types: dh of the previous step, in every lane of the unit, without a second trip through LDS (the earlier form split the gate
// rows over the waves, wrote its partial sums to LDS and summed them after the barrier: two hops per step).
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dhfinal, const float* __restrict__ whh,
                                                       const float* __restrict__ gates, const float* __restrict__ cs, const int* __restrict__ lens,
                                                       float* __restrict__ dgates, int T, int ndir, size_t whh_dir_stride) {
    __shared__ __attribute__((aligned(16))) float dg_lds[2][LG];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int len = lens[b];
    const int wave = j >> 6, lane = j & 63, q = lane >> 4;
    const int k = 16 * wave + (lane & 15);          // this lane's hidden unit
    f32x2 wt[LH / 2];                   // wt[i] = {W_hh[64 q + 2i][k], W_hh[64 q + 2i + 1][k]}: register pairs for v_pk_fma_f32
    const float* wr = whh + dir * whh_dir_stride + (size_t)(q * LH) * LH + k;
#pragma unroll
    for (int i = 0; i < LH; i += 2) wt[i / 2] = (f32x2){wr[(size_t)i * LH], wr[(size_t)(i + 1) * LH]};
    float dh = 0.f, dc = 0.f;
    if (dhfinal) dh = dhfinal[(size_t)b * (ndir * LH) + dir * LH + k];
    const int tstep = dir ? 1 : -1;                         // reverse of the forward processing order
    const int t0 = dir ? 0 : len - 1;
    // r = 0..len-1 counts backward steps; forward step index = len-1-r; time t = t0 + r*tstep
    float vc[BCH][7], vn[BCH][7];
    auto load_chunk = [&](int r0, float (&v)[BCH][7]) {
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = min(r0 + i, len - 1);
            const int t = t0 + r * tstep;
            const size_t rw = ((size_t)b * T + t) * ndir + dir;
            v[i][0] = gates[rw * LG + k]; v[i][1] = gates[rw * LG + LH + k]; v[i][2] = gates[rw * LG + 2 * LH + k]; v[i][3] = gates[rw * LG + 3 * LH + k];
            v[i][4] = cs[rw * LH + k];
            const int rp = min(r + 1, len - 1);             // previous forward step (clamped; masked below for the first step)
            v[i][5] = cs[(((size_t)b * T + (t0 + rp * tstep)) * ndir + dir) * LH + k];
            v[i][6] = dy ? dy[((size_t)b * T + t) * (ndir * LH) + dir * LH + k] : 0.f;
        }
    };
    for (int t = max(len, 0); t < T; ++t) dgates[(((size_t)b * T + t) * ndir + dir) * LG + j] = 0.f;       // padded steps: 1 KB per step
    if (len <= 0) return;                                   // (uniform)
    float* dgr = dgates + (((size_t)b * T + t0) * ndir + dir) * LG + q * LH + k;        // advanced by one time step per iteration
    const ptrdiff_t dg_inc = (ptrdiff_t)tstep * ndir * LG;
    load_chunk(0, vn);
    for (int r0 = 0; r0 < len; r0 += BCH) {
#pragma unroll
        for (int i = 0; i < BCH; ++i)
#pragma unroll
            for (int qq = 0; qq < 7; ++qq) vc[i][qq] = vn[i][qq];
        if (r0 + BCH < len) load_chunk(r0 + BCH, vn);
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = r0 + i;
            if (r < len) {                                  // (uniform; a guard, not a break, so that the chunk unrolls)
            const float ig = vc[i][0], fg = vc[i][1], gg = vc[i][2], og = vc[i][3], ct = vc[i][4];
            const float cprev = (r == len - 1) ? 0.f : vc[i][5];
            const float dht = dh + vc[i][6];
            const float tc = tanhf_(ct);
            const float d_o = dht * tc * og * (1.f - og);
            const float dct = dc + dht * og * (1.f - tc * tc);
            const float d_i = dct * gg * ig * (1.f - ig);
            const float d_f = dct * cprev * fg * (1.f - fg);
            const float d_g = dct * ig * (1.f - gg * gg);
            dc = dct * fg;
            const float mine = q == 0 ? d_i : q == 1 ? d_f : q == 2 ? d_g : d_o;
            float* dgw = dg_lds[i & 1];                     // (BCH is even: the step's parity is i & 1)
            dgw[q * LH + k] = mine;
            *dgr = mine;
            dgr += dg_inc;
            __syncthreads();
            f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};      // four independent chains of 8 packed FMAs
#pragma unroll
            for (int ii = 0; ii < LH; ii += 8) {
                const float4 dv0 = *reinterpret_cast<const float4*>(&dgw[q * LH + ii]);
                const float4 dv1 = *reinterpret_cast<const float4*>(&dgw[q * LH + ii + 4]);
                a0 = __builtin_elementwise_fma(wt[ii / 2], (f32x2){dv0.x, dv0.y}, a0);
                a1 = __builtin_elementwise_fma(wt[ii / 2 + 1], (f32x2){dv0.z, dv0.w}, a1);
                a2 = __builtin_elementwise_fma(wt[ii / 2 + 2], (f32x2){dv1.x, dv1.y}, a2);
                a3 = __builtin_elementwise_fma(wt[ii / 2 + 3], (f32x2){dv1.z, dv1.w}, a3);
            }
            const float part = ((a0[0] + a0[1]) + (a1[0] + a1[1])) + ((a2[0] + a2[1]) + (a3[0] + a3[1]));
            float e16, o16, lo, up;
            swap16(part, e16, o16);
            swap32(e16 + o16, lo, up);
            dh = lo + up;                                   // (i + f) + (g + o): the same sum in every lane of the unit
            }
        }
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

UNAST_DEFINE_RNG_EPOCH_SETTER(lstm)

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

// The recurrent dot products without LDS broadcasts (round 4, second pass).  A lane holds ONE value per group of 16 source elements (one
// ds_read_b32: lane c of every 16-lane row has element 16 G + c), and the 16 values of a row reach every lane of it by DPP row rotations
// folded into the multiply-add (v_fmac_f32_dpp row_ror:n, full rate).  The weight a lane multiplies rotation n with belongs to the lane
// that rotation n reads from; that lane index comes from the same DPP operation applied to the lane id at kernel start (RotSrc), so
// nothing here depends on which way the hardware calls "right".
// One asm block per group: hipcc (ROCm 7.2) neither folds a v_mov_b32_dpp into the multiply-add nor schedules 16 separate asm statements
// without an s_nop between every four.  The first multiply-add is the unrotated one and an s_nop follows it: two wait states between
// whatever wrote `v` and the first DPP read of it (the hazard recogniser does not look inside asm).
template <int N>
__device__ __forceinline__ int row_ror_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x120 + N, 0xf, 0xf, true); }
template <int N>
struct RotSrc { static __device__ __forceinline__ void fill(int c, int (&src)[16]) { src[N] = row_ror_i<N>(c); RotSrc<N + 1>::fill(c, src); } };
template <> struct RotSrc<0> { static __device__ __forceinline__ void fill(int c, int (&src)[16]) { src[0] = c; RotSrc<1>::fill(c, src); } };
template <> struct RotSrc<16> { static __device__ __forceinline__ void fill(int, int (&)[16]) {} };
__device__ __forceinline__ float row_ror8(float v) { return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x128, 0xf, 0xf, true)); }

// acc[n & 3] += w[n] * (v rotated by n) for n = 0..15: four independent chains.  FIRST: the chains START here (acc = the first four products:
// no zero-initialised accumulators -- every instruction of an in-order wave costs an issue slot of 4 cycles, a v_mov as much as a
// multiply-add).  VALU_SRC: `v` was written by a vector instruction (not an LDS read): the unrotated product first and an s_nop give the two
// wait states a DPP read of it needs.
template <bool FIRST, bool VALU_SRC>
__device__ __forceinline__ void dot16(const float (&w)[16], float v, float (&acc)[4]) {
#define D(n, a, wi) "v_fmac_f32_dpp %" #a ", %4, %" #wi " row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define M(n, a, wi) "v_mul_f32_dpp %" #a ", %4, %" #wi " row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define TAIL D(4, 0, 9) D(5, 1, 10) D(6, 2, 11) D(7, 3, 12) D(8, 0, 13) D(9, 1, 14) D(10, 2, 15) D(11, 3, 16) D(12, 0, 17) D(13, 1, 18) D(14, 2, 19) D(15, 3, 20)
#define OPS : "v"(v), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), \
              "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15])
    if constexpr (FIRST && VALU_SRC)
        asm("v_mul_f32_e32 %0, %4, %5\n\ts_nop 0\n\t" M(1, 1, 6) M(2, 2, 7) M(3, 3, 8) TAIL : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]) OPS);
    else if constexpr (FIRST)
        asm("v_mul_f32_e32 %0, %4, %5\n\t" M(1, 1, 6) M(2, 2, 7) M(3, 3, 8) TAIL : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]) OPS);
    else if constexpr (VALU_SRC)
        asm("v_fmac_f32_e32 %0, %4, %5\n\ts_nop 0\n\t" D(1, 1, 6) D(2, 2, 7) D(3, 3, 8) TAIL : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) OPS);
    else
        asm("v_fmac_f32_e32 %0, %4, %5\n\t" D(1, 1, 6) D(2, 2, 7) D(3, 3, 8) TAIL : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) OPS);
#undef D
#undef M
#undef TAIL
#undef OPS
}

#define LOG2E 1.4426950408889634f

// xproj [Bd,T,ndir*LG] (no bias), y [Bd,T,ndir*LH], gates [Bd,T,ndir,LG], cs [Bd,T,ndir,LH], hprev [Bd,T,ndir,LH], hfinal [Bd, ndir*LH].
// y and hprev of the padded steps t >= len are written as zeros here (they are GEMM operands of the next layer / of the weight
// gradients over all Bd*T rows): the caller allocates them uninitialised -- pre-zeroing them with a fill launch each was 26 MB of
// memset per tensor and call at config 3.
//
// Layout (round 4): the four gates of a hidden unit live in ONE wave.  Wave w owns units [16 w, 16 w + 16); lane l = 16 q + c computes gate
// q (i, f, g, o) of unit 16 w + c, i.e. row 64 q + 16 w + c of W_hh (64 floats in VGPRs, in rotation order -- dot16 above; its own wave's h
// comes from the register, the other waves' by one ds_read_b32 each).  After the activation three permlane swaps hand every lane the other
// three gates of its unit, the cell update happens in place (four times redundantly), and the only thing that crosses waves is h: 16 values
// per wave into a double-buffered LDS vector, ONE barrier per step, one LDS round trip on the step's dependent chain.  The 16 multiply-adds
// on the wave's OWN units run in front of the barrier, for the next step (under the LDS write and the wait for the slowest wave): 326 ->
// 283 us at 64 x 800; c, y and hprev leave in one store (lane rows 0, 1, 2) instead of three: -> 278 us.
// Measured and dropped (tools/lstm_knockout.sh): 512 threads with every dot product split over two lanes (32 multiply-adds per
// lane, two waves per SIMD) is SLOWER, 347 against 300 us at 64 x 800 -- a SIMD issues one unpacked fp32 instruction per 4 cycles whichever
// wave it comes from, so the split halves nothing and doubles the activation / swap / store instructions per SIMD.
// Global latency stays off the chain: the projections of FCH steps are register-resident while the next chunk's loads are in flight.
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ whh, const float* __restrict__ b_ih,
                                                       const float* __restrict__ b_hh, const int* __restrict__ lens, float* __restrict__ y,
                                                       float* __restrict__ gates, float* __restrict__ cs, float* __restrict__ hprev,
                                                       float* __restrict__ hfinal, int T, int ndir, size_t whh_dir_stride, size_t bias_dir_stride) {
    __shared__ __attribute__((aligned(16))) float h_lds[2][LH];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int wave = j >> 6, lane = j & 63, q = lane >> 4, c16 = lane & 15;
    const int u = 16 * wave + c16;                     // hidden unit of this lane
    const int row = q * LH + u;                        // its gate row
    const int len = lens[b];
    // w[jw][n] = W_hh[row][16 ((wave + jw) & 3) + src[n]]: the weights of source wave (wave + jw) & 3 in the order the rotations deliver h
    float w[4][16];
    int src[16];
    RotSrc<0>::fill(c16, src);
    const float* wr = whh + dir * whh_dir_stride + (size_t)row * LH;
#pragma unroll
    for (int jw = 0; jw < 4; ++jw)
#pragma unroll
        for (int n = 0; n < 16; ++n) w[jw][n] = wr[16 * ((wave + jw) & 3) + src[n]];
    const float bias = b_ih[dir * bias_dir_stride + row] + b_hh[dir * bias_dir_stride + row];
    const int hs1 = 16 * ((wave + 1) & 3) + c16, hs2 = 16 * ((wave + 2) & 3) + c16, hs3 = 16 * ((wave + 3) & 3) + c16;
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
    // The activated gates leave through a buffer descriptor over this sequence's slab: the per-lane part of the address is fixed for the
    // launch, the time step advances in a scalar register.
    const bool writer = q == 0;                                      // one lane row per wave stores the unit's state
    const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(gates + (size_t)b * T * ndir * LG, 0, (int)((size_t)T * ndir * LG * 4), 0x00020000);
    const uint32_t g_vo = (uint32_t)(dir * LG + row) * 4u;
    int g_so = t0 * ndir * LG * 4;                                   // scalar byte offset of the current time step
    const int g_inc = tstep * ndir * LG * 4;
    // sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 2 / (1 + 2^(-2 x log2 e)) - 1: one branch-free form with per-lane constants (lane row 2 = g)
    const float act_m = (q == 2) ? -2.f * LOG2E : -LOG2E, act_s = (q == 2) ? 2.f : 1.f, act_o = (q == 2) ? -1.f : 0.f;
    // The unit's state leaves in ONE store per step: lane row 0 writes c_t, row 1 y_t = h_t, row 2 hprev_t = h_{t-1} (every lane of the unit
    // has all three), row 3 nothing -- three 16-lane stores cost three issue slots of the in-order wave.
    const size_t st_row0 = ((size_t)b * T + t0) * ndir + dir;
    float* st_p = q == 0 ? cs + st_row0 * LH + u : q == 1 ? y + ((size_t)b * T + t0) * ((size_t)ndir * LH) + dir * LH + u : hprev + st_row0 * LH + u;
    const ptrdiff_t st_inc = (ptrdiff_t)tstep * ndir * LH;
    float xc[FCH], xn[FCH];
    auto load_chunk = [&](int s0, float (&x)[FCH]) {
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            const int st = min(s0 + i, len - 1);                   // clamped: a fixed number of loads per chunk
            x[i] = xp[(size_t)(t0 + st * tstep) * xs];
        }
    };
    load_chunk(0, xn);
    float accn[4] = {0.f, 0.f, 0.f, 0.f};                            // this wave's 16 units' share of the NEXT step's dot product (h = 0 before the first)
    for (int s0 = 0; s0 < len; s0 += FCH) {
#pragma unroll
        for (int i = 0; i < FCH; ++i) xc[i] = xn[i] + bias;          // the only wait for global loads: once per chunk
        if (s0 + FCH < len) load_chunk(s0 + FCH, xn);
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            if (s0 + i < len) {                                      // (uniform; a guard, not a break, so that the chunk unrolls and xc[i] is a register)
                const float* hl = h_lds[i & 1];                      // (FCH is even: the step's parity is i & 1)
                // (LKO_*: timing-only knock-out builds, tools/lstm_knockout.sh; never defined in the shipped library)
#ifdef LKO_LDS
                const float h1 = h, h2 = h, h3 = h;
#else
                const float h1 = hl[hs1], h2 = hl[hs2], h3 = hl[hs3];    // the other waves' units, one value per lane; this wave's are in `h`
#endif
                float acc[4] = {accn[0] + xc[i], accn[1], accn[2], accn[3]};
#ifdef LKO_DOT
                acc[1] = h * w[0][0]; acc[2] = h1 * w[1][0]; acc[3] = h2 * w[2][0] + h3 * w[3][0];
#else
                dot16<false, false>(w[1], h1, acc);
                dot16<false, false>(w[2], h2, acc);
                dot16<false, false>(w[3], h3, acc);
#endif
                const float pre = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#ifdef LKO_ACT
                const float act = __builtin_fmaf(pre, act_s, act_o);
#else
                const float act = __builtin_fmaf(__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(act_m * pre)), act_s, act_o);
#endif
#ifndef LKO_STORES
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(act), g_rs, g_vo, g_so, 0);
#endif
                const float h_before = h;
                float e16, o16, ig, gg, fg, og;
                swap16(act, e16, o16);                               // rows (0, 1): (i, f); rows (2, 3): (g, o)
                swap32(e16, ig, gg);
                swap32(o16, fg, og);
                c = fg * c + ig * gg;
#ifdef LKO_ACT
                h = og * c;
#else
                h = og * __builtin_fmaf(__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f((-2.f * LOG2E) * c)), 2.f, -1.f);
#endif
#ifndef LKO_STORES
                if (q < 3) *st_p = q == 0 ? c : q == 1 ? h : h_before;
                st_p += st_inc;
#endif
                if (writer) h_lds[(i + 1) & 1][u] = h;
                dot16<true, true>(w[0], h, accn);                                // the next step's share of this wave's own units: before the barrier
                g_so += g_inc;
#ifndef LKO_BARRIER
                __syncthreads();
#endif
            }
        }
    }
    if (writer) hfinal[(size_t)b * (ndir * LH) + dir * LH + u] = h;
}

// Backward through time.  dy [Bd,T,ndir*LH] (may be null), dhfinal [Bd,ndir*LH] (may be null),
// dgates [Bd,T,ndir,LG]: receives d(pre-activation gates) for the valid steps and zeros for the padded ones (t >= len).
// 256 threads: wave w owns units [16 w, 16 w + 16), lane l = 16 q + c is gate q (i, f, g, o) of unit 16 w + c.  (The forward's split of a
// dot product over two lanes does not carry over: here a 16-lane row must hold ONE gate type -- the rotations hand every lane of a row the
// same 16 gate gradients -- and 16 different units, which leaves four rows = four gate types per unit in a wave and nothing to split
// without a second trip through LDS.)  Every lane forms the cell's gradients of ITS unit from registers (saved gates / cell states / dy of
// BCH steps are register-resident while the next chunk's loads are in flight; everything that does not depend on dh / dc is computed
// ahead of the step), keeps the one of gate q, and publishes it in a double-buffered LDS vector of the 256 gate gradients -- ONE barrier --;
// then lane (q, c) multiplies the 64 gradients of gate type q by column 16 w + c of that gate's W_hh block (its own wave's 16 from the
// register, the other waves' by one ds_read_b32 each, all by row rotations) and two permlane swaps sum the four gate types: dh of the
// previous step, in every lane of the unit.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dhfinal, const float* __restrict__ whh,
                                                       const float* __restrict__ gates, const float* __restrict__ cs, const int* __restrict__ lens,
                                                       float* __restrict__ dgates, int T, int ndir, size_t whh_dir_stride) {
    __shared__ __attribute__((aligned(16))) float dg_lds[2][LG];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int len = lens[b];
    const int wave = j >> 6, lane = j & 63, q = lane >> 4, c16 = lane & 15;
    const int k = 16 * wave + c16;                  // this lane's hidden unit
    // wt[jw][n] = W_hh[64 q + 16 ((wave + jw) & 3) + src[n]][k]: column k of gate block q, in the order the rotations deliver the gate gradients
    float wt[4][16];
    int src[16];
    RotSrc<0>::fill(c16, src);
    const float* wr = whh + dir * whh_dir_stride + (size_t)(q * LH) * LH + k;
#pragma unroll
    for (int jw = 0; jw < 4; ++jw)
#pragma unroll
        for (int n = 0; n < 16; ++n) wt[jw][n] = wr[(size_t)(16 * ((wave + jw) & 3) + src[n]) * LH];
    const int ds1 = q * LH + 16 * ((wave + 1) & 3) + c16, ds2 = q * LH + 16 * ((wave + 2) & 3) + c16, ds3 = q * LH + 16 * ((wave + 3) & 3) + c16;
    float dh = 0.f, dc = 0.f;
    if (dhfinal) dh = dhfinal[(size_t)b * (ndir * LH) + dir * LH + k];
    const int tstep = dir ? 1 : -1;                         // reverse of the forward processing order
    const int t0 = dir ? 0 : len - 1;
    // r = 0..len-1 counts backward steps; forward step index = len-1-r; time t = t0 + r*tstep
    // Saved state through buffer descriptors over this sequence's slabs: the per-lane part of every address is fixed for the launch, the
    // time step is a scalar offset -- the 64-bit address arithmetic of 7 loads was a quarter of the step's vector instructions, and an
    // in-order wave pays 4 cycles for each.  dy = NULL becomes a descriptor of zero records: its loads return 0.
    const __amdgpu_buffer_rsrc_t gl_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gates) + (size_t)b * T * ndir * LG, 0, (int)((size_t)T * ndir * LG * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t cl_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cs) + (size_t)b * T * ndir * LH, 0, (int)((size_t)T * ndir * LH * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yl_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy ? dy : cs) + (size_t)b * T * ndir * LH, 0, dy ? (int)((size_t)T * ndir * LH * 4) : 0, 0x00020000);
    const int vo_g = (dir * LG + k) * 4, vo_c = (dir * LH + k) * 4;
    auto ldf = [](const __amdgpu_buffer_rsrc_t& rs, int vo, int so) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0)); };
    float vn[BCH][6], cnext;                                // gates i f g o, c_t, dy_t of BCH steps; c of the step after them
    // The steps beyond the sequence are not clamped to its last one: their loads leave the slab at its end (the range check returns 0), stop
    // at its first row, or read padded rows of it, and nothing computed from them is used.  Offsets advance by scalar adds.
    const int g_incb = tstep * ndir * LG * 4, c_incb = tstep * ndir * LH * 4;
    auto load_chunk = [&](int r0) {
        int so_g = (t0 + r0 * tstep) * ndir * LG * 4, so_c = (t0 + r0 * tstep) * ndir * LH * 4;
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int sg = max(so_g, 0), sc = max(so_c, 0);        // (the forward direction walks down past t = 0: no negative offsets)
            vn[i][0] = ldf(gl_rs, vo_g, sg); vn[i][1] = ldf(gl_rs, vo_g + LH * 4, sg); vn[i][2] = ldf(gl_rs, vo_g + 2 * LH * 4, sg); vn[i][3] = ldf(gl_rs, vo_g + 3 * LH * 4, sg);
            vn[i][4] = ldf(cl_rs, vo_c, sc);
            vn[i][5] = ldf(yl_rs, vo_c, sc);
            so_g += g_incb; so_c += c_incb;
        }
        cnext = ldf(cl_rs, vo_c, max(so_c, 0));
    };
    for (int t = max(len, 0); t < T; ++t) dgates[(((size_t)b * T + t) * ndir + dir) * LG + j] = 0.f;       // padded steps: 1 KB per step
    if (len <= 0) return;                                   // (uniform)
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(dgates + (size_t)b * T * ndir * LG, 0, (int)((size_t)T * ndir * LG * 4), 0x00020000);
    const uint32_t d_vo = (uint32_t)(dir * LG + q * LH + k) * 4u;
    int d_so = t0 * ndir * LG * 4;                          // scalar byte offset of the current time step
    const int d_inc = tstep * ndir * LG * 4;
    const bool q0 = q == 0, q1 = q == 1, q2 = q == 2, is_o = q == 3;
    load_chunk(0);
    for (int r0 = 0; r0 < len; r0 += BCH) {
        // Per step, off the dependent chain: with dht = dh + dy and dct = dc + dht B, the gate gradient of this lane is X Y where
        // X = dht (gate o) or dct (gates i, f, g) and Y = tc og (1 - og) (o), gg ig (1 - ig) (i), cprev fg (1 - fg) (f) or ig (1 - gg gg) (g):
        // Y = m1 m2 (1 - m2), or m1 (1 - m2 m2) for g, with the operands picked per lane row; dc = dct fg.
        float cy[BCH], cB[BCH], cf[BCH], cd[BCH];
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const float ig = vn[i][0], fg = vn[i][1], gg = vn[i][2], og = vn[i][3];
            const float cafter = i + 1 < BCH ? vn[i + 1][4] : cnext;           // c of the previous FORWARD step = the next backward step's
            const float cprev = (r0 + i >= len - 1) ? 0.f : cafter;           // (scalar condition; steps beyond the sequence are never used)
            const float tc = tanhf_(vn[i][4]);
            cB[i] = og * (1.f - tc * tc);
            cf[i] = fg;
            cd[i] = vn[i][5];
            const float m1 = q0 ? gg : q1 ? cprev : q2 ? ig : tc;
            const float m2 = q0 ? ig : q1 ? fg : q2 ? gg : og;
            const float ya = m1 * m2, yb = 1.f - m2;                          // m1 m2 (1 - m2)
            const float yc = 1.f - m2 * m2;                                   // m1 (1 - m2 m2)
            cy[i] = q2 ? m1 * yc : ya * yb;
        }
        if (r0 + BCH < len) load_chunk(r0 + BCH);
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = r0 + i;
            if (r < len) {                                  // (uniform; a guard, not a break, so that the chunk unrolls)
            const float dht = dh + cd[i];
            const float dct = __builtin_fmaf(dht, cB[i], dc);
            dc = dct * cf[i];
            const float mine = (is_o ? dht : dct) * cy[i];
            float* dgw = dg_lds[i & 1];                     // (BCH is even: the step's parity is i & 1)
            dgw[q * LH + k] = mine;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mine), d_rs, d_vo, d_so, 0);
            d_so += d_inc;
            float acc[4];
            dot16<true, true>(wt[0], mine, acc);            // this wave's own 16 gradients: in front of the barrier, under the LDS write
            __syncthreads();
            const float d1 = dgw[ds1], d2 = dgw[ds2], d3 = dgw[ds3];     // gate type q of the other waves' units
            dot16<false, false>(wt[1], d1, acc);
            dot16<false, false>(wt[2], d2, acc);
            dot16<false, false>(wt[3], d3, acc);
            float e16, o16, lo, up;
            swap16((acc[0] + acc[1]) + (acc[2] + acc[3]), e16, o16);
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

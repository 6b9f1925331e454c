// Row-panel GEMM for gfx950: C[M,N] = epilogue(A[M,K] * W[N,K]^T) for the K <= 256 contractions of the UNAST transformer layers
// (in-projections, out-projections, FFN linear1, cross-attention projections, the FFN linear2 input gradient; call sites
// src/module.py:273-274, 286-287 -> torch.nn.TransformerEncoderLayer / DecoderLayer).
//
// Why a second GEMM kernel.  With d = 256 the general tile kernel (gemm.hip) lives 8 k-steps per 128x128 tile: a workgroup is a
// cold prologue, 8 steps of load -> convert -> LDS -> barrier -> MFMA and a 64 KB epilogue, it re-reads and re-splits its A panel
// once per column tile, and on every launch all workgroups stream their C tiles at the same time.  Here the ACTIVATION is the
// stationary operand:
//   * a workgroup owns a panel of 128 (or 64) rows for ALL N columns; each wave loads the K <= 256 values of its 32 (16) rows
//     ONCE, splits them into hi/lo bf16 MFMA fragments and keeps them in registers (128 VGPRs) for the whole launch;
//   * the weights stream through a 4-slot LDS ring by LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no conversion): they
//     live in HBM a second time as TILED bf16 PLANES (unast_retile_weights below) -- per matrix a hi plane and a lo plane, cut
//     into 1-KB sub-tiles of 16 columns x 32 k laid out exactly as one wave's ds_read_b128 fragment read wants them
//     ([k-chunk g][column l15][8 k]); a DMA instruction copies one sub-tile, source and destination both linear;
//   * the ring holds two 64-column groups (64 KB each at K = 256, both planes): one raw s_barrier per group, the DMA of group g+1 is in
//     flight while group g is multiplied (counted s_waitcnt vmcnt, never 0 inside the loop), and C leaves through the epilogue of
//     group g-1, which every wave runs INSIDE group g -- half of a SIMD's waves before its first k-step, the other half in its middle.
// LayerNorm epilogue (LN = 1, N = 256): the workgroup holds complete rows, so y = LayerNorm(x + dropout(A W^T + b)) is finished
// here -- z, y, mean and rstd are written by the GEMM and the stand-alone LayerNorm launch disappears (post-LN sub-layers,
// SURVEY.md Appendix A).
#include "common.h"
#include <algorithm>
#include "../../include/unast_hip.h"
#include <type_traits>

static unsigned long long* g_panel_stamps = nullptr;       // diagnostic builds (-DPANEL_STAMPS): device buffer for per-wave cycle sums
#define PSUB 1024                  // bytes of one sub-tile: 16 columns x 32 k of bf16

struct PanelParams {
    const float* A; int lda;
    const unsigned char* W; size_t plane_bytes;     // tiled hi plane of this matrix (lo plane at + plane_bytes)
    float* C; int ldc;
    int M, N, K, ncg;                               // ncg = column groups of 64 (planes are zero-padded to them)
    const float* bias; const float* R; int ldr;
    const float* G; int ldg; float gate_scale;
    int act; uint32_t drop_thresh; float drop_scale; uint32_t seed, stream;
    int out_split;
    const float* gamma; const float* beta; float* Y; int ldy; float* mean; float* rstd; float eps;
    const float* Z; int ldz; float* part;           // LayerNorm-BACKWARD epilogue of the K-streamed kernel (kpanel_kernel<2>): saved pre-norm rows, column partials
    unsigned long long* gate_bits;                  // keep bits of an [M,N] activation: written when act == 1, read (as the gate) otherwise
    unsigned long long* stamps;                     // diagnostic build only (-DPANEL_STAMPS): per-wave cycle sums per loop phase
};

__device__ __forceinline__ void panel_wait_vm(int n) {
    // counted wait: every vector-memory operation of this wave except its n youngest is complete (n is wave-uniform).
    // A smaller count than necessary only waits longer, so odd counts are rounded down; a binary tree keeps it to 4 scalar branches.
#define PW(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
    if (n < 8) {
        if (n < 4) { if (n < 2) PW(0); else PW(2); }
        else       { if (n < 6) PW(4); else PW(6); }
    } else if (n < 16) {
        if (n < 12) { if (n < 10) PW(8); else PW(10); }
        else        { if (n < 14) PW(12); else PW(14); }
    } else {
        if (n < 20) PW(16); else PW(20);
    }
#undef PW
}

#ifdef PANEL_STAMPS
__device__ __forceinline__ unsigned long long panel_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define STAMP(var) const unsigned long long var = panel_stamp()
#define STAMP_ADD(slot, a, b) stamp_sum[slot] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

// KSTEPS = ceil(K / 32) (weights are zero-padded to it), RT = 16-row tiles per wave (2: 128-row panels, 1: 64-row panels),
// LN = 1: LayerNorm epilogue over N = 256 (all four column groups stay in the accumulators).
// Waves: wave = 2 * rg + ch owns rows [16 RT rg, 16 RT (rg + 1)) of the panel and, of every 64-column group, columns [32 ch, 32 ch + 32).
template <int KSTEPS, int RT, int LN, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void panel_kernel(const PanelParams p) {
    constexpr int NT = 64 * NW, ROWS = (NW / 2) * 16 * RT;      // NW = 8 or 16 waves: NW / 2 row groups x 2 column halves
    constexpr int NSUB = 8 * KSTEPS;                      // sub-tiles per 64-column group: 2 planes x 4 column tiles x KSTEPS
    constexpr int GROUP = NSUB * PSUB;                    // bytes of one group of weights (64 KB at K = 256)
    constexpr int DMAX = (NSUB + NW - 1) / NW;            // DMA instructions per wave and group (at most)
    constexpr int RING = 2 * GROUP;                       // the group being multiplied + the group in flight
    constexpr bool A_DMA = (KSTEPS == 8);                 // K = 256: the panel's rows come in by LDS-DMA, 1 KB per row and instruction
    static_assert(!A_DMA || RING >= ROWS * 1024, "the A image borrows the ring");
    __shared__ __attribute__((aligned(16))) unsigned char smem[(RING > ROWS * 1024 || !A_DMA ? RING : ROWS * 1024) + 4096 + 1024];      // ring | bias (N <= 1024) | LayerNorm row sums

    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    __builtin_assume(wave >= 0 && wave < NW);
    const int rg = wave >> 1, ch = wave & 1;
    const int m_wg = blockIdx.x * ROWS;
    const int m_wave = m_wg + rg * (16 * RT);
#ifdef PANEL_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // 0 wait, 1 barrier, 2 DMA issue, 3 epilogue, 4 MFMA phase, 5 prologue, 6 total
    const unsigned long long t_begin = panel_stamp();
#endif
    // dropout keys: rng_row_key(seed, stream, m) = pcg(m + rbase); rbase reads the RNG epoch from memory, once, before the ring starts
    const uint32_t rbase = p.drop_thresh ? rng_stream_base(p.seed, p.stream) : 0u;

    // LDS-DMA through buffer descriptors: the per-lane part of every address is the constant lane * 16 (one VGPR for the whole launch), the
    // rest is a scalar offset -- a DMA is then M0 + one instruction (global_load_lds with 64-bit per-lane addresses: ~110 cycles each,
    // 900 of a group's 7 000 by the s_memtime stamps)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.W), 0, 0x7FFFFFF0, 0x00020000);
    const int lane16 = lane * 16;
    auto issue_group = [&](int cg) {
#pragma unroll
        for (int q = 0; q < DMAX; ++q) {
            const int j = wave + NW * q;                              // this wave's sub-tile of the group: [plane][ct][ks]
            if (j >= NSUB) break;
            const int pl = j / (4 * KSTEPS), rem = j - pl * (4 * KSTEPS);          // rem = ct * KSTEPS + ks
            const uint32_t soff = (uint32_t)pl * (uint32_t)p.plane_bytes + (uint32_t)(4 * cg * KSTEPS + rem) * PSUB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(smem + (cg & 1) * GROUP + j * PSUB), 16, lane16, (int)soff, 0, 0);
        }
    };

    float* const sbias = reinterpret_cast<float*>(smem + RING);
    float* const sred = sbias + 1024;                     // LayerNorm: row sums of the partner wave
    const uint32_t sbias_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)sbias;       // LDS byte address
    const bool bias_lds = p.bias && p.N <= 1024;          // bias through LDS: an ordinary global load inside the loop would make hipcc drain the DMA ring
    if (bias_lds) {
        for (int i = t; i < 64 * p.ncg; i += NT) sbias[i] = (i < p.N) ? p.bias[i] : 0.f;
    }

    // ---- the panel's rows as MFMA fragments: lane holds A[m_wave + 16 rt + l15][32 ks + 8 g .. + 7], split once ------------
    bf16x8_t ah[RT][KSTEPS], al[RT][KSTEPS];
    if constexpr (A_DMA) {
        // Full 1-KB rows by LDS-DMA into the (still empty) ring: fragment-shaped register loads touch every 128-byte line of the
        // panel twice, 64 bytes at a time, and measured 15 of the kernel's 26 us at N = 256.  Row r of the panel sits at r * 1024; its
        // 16-byte chunk c is stored at position c ^ (r & 15) (source-side swizzle: LDS-DMA writes lane-linear), which makes the
        // fragment reads below conflict-free under ds_read_b128's lane groups.
        const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
        for (int q = 0; q < ROWS / NW; ++q) {
            const int r = wave * (ROWS / NW) + q;                     // wave-uniform
            const int row = min(m_wg + r, p.M - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(smem + r * 1024), 16, (lane ^ (r & 15)) << 4,
                                                     (int)((uint32_t)row * (uint32_t)p.lda * 4u), 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned char* rowp = smem + (rg * (16 * RT) + 16 * rt + l15) * 1024;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int c0 = 8 * ks + 2 * g;
                const float4 a = *reinterpret_cast<const float4*>(rowp + ((c0 ^ l15) << 4));
                const float4 c = *reinterpret_cast<const float4*>(rowp + (((c0 + 1) ^ l15) << 4));
                u32x2 h0, l0, h1, l1;
                split4<3>(a, h0, l0);
                split4<3>(c, h1, l1);
                ah[rt][ks] = __builtin_bit_cast(bf16x8_t, (u32x4){h0[0], h0[1], h1[0], h1[1]});
                al[rt][ks] = __builtin_bit_cast(bf16x8_t, (u32x4){l0[0], l0[1], l1[0], l1[1]});
            }
        }
        __syncthreads();                                  // every wave has its fragments (and the bias words are visible): the ring is free
    } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = min(m_wave + 16 * rt + l15, p.M - 1);
            const float* ar = p.A + (size_t)row * p.lda;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int k = 32 * ks + 8 * g;
                const int k0 = min(k, p.K - 4), k1 = min(k + 4, p.K - 4);
                float4 a = *reinterpret_cast<const float4*>(ar + k0);
                float4 c = *reinterpret_cast<const float4*>(ar + k1);
                if (k >= p.K) a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + 4 >= p.K) c = make_float4(0.f, 0.f, 0.f, 0.f);
                u32x2 h0, l0, h1, l1;
                split4<3>(a, h0, l0);
                split4<3>(c, h1, l1);
                ah[rt][ks] = __builtin_bit_cast(bf16x8_t, (u32x4){h0[0], h0[1], h1[0], h1[1]});
                al[rt][ks] = __builtin_bit_cast(bf16x8_t, (u32x4){l0[0], l0[1], l1[0], l1[1]});
            }
        }
        __syncthreads();                                  // the bias words are visible to every wave (a raw s_barrier orders nothing by itself)
    }
    issue_group(0);
#ifdef PANEL_STAMPS
    { const unsigned long long tp = panel_stamp(); stamp_sum[5] = tp - t_begin; }
#endif

    const bool vec_c = (p.ldc & 3) == 0;
    const bool vec_r = p.R && (p.ldr & 3) == 0 && ((((uintptr_t)p.R) & 15) == 0);
    const bool vec_g = p.G && (p.ldg & 3) == 0 && ((((uintptr_t)p.G) & 15) == 0);
    const int lane_off = (g * 16 + l15) * 16;

    // General epilogue of one 64-column group: lane holds C[m = m_wave + 16 rt + l15][n = 64 cg + 32 ch + 16 c2 + 4 g .. + 3]
    auto epilogue = [&](int cg, f32x4 (&ac)[RT][2]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = m_wave + 16 * rt + l15;
            if (m >= p.M) continue;
            uint32_t rkey = 0;
            if (p.drop_thresh) rkey = pcg_hash((uint32_t)m + rbase);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int n = 64 * cg + 32 * ch + 16 * c2 + 4 * g;
                if (n >= p.N) continue;
                float v[4] = {ac[rt][c2][0], ac[rt][c2][1], ac[rt][c2][2], ac[rt][c2][3]};
                float bv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f}, rv[4] = {0.f, 0.f, 0.f, 0.f};
                const bool full = n + 3 < p.N;
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = (n + r < p.N) ? p.bias[n + r] : 0.f;
                }
                if (p.G) {
                    if (full && vec_g) { const float4 t4 = *reinterpret_cast<const float4*>(p.G + (size_t)m * p.ldg + n); gv[0] = t4.x; gv[1] = t4.y; gv[2] = t4.z; gv[3] = t4.w; }
                    else { for (int r = 0; r < 4; ++r) if (n + r < p.N) gv[r] = p.G[(size_t)m * p.ldg + n + r]; }
                }
                if (p.R) {
                    if (full && vec_r) { const float4 t4 = *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n); rv[0] = t4.x; rv[1] = t4.y; rv[2] = t4.z; rv[3] = t4.w; }
                    else { for (int r = 0; r < 4; ++r) if (n + r < p.N) rv[r] = p.R[(size_t)m * p.ldr + n + r]; }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = v[r] + bv[r];
                    if (p.act == 1) x = fmaxf(x, 0.f);
                    if (p.drop_thresh) x = rng_keep(rkey, (uint32_t)(n + r), p.drop_thresh) ? x * p.drop_scale : 0.f;
                    if (p.G) x = (gv[r] > 0.f) ? x * p.gate_scale : 0.f;
                    v[r] = x + rv[r];
                }
                float* cp = p.C + (size_t)m * p.ldc + n;
                if (full && vec_c) {
                    const float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (p.out_split) *reinterpret_cast<uint4*>(cp) = split_chunk(o);
                    else *reinterpret_cast<float4*>(cp) = o;
                } else {
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = v[r];
                }
            }
        }
    };
    // Fast epilogue (whole 64-column group inside N, 16-byte row strides, no residual / gate operand, bias in LDS): exactly one
    // 16-byte store per (row tile, column tile) and NO global load, so the wave's vmcnt bookkeeping below stays exact.  It runs beside
    // the partner wave's MFMAs, which hold the SIMD's vector issue port half of the time: every instruction here costs ~8 cycles, and
    // measured with s_memtime stamps the first version (run-time flags tested per tile, 64-bit address arithmetic and an EXEC mask per
    // store, ~200 instructions) took 1 900 cycles per group against 2 x 770 cycles of MFMAs.  Hence: one specialised body per flag
    // combination, buffer stores (per-row 32-bit offsets computed once per launch, the group's offset in an SGPR, rows >= M dropped by
    // the descriptor's range check) and the bias pre-loaded by the caller.
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)min((size_t)0x7FFFFFF0, ((size_t)(p.M - 1) * p.ldc + p.N) * 4), 0x00020000);
    uint32_t c_voff[RT], c_rkey[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m_wave + 16 * rt + l15;
        c_voff[rt] = ((uint32_t)m * (uint32_t)p.ldc + (uint32_t)(32 * ch + 4 * g)) * 4u;
        c_rkey[rt] = pcg_hash((uint32_t)m + rbase);
    }
    // Keep bits (gate_bits): word ((m / 16) * (N / 16) + n / 16) * 4 + r holds, at bit 16 g + l15, whether element (m = 16 (m/16) + l15,
    // n = 16 (n/16) + 4 g + r) is > 0 -- i.e. one __ballot per accumulator register of a 16 x 16 tile.  linear1 writes them next to its
    // output (WBITS); the input-gradient GEMM through linear2 reads them with SCALAR loads (RGATE) instead of streaming the hidden
    // activation as a gate operand (105 MB per launch at config 3, and vector loads inside the ring's loop).
    const int ntile_n = (p.N + 15) >> 4, ntile_m = (p.M + 15) >> 4;
    // The buffer holds whole 128-row panels (ceil128(M) rows): a partial panel's rows beyond M get their words written too, and num_records
    // below covers exactly that.  (Round 3 first allocated the buffer for M rows while this descriptor already spanned whole panels: the
    // words of rows >= M were IN range of the descriptor and landed behind the allocation -- the fault of gpurun_out/r3/tpanel2.log.  The
    // note that stood here blamed the hardware ("8-byte buffer stores beyond num_records are not dropped"); tools/buffer_oob_probe.cpp shows
    // gfx950 drops 4-, 8- and 16-byte stores beyond num_records alike, byte-exactly, with the base in the vector or the scalar offset.)
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.gate_bits, 0, ((p.M + 127) >> 7) * 8 * ntile_n * 32, 0x00020000);
    auto epilogue_fast = [&](int cg, f32x4 (&ac)[RT][2], auto act_t, auto drop_t, auto split_t, auto wbits_t, auto rgate_t) {
        constexpr bool ACT = decltype(act_t)::value, DROP = decltype(drop_t)::value, SPLIT = decltype(split_t)::value;
        constexpr bool WBITS = decltype(wbits_t)::value, RGATE = decltype(rgate_t)::value;
        // (inline asm: before an ordinary LDS read hipcc waits vmcnt(0) for the LDS-DMA it believes may alias it, which would
        // drain the ring once per column group; the bias words were written before the first barrier)
        f32x4 b4[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (bias_lds) {
            const uint32_t ba = sbias_lds + 4u * (uint32_t)(64 * cg + 32 * ch + 4 * g);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b4[0]), "=&v"(b4[1]) : "v"(ba) : "memory");
        }
        const int soff = cg * 256;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                float v[4] = {ac[rt][c2][0] + b4[c2][0], ac[rt][c2][1] + b4[c2][1], ac[rt][c2][2] + b4[c2][2], ac[rt][c2][3] + b4[c2][3]};
                if (ACT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) asm("v_max_f32 %0, 0, %1" : "=v"(v[r]) : "v"(v[r]));     // (fmaxf adds a canonicalising v_max per value)
                }
                if (DROP) {
                    const uint32_t n = (uint32_t)(64 * cg + 32 * ch + 16 * c2 + 4 * g);
                    const uint32_t h01 = rng_pair(c_rkey[rt], n), h23 = rng_pair(c_rkey[rt], n + 2u);
                    v[0] = rng_keep_lo(h01, p.drop_thresh) ? v[0] * p.drop_scale : 0.f;
                    v[1] = rng_keep_hi(h01, p.drop_thresh) ? v[1] * p.drop_scale : 0.f;
                    v[2] = rng_keep_lo(h23, p.drop_thresh) ? v[2] * p.drop_scale : 0.f;
                    v[3] = rng_keep_hi(h23, p.drop_thresh) ? v[3] * p.drop_scale : 0.f;
                }
                if (WBITS || RGATE) {
                    const int tile_m = RGATE ? min((m_wave >> 4) + rt, ntile_m - 1) : (m_wave >> 4) + rt;       // (reads of rows beyond M stay inside the buffer)
                    const int widx = ((tile_m * ntile_n) + 4 * cg + 2 * ch + c2) * 4;        // wave-uniform
                    if (WBITS) {
                        const unsigned long long b0 = __ballot(v[0] > 0.f), b1 = __ballot(v[1] > 0.f), b2 = __ballot(v[2] > 0.f), b3 = __ballot(v[3] > 0.f);
                        const unsigned long long mine = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
                        // lanes 0-3 store the four words (rows >= M of a partial panel still have their tile inside the buffer: it is sized for ceil16(M))
                        if (lane < 4) __builtin_amdgcn_raw_buffer_store_b64((u32x2){(uint32_t)mine, (uint32_t)(mine >> 32)}, g_rsrc, (widx + lane) * 8, 0, 0);
                    } else {
                        // (constant address space + wave-uniform index = s_load: no vector-memory operation, nothing for the ring's vmcnt waits to see)
                        const __attribute__((address_space(4))) unsigned long long* wp = (const __attribute__((address_space(4))) unsigned long long*)(uintptr_t)p.gate_bits + widx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ((wp[r] >> lane) & 1ull) ? v[r] * p.gate_scale : 0.f;
                    }
                }
                u32x4 o;
                if (SPLIT) {
                    const uint4 sc = split_chunk(make_float4(v[0], v[1], v[2], v[3]));
                    o = (u32x4){sc.x, sc.y, sc.z, sc.w};
                } else {
                    o = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                }
                // (the group's offset goes into the VGPR offset, not into soffset: with an SGPR soffset the 16-byte store reads its data
                // registers late, and hipcc -- ROCm 7.2 -- let the next v_pk_add overwrite them: 5 % wrong elements in tile (rt 0, c2 0))
                __builtin_amdgcn_raw_buffer_store_b128(o, c_rsrc, (int)(c_voff[rt] + (uint32_t)soff) + 64 * c2, 0, 0);
            }
        }
    };
    // stores per fast epilogue: one buffer store per (row tile, column tile), issued whatever the rows (the range check drops rows >= M)
    const int s_fast = 2 * RT;
    int st_prev = 0, st_now = 0;                          // buffer stores issued in the previous group iteration / in this one
    const bool fast_ok = vec_c && !p.R && !p.G && (bias_lds || !p.bias);
    // which specialised body serves this launch (wave-uniform, fixed for the launch): 0 = general
    const int epi_kind = !fast_ok ? 0 : p.gate_bits ? (p.act == 1 ? (p.out_split ? 0 : p.drop_thresh ? 5 : 7) : (!p.drop_thresh && !p.out_split && !p.bias ? 6 : 0))
                       : p.drop_thresh ? (p.act == 1 && !p.out_split ? 1 : 0) : p.out_split ? (p.act == 1 ? 0 : 2) : (p.act == 1 ? 3 : 4);
    const int s_epi = s_fast + (epi_kind == 5 || epi_kind == 7 ? 2 * RT : 0);      // kinds 5 / 7 add one keep-bit store per tile
    auto run_epilogue = [&](int cg, f32x4 (&ac)[RT][2]) {
        STAMP(te0);
        using T = std::true_type; using F = std::false_type;
#ifdef PKO_PLAIN_EPILOGUE
        if (64 * cg + 64 <= p.N) { epilogue_fast(cg, ac, F{}, F{}, F{}, F{}, F{}); st_now += s_fast; return; }
#endif
#ifdef PKO_NO_EPILOGUE
        if (64 * cg + 64 <= p.N) { asm volatile("" :: "v"(ac[0][0]), "v"(ac[0][1])); return; }
#endif
        if (epi_kind == 0 || 64 * cg + 64 > p.N) {
            epilogue(cg, ac);
        } else {
            __builtin_amdgcn_s_setprio(1);                 // the SIMD's other waves are in their MFMA phase and wait for this one at the next barrier
            if (epi_kind == 1) epilogue_fast(cg, ac, T{}, T{}, F{}, F{}, F{});
            else if (epi_kind == 2) epilogue_fast(cg, ac, F{}, F{}, T{}, F{}, F{});
            else if (epi_kind == 3) epilogue_fast(cg, ac, T{}, F{}, F{}, F{}, F{});
            else if (epi_kind == 5) epilogue_fast(cg, ac, T{}, T{}, F{}, T{}, F{});
            else if (epi_kind == 6) epilogue_fast(cg, ac, F{}, F{}, F{}, F{}, T{});
            else if (epi_kind == 7) epilogue_fast(cg, ac, T{}, F{}, F{}, T{}, F{});
            else epilogue_fast(cg, ac, F{}, F{}, F{}, F{}, F{});
            __builtin_amdgcn_s_setprio(0);
            st_now += s_epi;
        }
        STAMP(te1);
        STAMP_ADD(3, te0, te1);
    };

    // group cg of the ring: wait for it, free the other half, refill that half with group cg + 1.
    // ONE barrier per 64-column group (96 MFMAs per wave at K = 256, RT = 2): with a barrier per 32-KB piece every wave's epilogue sat on
    // the critical path of its piece (s_memtime stamps: 950 of 4 000 cycles per piece parked at the barrier).
    auto group_sync = [&](int cg) {
        // Group cg's DMAs were issued at the start of iteration cg - 1; younger than them are only the stores of the epilogue that ran
        // in that iteration (group cg + 1 has not been requested yet).
        STAMP(ta);
        panel_wait_vm(st_prev);
        STAMP(tb);
        __builtin_amdgcn_s_barrier();                     // every wave's share of group cg is in LDS, and nobody reads group cg - 1 any more
        STAMP(tc);
        if (cg + 1 < p.ncg) issue_group(cg + 1);
        STAMP(td);
        STAMP_ADD(0, ta, tb); STAMP_ADD(1, tb, tc); STAMP_ADD(2, tc, td);
    };
    // k-steps [K0, K1) of group cg multiplied into `ac`
    auto group_mma = [&](int cg, auto k0_t, auto k1_t, f32x4 (&ac)[RT][2]) {
        constexpr int K0 = decltype(k0_t)::value, K1 = decltype(k1_t)::value;
        if constexpr (K0 < K1) {
            const unsigned char* base = smem + (cg & 1) * GROUP + lane_off + (2 * ch) * KSTEPS * PSUB;
            // weight fragments one k-step ahead of the MFMAs that use them: [buffer][column tile][hi, lo]
            bf16x8_t bf[2][2][2];
            auto load_b = [&](int buf, int ks) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    bf[buf][c2][0] = *reinterpret_cast<const bf16x8_t*>(base + ((0 * 4 + c2) * KSTEPS + ks) * PSUB);
                    bf[buf][c2][1] = *reinterpret_cast<const bf16x8_t*>(base + ((1 * 4 + c2) * KSTEPS + ks) * PSUB);
                }
            };
            STAMP(tm0);
            load_b(K0 & 1, K0);
#ifdef PKO_NO_LDS_READS
            load_b((K0 + 1) & 1, K0);
#endif
#pragma unroll
            for (int ks = K0; ks < K1; ++ks) {
#ifndef PKO_NO_LDS_READS
                if (ks + 1 < K1) load_b((ks + 1) & 1, ks + 1);
#endif
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const bf16x8_t bh = bf[ks & 1][c2][0], bl = bf[ks & 1][c2][1];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        // weights as the MFMA "A" (rows = n): each lane ends up with 4 consecutive n of row m = l15
#ifndef PKO_ONE_MFMA        // (PKO_*: timing-only knock-out builds, tools/panel_knockout.sh; never defined in the shipped library)
                        ac[rt][c2] = mfma16(bl, ah[rt][ks], ac[rt][c2]);
                        ac[rt][c2] = mfma16(bh, al[rt][ks], ac[rt][c2]);
#endif
                        ac[rt][c2] = mfma16(bh, ah[rt][ks], ac[rt][c2]);
                    }
                }
            }
#ifdef PANEL_STAMPS
            asm volatile("" :: "v"(ac[0][0]), "v"(ac[0][1]));      // the stamp below is taken when the last MFMA has ISSUED, not retired
#endif
            STAMP(tm1);
            STAMP_ADD(4, tm0, tm1);
        }
    };
    auto end_iteration = [&]() { st_prev = st_now; st_now = 0; };
    constexpr int KH = (KSTEPS + 1) / 2;
    using I0 = std::integral_constant<int, 0>; using IH = std::integral_constant<int, KH>; using IK = std::integral_constant<int, KSTEPS>;
    auto zero = [&](f32x4 (&ac)[RT][2]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { ac[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac[rt][1] = ac[rt][0]; }
    };

    if constexpr (LN) {
        f32x4 acc[4][RT][2];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            zero(acc[cg]);
            group_sync(cg);
            group_mma(cg, I0{}, IK{}, acc[cg]);
            end_iteration();
        }
        // z = R + dropout(acc + bias); y = LayerNorm(z) over the 256 columns of the row: this lane holds 32 of them, the three
        // other lane groups g of the wave 96 more, the partner wave (ch ^ 1) the other 128.
        float zv[RT][4][2][4];
        float s1[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = min(m_wave + 16 * rt + l15, p.M - 1);
            uint32_t rkey = 0;
            if (p.drop_thresh) rkey = pcg_hash((uint32_t)m + rbase);
            s1[rt] = 0.f;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int n = 64 * cg + 32 * ch + 16 * c2 + 4 * g;
                    const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 r4 = p.R ? *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[cg][rt][c2][r] + bb[r];
                        if (p.drop_thresh) x = rng_keep(rkey, (uint32_t)(n + r), p.drop_thresh) ? x * p.drop_scale : 0.f;
                        x += rr[r];
                        zv[rt][cg][c2][r] = x;
                        s1[rt] += x;
                    }
                }
            s1[rt] += __shfl_xor(s1[rt], 16, 64);
            s1[rt] += __shfl_xor(s1[rt], 32, 64);
        }
        // (the ring is idle now: every DMA has been waited for in the last iteration)
        if (g == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) sred[((rg * 2 + ch) * RT + rt) * 16 + l15] = s1[rt];
        }
        __syncthreads();
        float mu[RT], s2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            mu[rt] = (s1[rt] + sred[((rg * 2 + (ch ^ 1)) * RT + rt) * 16 + l15]) * (1.f / 256.f);
            s2[rt] = 0.f;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float d = zv[rt][cg][c2][r] - mu[rt]; s2[rt] += d * d; }
            s2[rt] += __shfl_xor(s2[rt], 16, 64);
            s2[rt] += __shfl_xor(s2[rt], 32, 64);
        }
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) sred[((rg * 2 + ch) * RT + rt) * 16 + l15] = s2[rt];
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = m_wave + 16 * rt + l15;
            const float rs = rsqrtf((s2[rt] + sred[((rg * 2 + (ch ^ 1)) * RT + rt) * 16 + l15]) * (1.f / 256.f) + p.eps);
            if (m >= p.M) continue;
            if (g == 0 && ch == 0) { p.mean[m] = mu[rt]; p.rstd[m] = rs; }
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int n = 64 * cg + 32 * ch + 16 * c2 + 4 * g;
                    const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + n), b4 = *reinterpret_cast<const float4*>(p.beta + n);
                    const float* z = zv[rt][cg][c2];
                    *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n) = make_float4(z[0], z[1], z[2], z[3]);
                    *reinterpret_cast<float4*>(p.Y + (size_t)m * p.ldy + n) =
                        make_float4((z[0] - mu[rt]) * rs * g4.x + b4.x, (z[1] - mu[rt]) * rs * g4.y + b4.y, (z[2] - mu[rt]) * rs * g4.z + b4.z, (z[3] - mu[rt]) * rs * g4.w + b4.w);
                }
        }
    } else {
        // All waves meet at the group's barrier, so an epilogue run by everyone at the same point leaves the matrix pipe idle for its
        // whole length.  Every wave runs the epilogue of group cg-1 inside group cg -- half of a SIMD's waves before the group's first
        // k-step, the other half (a workgroup's waves w, w + 4, w + 8 ... share a SIMD) in its middle: while one wave converts,
        // hashes and stores, its SIMD-mate multiplies.  The finished sums of a group wait in `pend`.
        const bool late = ((wave >> 2) & 1) != 0;
        f32x4 acc[RT][2], pend[RT][2];
        zero(acc);
        for (int cg = 0; cg < p.ncg; ++cg) {
            group_sync(cg);
            if (!late && cg > 0) run_epilogue(cg - 1, pend);
            group_mma(cg, I0{}, IH{}, acc);
            if (late && cg > 0) run_epilogue(cg - 1, pend);
            group_mma(cg, IH{}, IK{}, acc);
            end_iteration();
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { pend[rt][0] = acc[rt][0]; pend[rt][1] = acc[rt][1]; }
            zero(acc);
        }
        run_epilogue(p.ncg - 1, pend);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may be in flight when the workgroup's LDS is released
#ifdef PANEL_STAMPS
    if (p.stamps && lane == 0) {
        stamp_sum[6] = panel_stamp() - t_begin;
        for (int k = 0; k < 8; ++k) p.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + k] = stamp_sum[k];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The same row-panel GEMM on v_mfma_f32_32x32x16_bf16 (round 4; K = 256, N % 64 == 0, the plain / relu / relu + dropout / split epilogues;
// selected by rows_per_wg = 2128).  An experiment the round-3 review and the knock-out table of DESIGN 5d-6 asked for: the 16x16x32 loop
// is bound by vector ISSUE (an MFMA holds the SIMD's issue port 8 of its 16 cycles), the 32x32x16 form holds it 8 of 32 and reads half
// the weight fragments per FLOP.  8 waves x 32 rows: wave = 2 rg + ch owns rows [32 rg, 32 rg + 32) of the 128-row panel and columns
// [32 ch, 32 ch + 32) of every 64-column group; its activation rows stay in registers as B operands (lane: row l & 31, k = 16 ks + 8 (l >> 5)
// .. + 7: 128 VGPRs for hi and lo), the weights are the A operand, read from the SAME tiled planes as above (a 32-column x 16-k fragment
// is two 16-lane halves of two neighbouring sub-tiles), and a lane ends up with C[m = l & 31][n = 8 j + 4 (l >> 5) .. + 3], j = 0..3.
// The sums over k are formed 16 at a time instead of 32: results differ from the 16x16x32 kernels in the last bits.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ __launch_bounds__(512, 1) void panel32_kernel(const PanelParams p) {
    constexpr int KSTEPS = 8, NW = 8, NT = 64 * NW, ROWS = 128;
    constexpr int NSUB = 8 * KSTEPS, GROUP = NSUB * PSUB, DMAX = NSUB / NW, RING = 2 * GROUP;
    static_assert(RING >= ROWS * 1024, "the A image borrows the ring");
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING + 4096 + 1024];
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    __builtin_assume(wave >= 0 && wave < NW);
    const int rg = wave >> 1, ch = wave & 1;
    const int m_wg = blockIdx.x * ROWS, m_wave = m_wg + 32 * rg;
    const uint32_t rbase = p.drop_thresh ? rng_stream_base(p.seed, p.stream) : 0u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.W), 0, 0x7FFFFFF0, 0x00020000);
    const int lane16 = lane * 16;
    auto issue_group = [&](int cg) {
#pragma unroll
        for (int q = 0; q < DMAX; ++q) {
            const int j = wave + NW * q;                              // this wave's sub-tile of the group: [plane][ct][ks]
            const int pl = j / (4 * KSTEPS), rem = j - pl * (4 * KSTEPS);
            const uint32_t soff = (uint32_t)pl * (uint32_t)p.plane_bytes + (uint32_t)(4 * cg * KSTEPS + rem) * PSUB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(smem + (cg & 1) * GROUP + j * PSUB), 16, lane16, (int)soff, 0, 0);
        }
    };
    float* const sbias = reinterpret_cast<float*>(smem + RING);
    const uint32_t sbias_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)sbias;
    const bool bias_lds = p.bias != nullptr;
    if (bias_lds) for (int i = t; i < 64 * p.ncg; i += NT) sbias[i] = (i < p.N) ? p.bias[i] : 0.f;

    // the panel's rows as MFMA B fragments, split once
    bf16x8_t ah[16], al[16];
    {
        const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
        for (int q = 0; q < ROWS / NW; ++q) {
            const int r = wave * (ROWS / NW) + q;
            const int row = min(m_wg + r, p.M - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(smem + r * 1024), 16, (lane ^ (r & 15)) << 4,
                                                     (int)((uint32_t)row * (uint32_t)p.lda * 4u), 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned char* rowp = smem + (32 * rg + l31) * 1024;
        const int sw = l31 & 15;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int c0 = 4 * ks + 2 * kh;
            const float4 a = *reinterpret_cast<const float4*>(rowp + ((c0 ^ sw) << 4));
            const float4 c = *reinterpret_cast<const float4*>(rowp + (((c0 + 1) ^ sw) << 4));
            u32x2 h0, l0, h1, l1;
            split4<3>(a, h0, l0);
            split4<3>(c, h1, l1);
            ah[ks] = __builtin_bit_cast(bf16x8_t, (u32x4){h0[0], h0[1], h1[0], h1[1]});
            al[ks] = __builtin_bit_cast(bf16x8_t, (u32x4){l0[0], l0[1], l1[0], l1[1]});
        }
        __syncthreads();
    }
    issue_group(0);

    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)min((size_t)0x7FFFFFF0, ((size_t)(p.M - 1) * p.ldc + p.N) * 4), 0x00020000);
    const int m = m_wave + l31;
    const uint32_t c_voff = ((uint32_t)m * (uint32_t)p.ldc + (uint32_t)(32 * ch + 4 * kh)) * 4u;
    const uint32_t c_rkey = pcg_hash((uint32_t)m + rbase);
    auto epilogue = [&](int cg, const f32x16_t& ac, auto act_t, auto drop_t, auto split_t) {
        constexpr bool ACT = decltype(act_t)::value, DROP = decltype(drop_t)::value, SPLIT = decltype(split_t)::value;
        f32x4 b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias_lds) {
            const uint32_t ba = sbias_lds + 4u * (uint32_t)(64 * cg + 32 * ch + 4 * kh);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(b4[0]), "=&v"(b4[1]), "=&v"(b4[2]), "=&v"(b4[3]) : "v"(ba) : "memory");
        }
        const int soff = cg * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4] = {ac[4 * j] + b4[j][0], ac[4 * j + 1] + b4[j][1], ac[4 * j + 2] + b4[j][2], ac[4 * j + 3] + b4[j][3]};
            if (ACT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) asm("v_max_f32 %0, 0, %1" : "=v"(v[r]) : "v"(v[r]));
            }
            if (DROP) {
                const uint32_t n = (uint32_t)(64 * cg + 32 * ch + 8 * j + 4 * kh);
                const uint32_t h01 = rng_pair(c_rkey, n), h23 = rng_pair(c_rkey, n + 2u);
                v[0] = rng_keep_lo(h01, p.drop_thresh) ? v[0] * p.drop_scale : 0.f;
                v[1] = rng_keep_hi(h01, p.drop_thresh) ? v[1] * p.drop_scale : 0.f;
                v[2] = rng_keep_lo(h23, p.drop_thresh) ? v[2] * p.drop_scale : 0.f;
                v[3] = rng_keep_hi(h23, p.drop_thresh) ? v[3] * p.drop_scale : 0.f;
            }
            u32x4 o;
            if (SPLIT) {
                const uint4 sc = split_chunk(make_float4(v[0], v[1], v[2], v[3]));
                o = (u32x4){sc.x, sc.y, sc.z, sc.w};
            } else {
                o = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, c_rsrc, (int)(c_voff + (uint32_t)soff) + 32 * j, 0, 0);
        }
    };
    const int epi_kind = p.drop_thresh ? 1 : p.out_split ? 2 : p.act == 1 ? 3 : 4;
    int st_prev = 0, st_now = 0;
    auto run_epilogue = [&](int cg, const f32x16_t& ac) {
        using T = std::true_type; using F = std::false_type;
        __builtin_amdgcn_s_setprio(1);
        if (epi_kind == 1) epilogue(cg, ac, T{}, T{}, F{});
        else if (epi_kind == 2) epilogue(cg, ac, F{}, F{}, T{});
        else if (epi_kind == 3) epilogue(cg, ac, T{}, F{}, F{});
        else epilogue(cg, ac, F{}, F{}, F{});
        __builtin_amdgcn_s_setprio(0);
        st_now += 4;
    };
    auto group_sync = [&](int cg) {
        panel_wait_vm(st_prev);
        __builtin_amdgcn_s_barrier();
        if (cg + 1 < p.ncg) issue_group(cg + 1);
    };
    // weight fragment of (group, ks, plane): lane reads sub-tile (plane, ct = 2 ch + (l31 >> 4), ks / 2), unit (2 (ks & 1) + kh) * 16 + (l31 & 15)
    const int lane_off = (2 * ch + (l31 >> 4)) * KSTEPS * PSUB + (kh * 16 + (l31 & 15)) * 16;
    auto group_mma = [&](int cg, auto k0_t, auto k1_t, f32x16_t& ac) {
        constexpr int K0 = decltype(k0_t)::value, K1 = decltype(k1_t)::value;
        const unsigned char* base = smem + (cg & 1) * GROUP + lane_off;
        bf16x8_t bf[2][2];
        auto load_b = [&](int buf, int ks) {
            bf[buf][0] = *reinterpret_cast<const bf16x8_t*>(base + (ks >> 1) * PSUB + (ks & 1) * 512);
            bf[buf][1] = *reinterpret_cast<const bf16x8_t*>(base + 4 * KSTEPS * PSUB + (ks >> 1) * PSUB + (ks & 1) * 512);
        };
        load_b(K0 & 1, K0);
#pragma unroll
        for (int ks = K0; ks < K1; ++ks) {
            if (ks + 1 < K1) load_b((ks + 1) & 1, ks + 1);
            const bf16x8_t bh = bf[ks & 1][0], bl = bf[ks & 1][1];
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[ks], ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[ks], ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[ks], ac, 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using IH = std::integral_constant<int, 8>; using IK = std::integral_constant<int, 16>;
    const bool late = ((wave >> 2) & 1) != 0;            // waves w and w + 4 share a SIMD: one runs its epilogue before the group's first k-step, the other in its middle
    f32x16_t acc, pend;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; pend[i] = 0.f; }
    for (int cg = 0; cg < p.ncg; ++cg) {
        group_sync(cg);
        if (!late && cg > 0) run_epilogue(cg - 1, pend);
        group_mma(cg, I0{}, IH{}, acc);
        if (late && cg > 0) run_epilogue(cg - 1, pend);
        group_mma(cg, IH{}, IK{}, acc);
        st_prev = st_now; st_now = 0;
        pend = acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
    run_epilogue(p.ncg - 1, pend);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// Tiled bf16 planes of the weights.  One descriptor per 64 x 64 block of a destination matrix Wd[n][k] (n < N, k < K):
//   {src offset (floats), src row stride, transposed, N, K, n0, k0, KSTEPS, dst hi offset (bytes), plane bytes}
// transposed = 0: Wd[n][k] = src[n * ld + k]   (the forward operand: W as stored, [out][in])
// transposed = 1: Wd[n][k] = src[k * ld + n]   (the input-gradient operand W^T)
// Element (n, k) of a plane lives in sub-tile (n / 16, k / 32) at byte ((n/16) * KSTEPS + k/32) * 1024 + (((k % 32) / 8) * 16 + n % 16) * 16
// + (k % 8) * 2; rows n >= N and columns k >= K of the padded planes are zero.
// ---------------------------------------------------------------------------------------------------------------
struct RetileDesc { int src_off, ld, transposed, N, K, n0, k0, ksteps; long long dst_off, plane_bytes; };

__global__ __launch_bounds__(256) void retile_kernel(const float* __restrict__ src_base, unsigned char* __restrict__ dst_base, const RetileDesc* __restrict__ descs) {
    __shared__ float tile[64][65];                       // [n][k]
    const RetileDesc d = descs[blockIdx.x];
    const float* src = src_base + d.src_off;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i, r = idx >> 4, c4 = (idx & 15) * 4;          // r: slow source index, c4: fast source index
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (!d.transposed) {                              // source rows are n, fast axis k
            const int n = d.n0 + r;
            for (int e = 0; e < 4; ++e) if (n < d.N && d.k0 + c4 + e < d.K) v[e] = src[(size_t)n * d.ld + d.k0 + c4 + e];
            for (int e = 0; e < 4; ++e) tile[r][c4 + e] = v[e];
        } else {                                          // source rows are k, fast axis n
            const int k = d.k0 + r;
            for (int e = 0; e < 4; ++e) if (k < d.K && d.n0 + c4 + e < d.N) v[e] = src[(size_t)k * d.ld + d.n0 + c4 + e];
            for (int e = 0; e < 4; ++e) tile[c4 + e][r] = v[e];
        }
    }
    __syncthreads();
    // 4 x 2 sub-tiles of 64 16-byte units per plane = 512 units: two per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = t + 256 * i;
        const int st = u >> 6, w = u & 63;                // sub-tile (ct = st >> 1, ks = st & 1), unit w = g * 16 + l15
        const int ct = st >> 1, ksl = st & 1, gg = w >> 4, l = w & 15;
        const int ks = d.k0 / 32 + ksl;
        if (ks >= d.ksteps) continue;
        const int n = 16 * ct + l, k = 32 * ksl + 8 * gg;
        const float4 a = make_float4(tile[n][k], tile[n][k + 1], tile[n][k + 2], tile[n][k + 3]);
        const float4 c = make_float4(tile[n][k + 4], tile[n][k + 5], tile[n][k + 6], tile[n][k + 7]);
        u32x2 h0, l0, h1, l1;
        split4<3>(a, h0, l0);
        split4<3>(c, h1, l1);
        unsigned char* dst = dst_base + d.dst_off + ((size_t)((d.n0 / 16 + ct) * d.ksteps + ks)) * PSUB + w * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(h0[0], h0[1], h1[0], h1[1]);
        *reinterpret_cast<uint4*>(dst + d.plane_bytes) = make_uint4(l0[0], l0[1], l1[0], l1[1]);
    }
}

extern "C" int unast_retile_weights(const float* src_base, void* dst_base, const void* descs_dev, int ndesc, hipStream_t stream) {
    UNAST_REQUIRE(src_base && dst_base && descs_dev && ndesc > 0, "unast_retile_weights: bad arguments");
    UNAST_REQUIRE((((uintptr_t)dst_base) & 15) == 0, "unast_retile_weights: destination must be 16-byte aligned");
    hipLaunchKernelGGL(retile_kernel, dim3(ndesc), dim3(256), 0, stream, src_base, (unsigned char*)dst_base, (const RetileDesc*)descs_dev);
    return unast_check_launch("unast_retile_weights");
}

// ---------------------------------------------------------------------------------------------------------------
// K-streamed panel (K > 256, N a multiple of 256): the OUTPUT is the stationary operand.  A workgroup of 16 waves owns 128 rows x 256
// columns; wave (rg, ch) accumulates rows [32 rg, 32 rg + 32) x columns [64 ch, 64 ch + 64) -- two row tiles against four column
// tiles, so every weight fragment read from LDS feeds 6 MFMAs (the A-stationary kernel above: 3).  K streams in groups of 32 through two three-slot LDS rings filled by LDS-DMA two groups ahead: the weights of a group
// (256 n x 32 k, both planes = 32 KB, from the same tiled planes) and the activations of a group (128 rows x 32 k of fp32 = 16 KB, whole
// 128-byte row pieces, 8 rows per DMA instruction; every wave then reads its 32 rows as MFMA fragments and splits them).  One raw
// s_barrier per group.  (A first version loaded the activation fragments straight into registers: correct, and 1.5x SLOWER than the tile
// kernel -- every wave of a row group issued the same 16-rows-x-64-byte loads, 128 scattered load instructions per group and CU.)
// Serves the contractions of the train step that reduce over more than 256 values into 256 columns: FFN linear2 (K = 1024) with its
// residual + LayerNorm epilogue, and the input gradients of linear1 (K = 1024), of the self-attention in-projection (K = 768) and of
// the cross-attention key/value projection (K = 512).
// ---------------------------------------------------------------------------------------------------------------
template <int LN>
__global__ __launch_bounds__(1024, 4) void kpanel_kernel(const PanelParams p) {
    constexpr int NW = 16, ROWS = 128, WSLOT = 32768, ASLOT = 16384;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * WSLOT + 3 * ASLOT + 1024 + 2048];       // W ring | A ring | bias (256 floats) | LayerNorm row sums (4 x 128 floats)
    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    __builtin_assume(wave >= 0 && wave < NW);
    const int rg = wave >> 2, ch = wave & 3;
    const int m_wg = blockIdx.x * ROWS, n_wg = blockIdx.y * 256;
    const int m_wave = m_wg + 32 * rg;
    const int KG = p.K >> 5;                              // groups = k-steps of the planes
#ifdef PANEL_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // 0 wait for the activations, 1 split -> first weights there, 2 wait vmcnt, 3 barrier, 4 issue of reads + DMA, 5 first-half MFMAs -> all reads back, 6 total, 7 epilogue
    const unsigned long long t_begin = panel_stamp();
#endif
    const uint32_t rbase = p.drop_thresh ? rng_stream_base(p.seed, p.stream) : 0u;
    unsigned char* const aring = smem + 3 * WSLOT;

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.W), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
    const int lane16 = lane * 16;
    // A image of a group: row r (0..127) at r * 128, its 16-byte chunk c (4 k) at position c ^ key(r), key(r) = (r & 7) ^ ((r >> 3) & 1):
    // the fragment reads below (16 rows x one chunk position per ds_read_b128 lane group) then cover all 64 banks exactly once.
    // This wave copies rows 8 wave .. 8 wave + 7: lane -> row 8 wave + lane / 8, position lane % 8 (LDS-DMA writes lane-linear), so the
    // swizzle is applied on the SOURCE side.  Rows >= M are clamped (computed, never stored).
    const int ar = 8 * wave + (lane >> 3);
    const uint32_t a_voff = ((uint32_t)min(m_wg + ar, p.M - 1) * (uint32_t)p.lda + 4u * (uint32_t)((lane & 7) ^ ((ar & 7) ^ ((ar >> 3) & 1)))) * 4u;
    const uint32_t w_soff0 = (uint32_t)((n_wg >> 4) + wave) * (uint32_t)(KG * PSUB);           // this wave's column tile in a plane
    auto issue_group = [&](int kg, int slot) {
        // weights: sub-tile [plane q][column tile = wave] of k-step kg
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(smem + slot * WSLOT + (q * 16 + wave) * PSUB), 16, lane16,
                                                     (int)((uint32_t)q * (uint32_t)p.plane_bytes + w_soff0 + (uint32_t)kg * PSUB), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(aring + slot * ASLOT + wave * 1024), 16, (int)a_voff, kg * 128, 0, 0);
    };
    float* const sbias = reinterpret_cast<float*>(smem + 3 * WSLOT + 3 * ASLOT);
    float* const sred = sbias + 256;
    if (t < 256) sbias[t] = p.bias ? p.bias[n_wg + t] : 0.f;
    issue_group(0, 0);
    if (KG > 1) issue_group(1, 1);

    f32x4 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // LDS byte addresses of this lane's fragment reads inside a slot
    uint32_t a_rd[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = 32 * rg + 16 * rt + l15, key = (r & 7) ^ ((r >> 3) & 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) a_rd[rt][h] = (uint32_t)(r * 128 + (((2 * g + h) ^ key) << 4));
    }
    const uint32_t aring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)aring;
    const uint32_t wring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem + (uint32_t)(lane16 + (4 * ch) * PSUB);
    // Software pipeline over half groups: the LDS reads of group kg + 1 are issued right behind its barrier and are served while the MFMAs
    // of the second half of group kg run (s_memtime stamps of the first, unpipelined loop: of 3 080 ticks per group a wave spent 850
    // waiting for its fragments -- all 16 waves ask the LDS for 12 KB each at the same moment -- 430 issuing DMAs in front of those
    // reads and 890 at the barrier; the 24 MFMAs took 260).
    f32x4 raw[2][2];
    bf16x8_t b0h[2], b0l[2], b1h[2], b1l[2];             // weight fragments of column tiles 0, 1 (b0) and 2, 3 (b1)
    bf16x8_t ah[2], al[2];
#define KP_READ_A(slot_) do { const uint32_t ab = aring_lds + (uint32_t)(slot_) * ASLOT; \
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7" \
                     : "=&v"(raw[0][0]), "=&v"(raw[0][1]), "=&v"(raw[1][0]), "=&v"(raw[1][1]) \
                     : "v"(ab + a_rd[0][0]), "v"(ab + a_rd[0][1]), "v"(ab + a_rd[1][0]), "v"(ab + a_rd[1][1]) : "memory"); } while (0)
#define KP_READ_W0(slot_) do { const uint32_t wb = wring_lds + (uint32_t)(slot_) * WSLOT; \
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16384\n\tds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:17408" \
                     : "=&v"(b0h[0]), "=&v"(b0l[0]), "=&v"(b0h[1]), "=&v"(b0l[1]) : "v"(wb) : "memory"); } while (0)
#define KP_READ_W1(slot_) do { const uint32_t wb = wring_lds + (uint32_t)(slot_) * WSLOT; \
        asm volatile("ds_read_b128 %0, %4 offset:2048\n\tds_read_b128 %1, %4 offset:18432\n\tds_read_b128 %2, %4 offset:3072\n\tds_read_b128 %3, %4 offset:19456" \
                     : "=&v"(b1h[0]), "=&v"(b1l[0]), "=&v"(b1h[1]), "=&v"(b1l[1]) : "v"(wb) : "memory"); } while (0)
    // (the waits carry the registers as operands: to the compiler an asm's outputs are ready when the statement ends, and it moved the
    // conversions and the first MFMAs ahead of a bare s_waitcnt)
    int slot = 0;
    if (KG > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // group 0 is in LDS
    KP_READ_A(0);
    KP_READ_W0(0);
    if (KG > 2) issue_group(2, 2);
    for (int kg = 0; kg < KG; ++kg) {
        STAMP(ta);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
        STAMP(tb);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            u32x2 h0, l0, h1, l1;
            split4<3>(make_float4(raw[rt][0][0], raw[rt][0][1], raw[rt][0][2], raw[rt][0][3]), h0, l0);
            split4<3>(make_float4(raw[rt][1][0], raw[rt][1][1], raw[rt][1][2], raw[rt][1][3]), h1, l1);
            ah[rt] = __builtin_bit_cast(bf16x8_t, (u32x4){h0[0], h0[1], h1[0], h1[1]});
            al[rt] = __builtin_bit_cast(bf16x8_t, (u32x4){l0[0], l0[1], l1[0], l1[1]});
        }
        KP_READ_W1(slot);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b0h[0]), "+v"(b0l[0]), "+v"(b0h[1]), "+v"(b0l[1]) :: "memory");
        STAMP(tc);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt][ct] = mfma16(b0l[ct], ah[rt], acc[rt][ct]);
                acc[rt][ct] = mfma16(b0h[ct], al[rt], acc[rt][ct]);
                acc[rt][ct] = mfma16(b0h[ct], ah[rt], acc[rt][ct]);
            }
        // (pins the twelve MFMAs above in front of the wait below: otherwise hipcc sinks most of them behind it)
        asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[0][1]), "v"(acc[1][1]));
        // every LDS read of this group's slots has returned before the wave meets the others: the slots are refilled behind the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1h[0]), "+v"(b1l[0]), "+v"(b1h[1]), "+v"(b1l[1]) :: "memory");
        STAMP(td);
        if (kg + 1 < KG) {
            // outstanding vector-memory operations of this wave: group kg + 1 (3) and, if it exists, group kg + 2 (3)
            if (kg + 2 < KG) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(te);
            __builtin_amdgcn_s_barrier();                 // every wave's share of group kg + 1 is in LDS, and nobody reads group kg's slots any more
            STAMP(tf);
            const int nslot = slot == 2 ? 0 : slot + 1;
            KP_READ_A(nslot);
            KP_READ_W0(nslot);
            if (kg + 3 < KG) issue_group(kg + 3, slot);
            STAMP(tg);
            STAMP_ADD(2, td, te); STAMP_ADD(3, te, tf); STAMP_ADD(4, tf, tg);
            slot = nslot;
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt][2 + ct] = mfma16(b1l[ct], ah[rt], acc[rt][2 + ct]);
                acc[rt][2 + ct] = mfma16(b1h[ct], al[rt], acc[rt][2 + ct]);
                acc[rt][2 + ct] = mfma16(b1h[ct], ah[rt], acc[rt][2 + ct]);
            }
#ifdef PANEL_STAMPS
        asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][3]));        // the stamp below is taken when the last MFMA has ISSUED, not retired
#endif
        STAMP(th);
        STAMP_ADD(0, ta, tb); STAMP_ADD(1, tb, tc); STAMP_ADD(5, tc, td);
    }
#undef KP_READ_A
#undef KP_READ_W0
#undef KP_READ_W1
#ifdef PANEL_STAMPS
    const unsigned long long t_loop = panel_stamp();
#endif
    __syncthreads();                                      // (bias words visible whatever KG; the ring is idle: every DMA has been waited for)
    // lane holds C[m = m_wave + 16 rt + l15][n = n_wg + 64 ch + 16 ct + 4 g .. + 3]
    if constexpr (LN == 2) {
        // LayerNorm BACKWARD in the epilogue (round 4).  The GEMM is the input-gradient contraction that ends a sub-layer's backward,
        // dy = R + acc is the complete gradient of the PREVIOUS sub-layer's LayerNorm output, and the workgroup holds whole rows of it:
        //   g = dy gamma, xhat = (z - mean) rstd, dz = rstd (g - mean_row(g) - xhat mean_row(g xhat)),
        // dz -> C, dropout(dz) -> Y (the gradient of that sub-layer's dropped branch), column sums of dy xhat / dy over the panel's rows
        // -> part[panel][0 / 1][256] (summed by ln_bwd_finalize_kernel).  What the stand-alone kernel (norm.hip layernorm_bwd_kernel)
        // computes, in the same formulas; its launch, its read of dy and the GEMM's write of dy disappear.
        float* const red = reinterpret_cast<float*>(smem);              // the rings are idle: [2][4][128] row partials, then [2][4][256] column partials
        float xh[2][4][4], mu[2], rs[2], s1[2], s2[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int mr = m_wave + 16 * rt + l15, m = min(mr, p.M - 1);
            mu[rt] = p.mean[m]; rs[rt] = p.rstd[m];
            s1[rt] = 0.f; s2[rt] = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int n = 64 * ch + 16 * ct + 4 * g;
                const float4 r4 = p.R ? *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 z4 = *reinterpret_cast<const float4*>(p.Z + (size_t)m * p.ldz + n);
                const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + n);
                const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w}, gm[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dy = mr < p.M ? acc[rt][ct][r] + rr[r] : 0.f;          // rows beyond M add nothing to the column sums
                    const float x = (zz[r] - mu[rt]) * rs[rt];
                    const float gg = dy * gm[r];
                    acc[rt][ct][r] = dy; xh[rt][ct][r] = x;
                    s1[rt] += gg; s2[rt] += gg * x;
                }
            }
            s1[rt] += __shfl_xor(s1[rt], 16, 64); s1[rt] += __shfl_xor(s1[rt], 32, 64);
            s2[rt] += __shfl_xor(s2[rt], 16, 64); s2[rt] += __shfl_xor(s2[rt], 32, 64);
        }
        if (g == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                red[ch * 128 + 32 * rg + 16 * rt + l15] = s1[rt];
                red[512 + ch * 128 + 32 * rg + 16 * rt + l15] = s2[rt];
            }
        }
        __syncthreads();
        float cg[4][4], cb[4][4];                          // column partials over this lane's two rows
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) { cg[ct][r] = 0.f; cb[ct][r] = 0.f; }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int r0 = 32 * rg + 16 * rt + l15, m = m_wg + r0;
            const float c1 = ((red[r0] + red[128 + r0]) + (red[256 + r0] + red[384 + r0])) * (1.f / 256.f);
            const float c2 = ((red[512 + r0] + red[640 + r0]) + (red[768 + r0] + red[896 + r0])) * (1.f / 256.f);
            uint32_t rkey = 0;
            if (p.drop_thresh) rkey = pcg_hash((uint32_t)min(m, p.M - 1) + rbase);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int n = 64 * ch + 16 * ct + 4 * g;
                const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + n);         // (again: 32 registers less than keeping dy gamma)
                const float gm[4] = {g4.x, g4.y, g4.z, g4.w};
                float o[4], od[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = rs[rt] * (acc[rt][ct][r] * gm[r] - c1 - xh[rt][ct][r] * c2);
                    od[r] = (!p.drop_thresh || rng_keep(rkey, (uint32_t)(n + r), p.drop_thresh)) ? o[r] * p.drop_scale : 0.f;
                    cg[ct][r] += acc[rt][ct][r] * xh[rt][ct][r];
                    cb[ct][r] += acc[rt][ct][r];
                }
                if (m < p.M) {
                    *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                    if (p.Y) *reinterpret_cast<float4*>(p.Y + (size_t)m * p.ldy + n) = make_float4(od[0], od[1], od[2], od[3]);
                }
            }
        }
        __syncthreads();                                   // the row partials have been read: `red` becomes [2][4 row groups][256 columns]
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = cg[ct][r], b = cb[ct][r];        // sum over the 16 rows of the lane row (DPP-free form: four butterfly steps)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (l15 == 0) {
                    const int n = 64 * ch + 16 * ct + 4 * g + r;
                    red[rg * 256 + n] = a;
                    red[1024 + rg * 256 + n] = b;
                }
            }
        __syncthreads();
        if (p.part && t < 512) {
            const int q = t >> 8, n = t & 255;
            const float* rp = red + q * 1024 + n;
            p.part[((size_t)blockIdx.x * 2 + q) * 256 + n] = (rp[0] + rp[256]) + (rp[512] + rp[768]);
        }
    } else if constexpr (!LN) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int m = m_wave + 16 * rt + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int nl = 64 * ch + 16 * ct + 4 * g, n = n_wg + nl;
                const float4 b4 = *reinterpret_cast<const float4*>(sbias + nl);
                float4 o = make_float4(acc[rt][ct][0] + b4.x, acc[rt][ct][1] + b4.y, acc[rt][ct][2] + b4.z, acc[rt][ct][3] + b4.w);
                if (p.R) { const float4 r4 = *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n); o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w; }
                *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n) = o;
            }
        }
    } else {
        // z = R + dropout(acc + bias); y = LayerNorm(z) over the 256 columns of the row: this lane holds 16 of them, the three other lane
        // groups g of the wave 48 more, the waves ch' != ch of the row group the other 192.
        float mu[2], s1[2], s2[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int m = min(m_wave + 16 * rt + l15, p.M - 1);
            uint32_t rkey = 0;
            if (p.drop_thresh) rkey = pcg_hash((uint32_t)m + rbase);
            s1[rt] = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int n = 64 * ch + 16 * ct + 4 * g;
                const float4 b4 = *reinterpret_cast<const float4*>(sbias + n);
                const float4 r4 = p.R ? *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[rt][ct][r] + bb[r];
                    if (p.drop_thresh) x = rng_keep(rkey, (uint32_t)(n + r), p.drop_thresh) ? x * p.drop_scale : 0.f;
                    x += rr[r];
                    acc[rt][ct][r] = x;
                    s1[rt] += x;
                }
            }
            s1[rt] += __shfl_xor(s1[rt], 16, 64);
            s1[rt] += __shfl_xor(s1[rt], 32, 64);
        }
        if (g == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) sred[ch * 128 + 32 * rg + 16 * rt + l15] = s1[rt];
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int r0 = 32 * rg + 16 * rt + l15;
            mu[rt] = ((sred[r0] + sred[128 + r0]) + (sred[256 + r0] + sred[384 + r0])) * (1.f / 256.f);
            s2[rt] = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[rt][ct][r] - mu[rt]; s2[rt] += d * d; }
            s2[rt] += __shfl_xor(s2[rt], 16, 64);
            s2[rt] += __shfl_xor(s2[rt], 32, 64);
        }
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) sred[ch * 128 + 32 * rg + 16 * rt + l15] = s2[rt];
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int r0 = 32 * rg + 16 * rt + l15, m = m_wg + r0;
            const float rs = rsqrtf(((sred[r0] + sred[128 + r0]) + (sred[256 + r0] + sred[384 + r0])) * (1.f / 256.f) + p.eps);
            if (m >= p.M) continue;
            if (g == 0 && ch == 0) { p.mean[m] = mu[rt]; p.rstd[m] = rs; }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int n = 64 * ch + 16 * ct + 4 * g;
                const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + n), b4 = *reinterpret_cast<const float4*>(p.beta + n);
                const f32x4 z = acc[rt][ct];
                *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n) = make_float4(z[0], z[1], z[2], z[3]);
                *reinterpret_cast<float4*>(p.Y + (size_t)m * p.ldy + n) =
                    make_float4((z[0] - mu[rt]) * rs * g4.x + b4.x, (z[1] - mu[rt]) * rs * g4.y + b4.y, (z[2] - mu[rt]) * rs * g4.z + b4.z, (z[3] - mu[rt]) * rs * g4.w + b4.w);
            }
        }
    }
#ifdef PANEL_STAMPS
    if (p.stamps && lane == 0) {
        const unsigned long long t_end = panel_stamp();
        stamp_sum[6] = t_end - t_begin; stamp_sum[7] = t_end - t_loop;
        for (int k = 0; k < 8; ++k) p.stamps[((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * 16 + wave) * 8 + k] = stamp_sum[k];
    }
#endif
}

template <int KSTEPS>
static void panel_launch(const PanelParams& p, int rows, int ln, hipStream_t s) {
    // rows 64: 8 waves x 16 rows; 128: 8 waves x 32 rows (256 VGPRs, 2 waves / SIMD); 1128: 128 rows as 16 waves x 16 rows (128 VGPRs, 4 waves / SIMD)
    const int r = rows == 1128 ? 128 : rows;
    dim3 grid((p.M + r - 1) / r);
    if (rows == 1128) {
        if (ln) hipLaunchKernelGGL((panel_kernel<KSTEPS, 1, 1, 16>), grid, dim3(1024), 0, s, p);
        else    hipLaunchKernelGGL((panel_kernel<KSTEPS, 1, 0, 16>), grid, dim3(1024), 0, s, p);
    } else if (ln) {
        if (rows == 128) hipLaunchKernelGGL((panel_kernel<KSTEPS, 2, 1, 8>), grid, dim3(512), 0, s, p);
        else             hipLaunchKernelGGL((panel_kernel<KSTEPS, 1, 1, 8>), grid, dim3(512), 0, s, p);
    } else {
        if (rows == 128) hipLaunchKernelGGL((panel_kernel<KSTEPS, 2, 0, 8>), grid, dim3(512), 0, s, p);
        else             hipLaunchKernelGGL((panel_kernel<KSTEPS, 1, 0, 8>), grid, dim3(512), 0, s, p);
    }
}

// dz = LayerNorm-backward(R + A W^T) and its by-products in ONE launch (kpanel_kernel<2>): see the epilogue's comment.  K > 256, N = 256.
extern "C" int64_t unast_panel_gemm_lnbwd_ws_floats(int M) { return (int64_t)((M + 127) / 128) * 2 * 256; }

extern "C" int unast_panel_gemm_lnbwd(const float* A, int lda, const void* w_planes, int64_t plane_bytes, int M, int K, const float* R, int ldr,
                                      const float* z, int ldz, const float* mean, const float* rstd, const float* gamma,
                                      float* dz, int lddz, float* dz_drop, int lddrop, float* part, int64_t part_floats,
                                      float drop_p, unsigned int seed, unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(A && w_planes && z && mean && rstd && gamma && dz && M > 0, "unast_panel_gemm_lnbwd: null pointer");
    UNAST_REQUIRE(K > 256 && (K & 63) == 0 && (lda & 3) == 0 && (lddz & 3) == 0 && (ldz & 3) == 0 && (!R || (ldr & 3) == 0) && (!dz_drop || (lddrop & 3) == 0) &&
                  ((((uintptr_t)A) | ((uintptr_t)dz) | ((uintptr_t)w_planes) | ((uintptr_t)z) | ((uintptr_t)R) | ((uintptr_t)dz_drop) | ((uintptr_t)gamma)) & 15) == 0,
                  "unast_panel_gemm_lnbwd: needs K > 256, K %% 64 == 0, 16-byte row strides and aligned operands (K=%d)", K);
    UNAST_REQUIRE(!part || part_floats >= unast_panel_gemm_lnbwd_ws_floats(M), "unast_panel_gemm_lnbwd: partials buffer too small");
    UNAST_REQUIRE(dz_drop || drop_p <= 0.f, "unast_panel_gemm_lnbwd: dropout needs the second output");
    {
        const size_t ld_max = (size_t)std::max(std::max(lda, lddz), std::max(std::max(R ? ldr : 0, ldz), dz_drop ? lddrop : 0));
        UNAST_REQUIRE(((size_t)M + 127) / 128 * 128 * ld_max * 4 < (size_t)0x7FFFFFF0u, "unast_panel_gemm_lnbwd: operand exceeds the 2 GiB a buffer descriptor addresses");
    }
    PanelParams q = {};
    q.stamps = nullptr;
    q.A = A; q.lda = lda; q.W = (const unsigned char*)w_planes; q.plane_bytes = (size_t)plane_bytes; q.C = dz; q.ldc = lddz; q.M = M; q.N = 256; q.K = K; q.ncg = 4;
    q.R = R; q.ldr = ldr; q.Z = z; q.ldz = ldz; q.mean = const_cast<float*>(mean); q.rstd = const_cast<float*>(rstd); q.gamma = gamma;
    q.Y = dz_drop; q.ldy = lddrop; q.part = part;
    q.drop_thresh = dz_drop ? drop_threshold(drop_p) : 0u; q.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; q.seed = seed; q.stream = stream_id;
    hipLaunchKernelGGL((kpanel_kernel<2>), dim3((M + 127) / 128, 1), dim3(1024), 0, stream, q);
    return unast_check_launch("unast_panel_gemm_lnbwd");
}

// Diagnostic builds (-DPANEL_STAMPS, tools/panel_stamps.py): device buffer of 8 x 8 x workgroups uint64 that receives per-wave cycle sums.
extern "C" int unast_panel_debug_stamps(void* dev_buf) { g_panel_stamps = (unsigned long long*)dev_buf; return 0; }

extern "C" int unast_panel_gemm(const float* A, int lda, const void* w_planes, int64_t plane_bytes, float* C, int ldc, int M, int N, int K,
                                const float* bias, const float* R, int ldr, const float* G, int ldg, float gate_scale, int act,
                                float drop_p, unsigned int seed, unsigned int stream_id, int out_split,
                                const float* ln_gamma, const float* ln_beta, float* Y, int ldy, float* mean, float* rstd, float eps,
                                void* gate_bits, int rows_per_wg, hipStream_t stream) {
    UNAST_REQUIRE(A && w_planes && C && M > 0 && N > 0 && K >= 4, "unast_panel_gemm: bad arguments");
    {   // A, C, R, Y are addressed through buffer descriptors with 32-bit byte offsets: past 2 GiB loads would read zeros and stores be dropped
        const size_t ld_max = (size_t)std::max(std::max(lda, ldc), std::max(R ? ldr : 0, Y ? ldy : 0));
        UNAST_REQUIRE(((size_t)M + 127) / 128 * 128 * ld_max * 4 < (size_t)0x7FFFFFF0u, "unast_panel_gemm: operand of %d rows x %zu floats exceeds the 2 GiB a buffer descriptor addresses (use unast_gemm)", M, ld_max);
    }
    if (K > 256) {
        // K-streamed form: output-stationary 128 x 256 tiles
        UNAST_REQUIRE((K & 63) == 0 && (N & 255) == 0 && (lda & 3) == 0 && (ldc & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)C) | ((uintptr_t)w_planes)) & 15) == 0,
                      "unast_panel_gemm: K > 256 needs K %% 64 == 0, N %% 256 == 0, 16-byte row strides and aligned operands (N=%d K=%d)", N, K);
        UNAST_REQUIRE(!G && !act && !out_split && !gate_bits, "unast_panel_gemm: K > 256 serves the plain, residual and LayerNorm epilogues only");
        UNAST_REQUIRE(!R || ((ldr & 3) == 0 && (((uintptr_t)R) & 15) == 0), "unast_panel_gemm: residual operand must be 16-byte addressable");
        const int lnk = ln_gamma != nullptr;
        UNAST_REQUIRE(lnk || drop_p <= 0.f, "unast_panel_gemm: K > 256: dropout only inside the LayerNorm epilogue");
        UNAST_REQUIRE(!lnk || (N == 256 && ln_beta && Y && mean && rstd && (ldy & 3) == 0), "unast_panel_gemm: the LayerNorm epilogue needs N = 256, Y / mean / rstd and 16-byte row strides");
        PanelParams q = {};
        q.stamps = g_panel_stamps;
        q.A = A; q.lda = lda; q.W = (const unsigned char*)w_planes; q.plane_bytes = (size_t)plane_bytes; q.C = C; q.ldc = ldc; q.M = M; q.N = N; q.K = K; q.ncg = N / 64;
        q.bias = bias; q.R = R; q.ldr = ldr; q.drop_thresh = drop_threshold(drop_p); q.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; q.seed = seed; q.stream = stream_id;
        q.gamma = ln_gamma; q.beta = ln_beta; q.Y = Y; q.ldy = ldy; q.mean = mean; q.rstd = rstd; q.eps = eps;
        dim3 grid((M + 127) / 128, N / 256);
        if (lnk) hipLaunchKernelGGL((kpanel_kernel<1>), grid, dim3(1024), 0, stream, q);
        else     hipLaunchKernelGGL((kpanel_kernel<0>), grid, dim3(1024), 0, stream, q);
        return unast_check_launch("unast_panel_gemm");
    }
    UNAST_REQUIRE((K & 3) == 0 && K <= 256 && (lda & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)C) | ((uintptr_t)w_planes)) & 15) == 0,
                  "unast_panel_gemm: needs K %% 4 == 0, K <= 256, lda %% 4 == 0 and 16-byte aligned operands (K=%d lda=%d)", K, lda);
    UNAST_REQUIRE(rows_per_wg == 0 || rows_per_wg == 64 || rows_per_wg == 128 || rows_per_wg == 1128 || rows_per_wg == 2128, "unast_panel_gemm: rows_per_wg is 64, 128, 1128 (128 rows, 16 waves), 2128 (128 rows, 32x32x16 tiles) or 0 (auto)");
    UNAST_REQUIRE(!out_split || ((N & 3) == 0 && (ldc & 3) == 0), "unast_panel_gemm: out_split needs N %% 4 == 0 and ldc %% 4 == 0");
    const int ln = ln_gamma != nullptr;
    UNAST_REQUIRE(!ln || (N == 256 && ln_beta && Y && mean && rstd && (ldc & 3) == 0 && (ldy & 3) == 0 && !G && !act && !out_split &&
                          (!R || (ldr & 3) == 0)), "unast_panel_gemm: the LayerNorm epilogue needs N = 256, Y / mean / rstd and 16-byte row strides");
    PanelParams p;
    p.A = A; p.lda = lda; p.W = (const unsigned char*)w_planes; p.plane_bytes = (size_t)plane_bytes; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.ncg = (N + 63) / 64;
    p.bias = bias; p.R = R; p.ldr = ldr; p.G = G; p.ldg = ldg; p.gate_scale = gate_scale; p.act = act;
    p.drop_thresh = drop_threshold(drop_p); p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.seed = seed; p.stream = stream_id;
    p.out_split = out_split;
    p.gate_bits = (unsigned long long*)gate_bits;
    UNAST_REQUIRE(!gate_bits || ((N & 63) == 0 && !G && !R && !ln && (ldc & 3) == 0 && (((uintptr_t)gate_bits) & 15) == 0 && (act == 1 ? (!out_split && N <= 1024) : (!bias && !out_split && drop_p <= 0.f))),
                  "unast_panel_gemm: gate_bits needs N %% 64 == 0 and either the relu (+ dropout) epilogue (writes them) or the plain gated one (reads them)");
    p.stamps = g_panel_stamps;
    p.gamma = ln_gamma; p.beta = ln_beta; p.Y = Y; p.ldy = ldy; p.mean = mean; p.rstd = rstd; p.eps = eps;
    // vector epilogue operands must be 16-byte addressable
    UNAST_REQUIRE((ldc & 3) == 0 || !out_split, "unast_panel_gemm: ldc");
    if (rows_per_wg == 2128) {                           // the 32x32x16 form (experiment, see panel32_kernel)
        UNAST_REQUIRE(K == 256 && (N & 63) == 0 && (ldc & 3) == 0 && !R && !G && !ln && !gate_bits && N <= 1024 && !(out_split && (act || drop_p > 0.f)) && !(drop_p > 0.f && act != 1),
                      "unast_panel_gemm: rows_per_wg = 2128 serves K = 256, N %% 64 == 0 and the plain / relu / relu + dropout / split epilogues only");
        hipLaunchKernelGGL(panel32_kernel, dim3((M + 127) / 128), dim3(512), 0, stream, p);
        return unast_check_launch("unast_panel_gemm");
    }
    const int ksteps = (K + 31) / 32;
    UNAST_REQUIRE(ksteps != 8 || K == 256, "unast_panel_gemm: 224 < K < 256 is not built (the 8-k-step form copies whole 1-KB activation rows by LDS-DMA; K=%d)", K);
    const int rt = rows_per_wg ? rows_per_wg : (M >= 16384 ? 128 : 64);
    switch (ksteps) {
        case 8: panel_launch<8>(p, rt, ln, stream); break;
        case 4: panel_launch<4>(p, rt, ln, stream); break;
        case 3: panel_launch<3>(p, rt, ln, stream); break;
        case 2: panel_launch<2>(p, rt, ln, stream); break;
        case 1: panel_launch<1>(p, rt, ln, stream); break;
        case 6: panel_launch<6>(p, rt, ln, stream); break;
        default: return unast_set_error(UNAST_ERR_ARG, "unast_panel_gemm: K = %d (ceil(K/32) = %d k-steps) is not built (1, 2, 3, 4, 6, 8)", K, ksteps);
    }
    return unast_check_launch("unast_panel_gemm");
}

UNAST_DEFINE_RNG_EPOCH_SETTER(panel)

// Flash-style multi-head attention for gfx950 (head_dim 64): forward, dQ and dK/dV kernels.
// Replaces torch.nn.MultiheadAttention's softmax(QK^T/sqrt(d) + mask) -> dropout -> .V core inside
// torch.nn.TransformerEncoderLayer/DecoderLayer (src/module.py:273-274, 286-287; semantics SURVEY.md Appendix A):
// additive -inf mask = key >= lens_k[b] (key padding) OR key > query (causal); padded QUERIES are not masked.
// No T x T tensor ever reaches HBM: K/V tiles are staged in LDS as bf16 (split hi/lo when NSPLIT==3), scores and
// probabilities live in MFMA accumulators, online softmax statistics are per-lane.
//
// Layout trick (16x16x32 bf16 MFMA, C/D: col = lane&15, row = 4*(lane>>4)+reg):
//   forward / dQ kernel: S^T[key,q] = K.Q^T puts the QUERY on the lane, so max/sum over keys is registers + two
//   shuffles, and the P^T accumulators of two stacked key sub-tiles are directly the B operand of O^T = V^T.P^T
//   (k-slot j of lane-group g = key 4g+j of sub-tile 2u for j<4, of sub-tile 2u+1 for j>=4); V^T comes from
//   ds_read_b64_tr_b16 on the untransposed [key][d] image with the same key permutation.
//   dK/dV kernel: S[q,key] = Q.K^T puts the KEY on the lane; P and dS accumulators are then the B operands of
//   dV^T = dO^T.P and dK^T = Q^T.dS, with dO^T / Q^T read transposed from the same [q][d] LDS images that feed S, dP.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define HD 64
#define ALD 64                     // bf16 elements per LDS image row = 128 B, 16-B chunks XOR-swizzled by (row & 6)
// Byte offset of element (row, col) of a [rows][64] bf16 image.  The chunk swizzle makes both kinds of read conflict-free under
// the hardware's lane groups (ds_read_b128: {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; ds_read_b64_tr_b16: lanes 0-31 / 32-63):
// with plain 144-B padded rows every ds_read_b128 fragment read was 2-way conflicting (SQ_LDS_BANK_CONFLICT = 36-44 % of the
// LDS-active cycles, profiles/r01_pmc_attn_counters.txt) -- no row padding can be conflict-free for those groups.
__device__ __forceinline__ int img_off(int row, int col) { return row * (ALD * 2) + ((((col >> 3) ^ (row & 6))) << 4) + ((col & 7) << 1); }
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define NEG_BIG (-1.0e30f)

struct AttnParams {
    const float* Q; const float* K; const float* V; int ldq, ldk, ldv;
    float* O; int ldo;                 // fwd: output; dq mode: dQ output
    float* LSE;                        // [B,H,Tq]  fwd: written; bwd: read
    const float* dO; int lddo;         // bwd
    const float* Delta;                // [B,H,Tq]  bwd
    float* dK; float* dV; int lddk, lddv;
    const int* lens_k;
    const int* lens_q;                 // bwd, optional: queries t >= lens_q[b] carry a zero dO (the caller's guarantee): their tiles are not visited
    int B, H, Tq, Tk, causal;
    float scale;
    uint32_t drop_thresh; float drop_scale; uint32_t seed, stream;
    int qkv_split;                     // Q, K, V (and dO) arrive in the pre-split operand format (GEMM out_split): no fp32 -> hi/lo conversion here
    int order;                         // block dispatch order, see xcd_remap
};

template <int NSPLIT>
__device__ __forceinline__ void split8(const float4& a, const float4& b, bf16x8_t& hi, bf16x8_t& lo) {
    u32x2 h0, l0, h1, l1;
    split4<NSPLIT>(a, h0, l0);
    split4<NSPLIT>(b, h1, l1);
    u32x4 h = {h0[0], h0[1], h1[0], h1[1]}, l = {l0[0], l0[1], l1[0], l1[1]};
    hi = __builtin_bit_cast(bf16x8_t, h);
    lo = __builtin_bit_cast(bf16x8_t, l);
}

// 8 consecutive operand elements at `src` -> hi/lo fragments: fp32 values (optionally scaled) are split here; pre-split data
// ([hi x4 | lo x4] per 16-byte chunk) is only re-packed.  `ok` false -> zeros.
template <int NSPLIT>
__device__ __forceinline__ void load_frag8(const float* __restrict__ src, bool ok, bool split_in, float pre_scale, bf16x8_t& hi, bf16x8_t& lo) {
    float4 a = make_float4(0, 0, 0, 0), c = a;
    if (ok) {
        a = *reinterpret_cast<const float4*>(src);
        c = *reinterpret_cast<const float4*>(src + 4);
    }
    if (split_in) {
        u32x4 h = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(c.x), __float_as_uint(c.y)};
        u32x4 l = {__float_as_uint(a.z), __float_as_uint(a.w), __float_as_uint(c.z), __float_as_uint(c.w)};
        hi = __builtin_bit_cast(bf16x8_t, h);
        lo = __builtin_bit_cast(bf16x8_t, l);
    } else {
        a.x *= pre_scale; a.y *= pre_scale; a.z *= pre_scale; a.w *= pre_scale; c.x *= pre_scale; c.y *= pre_scale; c.z *= pre_scale; c.w *= pre_scale;
        split8<NSPLIT>(a, c, hi, lo);
    }
}

// 8 fp32 values (two accumulator quads) -> hi/lo bf16x8 operand fragments
template <int NSPLIT>
__device__ __forceinline__ void pack_acc(const f32x4& a, const f32x4& b, bf16x8_t& hi, bf16x8_t& lo) {
    split8<NSPLIT>(make_float4(a[0], a[1], a[2], a[3]), make_float4(b[0], b[1], b[2], b[3]), hi, lo);
}

// fragment [row = r0 + (lane&15)][k = 32*kst + 8g .. +7] of a [rows][ALD] image
__device__ __forceinline__ bf16x8_t row_frag(const unsigned char* img, int r0, int kst, int l15, int g) {
    // r0 is a multiple of 16: (row & 6) = (l15 & 6) and chunk 4*kst + g = (kst << 2) ^ g, so the swizzle term is one lane constant
    const int sw = g ^ (l15 & 6);
    return *reinterpret_cast<const bf16x8_t*>(img + (r0 + l15) * (ALD * 2) + (((kst << 2) ^ sw) << 4));
}
// transposed fragment: lane gets [col = c0 + (lane&15)][k-slots: rows k0+4g+0..3 then k0+16+4g+0..3]
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* img, int k0, int c0, int l15, int g) {
    const int q = l15 >> 2, p = l15 & 3;
    // k0 and c0 are multiples of 16: row & 6 = (4g + q) & 6 for both rows, chunk = (c0 >> 3) ^ (p >> 1)
    const int lane_off = (4 * g + q) * (ALD * 2) + ((p & 1) << 3);
    const int sw = (p >> 1) ^ ((4 * g + q) & 6);
    const unsigned char* base = img + k0 * (ALD * 2) + lane_off + ((((c0 >> 3) ^ sw)) << 4);
    s16x4 v0 = lds_read_tr16(base);
    s16x4 v1 = lds_read_tr16(base + 16 * (ALD * 2));
    s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int NSPLIT>
__device__ __forceinline__ f32x4 mma3(const bf16x8_t& ah, const bf16x8_t& al, const bf16x8_t& bh, const bf16x8_t& bl, f32x4 c) {
    if (NSPLIT == 3) {
        c = mfma16(al, bh, c);
        c = mfma16(ah, bl, c);
    }
    return mfma16(ah, bh, c);
}

// Products with one operand made by the kernel itself (P, dS) in ONE bf16 part: that operand's low part is left out, i.e. two MFMAs.
// mma2_a: A = a_hi only (a_hi*b_lo + a_hi*b_hi); mma2_b: B = b_hi only (a_lo*b_hi + a_hi*b_hi).
template <int NSPLIT>
__device__ __forceinline__ f32x4 mma2_a(const bf16x8_t& ah, const bf16x8_t& bh, const bf16x8_t& bl, f32x4 c) {
    if (NSPLIT == 3) c = mfma16(ah, bl, c);
    return mfma16(ah, bh, c);
}
template <int NSPLIT>
__device__ __forceinline__ f32x4 mma2_b(const bf16x8_t& ah, const bf16x8_t& al, const bf16x8_t& bh, f32x4 c) {
    if (NSPLIT == 3) c = mfma16(al, bh, c);
    return mfma16(ah, bh, c);
}

// global [rows x 64] fp32 tile -> registers (NF float4 per thread); rows >= rows_valid read as zero
template <int NROWS, int NT = 256>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int ld, int rows_valid, float4 (&r)[NROWS * 16 / NT], int t) {
#pragma unroll
    for (int i = 0; i < NROWS * 16 / NT; ++i) {
        const int idx = t + NT * i;
        const int row = idx >> 4, dq = (idx & 15) * 4;
        r[i] = (row < rows_valid) ? *reinterpret_cast<const float4*>(src + (size_t)row * ld + dq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int NROWS, int NSPLIT, int NT = 256>
__device__ __forceinline__ void tile_store(const float4 (&r)[NROWS * 16 / NT], unsigned char* hi_img, unsigned char* lo_img, int t, bool split_in = false) {
#pragma unroll
    for (int i = 0; i < NROWS * 16 / NT; ++i) {
        const int idx = t + NT * i;
        const int row = idx >> 4, dq = (idx & 15) * 4;
        u32x2 hi, lo;
        if (split_in) {     // already [hi x4 | lo x4]
            hi[0] = __float_as_uint(r[i].x); hi[1] = __float_as_uint(r[i].y); lo[0] = __float_as_uint(r[i].z); lo[1] = __float_as_uint(r[i].w);
        } else {
            split4<NSPLIT>(r[i], hi, lo);
        }
        *reinterpret_cast<u32x2*>(hi_img + img_off(row, dq)) = hi;
        if (NSPLIT == 3) *reinterpret_cast<u32x2*>(lo_img + img_off(row, dq)) = lo;
    }
}

// XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so with the natural
// (x fastest) order the nx blocks that sweep the SAME (batch, head) K/V (or Q/dO) land on nx different L2s and each one
// re-fetches it from HBM (measured: FETCH_SIZE 2.3x the algorithmic bytes).  Here all nx blocks of one (b,h) get linear
// ids that are equal mod 8.  Placement is a speed matter only; the grid is padded to a multiple of 8 (b,h) groups.
// order = 1: the blocks of ONE position x of all (b,h) of an XCD before those of the next x -- with ragged lengths (800 = 6 x 128 + 32) the
// short last block of every (b,h) is then dispatched at the END of the launch, where it fills the tail instead of leaving a full block
// to finish alone (at the price of a (b,h)'s blocks no longer meeting in the L2: its K/V come from the MALL the later times).
__device__ __forceinline__ bool xcd_remap(int nx, int nbh, int& x, int& bh, int order = 0) {
    const int L = blockIdx.x;
    const int g = L & 7, s = L >> 3;
    if (order) {
        const int nbh8 = (nbh + 7) >> 3;
        x = s / nbh8;
        bh = (s - x * nbh8) * 8 + g;
    } else {
        bh = (s / nx) * 8 + g;
        x = s % nx;
    }
    return bh < nbh;
}
static inline unsigned xcd_grid(int nx, int nbh) { return (unsigned)(((nbh + 7) / 8) * 8 * nx); }

// ------------------------------------------------------------------------------------------------------------
// MODE 0: forward (O, LSE).  MODE 1: dQ (recomputes P from LSE; dQ = scale * dS.K).
// 1-D grid of ceil(Tq/128) * B*H workgroups (XCD-remapped), 256 threads; wave w owns queries [128*bx + 32w, +32).
// ------------------------------------------------------------------------------------------------------------
// QS = 16-query sub-tiles per wave: 2 -> 4 waves x 32 queries (256 VGPRs, 2 waves/SIMD); 1 -> 8 waves x 16 queries, whose
// halved per-wave state (Q fragments, S and O accumulators, prefetch registers) fits 128 VGPRs = 4 waves/SIMD: the same MFMA
// and vector work per workgroup, twice the waves to hide LDS / exp / barrier latencies behind.
template <int NSPLIT, int MODE, int QS>
__global__ __launch_bounds__(128 * (4 / QS), (QS == 1 ? 4 : 2)) void attn_q_kernel(const AttnParams p) {
    constexpr int PARTS = (NSPLIT == 3) ? 2 : 1;
    constexpr int NT = 128 * (4 / QS);                        // 512 threads (QS = 1) or 256 (QS = 2): 128 queries per workgroup either way
    constexpr int IMG = 64 * ALD * 2;                         // one 64-row image
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PARTS * IMG];
    unsigned char* sK[2] = {smem, smem + (PARTS - 1) * IMG};
    unsigned char* sV[2] = {smem + PARTS * IMG, smem + PARTS * IMG + (PARTS - 1) * IMG};

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    int bx, bh;
    if (!xcd_remap((p.Tq + 127) / 128, p.B * p.H, bx, bh, p.order)) return;       // whole workgroup exits (EXEC stays full elsewhere)
    if (p.causal) bx = (p.Tq + 127) / 128 - 1 - bx;                     // late query blocks see the most keys: dispatch them first
    const int h = bh % p.H, b = bh / p.H;
    const int qblk = bx * 128;
    const int q0 = qblk + wave * 16 * QS;
    const bool wave_live = __builtin_amdgcn_readfirstlane((int)(q0 < p.Tq)) != 0;
    const int klen = p.lens_k ? min(p.Tk, p.lens_k[b]) : p.Tk;
    int kmax = klen;
    if (p.causal) kmax = min(kmax, qblk + 128);
    const int nkt = (kmax + 63) / 64;

    const float* Qb = p.Q + (size_t)b * p.Tq * p.ldq + h * HD;
    const float* Kb = p.K + (size_t)b * p.Tk * p.ldk + h * HD;
    const float* Vb = p.V + (size_t)b * p.Tk * p.ldv + h * HD;
    const float sc = p.scale * LOG2E;

    // per-wave query-side operand fragments: lane holds X[q = q0+16qs+l15][d = 32kst+8g..+7]
    // Scores are wanted in log2 units: fp32 Q is multiplied by scale*log2(e) before it is split; pre-split Q cannot be, so
    // there the factor rides in the exponent's fma (ssc) instead -- the same instruction count either way.
    const bool split_in = p.qkv_split != 0;
    const float ssc = split_in ? sc : 1.f;
    bf16x8_t qf[QS][2][PARTS], dof[QS][2][PARTS];
    float lse2[QS], delta[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) { lse2[qs] = 0.f; delta[qs] = 0.f; }
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int q = q0 + 16 * qs + l15;
        const bool ok = q < p.Tq;
#pragma unroll
        for (int kst = 0; kst < 2; ++kst) {
            bf16x8_t hi, lo;
            load_frag8<NSPLIT>(Qb + (size_t)q * p.ldq + 32 * kst + 8 * g, ok, split_in, sc, hi, lo);
            qf[qs][kst][0] = hi;
            if (PARTS == 2) qf[qs][kst][PARTS - 1] = lo;
            if (MODE == 1) {
                load_frag8<NSPLIT>(p.dO + ((size_t)b * p.Tq + q) * p.lddo + h * HD + 32 * kst + 8 * g, ok, split_in, 1.f, hi, lo);
                dof[qs][kst][0] = hi;
                if (PARTS == 2) dof[qs][kst][PARTS - 1] = lo;
            }
        }
        if (MODE == 1 && ok) {
            lse2[qs] = p.LSE[((size_t)b * p.H + h) * p.Tq + q] * LOG2E;
            delta[qs] = p.Delta[((size_t)b * p.H + h) * p.Tq + q];
        }
    }

    uint32_t rkeys[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) rkeys[qs] = 0u;
    if (p.drop_thresh) {
#pragma unroll
        for (int qs = 0; qs < QS; ++qs)
            rkeys[qs] = rng_row_key(p.seed, p.stream, (uint32_t)(((size_t)b * p.H + h) * p.Tq + q0 + 16 * qs + l15));
    }
    float m[QS], lsum[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) { m[qs] = NEG_BIG; lsum[qs] = 0.f; }
    f32x4 o[4][QS];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) o[dt][qs] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 rk[1024 / NT], rv[1024 / NT];
    if (nkt > 0) {
        tile_load<64, NT>(Kb, p.ldk, min(64, p.Tk), rk, t);
        tile_load<64, NT>(Vb, p.ldv, min(64, p.Tk), rv, t);
    }
    for (int kt = 0; kt < nkt; ++kt) {
        tile_store<64, NSPLIT, NT>(rk, sK[0], sK[PARTS - 1], t, split_in);
        tile_store<64, NSPLIT, NT>(rv, sV[0], sV[PARTS - 1], t, split_in);
        __syncthreads();
        if (kt + 1 < nkt) {
            const int kr = (kt + 1) * 64;
            tile_load<64, NT>(Kb + (size_t)kr * p.ldk, p.ldk, min(64, p.Tk - kr), rk, t);
            tile_load<64, NT>(Vb + (size_t)kr * p.ldv, p.ldv, min(64, p.Tk - kr), rv, t);
        }
        // A wave whose queries all lie past Tq (the overhang of the last 128-query block: three of four waves at Tq = 800) keeps loading,
        // storing and meeting the barriers, and skips the products and the softmax.
        if (wave_live) {
        // ---- S^T[key,q] (and dP^T in dQ mode) -----------------------------------------------------------
        f32x4 s[4][QS], dp[4][QS];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) { s[ks][qs] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[ks][qs] = s[ks][qs]; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kst = 0; kst < 2; ++kst) {
                const bf16x8_t kh = row_frag(sK[0], 16 * ks, kst, l15, g);
                const bf16x8_t kl = (PARTS == 2) ? row_frag(sK[PARTS - 1], 16 * ks, kst, l15, g) : kh;
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) s[ks][qs] = mma3<NSPLIT>(kh, kl, qf[qs][kst][0], qf[qs][kst][PARTS - 1], s[ks][qs]);
                if (MODE == 1) {
                    const bf16x8_t vh = row_frag(sV[0], 16 * ks, kst, l15, g);
                    const bf16x8_t vl = (PARTS == 2) ? row_frag(sV[PARTS - 1], 16 * ks, kst, l15, g) : vh;
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs) dp[ks][qs] = mma3<NSPLIT>(vh, vl, dof[qs][kst][0], dof[qs][kst][PARTS - 1], dp[ks][qs]);
                }
            }
        // ---- softmax (fwd: online; dQ: from saved LSE) ---------------------------------------------------
        // Interior tiles (every key < klen and, if causal, below the diagonal for all 32 queries of this wave) take a
        // path without per-element mask tests; the condition is wave-uniform.
        const bool interior = (kt * 64 + 64 <= klen) && (!p.causal || kt * 64 + 63 <= q0) && (MODE == 0 || q0 + 16 * QS - 1 < p.Tq);
        auto softmax_tile = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                const int q = q0 + 16 * qs + l15;
                const uint32_t rkey = rkeys[qs];
                float mref;
                if (MODE == 0) {
                    float tmax = NEG_BIG;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = s[ks][qs][r];
                            if (MASKED) {
                                const int key = kt * 64 + 16 * ks + 4 * g + r;
                                const bool valid = key < klen && (!p.causal || key <= q);
                                v = valid ? v : NEG_BIG;
                                s[ks][qs][r] = v;
                            }
                            tmax = fmaxf(tmax, v);
                        }
                    tmax = rows4_max(tmax);
                    const float mnew = fmaxf(m[qs], tmax * ssc);
                    const float alpha = __builtin_amdgcn_exp2f(m[qs] - mnew);
                    m[qs] = mnew;
                    mref = mnew;
                    lsum[qs] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) { o[dt][qs][0] *= alpha; o[dt][qs][1] *= alpha; o[dt][qs][2] *= alpha; o[dt][qs][3] *= alpha; }
                } else {
                    mref = lse2[qs];
                }
                float rs = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int key0 = kt * 64 + 16 * ks + 4 * g;
                    uint32_t h01 = 0, h23 = 0;
                    if (p.drop_thresh) { h01 = rng_pair(rkey, (uint32_t)key0); h23 = rng_pair(rkey, (uint32_t)key0 + 2u); }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t hh = (r < 2) ? h01 : h23;
                        const bool keep = !p.drop_thresh || ((r & 1) ? rng_keep_hi(hh, p.drop_thresh) : rng_keep_lo(hh, p.drop_thresh));
                        float pv;
                        if (MODE == 0) {
                            const float v = s[ks][qs][r];
                            pv = __builtin_amdgcn_exp2f(__builtin_fmaf(v, ssc, -mref));      // masked entries: exp2(-1e30 * ssc - m) = 0
                            rs += pv;
                            s[ks][qs][r] = keep ? pv : 0.f;          // 1/(1-p) is applied once, in the epilogue's normalisation
                        } else {
                            pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ks][qs][r], ssc, -mref));
                            if (MASKED) {
                                const int key = key0 + r;
                                const bool valid = key < klen && (!p.causal || key <= q) && q < p.Tq;
                                pv = valid ? pv : 0.f;
                            }
                            const float dpe = keep ? dp[ks][qs][r] * p.drop_scale : 0.f;
                            s[ks][qs][r] = pv * (dpe - delta[qs]);           // dS^T
                        }
                    }
                }
                if (MODE == 0) {
                    lsum[qs] += rows4_sum(rs);
                }
            }
        };
        if (interior) softmax_tile(std::false_type{}); else softmax_tile(std::true_type{});
        // ---- O^T += V^T . P^T   (dQ mode: dQ^T += K^T . dS^T) ---------------------------------------------
        unsigned char* const* sX = (MODE == 0) ? sV : sK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8_t pf[QS][PARTS];
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                bf16x8_t hi, lo;
                pack_acc<NSPLIT>(s[2 * u][qs], s[2 * u + 1][qs], hi, lo);
                pf[qs][0] = hi;
                if (PARTS == 2) pf[qs][PARTS - 1] = lo;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t xh = tr_frag(sX[0], 32 * u, 16 * dt, l15, g);
                const bf16x8_t xl = (PARTS == 2) ? tr_frag(sX[PARTS - 1], 32 * u, 16 * dt, l15, g) : xh;
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) o[dt][qs] = mma3<NSPLIT>(xh, xl, pf[qs][0], pf[qs][PARTS - 1], o[dt][qs]);
            }
        }
        }   // wave_live
        __syncthreads();
    }

    // ---- epilogue: lane holds O^T[d = 16dt+4g+r][q = q0+16qs+l15] ------------------------------------------
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int q = q0 + 16 * qs + l15;
        if (q >= p.Tq) continue;
        const float f = (MODE == 0) ? (lsum[qs] > 0.f ? p.drop_scale / lsum[qs] : 0.f) : p.scale;
        float* dst = p.O + ((size_t)b * p.Tq + q) * p.ldo + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<float4*>(dst + 16 * dt) = make_float4(o[dt][qs][0] * f, o[dt][qs][1] * f, o[dt][qs][2] * f, o[dt][qs][3] * f);
        if (MODE == 0 && g == 0) p.LSE[((size_t)b * p.H + h) * p.Tq + q] = (m[qs] + log2f(lsum[qs])) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Forward on v_mfma_f32_32x32x16_bf16 (round 4).  Same work split as attn_q_kernel<*, 0, 2> -- 128 queries per workgroup, wave w owns
// queries [128 bx + 32 w, +32), 64-key tiles through LDS -- but one 32-query x 32-key accumulator tile per MFMA: the instruction holds
// the SIMD's vector issue for 8 of its 32 cycles instead of 8 of 16 (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), so the
// softmax's vector instructions have three times the room under the matrix pipe, and every K fragment read from LDS feeds 32 queries.
// Layout (C/D of 32x32x16: column n = lane & 31, row = 8 (i / 4) + 4 (lane >> 5) + i % 4 for register i): S^T[key, q] = K.Q^T puts the
// QUERY on the lane, a lane holds 16 of the 32 keys of a tile and its partner lane ^ 32 the other 16: the row maximum is 15 v_max3 +
// one permlane32 swap, the row sum stays a per-lane partial until the epilogue.  Registers 8v .. 8v+7 of a tile are the B operand of
// O^T += V^T.P^T for keys [16 v, 16 v + 16) in the k-slot order 8 (j / 4) + 4 hi + j % 4, which two ds_read_b64_tr_b16 of the untransposed
// [key][d] image of V deliver.  The running maximum is only raised when some query's tile maximum exceeds it by more than 8 (in log2
// units: P <= 256, exact in fp32 and in the hi/lo split), so the rescaling pass over the 32 output accumulators leaves the loop.
// Same masks, same dropout decisions (rng_pair over adjacent keys) and the same three-term products as the 16x16x32 kernel.
// Measured and removed again (round 4; git history has it): the two 32-key halves of a tile software-pipelined inside the wave -- online
// softmax per half, dropout as a template parameter so that an interior iteration is one basic block, sched_group_barrier pinning one
// MFMA of S(h+1) / P.V(h-1) per ~22 vector instructions of softmax(h), which hipcc did emit -- ran 97 vs 92 us at 800 x 800 with dropout
// (103 vs 108 without): interleaving the two instruction kinds inside a wave buys nothing over what the SIMD's second wave already covers.
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ f32x16 mfma32(const bf16x8_t& a, const bf16x8_t& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int NSPLIT>
__device__ __forceinline__ f32x16 mma3_32(const bf16x8_t& ah, const bf16x8_t& al, const bf16x8_t& bh, const bf16x8_t& bl, f32x16 c) {
    if (NSPLIT == 3) {
        c = mfma32(al, bh, c);
        c = mfma32(ah, bl, c);
    }
    return mfma32(ah, bh, c);
}
// K image [64 keys][64 d] bf16 (128-B rows), read by rows as the A operand (lane: key = lane & 31, 16 B at d = 16 kst + 8 hi): the
// 16-B chunk index is XOR-ed with (row >> 1) & 7, which spreads the rows of every ds_read_b128 lane group ({0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}) over all 16 granules of the 64 banks.  V image, read transposed (4 keys x 16 d per 16-lane group): chunk bit 2
// is flipped for rows with (row >> 1) & 1, so the four rows of a group (two per bank half) take disjoint 64-B spans.
__device__ __forceinline__ int kimg_off(int row, int col) { return row * 128 + ((((col >> 3) ^ (row >> 1)) & 7) << 4) + ((col & 7) << 1); }
__device__ __forceinline__ int vimg_off(int row, int col) { return row * 128 + ((((col >> 3) ^ (((row >> 1) & 1) << 2))) << 4) + ((col & 7) << 1); }
template <int NROWS, int NSPLIT, bool VIMG, int NT = 256>
__device__ __forceinline__ void tile_store32(const float4 (&r)[NROWS * 16 / NT], unsigned char* hi_img, unsigned char* lo_img, int t, bool split_in) {
#pragma unroll
    for (int i = 0; i < NROWS * 16 / NT; ++i) {
        const int idx = t + NT * i;
        const int row = idx >> 4, dq = (idx & 15) * 4;
        u32x2 hi, lo;
        if (split_in) {
            hi[0] = __float_as_uint(r[i].x); hi[1] = __float_as_uint(r[i].y); lo[0] = __float_as_uint(r[i].z); lo[1] = __float_as_uint(r[i].w);
        } else {
            split4<NSPLIT>(r[i], hi, lo);
        }
        const int off = VIMG ? vimg_off(row, dq) : kimg_off(row, dq);
        *reinterpret_cast<u32x2*>(hi_img + off) = hi;
        if (NSPLIT == 3) *reinterpret_cast<u32x2*>(lo_img + off) = lo;
    }
}
// lane (hi, m): V[kb + 8 (j / 4) + 4 hi + j % 4][d0 + m], j = 0..7 (kb multiple of 16, d0 of 32)
__device__ __forceinline__ bf16x8_t vtr_frag32(const unsigned char* img, int kb, int d0, int lane) {
    const int hi = lane >> 5, dh = (lane >> 4) & 1, li = lane & 15;
    const int row = kb + 4 * hi + (li >> 2), col = d0 + 16 * dh + 4 * (li & 3);
    s16x4 v0 = lds_read_tr16(img + vimg_off(row, col));
    s16x4 v1 = lds_read_tr16(img + vimg_off(row + 8, col));
    s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ float half_max(float v) {         // max over the lane pair (l, l ^ 32)
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float z = __uint_as_float(b[0]), w = __uint_as_float(b[1]);
    asm("v_max_f32 %0, %1, %2" : "=v"(z) : "v"(z), "v"(w));
    return z;
}
__device__ __forceinline__ float half_sum(float v) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
#define ATTN_DEFER_MAX 8.0f

template <int NSPLIT>
__global__ __launch_bounds__(256, 2) void attn_fwd32_kernel(const AttnParams p) {
    constexpr int PARTS = (NSPLIT == 3) ? 2 : 1;
    constexpr int IMG = 64 * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PARTS * IMG];
    unsigned char* sK[2] = {smem, smem + (PARTS - 1) * IMG};
    unsigned char* sV[2] = {smem + PARTS * IMG, smem + PARTS * IMG + (PARTS - 1) * IMG};

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n = lane & 31, hi = lane >> 5;
    int bx, bh;
    if (!xcd_remap((p.Tq + 127) / 128, p.B * p.H, bx, bh, p.order)) return;
    if (p.causal) bx = (p.Tq + 127) / 128 - 1 - bx;
    const int h = bh % p.H, b = bh / p.H;
    const int qblk = bx * 128;
    const int q0 = qblk + wave * 32;
    const bool wave_live = __builtin_amdgcn_readfirstlane((int)(q0 < p.Tq)) != 0;
    const int klen = p.lens_k ? min(p.Tk, p.lens_k[b]) : p.Tk;
    int kmax = klen;
    if (p.causal) kmax = min(kmax, qblk + 128);
    const int nkt = (kmax + 63) / 64;

    const float* Qb = p.Q + (size_t)b * p.Tq * p.ldq + h * HD;
    const float* Kb = p.K + (size_t)b * p.Tk * p.ldk + h * HD;
    const float* Vb = p.V + (size_t)b * p.Tk * p.ldv + h * HD;
    const float sc = p.scale * LOG2E;
    const bool split_in = p.qkv_split != 0;
    const float ssc = split_in ? sc : 1.f;
    const int q = q0 + n;
    const bool qok = q < p.Tq;

    bf16x8_t qf[4][PARTS];                 // B operand of S^T: lane holds Q[q][16 kst + 8 hi .. +7]
#pragma unroll
    for (int kst = 0; kst < 4; ++kst) {
        bf16x8_t fh, fl;
        load_frag8<NSPLIT>(Qb + (size_t)q * p.ldq + 16 * kst + 8 * hi, qok, split_in, sc, fh, fl);
        qf[kst][0] = fh;
        if (PARTS == 2) qf[kst][PARTS - 1] = fl;
    }
    // dropout: (key >> 1) = 32 kt + 2 hi + c with c = 16 t2 + 4 (i / 4) + (i % 4) / 2 < 32 in bit fields of its own, so the sum is an XOR
    const uint32_t rkey = p.drop_thresh ? rng_row_key(p.seed, p.stream, (uint32_t)(((size_t)b * p.H + h) * p.Tq + q)) : 0u;
    float m = NEG_BIG, lsum = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;

    float4 rk[4], rv[4];
    if (nkt > 0) {
        tile_load<64>(Kb, p.ldk, min(64, p.Tk), rk, t);
        tile_load<64>(Vb, p.ldv, min(64, p.Tk), rv, t);
    }
    // KO_*: timing-only knock-out builds (tools/attn_knockout.sh; never defined in the shipped library; results are wrong with any of them)
    for (int kt = 0; kt < nkt; ++kt) {
#ifndef KO_LDS_STORE
        tile_store32<64, NSPLIT, false>(rk, sK[0], sK[PARTS - 1], t, split_in);
        tile_store32<64, NSPLIT, true>(rv, sV[0], sV[PARTS - 1], t, split_in);
#endif
#ifndef KO_BARRIER
        __syncthreads();
#endif
#ifndef KO_GLOAD
        if (kt + 1 < nkt) {
            const int kr = (kt + 1) * 64;
            tile_load<64>(Kb + (size_t)kr * p.ldk, p.ldk, min(64, p.Tk - kr), rk, t);
            tile_load<64>(Vb + (size_t)kr * p.ldv, p.ldv, min(64, p.Tk - kr), rv, t);
        }
#endif
        if (wave_live) {
        f32x16 s[2];
        // Every K fragment of the tile is requested from LDS BEFORE the first MFMA (16 ds_read_b128, 64 VGPRs that are free in this phase):
        // left to itself hipcc placed each read right in front of the one or two MFMAs that use it, with an s_waitcnt in between -- the full
        // LDS latency in front of every MFMA pair, and the S product at half the matrix pipe's rate (the knock-out builds priced 24 MFMAs
        // at 30 us of a 114-us launch; the pipe needs 15).  The empty asm pins the order: its operands must all have arrived.
        bf16x8_t kfh[2][4], kfl[2][4];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int kst = 0; kst < 4; ++kst) {
                const int off = kimg_off(32 * t2 + n, 16 * kst + 8 * hi);
                kfh[t2][kst] = *reinterpret_cast<const bf16x8_t*>(sK[0] + off);
                kfl[t2][kst] = (PARTS == 2) ? *reinterpret_cast<const bf16x8_t*>(sK[PARTS - 1] + off) : kfh[t2][kst];
            }
        asm volatile("" : "+v"(kfh[0][0]), "+v"(kfh[0][1]), "+v"(kfh[0][2]), "+v"(kfh[0][3]), "+v"(kfh[1][0]), "+v"(kfh[1][1]), "+v"(kfh[1][2]), "+v"(kfh[1][3]));
        if (PARTS == 2) asm volatile("" : "+v"(kfl[0][0]), "+v"(kfl[0][1]), "+v"(kfl[0][2]), "+v"(kfl[0][3]), "+v"(kfl[1][0]), "+v"(kfl[1][1]), "+v"(kfl[1][2]), "+v"(kfl[1][3]));
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[t2][i] = 0.f;
#pragma unroll
            for (int kst = 0; kst < 4; ++kst) {
#ifdef KO_SMFMA
                s[t2][kst] += __builtin_bit_cast(f32x4, kfh[t2][kst])[0] + __builtin_bit_cast(f32x4, kfl[t2][kst])[1];
#else
                s[t2] = mma3_32<NSPLIT>(kfh[t2][kst], kfl[t2][kst], qf[kst][0], qf[kst][PARTS - 1], s[t2]);
#endif
            }
        }
        const bool interior = (kt * 64 + 64 <= klen) && (!p.causal || kt * 64 + 63 <= q0);
        auto softmax_tile = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            float tmax = NEG_BIG;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = s[t2][i];
                    if (MASKED) {
                        const int key = kt * 64 + 32 * t2 + 8 * (i >> 2) + 4 * hi + (i & 3);
                        const bool valid = key < klen && (!p.causal || key <= q);
                        v = valid ? v : NEG_BIG;
                        s[t2][i] = v;
                    }
                    tmax = fmaxf(tmax, v);
                }
            const float ts = half_max(tmax) * ssc;
            // The running maximum follows a query's tile maximum only once that exceeds it by more than ATTN_DEFER_MAX (P <= 2^8 meanwhile:
            // exact in fp32 and in the hi/lo split), so alpha is exactly 1 in most iterations.  Branch-free on purpose: with the rescaling
            // pass under a wave-uniform branch the compiler kept three copies of the 32 output accumulators (94 v_mov_b64 per iteration).
            const float mnew = (ts > m + ATTN_DEFER_MAX) ? ts : m;
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            lsum *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt] = o[dt] * alpha;
            const uint32_t rk_it = rkey ^ (uint32_t)(kt * 32 + 2 * hi);
            float rs = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ip = 0; ip < 8; ++ip) {
                    const int i = 2 * ip;
                    uint32_t hh = 0;
                    if (p.drop_thresh) {
                        hh = (rk_it ^ (uint32_t)(16 * t2 + 4 * (i >> 2) + ((i & 3) >> 1))) * 0x9E3779B1u;
                        hh ^= hh >> 15;
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const bool keep = !p.drop_thresh || (e ? rng_keep_hi(hh, p.drop_thresh) : rng_keep_lo(hh, p.drop_thresh));
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t2][i + e], ssc, -m));      // masked entries: exp2(-1e30 * ssc - m) = 0
                        rs += pv;
                        s[t2][i + e] = keep ? pv : 0.f;
                    }
                }
            lsum += rs;
        };
#ifndef KO_SOFTMAX
        if (interior) softmax_tile(std::false_type{}); else softmax_tile(std::true_type{});
#endif
        // the same for the V fragments, one 32-key half ahead of its MFMAs (16 transposed reads = 32 VGPRs per half)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            bf16x8_t vfh[2][2], vfl[2][2];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    vfh[v][dt] = vtr_frag32(sV[0], 32 * t2 + 16 * v, 32 * dt, lane);
                    vfl[v][dt] = (PARTS == 2) ? vtr_frag32(sV[PARTS - 1], 32 * t2 + 16 * v, 32 * dt, lane) : vfh[v][dt];
                }
            bf16x8_t ph[2], pl[2];
#pragma unroll
            for (int v = 0; v < 2; ++v)
                split8<NSPLIT>(make_float4(s[t2][8 * v], s[t2][8 * v + 1], s[t2][8 * v + 2], s[t2][8 * v + 3]),
                               make_float4(s[t2][8 * v + 4], s[t2][8 * v + 5], s[t2][8 * v + 6], s[t2][8 * v + 7]), ph[v], pl[v]);
            asm volatile("" : "+v"(vfh[0][0]), "+v"(vfh[0][1]), "+v"(vfh[1][0]), "+v"(vfh[1][1]));
            if (PARTS == 2) asm volatile("" : "+v"(vfl[0][0]), "+v"(vfl[0][1]), "+v"(vfl[1][0]), "+v"(vfl[1][1]));
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
#ifdef KO_PVMFMA
                    o[dt][2 * t2 + v] += __builtin_bit_cast(f32x4, vfh[v][dt])[0] + __builtin_bit_cast(f32x4, vfl[v][dt])[1] + __builtin_bit_cast(f32x4, ph[v])[0] + __builtin_bit_cast(f32x4, pl[v])[1];
#else
                    o[dt] = mma3_32<NSPLIT>(vfh[v][dt], vfl[v][dt], ph[v], pl[v], o[dt]);
#endif
                }
        }
        }   // wave_live
#ifndef KO_BARRIER
        __syncthreads();
#endif
    }
    // ---- epilogue: lane holds O^T[d = 32 dt + 8 (i / 4) + 4 hi + i % 4][q] --------------------------------------
    const float ltot = half_sum(lsum);
    if (qok) {
        const float f = ltot > 0.f ? p.drop_scale / ltot : 0.f;
        float* dst = p.O + ((size_t)b * p.Tq + q) * p.ldo + h * HD + 4 * hi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                *reinterpret_cast<float4*>(dst + 32 * dt + 8 * i4) = make_float4(o[dt][4 * i4] * f, o[dt][4 * i4 + 1] * f, o[dt][4 * i4 + 2] * f, o[dt][4 * i4 + 3] * f);
        if (hi == 0) p.LSE[((size_t)b * p.H + h) * p.Tq + q] = (m + log2f(ltot)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------------------
// dK / dV.  grid (ceil(Tk/128), H, B); wave w owns keys [128*bx + 32w, +32) and keeps dK^T, dV^T [64 x 32] in
// accumulators while the workgroup sweeps 32-query tiles of Q and dO through LDS.
// ------------------------------------------------------------------------------------------------------------
// FUSE_DQ = 1: the same pass also produces dQ (5 MFMA products per (query, key) tile instead of the 7 of dQ-kernel + dK/dV-kernel,
// which both recompute S and dP).  dQ[q,d] = scale * sum_key dS[q,key] K[key,d] contracts over the key, which sits on the LANE of
// the dS accumulators, so dS crosses LDS once: every wave writes its [32 keys x 32 queries] block (the hi/lo bf16 words it packs
// for the dK product anyway) into a [128 keys][2 x 32 queries] image (two query tiles side by side = double buffer), and in the
// NEXT iteration -- behind the barrier that is there anyway -- wave w multiplies the whole 128-key column block by the
// K image: dQ[32 q x 16 d (slice w)] = dS^T-fragments (transposed reads) x K-fragments (transposed reads, same
// key permutation).  The result is the workgroup's share of dQ; key blocks are summed with fp32 atomics (each wave-instruction
// adds 4 rows x 64 contiguous bytes), dQ is zeroed by the caller.
// PLO = 0 (the shipped one-pass form): the probabilities P and the score gradients dS, which the kernel makes itself, enter the dV, dK and
// dQ products as ONE bf16 part each -- the terms P_lo.dO_hi, dS_lo.Q_hi and dS_lo.K_hi are left out: 12 MFMAs per tile product instead of
// 15, no low-part split of P and dS (2.5 vector instructions per score each), half the dS image.  tools/oracle_emu_attn_terms.py
// (fp64 oracle, the kernels' products emulated term by term) prices it: gradient error against fp64 median 1.6e-4 -> 1.7e-4 / 4.5e-5 ->
// 5.0e-5, the worst gradient-norm error unchanged (1.6e-4; 1.2e-3 -> 1.5e-3 on the tensors DESIGN section 3 lists), outputs untouched.
// (The FORWARD's P_lo.V_hi term is a different matter: without it the outputs move by 1.2e-3 - 2.1e-3, beyond north_star's 1e-3; kept.)
template <int NSPLIT, int FUSE_DQ, int PLO>
__global__ __launch_bounds__(256, 2) void attn_dkv_kernel(const AttnParams p) {
    constexpr int PARTS = (NSPLIT == 3) ? 2 : 1;
    constexpr int SPARTS = PLO ? PARTS : 1;                   // parts of P / dS
    constexpr int IMG = 32 * ALD * 2;
    constexpr int IMG128 = 128 * ALD * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PARTS * IMG + (FUSE_DQ ? (PARTS + SPARTS) * IMG128 : 0)];
    unsigned char* sQ[2] = {smem, smem + (PARTS - 1) * IMG};
    unsigned char* sD[2] = {smem + PARTS * IMG, smem + PARTS * IMG + (PARTS - 1) * IMG};
    unsigned char* const fbase = smem + 2 * PARTS * IMG;
    unsigned char* sKs[2] = {fbase, fbase + (PARTS - 1) * IMG128};                               // K * scale, [128 keys][64 d]
    unsigned char* sS[2] = {fbase + PARTS * IMG128, fbase + PARTS * IMG128 + (SPARTS - 1) * IMG128};  // dS, [128 keys][buf*32 + q]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    int bx, bh;
    if (!xcd_remap((p.Tk + 127) / 128, p.B * p.H, bx, bh, p.order)) return;
    const int h = bh % p.H, b = bh / p.H;
    const int kblk = bx * 128;
    const int k0 = kblk + wave * 32;
    const int klen = p.lens_k ? min(p.Tk, p.lens_k[b]) : p.Tk;
    // A wave whose 32 keys are all masked (the tail of a sequence shorter than the block: lengths are ragged) contributes nothing: it
    // keeps loading, storing, meeting the barriers and taking its share of the dQ product, and skips S, dP, the pointwise phase and
    // the dK / dV products; its accumulators stay zero (what the epilogue writes for masked keys).  The dQ product then covers the
    // live 32-key chunks of the block only (the dead waves' rows of the dS image are never written).
    const bool wave_live = __builtin_amdgcn_readfirstlane((int)(k0 < klen)) != 0;
    const int kk_live = min(4, (max(klen - kblk, 0) + 31) >> 5);
    const float sc = p.scale * LOG2E;
    const float* Qb = p.Q + (size_t)b * p.Tq * p.ldq + h * HD;
    const float* Kb = p.K + (size_t)b * p.Tk * p.ldk + h * HD;
    const float* Vb = p.V + (size_t)b * p.Tk * p.ldv + h * HD;
    const float* dOb = p.dO + (size_t)b * p.Tq * p.lddo + h * HD;
    const float* lseb = p.LSE + ((size_t)b * p.H + h) * p.Tq;
    const float* delb = p.Delta + ((size_t)b * p.H + h) * p.Tq;

    // key-side operand fragments in registers: lane holds X[key = k0+16ks+l15][d = 32kst+8g..+7]
    const bool split_in = p.qkv_split != 0;
    bf16x8_t kf[2][2][PARTS], vf[2][2][PARTS];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int key = k0 + 16 * ks + l15;
        const bool ok = key < p.Tk;
#pragma unroll
        for (int kst = 0; kst < 2; ++kst) {
            bf16x8_t hi, lo;
            if (!FUSE_DQ) {      // (one-pass form: the K fragments are re-read from the workgroup's K image every iteration, see below)
                load_frag8<NSPLIT>(Kb + (size_t)key * p.ldk + 32 * kst + 8 * g, ok, split_in, 1.f, hi, lo);
                kf[ks][kst][0] = hi; if (PARTS == 2) kf[ks][kst][PARTS - 1] = lo;
            }
            load_frag8<NSPLIT>(Vb + (size_t)key * p.ldv + 32 * kst + 8 * g, ok, split_in, 1.f, hi, lo);
            vf[ks][kst][0] = hi; if (PARTS == 2) vf[ks][kst][PARTS - 1] = lo;
        }
    }
    f32x4 dk[4][2], dv[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { dk[dt][ks] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt][ks] = dk[dt][ks]; }

    const uint32_t rbase = rng_stream_base(p.seed, p.stream);      // rng_row_key(seed, stream, row) = pcg(row + rbase)
    const bool block_live = kblk < klen;                    // all keys of the block masked -> gradients are zero
    const int qt_begin = p.causal ? (kblk / 32) : 0;
    const int qlen = p.lens_q ? min(p.Tq, max(p.lens_q[b], 0)) : p.Tq;
    const int qt_end = block_live ? (qlen + 31) / 32 : qt_begin;
    float4 rq[2], rd[2];
    // Softmax statistics of the query tile: lane L of every wave prefetches LSE (lanes 0-31) or Delta (lanes 32-63) of query
    // L & 31 together with the Q / dO tile; the pointwise phase fetches its 2 x 4 rows with lane shuffles instead of eight
    // dependent global loads in the middle of the iteration (the workgroup's LDS is full: 2 x 80 KB per CU).
    float rl = 0.f, rl_cur = 0.f;
    const float* const stat_src = (lane < 32) ? lseb : delb;
    const float stat_mul = (lane < 32) ? LOG2E : 1.f;
    auto stat_load = [&](int qtile) {          // a bare load: arithmetic on the result here would wait for it on the spot
        rl = stat_src[min(qtile * 32 + (lane & 31), p.Tq - 1)];      // rows past Tq: any finite value, their probabilities are masked
    };
    if (FUSE_DQ && qt_begin < qt_end) {          // the workgroup's 128 keys as the B operand of the dQ product (scale applied to dQ itself)
        float4 rk8[8];
        tile_load<128>(Kb + (size_t)kblk * p.ldk, p.ldk, min(128, p.Tk - kblk), rk8, t);
        tile_store<128, NSPLIT>(rk8, sKs[0], sKs[PARTS - 1], t, split_in);
    }
    if (qt_begin < qt_end) {
        tile_load<32>(Qb + (size_t)qt_begin * 32 * p.ldq, p.ldq, min(32, p.Tq - qt_begin * 32), rq, t);
        tile_load<32>(dOb + (size_t)qt_begin * 32 * p.lddo, p.lddo, min(32, p.Tq - qt_begin * 32), rd, t);
        stat_load(qt_begin);
    }
    // dQ of query tile `qtile` from the dS image's half `buf`: wave w owns head-dim columns [16w, 16w+16)
    auto dq_phase = [&](int qtile, int buf) {
        f32x4 dq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk >= kk_live) break;                     // (workgroup-uniform)
            const bf16x8_t bh = tr_frag(sKs[0], 32 * kk, 16 * wave, l15, g);
            const bf16x8_t bl = (PARTS == 2) ? tr_frag(sKs[PARTS - 1], 32 * kk, 16 * wave, l15, g) : bh;
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                const bf16x8_t ah = tr_frag(sS[0], 32 * kk, 32 * buf + 16 * qs, l15, g);
                if (SPARTS == 2) {
                    const bf16x8_t al = tr_frag(sS[SPARTS - 1], 32 * kk, 32 * buf + 16 * qs, l15, g);
                    dq[qs] = mma3<NSPLIT>(ah, al, bh, bl, dq[qs]);          // D[m = q][n = d]
                } else {
                    dq[qs] = mma2_a<NSPLIT>(ah, bh, bl, dq[qs]);
                }
            }
        }
        float* dqb = p.O + (size_t)b * p.Tq * p.ldo + h * HD + 16 * wave + l15;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qtile * 32 + 16 * qs + 4 * g + r;
                if (q < p.Tq) unsafeAtomicAdd(dqb + (size_t)q * p.ldo, dq[qs][r] * p.scale);
            }
    };
    for (int qt = qt_begin; qt < qt_end; ++qt) {
        tile_store<32, NSPLIT>(rq, sQ[0], sQ[PARTS - 1], t, split_in);
        tile_store<32, NSPLIT>(rd, sD[0], sD[PARTS - 1], t, split_in);
        rl_cur = rl * stat_mul;
        // dropout row keys of the tile's 32 queries: lane L hashes query L & 31 ONCE; the pointwise phase fetches its eight rows by lane
        // shuffle (eight pcg hashes per lane and iteration before: 112 of the loop's ~1 000 vector issue slots)
        const uint32_t rk_cur = p.drop_thresh ? pcg_hash((uint32_t)(((size_t)b * p.H + h) * p.Tq + qt * 32 + (lane & 31)) + rbase) : 0u;
        __syncthreads();
        if (FUSE_DQ && qt > qt_begin) dq_phase(qt - 1, (qt - 1) & 1);
        if (qt + 1 < qt_end) {
            const int qr = (qt + 1) * 32;
            tile_load<32>(Qb + (size_t)qr * p.ldq, p.ldq, min(32, p.Tq - qr), rq, t);
            tile_load<32>(dOb + (size_t)qr * p.lddo, p.lddo, min(32, p.Tq - qr), rd, t);
            stat_load(qt + 1);
        }
        if (wave_live) {
        // ---- S[q,key], dP[q,key] --------------------------------------------------------------------------
        f32x4 s[2][2], dp[2][2];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { s[qs][ks] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[qs][ks] = s[qs][ks]; }
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int kst = 0; kst < 2; ++kst) {
                const bf16x8_t qh = row_frag(sQ[0], 16 * qs, kst, l15, g);
                const bf16x8_t ql = (PARTS == 2) ? row_frag(sQ[PARTS - 1], 16 * qs, kst, l15, g) : qh;
                const bf16x8_t dh = row_frag(sD[0], 16 * qs, kst, l15, g);
                const bf16x8_t dl = (PARTS == 2) ? row_frag(sD[PARTS - 1], 16 * qs, kst, l15, g) : dh;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (FUSE_DQ) {
                        // The kernel sits at the 256-VGPR limit of 2 waves/SIMD; holding K in registers as well (32 VGPRs) made it
                        // spill 18 of them inside this loop.  8 more ds_read_b128 per iteration instead: 281 -> 250 us at 800x800.
                        const bf16x8_t kh = row_frag(sKs[0], wave * 32 + 16 * ks, kst, l15, g);
                        const bf16x8_t kl = (PARTS == 2) ? row_frag(sKs[PARTS - 1], wave * 32 + 16 * ks, kst, l15, g) : kh;
                        s[qs][ks] = mma3<NSPLIT>(qh, ql, kh, kl, s[qs][ks]);
                    } else {
                        s[qs][ks] = mma3<NSPLIT>(qh, ql, kf[ks][kst][0], kf[ks][kst][PARTS - 1], s[qs][ks]);
                    }
                    dp[qs][ks] = mma3<NSPLIT>(dh, dl, vf[ks][kst][0], vf[ks][kst][PARTS - 1], dp[qs][ks]);
                }
            }
        // ---- P (dropped) and dS; lane: key = k0+16ks+l15, q = 32qt+16qs+4g+r ------------------------------
        // interior (wave-uniform): every (query, key) pair of this 32x32 sub-problem is unmasked
        const bool interior = (qt * 32 + 32 <= p.Tq) && (k0 + 32 <= klen) && (!p.causal || k0 + 31 <= qt * 32);
        // Dropout decisions: one hash serves the two adjacent keys of a lane pair (rng_pair), and here the key sits on the lane -- so the
        // even lane of a pair hashes rows r = 0, 1 of each 4-row group, the odd lane rows 2, 3, and a quad-permute DPP move hands each
        // lane its partner's two (half the multiplies of hashing per element: 16 v_mul_lo_u32 fewer per 32 x 32 tile and wave).
        auto pointwise = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            const int par = l15 & 1;
            const uint32_t sh16 = par ? 0u : 16u, thr_hi = p.drop_thresh << 16;
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                float l2[4], de[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    l2[r] = __shfl(rl_cur, 16 * qs + 4 * g + r, 64);
                    de[r] = __shfl(rl_cur, 32 + 16 * qs + 4 * g + r, 64);
                }
                uint32_t rkm[2] = {0u, 0u};
                if (p.drop_thresh) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) rkm[j] = (uint32_t)__shfl((int)rk_cur, 16 * qs + 4 * g + 2 * par + j, 64);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int key = k0 + 16 * ks + l15;
                    uint32_t hh[4] = {0u, 0u, 0u, 0u};
                    if (p.drop_thresh) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const uint32_t hm = rng_pair(rkm[j], (uint32_t)key);
                            hh[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hm, 0xA0, 0xF, 0xF, true);          // quad_perm [0,0,2,2]: the even lane's
                            hh[2 + j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hm, 0xF5, 0xF, 0xF, true);      // quad_perm [1,1,3,3]: the odd lane's
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(s[qs][ks][r] * sc - l2[r]);
                        if (MASKED) {
                            const int q = qt * 32 + 16 * qs + 4 * g + r;
                            const bool valid = q < p.Tq && key < klen && (!p.causal || key <= q);
                            pv = valid ? pv : 0.f;
                        }
                        float pd = pv, dpe = dp[qs][ks][r];
                        if (p.drop_thresh) {
                            // this lane's half of the pair hash (odd key: upper, even key: lower) moved to the top and compared as a whole word
                            const bool keep = (hh[r] << sh16) >= thr_hi;
                            pd = keep ? pv : 0.f;
                            dpe = keep ? dpe : 0.f;
                        }
                        s[qs][ks][r] = pd;                       // kept probabilities (B operand of dV; 1/(1-p) is applied to dV in the epilogue)
                        dp[qs][ks][r] = pv * __builtin_fmaf(dpe, p.drop_scale, -de[r]);         // dS (B operand of dK)
                    }
                }
            }
        };
        if (interior) pointwise(std::false_type{}); else pointwise(std::true_type{});
        // ---- dV^T += dO^T . Pd ;  dK^T += Q^T . dS  (sum over the 32 queries of the tile) ------------------
        bf16x8_t pf[2][SPARTS], sf[2][SPARTS];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t hi, lo;
            pack_acc<(SPARTS == 2 ? NSPLIT : 1)>(s[0][ks], s[1][ks], hi, lo);
            pf[ks][0] = hi; if (SPARTS == 2) pf[ks][SPARTS - 1] = lo;
            pack_acc<(SPARTS == 2 ? NSPLIT : 1)>(dp[0][ks], dp[1][ks], hi, lo);
            sf[ks][0] = hi; if (SPARTS == 2) sf[ks][SPARTS - 1] = lo;
            if (FUSE_DQ) {     // the same words, as row `key` of the dS image: queries 4g..4g+3 of both 16-row sub-tiles
                const int row = wave * 32 + 16 * ks + l15, col = 32 * (qt & 1) + 4 * g;
                const u32x4 hw = __builtin_bit_cast(u32x4, hi), lw = __builtin_bit_cast(u32x4, lo);
                *reinterpret_cast<u32x2*>(sS[0] + img_off(row, col)) = (u32x2){hw[0], hw[1]};
                *reinterpret_cast<u32x2*>(sS[0] + img_off(row, col + 16)) = (u32x2){hw[2], hw[3]};
                if (SPARTS == 2) {
                    *reinterpret_cast<u32x2*>(sS[SPARTS - 1] + img_off(row, col)) = (u32x2){lw[0], lw[1]};
                    *reinterpret_cast<u32x2*>(sS[SPARTS - 1] + img_off(row, col + 16)) = (u32x2){lw[2], lw[3]};
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8_t doh = tr_frag(sD[0], 0, 16 * dt, l15, g);
            const bf16x8_t dol = (PARTS == 2) ? tr_frag(sD[PARTS - 1], 0, 16 * dt, l15, g) : doh;
            const bf16x8_t qh = tr_frag(sQ[0], 0, 16 * dt, l15, g);
            const bf16x8_t ql = (PARTS == 2) ? tr_frag(sQ[PARTS - 1], 0, 16 * dt, l15, g) : qh;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (SPARTS == 2) {
                    dv[dt][ks] = mma3<NSPLIT>(doh, dol, pf[ks][0], pf[ks][SPARTS - 1], dv[dt][ks]);
                    dk[dt][ks] = mma3<NSPLIT>(qh, ql, sf[ks][0], sf[ks][SPARTS - 1], dk[dt][ks]);
                } else {
                    dv[dt][ks] = mma2_b<NSPLIT>(doh, dol, pf[ks][0], dv[dt][ks]);
                    dk[dt][ks] = mma2_b<NSPLIT>(qh, ql, sf[ks][0], dk[dt][ks]);
                }
            }
        }
        }   // wave_live
        __syncthreads();
    }
    if (FUSE_DQ && qt_begin < qt_end) dq_phase(qt_end - 1, (qt_end - 1) & 1);
    // ---- epilogue: lane holds dK^T[d = 16dt+4g+r][key = k0+16ks+l15] ---------------------------------------
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int key = k0 + 16 * ks + l15;
        if (key >= p.Tk) continue;
        float* dkp = p.dK + ((size_t)b * p.Tk + key) * p.lddk + h * HD + 4 * g;
        float* dvp = p.dV + ((size_t)b * p.Tk + key) * p.lddv + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *reinterpret_cast<float4*>(dkp + 16 * dt) = make_float4(dk[dt][ks][0] * p.scale, dk[dt][ks][1] * p.scale, dk[dt][ks][2] * p.scale, dk[dt][ks][3] * p.scale);
            *reinterpret_cast<float4*>(dvp + 16 * dt) = make_float4(dv[dt][ks][0] * p.drop_scale, dv[dt][ks][1] * p.drop_scale, dv[dt][ks][2] * p.drop_scale, dv[dt][ks][3] * p.drop_scale);
        }
    }
}

// Delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]; one wave per (b,q) row of H*64 = 256 columns (H == 4) or generic H.
// zero_dq (may be NULL): the same rows of dQ are cleared on the way (the one-pass backward accumulates into dQ with atomics).
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dO, int lddo, const float* __restrict__ O, int ldo,
                                                         float* __restrict__ delta, int rows, int Tq, int H, float* __restrict__ zero_dq, int lddq,
                                                         int do_split) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / Tq, q = row - b * Tq;
    for (int h0 = 0; h0 < H; h0 += 4) {                   // 4 heads (256 columns) per pass, 16 lanes per head
        const int h = h0 + (lane >> 4);
        float s = 0.f;
        if (h < H) {
            const int c = h * HD + (lane & 15) * 4;
            float4 a = *reinterpret_cast<const float4*>(dO + (size_t)row * lddo + c);
            if (do_split) {         // [hi x4 | lo x4] -> the 4 values (hi + lo, to the ~16 bits the products see anyway)
                const uint32_t h0 = __float_as_uint(a.x), h1 = __float_as_uint(a.y), l0 = __float_as_uint(a.z), l1 = __float_as_uint(a.w);
                a.x = __uint_as_float(h0 << 16) + __uint_as_float(l0 << 16);
                a.y = __uint_as_float(h0 & 0xFFFF0000u) + __uint_as_float(l0 & 0xFFFF0000u);
                a.z = __uint_as_float(h1 << 16) + __uint_as_float(l1 << 16);
                a.w = __uint_as_float(h1 & 0xFFFF0000u) + __uint_as_float(l1 & 0xFFFF0000u);
            }
            float4 o = *reinterpret_cast<const float4*>(O + (size_t)row * ldo + c);
            s = (a.x * o.x + a.y * o.y) + (a.z * o.z + a.w * o.w);
            if (zero_dq) *reinterpret_cast<float4*>(zero_dq + (size_t)row * lddq + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        if (h < H && (lane & 15) == 0) delta[((size_t)b * H + h) * Tq + q] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static int fill_common(AttnParams& p, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const int* lens_k,
                       int B, int H, int Tq, int Tk, int head_dim, int causal, float scale, float drop_p, unsigned seed, unsigned stream_id, int qkv_split) {
    UNAST_REQUIRE(Q && K && V, "unast_attn: null Q/K/V");
    UNAST_REQUIRE(head_dim == HD, "unast_attn: this build supports head_dim=%d only (got %d)", HD, head_dim);
    UNAST_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "unast_attn: bad dims");
    UNAST_REQUIRE(al16(Q) && al16(K) && al16(V) && (ldq & 3) == 0 && (ldk & 3) == 0 && (ldv & 3) == 0, "unast_attn: Q/K/V must be 16-byte aligned with ld%%4==0");
    UNAST_REQUIRE(!causal || Tq == Tk, "unast_attn: causal masking requires Tq == Tk");
    p.Q = Q; p.K = K; p.V = V; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.lens_k = lens_k;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.causal = causal; p.scale = scale;
    p.drop_thresh = drop_threshold(drop_p); p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.seed = seed; p.stream = stream_id;
    p.qkv_split = qkv_split;
    // position-major dispatch for the causal launches (blocks of unequal work: heaviest positions first, the light ones fill the tail:
    // forward 131 -> 121 us, one-pass backward 323 -> 299 us at 64 x 4 heads x 800 x 800, 1 430 -> 1 315 at 2 000 x 2 000), head-major for the
    // others (their forward loses 4-8 % to the lost L2 locality; head-major with only the short last blocks moved to the end measured equal); tools/ab_attn_order.py, UNAST_ATTN_ORDER = 0 / 1 forces one.
    static const int forced = [] { const char* e = getenv("UNAST_ATTN_ORDER"); return e ? atoi(e) : -1; }();
    p.order = forced >= 0 ? forced : (causal ? 1 : 0);
    p.lens_q = nullptr;
    p.O = nullptr; p.LSE = nullptr; p.dO = nullptr; p.Delta = nullptr; p.dK = nullptr; p.dV = nullptr; p.ldo = p.lddo = p.lddk = p.lddv = 0;
    return UNAST_OK;
}

// 1 (default, UNAST_ATTN_FWD_M32): forward on 32x32x16 MFMAs (attn_fwd32_kernel); 0: the 16x16x32 kernel.  unast_attn_fwd_variant
// switches at run time (tests compare the two).
static int g_attn_fwd_m32 = [] { const char* e = getenv("UNAST_ATTN_FWD_M32"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int unast_attn_fwd_variant(int m32) {
    const int old = g_attn_fwd_m32;
    if (m32 >= 0) g_attn_fwd_m32 = m32 ? 1 : 0;
    return old;
}

extern "C" int unast_attn_fwd(int nsplit, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                              float* LSE, const int* lens_k, int B, int H, int Tq, int Tk, int head_dim, int causal, float scale,
                              float drop_p, unsigned int seed, unsigned int stream_id, int qkv_split, hipStream_t stream) {
    AttnParams p;
    int rc = fill_common(p, Q, ldq, K, ldk, V, ldv, lens_k, B, H, Tq, Tk, head_dim, causal, scale, drop_p, seed, stream_id, qkv_split);
    if (rc) return rc;
    UNAST_REQUIRE(O && LSE && al16(O) && (ldo & 3) == 0, "unast_attn_fwd: bad output");
    UNAST_REQUIRE(nsplit == 1 || nsplit == 3, "unast_attn_fwd: nsplit must be 1 or 3");
    p.O = O; p.ldo = ldo; p.LSE = LSE;
    // 8 waves x 16 queries (128 VGPRs, 4 waves/SIMD) instead of 4 x 32 (248 VGPRs, 2 waves/SIMD): measured equal on MI355X (109 vs 108 us at
    // 800x800, every config-3 shape within 3 %) -- occupancy is not what bounds this kernel -- so the 4-wave form stays the default.
    static const bool q16 = [] { const char* e = getenv("UNAST_ATTN_FWD_Q16"); return e && e[0] == '1'; }();
    dim3 grid(xcd_grid((Tq + 127) / 128, B * H));
    if (g_attn_fwd_m32 && !q16) {
        if (nsplit == 3) hipLaunchKernelGGL((attn_fwd32_kernel<3>), grid, dim3(256), 0, stream, p);
        else             hipLaunchKernelGGL((attn_fwd32_kernel<1>), grid, dim3(256), 0, stream, p);
        return unast_check_launch("unast_attn_fwd");
    }
    if (nsplit == 3) if (q16) hipLaunchKernelGGL((attn_q_kernel<3, 0, 1>), grid, dim3(512), 0, stream, p); else hipLaunchKernelGGL((attn_q_kernel<3, 0, 2>), grid, dim3(256), 0, stream, p);
    else             if (q16) hipLaunchKernelGGL((attn_q_kernel<1, 0, 1>), grid, dim3(512), 0, stream, p); else hipLaunchKernelGGL((attn_q_kernel<1, 0, 2>), grid, dim3(256), 0, stream, p);
    return unast_check_launch("unast_attn_fwd");
}

extern "C" int unast_attn_bwd(int nsplit, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                              const float* dO, int lddo, const float* LSE, float* delta_ws, float* dQ, int lddq, float* dK, int lddk,
                              float* dV, int lddv, const int* lens_k, int B, int H, int Tq, int Tk, int head_dim, int causal, float scale,
                              float drop_p, unsigned int seed, unsigned int stream_id, int fused, int qkv_split, const int* lens_q, hipStream_t stream) {
    AttnParams p;
    int rc = fill_common(p, Q, ldq, K, ldk, V, ldv, lens_k, B, H, Tq, Tk, head_dim, causal, scale, drop_p, seed, stream_id, qkv_split);
    if (rc) return rc;
    UNAST_REQUIRE(O && dO && LSE && delta_ws && dQ && dK && dV, "unast_attn_bwd: null pointer");
    UNAST_REQUIRE(!lens_q || fused, "unast_attn_bwd: lens_q is served by the one-pass backward (fused = 1)");
    p.lens_q = lens_q;
    UNAST_REQUIRE(al16(O) && al16(dO) && al16(dQ) && al16(dK) && al16(dV) && ((ldo | lddo | lddq | lddk | lddv) & 3) == 0,
                  "unast_attn_bwd: operands must be 16-byte aligned with ld%%4==0");
    UNAST_REQUIRE(nsplit == 1 || nsplit == 3, "unast_attn_bwd: nsplit must be 1 or 3");
    const int rows = B * Tq;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, dO, lddo, O, ldo, delta_ws, rows, Tq, H, fused ? dQ : nullptr, lddq, qkv_split);
    p.dO = dO; p.lddo = lddo; p.LSE = const_cast<float*>(LSE); p.Delta = delta_ws;
    p.O = dQ; p.ldo = lddq; p.dK = dK; p.dV = dV; p.lddk = lddk; p.lddv = lddv;
    dim3 gq(xcd_grid((Tq + 127) / 128, B * H)), gk(xcd_grid((Tk + 127) / 128, B * H));
    // fused: 0 = dQ kernel + dK/dV kernel; 1 = one pass, all three terms in every product; 2 = one pass, P and dS as one bf16 part
    if (fused) {
        if (nsplit == 3) { if (fused == 1) hipLaunchKernelGGL((attn_dkv_kernel<3, 1, 1>), gk, dim3(256), 0, stream, p); else hipLaunchKernelGGL((attn_dkv_kernel<3, 1, 0>), gk, dim3(256), 0, stream, p); }
        else             hipLaunchKernelGGL((attn_dkv_kernel<1, 1, 1>), gk, dim3(256), 0, stream, p);
    } else if (nsplit == 3) {
        hipLaunchKernelGGL((attn_q_kernel<3, 1, 2>), gq, dim3(256), 0, stream, p);
        hipLaunchKernelGGL((attn_dkv_kernel<3, 0, 1>), gk, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((attn_q_kernel<1, 1, 2>), gq, dim3(256), 0, stream, p);
        hipLaunchKernelGGL((attn_dkv_kernel<1, 0, 1>), gk, dim3(256), 0, stream, p);
    }
    return unast_check_launch("unast_attn_bwd");
}

UNAST_DEFINE_RNG_EPOCH_SETTER(attention)

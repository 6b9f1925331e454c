// Kernels of ONE autoregressive decoding position (unast_amd/inference.py; reference: infer_sequence, src/network.py:219-252 and
// 455-481, which re-runs the whole decoder over the prefix at every position).  A position pushes B rows (one per sequence)
// through the decoder: every contraction has M = B <= a few dozen rows, and attention has one query per (sequence, head).  The
// general kernels (gemm.hip: 128x128 tiles staged through LDS; attention.hip: 64-query tiles) spend 12-25 us per launch on
// such shapes -- their tiles are >75 % padding and their k-loops are a chain of dependent load -> LDS -> barrier rounds -- and a
// position is ~35 such steps in sequence.  The code here is built for latency instead:
//   * linear tile: Y[M,N] = epilogue(X'[M,K] . W[N,K]^T).  A workgroup owns 32 rows x 16 output columns (N/16 workgroups:
//     16-64 CUs pull the weight rows in parallel); its 8 waves split K, load their MFMA fragments straight from global memory
//     (no LDS staging, the next round requested before the current one is multiplied), and meet once in LDS to add their partial
//     sums.  Same split-bf16 arithmetic (3 MFMAs per product, fp32 accumulate) and the same epilogue order / dropout streams
//     as gemm.hip.  What would be separate steps of the chain is folded in: the INPUT rows can be produced on the fly (LayerNorm
//     of X, LayerNorm + dropout, token embedding + positional encoding, X*scale + positional encoding -- every workgroup
//     recomputes the 32 rows from L2, the first column tile writes them out for later residual use), the input can be the frame
//     of a device-resident position, and output columns >= split_col are appended to the K/V cache row of that position.
//   * attention tile: one workgroup per (sequence, head); scores, softmax and P.V in fp32 on the vector ALUs (2 x 64 x Tk FMAs:
//     nothing), K and V each read once with 16 lanes per 256-byte row; a 16-lane group owns every NT/16-th key with a private
//     online-softmax state, so the key loop has no barrier; the states merge once at the end.
//   * end: greedy choice / frame write-back, stop rule, position += 1.
// A position is 34-36 of these launches, replayed from one captured graph.  Measured and dropped: the same tile functions as
// the phases of ONE persistent launch per position (128 workgroups walking a recorded phase list, a grid-wide barrier between
// phases -- arrival counter or per-workgroup flags, relaxed polling, one device-scope release / acquire per workgroup): results
// identical, 387 us per position against 264 us for the separate launches.  A barrier has to write back and invalidate the
// XCDs' L2s just as a kernel boundary does, so every phase still starts cold, and the boundary itself is the cheaper of the two.
#include "common.h"
#include <string.h>

enum { PRO_NONE = 0, PRO_LN = 1, PRO_LN_DROP = 2, PRO_EMBED = 3, PRO_POSENC = 4 };

struct DecLinParams {
    const float* X; int ldx; long long x_pos_stride;
    const float* W; int ldw;
    const float* bias;
    float* Y; int ldy;
    int M, N, K, act;
    uint32_t drop_thresh; float drop_scale; uint32_t seed, stream;
    const float* R; int ldr;
    int pro;
    const float* ln_g; const float* ln_b; float ln_eps;
    const int64_t* tokens; int ld_tok; const float* emb; const float* pe; float pro_scale;
    uint32_t pro_thresh1; float pro_dscale1; uint32_t pro_stream1;
    uint32_t pro_thresh2; float pro_dscale2; uint32_t pro_stream2;
    float* xn_out; int ld_xn;
    float* cache; int ld_cache; int cache_rows; int split_col; const int64_t* pos;
};

struct DecAttnParams {
    const float* Q; int ldq;
    const float* K; const float* V; int ldkv; int rows_per_seq;
    const int* lens; const int64_t* stop_lens; const int64_t* pos;     // lens, or min(stop_lens[b] + 1, *pos + 1)
    float* O; int ldo;
    int B, H; float scale;
    uint32_t drop_thresh; float drop_scale; uint32_t seed, stream;
};

struct DecEndParams {
    const float* head; int ld; int width; int B;          // text: logits [B, ld], width = vocabulary; speech: [mel | stop] rows, width = num_mels
    int64_t* tokens; int ld_tok; int eos;                 // text
    float* outputs; int ld_out; float* stops; int ld_stop; // speech
    int64_t* stop_lens; int64_t max_len; int64_t* pos; int* epoch;
};

#define DL_ROWS 32
#define DL_COLS 16
#define DL_WAVES 8
#define DL_THREADS (64 * DL_WAVES)
#define DL_LNK 256                  // rows produced on the fly have up to this many features (d_model)
#define DL_XLD (DL_LNK + 4)

struct DecShared {
    float red[DL_WAVES][DL_ROWS][DL_COLS];
    float xs[DL_ROWS * DL_XLD];
    float wm[16], wl[16];
    float wacc[16][64];
};

__device__ __forceinline__ void dl_split8(const float4& a, const float4& b, bf16x8_t& hi, bf16x8_t& lo) {
    u32x2 h0, l0, h1, l1;
    split4<3>(a, h0, l0);
    split4<3>(b, h1, l1);
    u32x4 h = {h0[0], h0[1], h1[0], h1[1]}, l = {l0[0], l0[1], l1[0], l1[1]};
    hi = __builtin_bit_cast(bf16x8_t, h);
    lo = __builtin_bit_cast(bf16x8_t, l);
}

// 8 consecutive k of one row (two 16-B loads), zero outside [0, K) or when the row is out of range
__device__ __forceinline__ void dl_load8(const float* row, bool row_ok, int k, int K, float4& a, float4& b) {
    a = (row_ok && k < K) ? *reinterpret_cast<const float4*>(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    b = (row_ok && k + 4 < K) ? *reinterpret_cast<const float4*>(row + k + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float4 dl_drop4(const float4& v, uint32_t rkey, int k, uint32_t thresh, float scale) {
    float4 o;
    o.x = rng_keep(rkey, (uint32_t)k, thresh) ? v.x * scale : 0.f;
    o.y = rng_keep(rkey, (uint32_t)k + 1u, thresh) ? v.y * scale : 0.f;
    o.z = rng_keep(rkey, (uint32_t)k + 2u, thresh) ? v.z * scale : 0.f;
    o.w = rng_keep(rkey, (uint32_t)k + 3u, thresh) ? v.w * scale : 0.f;
    return o;
}

// One 32 x 16 output tile (bx = column tile, by = row block) by a 512-thread workgroup.  ROWS: the input rows are produced into
// LDS first (p.pro != PRO_NONE).
template <bool ROWS>
__device__ __forceinline__ void dl_tile(const DecLinParams& p, int bx, int by, DecShared& sh) {
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, l15 = lane & 15, g = lane >> 4;
    const int n0 = bx * DL_COLS, m0 = by * DL_ROWS;
    const int K = p.K;
    const int64_t posv = p.pos ? p.pos[0] : 0;
    const float* X = p.X ? p.X + (size_t)posv * (size_t)p.x_pos_stride : nullptr;

    // Everything that does not depend on the produced rows or on the products is requested first, so that the tile is one
    // memory round trip deep where it can be: this thread's epilogue operands and the wave's first round of fragments.
    const int er = t >> 4, ec = t & 15, em = m0 + er, en = n0 + ec;
    const bool e_ok = em < p.M && en < p.N;
    const float e_bias = (e_ok && p.bias) ? p.bias[en] : 0.f;
    const float e_res = (e_ok && p.R) ? p.R[(size_t)em * p.ldr + en] : 0.f;
    const int ksteps = (K + 31) >> 5;
    const int n = n0 + l15;
    const bool n_ok = n < p.N;
    const float* wrow = p.W + (size_t)(n_ok ? n : 0) * p.ldw;
    const int ma = m0 + l15, mb_ = m0 + 16 + l15;
    const bool a_ok = ma < p.M, b_ok = mb_ < p.M;
    const float* xa = ROWS ? nullptr : X + (size_t)(a_ok ? ma : 0) * p.ldx;
    const float* xb = ROWS ? nullptr : X + (size_t)(b_ok ? mb_ : 0) * p.ldx;
    float4 wv[2][2], av[2][2], bv[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = (w + u * DL_WAVES) * 32 + 8 * g;           // k >= K for a step past the end: all zeros, nothing requested
        dl_load8(wrow, n_ok, k, K, wv[u][0], wv[u][1]);
        if constexpr (!ROWS) {
            dl_load8(xa, a_ok, k, K, av[u][0], av[u][1]);
            dl_load8(xb, b_ok, k, K, bv[u][0], bv[u][1]);
        }
    }

    if constexpr (ROWS) {
        // ---- the 32 input rows, 16 lanes per row (K <= 256: up to 4 x 16 B per lane)
        const int r = t >> 4, c = t & 15, m = m0 + r;
        const bool ok = m < p.M;
        const int pro = p.pro;
        const float* src;
        if (pro == PRO_EMBED) src = p.emb + (size_t)(ok ? p.tokens[(size_t)m * p.ld_tok + posv] : 0) * K;
        else src = X + (size_t)(ok ? m : 0) * p.ldx;
        float4 v[DL_LNK / 64];
#pragma unroll
        for (int i = 0; i < DL_LNK / 64; ++i) {
            const int k = (c + 16 * i) * 4;
            v[i] = (ok && k < K) ? *reinterpret_cast<const float4*>(src + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (pro == PRO_LN || pro == PRO_LN_DROP) {               // LayerNorm: biased variance, two passes over registers
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < DL_LNK / 64; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mu = s / (float)K;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < DL_LNK / 64; ++i) {
                const int k = (c + 16 * i) * 4;
                if (k < K) {
                    const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                    q += (a * a + b * b) + (cc * cc + d * d);
                }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            const float rs = rsqrtf(q / (float)K + p.ln_eps);
#pragma unroll
            for (int i = 0; i < DL_LNK / 64; ++i) {
                const int k = (c + 16 * i) * 4;
                if (k < K) {
                    const float4 gm = *reinterpret_cast<const float4*>(p.ln_g + k);
                    const float4 bt = *reinterpret_cast<const float4*>(p.ln_b + k);
                    v[i].x = (v[i].x - mu) * rs * gm.x + bt.x;
                    v[i].y = (v[i].y - mu) * rs * gm.y + bt.y;
                    v[i].z = (v[i].z - mu) * rs * gm.z + bt.z;
                    v[i].w = (v[i].w - mu) * rs * gm.w + bt.w;
                }
            }
        } else {                                                 // (dropout(embedding) | X) * scale + pe[pos]
            const uint32_t k1 = (pro == PRO_EMBED && p.pro_thresh1) ? rng_row_key(p.seed, p.pro_stream1, (uint32_t)m) : 0u;
#pragma unroll
            for (int i = 0; i < DL_LNK / 64; ++i) {
                const int k = (c + 16 * i) * 4;
                if (k < K) {
                    if (pro == PRO_EMBED && p.pro_thresh1) v[i] = dl_drop4(v[i], k1, k, p.pro_thresh1, p.pro_dscale1);
                    const float4 pe4 = *reinterpret_cast<const float4*>(p.pe + (size_t)posv * K + k);
                    v[i].x = v[i].x * p.pro_scale + pe4.x; v[i].y = v[i].y * p.pro_scale + pe4.y;
                    v[i].z = v[i].z * p.pro_scale + pe4.z; v[i].w = v[i].w * p.pro_scale + pe4.w;
                }
            }
        }
        const uint32_t k2 = (pro != PRO_LN && p.pro_thresh2) ? rng_row_key(p.seed, p.pro_stream2, (uint32_t)m) : 0u;
#pragma unroll
        for (int i = 0; i < DL_LNK / 64; ++i) {
            const int k = (c + 16 * i) * 4;
            if (k < K) {
                float4 o4 = v[i];
                if (pro != PRO_LN && p.pro_thresh2) o4 = dl_drop4(o4, k2, k, p.pro_thresh2, p.pro_dscale2);
                if (!ok) o4 = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&sh.xs[r * DL_XLD + k]) = o4;
                if (ok && p.xn_out && bx == 0) *reinterpret_cast<float4*>(p.xn_out + (size_t)m * p.ld_xn + k) = o4;
            }
        }
        __syncthreads();
    }

    // ---- partial products of this wave's k-steps (two per round, the next round requested before this one is multiplied):
    // lane holds C[m = 16*mb + l15][n = 4g .. 4g+3]
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int s0 = w; s0 < ksteps; s0 += 2 * DL_WAVES) {
        if constexpr (ROWS) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = (s0 + u * DL_WAVES) * 32 + 8 * g;
                const bool kin = k < K, kin4 = k + 4 < K;
                av[u][0] = kin ? *reinterpret_cast<const float4*>(&sh.xs[l15 * DL_XLD + k]) : make_float4(0.f, 0.f, 0.f, 0.f);
                av[u][1] = kin4 ? *reinterpret_cast<const float4*>(&sh.xs[l15 * DL_XLD + k + 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[u][0] = kin ? *reinterpret_cast<const float4*>(&sh.xs[(16 + l15) * DL_XLD + k]) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[u][1] = kin4 ? *reinterpret_cast<const float4*>(&sh.xs[(16 + l15) * DL_XLD + k + 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float4 wn[2][2], an[2][2], bn[2][2];
        const int s1 = s0 + 2 * DL_WAVES;
        if (s1 < ksteps) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = (s1 + u * DL_WAVES) * 32 + 8 * g;
                dl_load8(wrow, n_ok, k, K, wn[u][0], wn[u][1]);
                if constexpr (!ROWS) {
                    dl_load8(xa, a_ok, k, K, an[u][0], an[u][1]);
                    dl_load8(xb, b_ok, k, K, bn[u][0], bn[u][1]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8_t wh, wl, ah, al, bh, bl;
            dl_split8(wv[u][0], wv[u][1], wh, wl);
            dl_split8(av[u][0], av[u][1], ah, al);
            dl_split8(bv[u][0], bv[u][1], bh, bl);
            acc[0] = mfma16(wl, ah, acc[0]);               // small terms first
            acc[0] = mfma16(wh, al, acc[0]);
            acc[0] = mfma16(wh, ah, acc[0]);
            acc[1] = mfma16(wl, bh, acc[1]);
            acc[1] = mfma16(wh, bl, acc[1]);
            acc[1] = mfma16(wh, bh, acc[1]);
        }
        if (s1 < ksteps) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    wv[u][q] = wn[u][q];
                    if constexpr (!ROWS) { av[u][q] = an[u][q]; bv[u][q] = bn[u][q]; }
                }
        }
    }
    *reinterpret_cast<float4*>(&sh.red[w][l15][4 * g]) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    *reinterpret_cast<float4*>(&sh.red[w][16 + l15][4 * g]) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
    __syncthreads();

    // ---- sum of the 8 partials + epilogue (order as gemm.hip: bias, activation, dropout, residual): one output per thread
    if (e_ok) {
        float x = 0.f;
#pragma unroll
        for (int ww = 0; ww < DL_WAVES; ++ww) x += sh.red[ww][er][ec];
        x += e_bias;
        if (p.act == 1) x = fmaxf(x, 0.f);
        if (p.drop_thresh) x = rng_keep(rng_row_key(p.seed, p.stream, (uint32_t)em), (uint32_t)en, p.drop_thresh) ? x * p.drop_scale : 0.f;
        x += e_res;
        if (p.cache && en >= p.split_col)
            p.cache[((size_t)em * p.cache_rows + (size_t)posv) * p.ld_cache + (en - p.split_col)] = x;
        else
            p.Y[(size_t)em * p.ldy + en] = x;
    }
    __syncthreads();                                        // red / xs are reused by the workgroup's next tile
}

// Single-query attention of (sequence b, head h) by an NT-thread workgroup:
// O[b, 64h..] = dropout(softmax(scale * q . K^T over the valid keys)) . V (head dim 64; torch's multi_head_attention_forward as the
// decoder layers call it at the last position of the prefix).
template <int NT>
__device__ __forceinline__ void da_tile(const DecAttnParams& p, int bh, DecShared& sh) {
    constexpr int SLOTS = NT / 16, NW = NT / 64;
    const int b = bh / p.H, h = bh - b * p.H;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, sg = lane >> 4, c = lane & 15;
    const float* kb = p.K + (size_t)b * p.rows_per_seq * p.ldkv + h * 64 + c * 4;
    const float* vb = p.V + (size_t)b * p.rows_per_seq * p.ldkv + h * 64 + c * 4;
    // The first round of K / V rows is requested before the sequence's length is known (any row below rows_per_seq is
    // readable; rows past the length are masked below), together with the length and the query: one memory round trip.
    float4 k4[4], v4[4];
    const int slot = w * 4 + sg;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int key = slot + u * SLOTS;
        const size_t off = (size_t)(key < p.rows_per_seq ? key : 0) * p.ldkv;
        k4[u] = *reinterpret_cast<const float4*>(kb + off);
        v4[u] = *reinterpret_cast<const float4*>(vb + off);
    }
    int n;
    if (p.stop_lens) {                                      // the reference's dec_mask as a valid-prefix length (src/network.py:226-231, 461-465)
        const int64_t a = p.stop_lens[b] + 1, bb = p.pos[0] + 1;
        n = (int)(a < bb ? a : bb);
    } else {
        n = p.lens[b];
    }
    n = n < 0 ? 0 : (n > p.rows_per_seq ? p.rows_per_seq : n);
    float4 q4 = *reinterpret_cast<const float4*>(p.Q + (size_t)b * p.ldq + h * 64 + c * 4);
    q4.x *= p.scale; q4.y *= p.scale; q4.z *= p.scale; q4.w *= p.scale;         // a power of two for head dim 64: exact
    const uint32_t rkey = p.drop_thresh ? rng_row_key(p.seed, p.stream, (uint32_t)(b * p.H + h)) : 0u;

    float m = -INFINITY, l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = slot; k0 < n; k0 += 4 * SLOTS) {
        float4 kn[4], vn[4];
        const int k1 = k0 + 4 * SLOTS;
        if (k1 < n) {                                       // next round in flight while this one is reduced
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = k1 + u * SLOTS;
                const size_t off = (size_t)(key < n ? key : k1) * p.ldkv;
                kn[u] = *reinterpret_cast<const float4*>(kb + off);
                vn[u] = *reinterpret_cast<const float4*>(vb + off);
            }
        }
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s = (q4.x * k4[u].x + q4.y * k4[u].y) + (q4.z * k4[u].z + q4.w * k4[u].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            sc[u] = (k0 + u * SLOTS < n) ? s : -INFINITY;
        }
        const float mn = fmaxf(fmaxf(m, sc[0]), fmaxf(fmaxf(sc[1], sc[2]), sc[3]));      // finite: key k0 is valid
        const float corr = __expf(m - mn);
        l *= corr; acc.x *= corr; acc.y *= corr; acc.z *= corr; acc.w *= corr;
        m = mn;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float e = __expf(sc[u] - mn);
            l += e;                                         // dropout applies to the normalised probabilities: the sum is taken before it
            const float pk = (!p.drop_thresh || rng_keep(rkey, (uint32_t)(k0 + u * SLOTS), p.drop_thresh)) ? e * p.drop_scale : 0.f;
            acc.x += pk * v4[u].x; acc.y += pk * v4[u].y; acc.z += pk * v4[u].z; acc.w += pk * v4[u].w;
        }
        if (k1 < n) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { k4[u] = kn[u]; v4[u] = vn[u]; }
        }
    }
    // ---- merge the 4 groups of the wave
    float mw = fmaxf(m, __shfl_xor(m, 16, 64));
    mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
    const float f = (m == -INFINITY) ? 0.f : __expf(m - mw);                      // a group without keys contributes nothing
    l *= f; acc.x *= f; acc.y *= f; acc.z *= f; acc.w *= f;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        l += __shfl_xor(l, o, 64);
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (sg == 0) *reinterpret_cast<float4*>(&sh.wacc[w][c * 4]) = acc;
    if (lane == 0) { sh.wm[w] = mw; sh.wl[w] = l; }
    __syncthreads();
    // ---- ... and the waves
    if (t < 64) {
        float M = sh.wm[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) M = fmaxf(M, sh.wm[i]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float fi = (sh.wm[i] == -INFINITY) ? 0.f : __expf(sh.wm[i] - M);
            L += sh.wl[i] * fi;
            o += sh.wacc[i][t] * fi;
        }
        p.O[(size_t)b * p.ldo + h * 64 + t] = n > 0 ? o / L : 0.f;
    }
    __syncthreads();
}

// End of a position, one workgroup: the prediction goes to position pos+1, the stop rule is applied (argmax == EOS,
// src/network.py:470-472; sigmoid(stop) >= .5, src/network.py:240-243), then pos and the RNG epoch advance (after a barrier
// behind every read of pos).
__device__ __forceinline__ void end_text(const DecEndParams& p) {
    const int64_t pv = p.pos[0];
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
        const float* xr = p.head + (size_t)b * p.ld;
        float best = xr[0];
        int bi = 0;
        for (int c = 1; c < p.width; ++c) {               // first maximum, as torch.argmax
            const float v = xr[c];
            if (v > best) { best = v; bi = c; }
        }
        p.tokens[(size_t)b * p.ld_tok + pv + 1] = bi;
        if (bi == p.eos && p.stop_lens[b] == p.max_len) p.stop_lens[b] = pv + 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) { p.pos[0] = pv + 1; if (p.epoch) p.epoch[0] += 1; }
}

__device__ __forceinline__ void end_speech(const DecEndParams& p) {
    const int64_t pv = p.pos[0];
    const int M = p.width;
    for (int i = threadIdx.x; i < p.B * M; i += blockDim.x) {
        const int b = i / M, c = i - b * M;
        p.outputs[(size_t)b * p.ld_out + (size_t)(pv + 1) * M + c] = p.head[(size_t)b * p.ld + c];
    }
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
        const float x = p.head[(size_t)b * p.ld + M];
        p.stops[(size_t)b * p.ld_stop + pv + 1] = x;
        if (1.f / (1.f + expf(-x)) >= .5f && p.stop_lens[b] == p.max_len) p.stop_lens[b] = pv + 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) { p.pos[0] = pv + 1; if (p.epoch) p.epoch[0] += 1; }
}

// ------------------------------------------------------------------------------------------------------------
// Separate launches
// ------------------------------------------------------------------------------------------------------------
template <bool ROWS>
__global__ __launch_bounds__(DL_THREADS) void decode_linear_kernel(const DecLinParams p) {
    __shared__ DecShared sh;
    dl_tile<ROWS>(p, blockIdx.x, blockIdx.y, sh);
}
__global__ __launch_bounds__(1024) void decode_attn_kernel(const DecAttnParams p) {
    __shared__ DecShared sh;
    da_tile<1024>(p, blockIdx.x, sh);
}
__global__ __launch_bounds__(256) void decode_end_text_kernel(const DecEndParams p) { end_text(p); }
__global__ __launch_bounds__(256) void decode_end_speech_kernel(const DecEndParams p) { end_speech(p); }

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" int unast_decode_linear(const float* X, int ldx, int64_t x_pos_stride, const float* W, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K,
                                   int act, float drop_p, unsigned int seed, unsigned int stream_id, const float* R, int ldr,
                                   int prologue, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                   const int64_t* tokens, int ld_tok, const float* emb, const float* pe, float pro_scale,
                                   float pro_drop1, unsigned int pro_stream1, float pro_drop2, unsigned int pro_stream2,
                                   float* xn_out, int ld_xn,
                                   float* cache, int ld_cache, int cache_rows, int split_col, const int64_t* pos, hipStream_t stream) {
    UNAST_REQUIRE(W && M > 0 && N > 0 && K >= 4 && (K & 3) == 0, "unast_decode_linear: bad arguments (M=%d N=%d K=%d; K %% 4 == 0)", M, N, K);
    UNAST_REQUIRE(prologue >= PRO_NONE && prologue <= PRO_POSENC, "unast_decode_linear: unknown prologue %d", prologue);
    UNAST_REQUIRE(prologue == PRO_EMBED || (X && (ldx & 3) == 0 && (((uintptr_t)X) & 15) == 0 && (x_pos_stride & 3) == 0), "unast_decode_linear: X rows must be 16-byte aligned");
    UNAST_REQUIRE((ldw & 3) == 0 && (((uintptr_t)W) & 15) == 0, "unast_decode_linear: W rows must be 16-byte aligned");
    UNAST_REQUIRE(Y || (cache && split_col == 0), "unast_decode_linear: no destination");
    UNAST_REQUIRE(!cache || (pos && cache_rows > 0 && ld_cache >= N - split_col && split_col >= 0), "unast_decode_linear: bad cache arguments");
    UNAST_REQUIRE(x_pos_stride == 0 || pos, "unast_decode_linear: a position-indexed input needs pos");
    if (prologue != PRO_NONE) {
        UNAST_REQUIRE(K <= DL_LNK, "unast_decode_linear: rows produced on the fly have at most %d features (K=%d)", DL_LNK, K);
        UNAST_REQUIRE(!xn_out || ((ld_xn & 3) == 0 && (((uintptr_t)xn_out) & 15) == 0), "unast_decode_linear: xn_out rows must be 16-byte aligned");
    }
    if (prologue == PRO_LN || prologue == PRO_LN_DROP)
        UNAST_REQUIRE(ln_gamma && ln_beta && (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0, "unast_decode_linear: LayerNorm needs 16-byte aligned gamma and beta");
    if (prologue == PRO_EMBED) UNAST_REQUIRE(tokens && emb && pos && (((uintptr_t)emb) & 15) == 0, "unast_decode_linear: the embedding prologue needs tokens, emb and pos");
    if (prologue == PRO_EMBED || prologue == PRO_POSENC) UNAST_REQUIRE(pe && pos && (((uintptr_t)pe) & 15) == 0, "unast_decode_linear: the positional-encoding prologue needs pe and pos");
    DecLinParams p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.ldx = ldx; p.x_pos_stride = x_pos_stride; p.W = W; p.ldw = ldw; p.bias = bias; p.Y = Y; p.ldy = ldy; p.M = M; p.N = N; p.K = K; p.act = act;
    p.drop_thresh = drop_threshold(drop_p); p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.seed = seed; p.stream = stream_id;
    p.R = R; p.ldr = ldr; p.pro = prologue; p.ln_g = ln_gamma; p.ln_b = ln_beta; p.ln_eps = ln_eps;
    p.tokens = tokens; p.ld_tok = ld_tok; p.emb = emb; p.pe = pe; p.pro_scale = pro_scale;
    p.pro_thresh1 = drop_threshold(pro_drop1); p.pro_dscale1 = pro_drop1 > 0.f ? 1.f / (1.f - pro_drop1) : 1.f; p.pro_stream1 = pro_stream1;
    p.pro_thresh2 = drop_threshold(pro_drop2); p.pro_dscale2 = pro_drop2 > 0.f ? 1.f / (1.f - pro_drop2) : 1.f; p.pro_stream2 = pro_stream2;
    p.xn_out = xn_out; p.ld_xn = ld_xn;
    p.cache = cache; p.ld_cache = ld_cache; p.cache_rows = cache_rows; p.split_col = split_col; p.pos = pos;
    const dim3 grid((N + DL_COLS - 1) / DL_COLS, (M + DL_ROWS - 1) / DL_ROWS);
    if (prologue != PRO_NONE) hipLaunchKernelGGL((decode_linear_kernel<true>), grid, dim3(DL_THREADS), 0, stream, p);
    else                      hipLaunchKernelGGL((decode_linear_kernel<false>), grid, dim3(DL_THREADS), 0, stream, p);
    return unast_check_launch("unast_decode_linear");
}

extern "C" int unast_decode_attn(const float* Q, int ldq, const float* K, const float* V, int ldkv, int rows_per_seq, const int* lens,
                                 const int64_t* stop_lens, const int64_t* pos, float* O, int ldo,
                                 int B, int H, float scale, float drop_p, unsigned int seed, unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(Q && K && V && O && B > 0 && H > 0 && rows_per_seq > 0, "unast_decode_attn: bad arguments");
    UNAST_REQUIRE(lens || (stop_lens && pos), "unast_decode_attn: give lens, or stop_lens and pos");
    UNAST_REQUIRE((ldq & 3) == 0 && (ldkv & 3) == 0 && ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V)) & 15) == 0, "unast_decode_attn: rows must be 16-byte aligned");
    DecAttnParams p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.ldq = ldq; p.K = K; p.V = V; p.ldkv = ldkv; p.rows_per_seq = rows_per_seq; p.lens = lens; p.stop_lens = stop_lens; p.pos = pos;
    p.O = O; p.ldo = ldo; p.B = B; p.H = H; p.scale = scale;
    p.drop_thresh = drop_threshold(drop_p); p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.seed = seed; p.stream = stream_id;
    hipLaunchKernelGGL(decode_attn_kernel, dim3(B * H), dim3(1024), 0, stream, p);
    return unast_check_launch("unast_decode_attn");
}

extern "C" int unast_decode_end_text(const float* logits, int ld, int V, int B, int64_t* tokens, int ld_tok, int64_t* stop_lens, int64_t max_len, int eos,
                                     int64_t* pos, int* epoch, hipStream_t stream) {
    UNAST_REQUIRE(logits && tokens && stop_lens && pos && B > 0 && V > 0 && ld >= V && ld_tok > max_len, "unast_decode_end_text: bad arguments");
    DecEndParams p;
    memset(&p, 0, sizeof(p));
    p.head = logits; p.ld = ld; p.width = V; p.B = B; p.tokens = tokens; p.ld_tok = ld_tok; p.eos = eos;
    p.stop_lens = stop_lens; p.max_len = max_len; p.pos = pos; p.epoch = epoch;
    hipLaunchKernelGGL(decode_end_text_kernel, dim3(1), dim3(256), 0, stream, p);
    return unast_check_launch("unast_decode_end_text");
}

extern "C" int unast_decode_end_speech(const float* head, int ld, int M, int B, float* outputs, int ld_out, float* stops, int ld_stop, int64_t* stop_lens,
                                       int64_t max_len, int64_t* pos, int* epoch, hipStream_t stream) {
    UNAST_REQUIRE(head && outputs && stops && stop_lens && pos && B > 0 && M > 0 && ld > M && ld_out >= (max_len + 1) * M && ld_stop > max_len,
                  "unast_decode_end_speech: bad arguments");
    DecEndParams p;
    memset(&p, 0, sizeof(p));
    p.head = head; p.ld = ld; p.width = M; p.B = B; p.outputs = outputs; p.ld_out = ld_out; p.stops = stops; p.ld_stop = ld_stop;
    p.stop_lens = stop_lens; p.max_len = max_len; p.pos = pos; p.epoch = epoch;
    hipLaunchKernelGGL(decode_end_speech_kernel, dim3(1), dim3(256), 0, stream, p);
    return unast_check_launch("unast_decode_end_speech");
}

UNAST_DEFINE_RNG_EPOCH_SETTER(decode)

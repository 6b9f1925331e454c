// Kernels of ONE autoregressive decoding position (unast_amd/inference.py; reference: infer_sequence, src/network.py:219-252 and
// 455-481, which re-runs the whole decoder over the prefix at every position).  A position pushes B rows (one per sequence)
// through the decoder: every contraction has M = B <= a few dozen rows, and attention has one query per (sequence, head).  The
// general kernels (gemm.hip: 128x128 tiles staged through LDS; attention.hip: 64-query tiles) spend 12-25 us per launch on
// such shapes -- their tiles are >75 % padding and their k-loops are a chain of dependent load -> LDS -> barrier rounds -- and a
// position is ~35 such launches in sequence.  The kernels here are built for latency instead:
//   * decode_linear: Y[M,N] = epilogue(X[M,K] . W[N,K]^T).  A workgroup owns 32 rows x 16 output columns (N/16 workgroups:
//     16-64 CUs pull the weight rows in parallel); its 8 waves split K, load their MFMA fragments straight from global memory
//     (no LDS staging, one or two rounds of loads in flight), and meet once in LDS to add their partial sums.  Same split-bf16
//     arithmetic (3 MFMAs per product, fp32 accumulate) and the same epilogue order / dropout streams as gemm.hip.  Optional
//     fusions that remove launches from the chain: LayerNorm of the INPUT rows (every workgroup recomputes the 32 row statistics
//     from L2 -- 32 KB -- and workgroup 0 writes the normalised rows out for later residual use), and appending the output
//     columns >= split_col to a K/V cache row selected by a position kept in device memory.
//   * decode_attn: one 16-wave workgroup per (sequence, head); scores, softmax and P.V in fp32 on the vector ALUs (2 x 64 x Tk
//     FMAs: nothing), K and V each read once with 16 lanes per 256-byte row and no synchronisation inside the key loop; bound
//     by how fast one CU streams its 2 x Tk x 256 B.
#include "common.h"

struct DecLinParams {
    const float* X; int ldx;
    const float* W; int ldw;
    const float* bias;
    float* Y; int ldy;
    int M, N, K, act;
    uint32_t drop_thresh; float drop_scale; uint32_t seed, stream;
    const float* R; int ldr;
    const float* ln_g; const float* ln_b; float ln_eps; float* xn_out; int ld_xn;
    float* cache; int ld_cache; int cache_rows; int split_col; const int64_t* pos;
};

#define DL_ROWS 32
#define DL_COLS 16
#define DL_WAVES 8
#define DL_LNK 256                  // the fused LayerNorm handles rows of up to this many features (d_model)
#define DL_XLD (DL_LNK + 4)

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

template <bool LN>
__global__ __launch_bounds__(512) void decode_linear_kernel(const DecLinParams p) {
    __shared__ float red[DL_WAVES][DL_ROWS][DL_COLS];
    __shared__ float xs[LN ? DL_ROWS * DL_XLD : 4];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * DL_COLS, m0 = blockIdx.y * DL_ROWS;
    const int K = p.K;

    if constexpr (LN) {
        // ---- LayerNorm of the 32 input rows (biased variance, two passes over registers): 16 lanes per row
        const int r = t >> 4, c = t & 15, m = m0 + r;
        const bool ok = m < p.M;
        const float* zr = p.X + (size_t)m * p.ldx;
        float4 v[DL_LNK / 64];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < DL_LNK / 64; ++i) {
            const int k = (c + 16 * i) * 4;
            v[i] = (ok && k < K) ? *reinterpret_cast<const float4*>(zr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
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
                float4 o4;
                o4.x = ok ? (v[i].x - mu) * rs * gm.x + bt.x : 0.f;
                o4.y = ok ? (v[i].y - mu) * rs * gm.y + bt.y : 0.f;
                o4.z = ok ? (v[i].z - mu) * rs * gm.z + bt.z : 0.f;
                o4.w = ok ? (v[i].w - mu) * rs * gm.w + bt.w : 0.f;
                *reinterpret_cast<float4*>(&xs[r * DL_XLD + k]) = o4;
                if (ok && p.xn_out && blockIdx.x == 0) *reinterpret_cast<float4*>(p.xn_out + (size_t)m * p.ld_xn + k) = o4;
            }
        }
        __syncthreads();
    }

    // ---- partial products of this wave's k-steps: lane holds C[m = 16*mb + l15][n = 4g .. 4g+3]
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ksteps = (K + 31) >> 5;
    const int n = n0 + l15;
    const bool n_ok = n < p.N;
    const float* wrow = p.W + (size_t)(n_ok ? n : 0) * p.ldw;
    const int ma = m0 + l15, mb_ = m0 + 16 + l15;
    const bool a_ok = ma < p.M, b_ok = mb_ < p.M;
    const float* xa = p.X + (size_t)(a_ok ? ma : 0) * p.ldx;
    const float* xb = p.X + (size_t)(b_ok ? mb_ : 0) * p.ldx;
    for (int s0 = w; s0 < ksteps; s0 += 2 * DL_WAVES) {
        float4 wv[2][2], av[2][2], bv[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {                      // both steps' loads are issued before either is consumed
            const int k = (s0 + u * DL_WAVES) * 32 + 8 * g;      // k >= K for a step past the end: all zeros
            dl_load8(wrow, n_ok, k, K, wv[u][0], wv[u][1]);
            if constexpr (LN) {
                const bool kin = k < K, kin4 = k + 4 < K;
                av[u][0] = kin ? *reinterpret_cast<const float4*>(&xs[l15 * DL_XLD + k]) : make_float4(0.f, 0.f, 0.f, 0.f);
                av[u][1] = kin4 ? *reinterpret_cast<const float4*>(&xs[l15 * DL_XLD + k + 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[u][0] = kin ? *reinterpret_cast<const float4*>(&xs[(16 + l15) * DL_XLD + k]) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[u][1] = kin4 ? *reinterpret_cast<const float4*>(&xs[(16 + l15) * DL_XLD + k + 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                dl_load8(xa, a_ok, k, K, av[u][0], av[u][1]);
                dl_load8(xb, b_ok, k, K, bv[u][0], bv[u][1]);
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
    }
    *reinterpret_cast<float4*>(&red[w][l15][4 * g]) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    *reinterpret_cast<float4*>(&red[w][16 + l15][4 * g]) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
    __syncthreads();

    // ---- sum of the 8 partials + epilogue (order as gemm.hip: bias, activation, dropout, residual): one output per thread
    const int r = t >> 4, c = t & 15, m = m0 + r, nn = n0 + c;
    if (m >= p.M || nn >= p.N) return;
    float x = 0.f;
#pragma unroll
    for (int ww = 0; ww < DL_WAVES; ++ww) x += red[ww][r][c];
    if (p.bias) x += p.bias[nn];
    if (p.act == 1) x = fmaxf(x, 0.f);
    if (p.drop_thresh) x = rng_keep(rng_row_key(p.seed, p.stream, (uint32_t)m), (uint32_t)nn, p.drop_thresh) ? x * p.drop_scale : 0.f;
    if (p.R) x += p.R[(size_t)m * p.ldr + nn];
    if (p.cache && nn >= p.split_col)
        p.cache[((size_t)m * p.cache_rows + (size_t)p.pos[0]) * p.ld_cache + (nn - p.split_col)] = x;
    else
        p.Y[(size_t)m * p.ldy + nn] = x;
}

extern "C" int unast_decode_linear(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K, int act,
                                   float drop_p, unsigned int seed, unsigned int stream_id, const float* R, int ldr,
                                   const float* ln_gamma, const float* ln_beta, float ln_eps, float* xn_out, int ld_xn,
                                   float* cache, int ld_cache, int cache_rows, int split_col, const int64_t* pos, hipStream_t stream) {
    UNAST_REQUIRE(X && W && M > 0 && N > 0 && K >= 4 && (K & 3) == 0, "unast_decode_linear: bad arguments (M=%d N=%d K=%d; K %% 4 == 0)", M, N, K);
    UNAST_REQUIRE((ldx & 3) == 0 && (ldw & 3) == 0 && ((((uintptr_t)X) | ((uintptr_t)W)) & 15) == 0, "unast_decode_linear: X, W rows must be 16-byte aligned");
    UNAST_REQUIRE(Y || (cache && split_col == 0), "unast_decode_linear: no destination");
    UNAST_REQUIRE(!cache || (pos && cache_rows > 0 && ld_cache >= N - split_col && split_col >= 0), "unast_decode_linear: bad cache arguments");
    UNAST_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "unast_decode_linear: LayerNorm needs gamma and beta");
    UNAST_REQUIRE(!ln_gamma || (K <= DL_LNK && (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0 && (!xn_out || ((ld_xn & 3) == 0 && (((uintptr_t)xn_out) & 15) == 0))),
                  "unast_decode_linear: fused LayerNorm handles K <= %d, 16-byte aligned operands", DL_LNK);
    DecLinParams p;
    p.X = X; p.ldx = ldx; p.W = W; p.ldw = ldw; p.bias = bias; p.Y = Y; p.ldy = ldy; p.M = M; p.N = N; p.K = K; p.act = act;
    p.drop_thresh = drop_threshold(drop_p); p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.seed = seed; p.stream = stream_id;
    p.R = R; p.ldr = ldr; p.ln_g = ln_gamma; p.ln_b = ln_beta; p.ln_eps = ln_eps; p.xn_out = xn_out; p.ld_xn = ld_xn;
    p.cache = cache; p.ld_cache = ld_cache; p.cache_rows = cache_rows; p.split_col = split_col; p.pos = pos;
    const dim3 grid((N + DL_COLS - 1) / DL_COLS, (M + DL_ROWS - 1) / DL_ROWS);
    if (ln_gamma) hipLaunchKernelGGL((decode_linear_kernel<true>), grid, dim3(512), 0, stream, p);
    else          hipLaunchKernelGGL((decode_linear_kernel<false>), grid, dim3(512), 0, stream, p);
    return unast_check_launch("unast_decode_linear");
}

// ------------------------------------------------------------------------------------------------------------
// Single-query attention over a K/V cache: O[b, h*64..] = dropout(softmax(scale * q . K^T over keys < lens[b])) . V
// (head dim 64; src/module.py decoder layers through torch's multi_head_attention_forward, SURVEY.md Appendix A: MHA).
// ------------------------------------------------------------------------------------------------------------
struct DecAttnParams {
    const float* Q; int ldq;
    const float* K; const float* V; int ldkv; int rows_per_seq;
    const int* lens;
    float* O; int ldo;
    int H; float scale;
    uint32_t drop_thresh; float drop_scale; uint32_t seed, stream;
};

// 16 waves per (sequence, head).  A group of 16 lanes owns the keys slot, slot+64, ... (slot = 4*wave + lane/16) and keeps a
// private online-softmax state (running max, sum, 4 of the 64 output features per lane), so the key loop has no workgroup-wide
// synchronisation and the K and V rows of 4 keys per group (8 x 16 B per lane, 128 KB per workgroup) are in flight at once;
// the 64 groups' states are merged once at the end (shuffles inside a wave, LDS across waves).
#define DA_THREADS 1024
#define DA_SLOTS (DA_THREADS / 16)
__global__ __launch_bounds__(DA_THREADS) void decode_attn_kernel(const DecAttnParams p) {
    __shared__ float wm[DA_THREADS / 64], wl[DA_THREADS / 64];
    __shared__ float wacc[DA_THREADS / 64][64];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, sg = lane >> 4, c = lane & 15;
    int n = p.lens[b];
    n = n < 0 ? 0 : (n > p.rows_per_seq ? p.rows_per_seq : n);
    float4 q4 = *reinterpret_cast<const float4*>(p.Q + (size_t)b * p.ldq + h * 64 + c * 4);
    q4.x *= p.scale; q4.y *= p.scale; q4.z *= p.scale; q4.w *= p.scale;         // a power of two for head dim 64: exact
    const float* kb = p.K + (size_t)b * p.rows_per_seq * p.ldkv + h * 64 + c * 4;
    const float* vb = p.V + (size_t)b * p.rows_per_seq * p.ldkv + h * 64 + c * 4;
    const uint32_t rkey = p.drop_thresh ? rng_row_key(p.seed, p.stream, (uint32_t)(b * p.H + h)) : 0u;

    float m = -INFINITY, l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = w * 4 + sg; k0 < n; k0 += 4 * DA_SLOTS) {
        float4 k4[4], v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + u * DA_SLOTS;
            const size_t off = (size_t)(key < n ? key : k0) * p.ldkv;
            k4[u] = *reinterpret_cast<const float4*>(kb + off);
            v4[u] = *reinterpret_cast<const float4*>(vb + off);
        }
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s = (q4.x * k4[u].x + q4.y * k4[u].y) + (q4.z * k4[u].z + q4.w * k4[u].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            sc[u] = (k0 + u * DA_SLOTS < n) ? s : -INFINITY;
        }
        const float mn = fmaxf(fmaxf(m, sc[0]), fmaxf(fmaxf(sc[1], sc[2]), sc[3]));      // finite: key k0 is valid
        const float corr = __expf(m - mn);
        l *= corr; acc.x *= corr; acc.y *= corr; acc.z *= corr; acc.w *= corr;
        m = mn;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float e = __expf(sc[u] - mn);
            l += e;                                         // dropout applies to the normalised probabilities: the sum is taken before it
            const float pk = (!p.drop_thresh || rng_keep(rkey, (uint32_t)(k0 + u * DA_SLOTS), p.drop_thresh)) ? e * p.drop_scale : 0.f;
            acc.x += pk * v4[u].x; acc.y += pk * v4[u].y; acc.z += pk * v4[u].z; acc.w += pk * v4[u].w;
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
    if (sg == 0) *reinterpret_cast<float4*>(&wacc[w][c * 4]) = acc;
    if (lane == 0) { wm[w] = mw; wl[w] = l; }
    __syncthreads();
    // ---- ... and the 16 waves
    if (t < 64) {
        float M = wm[0];
#pragma unroll
        for (int i = 1; i < DA_THREADS / 64; ++i) M = fmaxf(M, wm[i]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int i = 0; i < DA_THREADS / 64; ++i) {
            const float fi = (wm[i] == -INFINITY) ? 0.f : __expf(wm[i] - M);
            L += wl[i] * fi;
            o += wacc[i][t] * fi;
        }
        p.O[(size_t)b * p.ldo + h * 64 + t] = n > 0 ? o / L : 0.f;
    }
}

extern "C" int unast_decode_attn(const float* Q, int ldq, const float* K, const float* V, int ldkv, int rows_per_seq, const int* lens, float* O, int ldo,
                                 int B, int H, float scale, float drop_p, unsigned int seed, unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(Q && K && V && lens && O && B > 0 && H > 0 && rows_per_seq > 0, "unast_decode_attn: bad arguments");
    UNAST_REQUIRE((ldq & 3) == 0 && (ldkv & 3) == 0 && ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V)) & 15) == 0, "unast_decode_attn: rows must be 16-byte aligned");
    DecAttnParams p;
    p.Q = Q; p.ldq = ldq; p.K = K; p.V = V; p.ldkv = ldkv; p.rows_per_seq = rows_per_seq; p.lens = lens; p.O = O; p.ldo = ldo; p.H = H; p.scale = scale;
    p.drop_thresh = drop_threshold(drop_p); p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.seed = seed; p.stream = stream_id;
    hipLaunchKernelGGL(decode_attn_kernel, dim3(B * H), dim3(DA_THREADS), 0, stream, p);
    return unast_check_launch("unast_decode_attn");
}

UNAST_DEFINE_RNG_EPOCH_SETTER(decode)

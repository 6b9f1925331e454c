// Split-bf16 MFMA GEMM for gfx950: C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T).
//
// One kernel template serves every dense contraction of the UNAST train step:
//   linear forward   Y = X W^T        A K-contiguous,  B K-contiguous
//   linear dgrad     dX = dY W        A K-contiguous,  B row-contiguous ([K][N])
//   linear wgrad     dW = dY^T X      A row-contiguous ([K][M]), B row-contiguous, split-K + fp32 atomics
//   conv1d k5 forward / dgrad / wgrad as implicit GEMM over (tap, channel) with time-shifted row gathers
//   (TextPrenet src/module.py:199-230 'same' padding, SpeechPostnet src/module.py:155-168 causal padding).
//
// Operands live in HBM as fp32; each 128x32 tile is converted on the way into LDS to bf16 (NSPLIT=1) or to a
// hi/lo bf16 pair (NSPLIT=3: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate) so that the contraction keeps
// ~16 mantissa bits per operand while still running on the bf16 MFMA pipe (v_mfma_f32_16x16x32_bf16).
// K-contiguous operands are staged as [row][32 k] images (XOR-swizzled 16-B chunks, read with ds_read_b128);
// row-contiguous operands are staged untransposed as [k][row] images and read with ds_read_b64_tr_b16.
// 256 threads = 4 waves (2x2), each wave owns a 64x64 sub-tile = 4x4 MFMA tiles, accumulators stay in VGPRs.
#include "common.h"
#include "../../include/unast_hip.h"

#define GBM 128
#define GBK 32
// Block tile = 128 x (64*WN): WN = 2 (256 threads) or 4 (512 threads, 128x256 tile).  The wide tile halves the A-operand
// bytes a CU has to pull through its vector-memory path per FLOP, which is what bounds these fp32-operand GEMMs.
__host__ __device__ constexpr int rc_stride(int rows) { return rows + 16; }            // bf16 elements per k-row of a [k][row] image
__host__ __device__ constexpr int kc_bytes(int rows) { return rows * GBK * 2; }
__host__ __device__ constexpr int rc_bytes(int rows) { return GBK * rc_stride(rows) * 2; }

enum { OP_KC = 0, OP_KC_CONV = 1, OP_RC = 2, OP_RC_CONV_DGRAD = 3, OP_RC_CONV_WGRAD = 4 };

struct GemmParams {
    const float* A; const float* B; float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int T, ca, cb, shift, KS;                // implicit-conv geometry
    const float* bias; const float* R; int ldr; const float* G; int ldg; float gate_scale;
    float alpha; int beta; int act;
    uint32_t drop_thresh; float drop_scale; uint32_t seed, stream;
    int kchunk; int atomic; int tiles_m, tiles_n, nsplitk;
    float* slab; int ld_slab; size_t slab_stride;     // split-K partial slabs [z][M][ld_slab]
    int kb_valid;                                     // rows of a row-contiguous B that exist (K may be zero-padded above it)
    float* rowsum_a;                                  // optional: rowsum_a[m] += sum_k A[m][k] (bias gradient fused into wgrad)
    int group;                                        // launched as one problem of a grouped weight-gradient launch (split-K tile map, slabs)
    int out_split;                                    // store C in the pre-split operand format (16-B chunks [hi x4 | lo x4]) for the attention kernels
    double* colstats;                                 // conv forward only: colstats[n] += sum_m C[m][n], colstats[N + n] += sum_m C[m][n]^2 (BatchNorm batch statistics)
};

__device__ __forceinline__ int swz_h(int row) { return (0x1320 >> (((row >> 2) & 3) << 2)) & 3; }
__device__ __forceinline__ int kc_off(int row, int k) {      // byte offset of element (row,k) in a [128][32] image
    return row * 64 + ((((k >> 3) ^ swz_h(row))) << 4) + ((k & 7) << 1);
}

// Tile loads are UNCONDITIONAL (clamped address, then select-to-zero): branches around loads make hipcc wait for each
// load separately and serialise the staging.  Contract checked on the host: K % 4 == 0 for K-contiguous operands,
// ld >= ceil4(rows) for row-contiguous operands (junk in the pad only reaches rows/columns that are never stored).
template <int MODE>
__device__ __forceinline__ float4 load_kc(const GemmParams& p, const float* __restrict__ src, int ld, int nrows,
                                          int row, int k, int kend, int conv_rb, int conv_rt) {
    bool ok = row < nrows && k < kend;
    const int rr = min(row, nrows - 1);
    const int kk = min(k, p.K - 4);
    const float* ptr;
    if (MODE == OP_KC) {
        ptr = src + (size_t)rr * ld + kk;
    } else {       // OP_KC_CONV: A[(b,t)][(j,c)] = X[b, t + j - shift, c]
        const int j = kk / p.ca;
        const int c = kk - j * p.ca;
        const int ts = conv_rt + j - p.shift;
        ok = ok && ts >= 0 && ts < p.T;
        const int tc = min(max(ts, 0), p.T - 1);
        ptr = src + (size_t)(conv_rb + tc) * ld + c;
    }
    float4 v = *reinterpret_cast<const float4*>(ptr);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

template <int MODE>
__device__ __forceinline__ float4 load_rc(const GemmParams& p, const float* __restrict__ src, int ld, int nrows,
                                          int row, int k, int kend, int kvalid, int conv_j, int conv_c) {
    bool ok = row < nrows && k < kend && k < kvalid;
    const int rr = (row < nrows) ? row : 0;
    const int kk = min(k, kvalid - 1);
    const float* ptr;
    if (MODE == OP_RC_CONV_WGRAD) {     // B[(b,t)][(j,c)] = X[b, t + j - shift, c]
        const int b = kk / p.T;
        const int tt = kk - b * p.T;
        const int ts = tt + conv_j - p.shift;
        ok = ok && ts >= 0 && ts < p.T;
        const int tc = min(max(ts, 0), p.T - 1);
        ptr = src + (size_t)(b * p.T + tc) * ld + ((row < nrows) ? conv_c : 0);
    } else if (MODE == OP_RC) {
        ptr = src + (size_t)kk * ld + rr;
    } else {                            // OP_RC_CONV_DGRAD: B[c][(j',o)] = Wp[o][KS-1-j'][c], Wp = [O][KS][C]
        const int jj = kk / p.cb;
        const int o = kk - jj * p.cb;
        ptr = src + ((size_t)o * p.KS + (p.KS - 1 - jj)) * nrows + rr;
    }
    float4 v = *reinterpret_cast<const float4*>(ptr);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

// Epilogue shared by the GEMM kernels: lane holds C[m = ..+l15][n = ..+4g .. 4g+3] of MI x 4 MFMA tiles; the wave's
// sub-tile starts at (m0 + wm*16*MI, n0 + wn*64).
// 16-lane row reduction (lanes that differ in l15 only) with DPP adds: every lane of the row ends up with the row's sum.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// Column sums of the tile (BatchNorm's batch statistics of a convolution's output, taken where the output is produced instead
// of by a second pass over it): in-lane over the wave's MI row sub-tiles, DPP over the 16 rows of a sub-tile, LDS atomics over
// the workgroup's waves, one fp64 atomic per column and moment per workgroup.  `red` = the (drained) operand stages.
template <int MI, int GBN_, int WM_>
__device__ __forceinline__ void gemm_colstats(const GemmParams& p, const f32x4 (&acc)[MI][4], float* red, int m0, int n0, int wm, int wn, int l15, int g, int t, int nthreads) {
    __syncthreads();                                            // every wave is done reading the operand stages
    // red[moment][wm][column]: one slot per wave row (no float atomics: the sums must not depend on which wave arrives first)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + 4 * g;                // column inside the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + nl + r;
            const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (16 * MI) + i * 16 + l15;
                const float x = (m < p.M) ? acc[i][j][r] * p.alpha + bv : 0.f;
                s1 += x; s2 += x * x;
            }
            s1 = row16_sum(s1); s2 = row16_sum(s2);
            if (l15 == 0) { red[wm * GBN_ + nl + r] = s1; red[(WM_ + wm) * GBN_ + nl + r] = s2; }
        }
    }
    __syncthreads();
    for (int i = t; i < 2 * GBN_; i += nthreads) {
        const int mom = i / GBN_, c = i % GBN_, n = n0 + c;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WM_; ++w) v += red[(mom * WM_ + w) * GBN_ + c];
        if (n < p.N) unsafeAtomicAdd(p.colstats + (size_t)mom * p.N + n, (double)v);
    }
}

template <int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MI][4], int m0, int n0, int wm, int wn, int l15, int g, int zsplit) {
    // ---- epilogue: lane holds C[m = ..+l15][n = ..+4g .. 4g+3] --------------------------------------
    const bool first_split = (zsplit == 0);
    if (p.slab) {                                   // split-K: raw partial sums to this split's slab (plain 16-B stores)
        float* slab = p.slab + (size_t)zsplit * p.slab_stride;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * (16 * MI) + i * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + 4 * g;
                if (n >= p.N) continue;
                *reinterpret_cast<float4*>(slab + (size_t)m * p.ld_slab + n) =
                    make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
            }
        }
        return;
    }
    const bool vec_c = ((p.ldc & 3) == 0);
    const bool vec_r = p.R && ((p.ldr & 3) == 0) && ((((uintptr_t)p.R) & 15) == 0);
    const bool vec_g = p.G && ((p.ldg & 3) == 0) && ((((uintptr_t)p.G) & 15) == 0);
    float bias_v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_v[j][r] = (p.bias && first_split && n + r < p.N) ? p.bias[n + r] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (16 * MI) + i * 16 + l15;
        if (m >= p.M) continue;
        uint32_t rkey = 0;
        if (p.drop_thresh) rkey = rng_row_key(p.seed, p.stream, (uint32_t)m);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * g;
            if (n >= p.N) continue;
            const bool full = (n + 3 < p.N);
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            float gv[4] = {1.f, 1.f, 1.f, 1.f}, rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.G) {
                if (vec_g && full) { float4 t4 = *reinterpret_cast<const float4*>(p.G + (size_t)m * p.ldg + n); gv[0] = t4.x; gv[1] = t4.y; gv[2] = t4.z; gv[3] = t4.w; }
                else { for (int r = 0; r < 4; ++r) if (n + r < p.N) gv[r] = p.G[(size_t)m * p.ldg + n + r]; }
            }
            if (p.R && first_split) {
                if (vec_r && full) { float4 t4 = *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n); rv[0] = t4.x; rv[1] = t4.y; rv[2] = t4.z; rv[3] = t4.w; }
                else { for (int r = 0; r < 4; ++r) if (n + r < p.N) rv[r] = p.R[(size_t)m * p.ldr + n + r]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r] * p.alpha + bias_v[j][r];
                if (p.act == 1) x = fmaxf(x, 0.f);
                if (p.drop_thresh) x = rng_keep(rkey, (uint32_t)(n + r), p.drop_thresh) ? x * p.drop_scale : 0.f;
                if (p.G) x = (gv[r] > 0.f) ? x * p.gate_scale : 0.f;
                v[r] = x + rv[r];
            }
            float* cp = p.C + (size_t)m * p.ldc + n;
            if (p.atomic) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) atomicAdd(cp + r, v[r]);
            } else if (vec_c && full) {
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (p.beta) {
                    float4 c = *reinterpret_cast<const float4*>(cp);
                    o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
                }
                if (p.out_split) *reinterpret_cast<uint4*>(cp) = split_chunk(o);
                else *reinterpret_cast<float4*>(cp) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) cp[r] = p.beta ? cp[r] + v[r] : v[r];
            }
        }
    }
}

// FLAGS bit 0 (FULL): every tile is interior (M % 128 == 0, N % tile == 0, K and the K-slices % 32 == 0, plain operands):
//   loads carry no bounds handling and address as uniform base (advanced per k-step on the scalar unit) + a per-thread
//   32-bit byte offset computed once -- the selects and 64-bit address arithmetic were a quarter of the loop's vector
//   instructions, and this loop is bound by vector-instruction issue (MFMA issue + operand conversion), not by MFMA rate.
// FLAGS bit 1 (BSPLIT): B is a weight kept in the pre-split chunk format (common.h split_chunk; written by the AdamW
//   kernel): the loader copies hi/lo straight to LDS instead of re-splitting the same weights in every row panel.
template <int AM, int BMODE, int NSPLIT, int WN, int WM, int FLAGS>
__device__ __forceinline__ void gemm_body(const GemmParams& p, const int pid) {
    constexpr bool FULL = (FLAGS & 1) != 0, BSPLIT = (FLAGS & 2) != 0;
    static_assert(!FULL || ((AM == OP_KC || AM == OP_RC) && (BMODE == OP_KC || BMODE == OP_RC)), "FULL: plain operands only");
    constexpr bool A_KC = (AM == OP_KC || AM == OP_KC_CONV);
    constexpr bool B_KC = (BMODE == OP_KC);
    constexpr int PARTS = (NSPLIT == 3) ? 2 : 1;
    constexpr int NT = 64 * WM * WN;                   // threads
    constexpr int MI = 8 / WM;                         // 16-row sub-tiles per wave along M (wave tile = 16*MI x 64)
    constexpr int GBN = 64 * WN;                       // block tile columns
    constexpr int A_BYTES = A_KC ? kc_bytes(GBM) : rc_bytes(GBM);
    constexpr int B_BYTES = B_KC ? kc_bytes(GBN) : rc_bytes(GBN);
    constexpr int A_RCS = rc_stride(GBM), B_RCS = rc_stride(GBN);
    // loader geometry: K-contiguous: 8 threads per 128-B row piece; row-contiguous: ROWS/4 threads per k-row
    constexpr int A_PASS = (GBM * 8) / NT, B_PASS = (GBN * 8) / NT;          // float4 per thread per tile
    constexpr int KC_RPP = NT / 8;                                           // rows per pass (K-contiguous)
    constexpr int A_TPK = GBM / 4, B_TPK = GBN / 4;                          // threads per k-row (row-contiguous)
    constexpr int A_KPP = NT / A_TPK, B_KPP = NT / B_TPK;                    // k-rows per pass
    constexpr int STAGE = (A_BYTES + B_BYTES) * PARTS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];      // two stages: one barrier per k-step

    // XCD-aware tile mapping: blocks b and b+8 (same XCD under round-robin dispatch) share the A row panel.
    int tile_m, tile_n, zsplit = 0;
    if (p.tiles_m >= 8 && p.nsplitk == 1 && !p.group) {
        const int G = 8 * p.tiles_n;
        const int grp = pid / G, rem = pid - grp * G;
        tile_m = grp * 8 + (rem & 7);
        tile_n = rem >> 3;
    } else {
        // few row panels (weight gradients, split-K): all tiles of one K-slice read the same operand rows, so they get
        // linear ids that are equal mod 8 (same XCD / L2 under round-robin dispatch); grid padded to 8 slices.
        const int ntile = p.tiles_m * p.tiles_n;
        const int g8 = pid & 7, sidx = pid >> 3;
        zsplit = (sidx / ntile) * 8 + g8;
        const int tile = sidx % ntile;
        tile_m = tile % p.tiles_m;
        tile_n = tile / p.tiles_m;
        if (zsplit >= p.nsplitk) return;
    }
    if (tile_m >= p.tiles_m) return;        // whole block exits together (keeps EXEC full for tr reads)
    const int m0 = tile_m * GBM, n0 = tile_n * GBN;
    const int kbeg = zsplit * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg + GBK - 1) / GBK;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, g = lane >> 4;

    // ---- per-thread loader coordinates -------------------------------------------------------
    int a_rb[A_PASS], a_rt[A_PASS];   // conv A: batch row base / time index of the tile rows of this thread
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) { a_rb[i] = 0; a_rt[i] = 0; }
    if (AM == OP_KC_CONV) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            int row = min(m0 + (t >> 3) + KC_RPP * i, p.M - 1);     // clamped: loads are unconditional (rows >= M are zeroed after the load)
            int b = row / p.T;
            a_rb[i] = b * p.T;
            a_rt[i] = row - b * p.T;
        }
    }
    int b_cj = 0, b_cc = 0;                 // conv-wgrad B: tap and channel of this thread's 4 columns
    if (BMODE == OP_RC_CONV_WGRAD) {
        int row = min(n0 + (t % B_TPK) * 4, p.N - 4);       // clamped for the same reason (N = 5*cb is a multiple of 4)
        b_cj = row / p.cb;
        b_cc = row - b_cj * p.cb;
    }

    uint32_t a_off[A_PASS], b_off[B_PASS];          // FULL: per-thread byte offsets inside the current k-slab
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
        a_off[i] = FULL ? 4u * (A_KC ? (uint32_t)(m0 + (t >> 3) + KC_RPP * i) * (uint32_t)p.lda + (t & 7) * 4
                                     : (uint32_t)(t / A_TPK + A_KPP * i) * (uint32_t)p.lda + m0 + (t % A_TPK) * 4) : 0u;
#pragma unroll
    for (int i = 0; i < B_PASS; ++i)
        b_off[i] = FULL ? 4u * (B_KC ? (uint32_t)(n0 + (t >> 3) + KC_RPP * i) * (uint32_t)p.ldb + (t & 7) * 4
                                     : (uint32_t)(t / B_TPK + B_KPP * i) * (uint32_t)p.ldb + n0 + (t % B_TPK) * 4) : 0u;

    auto load_tiles = [&](int kt, float4 (&ra)[A_PASS], float4 (&rb)[B_PASS]) {
        if (FULL) {     // prefetches past the last k-step re-read the last one (never consumed) instead of leaving the operand
            const size_t k0 = (size_t)(kbeg + min(kt, nk - 1) * GBK);
            const char* Ab = reinterpret_cast<const char*>(p.A + (A_KC ? k0 : k0 * (size_t)p.lda));
            const char* Bb = reinterpret_cast<const char*>(p.B + (B_KC ? k0 : k0 * (size_t)p.ldb));
#pragma unroll
            for (int i = 0; i < A_PASS; ++i) ra[i] = *reinterpret_cast<const float4*>(Ab + a_off[i]);
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) rb[i] = *reinterpret_cast<const float4*>(Bb + b_off[i]);
            return;
        }
        const int k0 = kbeg + kt * GBK;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            if (A_KC) ra[i] = load_kc<AM>(p, p.A, p.lda, p.M, m0 + (t >> 3) + KC_RPP * i, k0 + (t & 7) * 4, kend, a_rb[i], a_rt[i]);
            else      ra[i] = load_rc<AM>(p, p.A, p.lda, p.M, m0 + (t % A_TPK) * 4, k0 + t / A_TPK + A_KPP * i, kend, p.K, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            if (B_KC) rb[i] = load_kc<BMODE>(p, p.B, p.ldb, p.N, n0 + (t >> 3) + KC_RPP * i, k0 + (t & 7) * 4, kend, 0, 0);
            else      rb[i] = load_rc<BMODE>(p, p.B, p.ldb, p.N, n0 + (t % B_TPK) * 4, k0 + t / B_TPK + B_KPP * i, kend, p.kb_valid, b_cj, b_cc);
        }
    };
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = (AM == OP_RC) && p.rowsum_a && tile_n == 0;
    // `live`: the tile exists (FULL prefetches past the end re-read the last tile; the general path zero-fills them)
    auto store_tiles = [&](const float4 (&ra)[A_PASS], const float4 (&rb)[B_PASS], unsigned char* sA, unsigned char* sB, bool live) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            u32x2 hi, lo;
            if (AM == OP_RC && do_rowsum && live) { bsum[0] += ra[i].x; bsum[1] += ra[i].y; bsum[2] += ra[i].z; bsum[3] += ra[i].w; }
            split4<NSPLIT>(ra[i], hi, lo);
            const int off = A_KC ? kc_off((t >> 3) + KC_RPP * i, (t & 7) * 4) : ((t / A_TPK + A_KPP * i) * A_RCS + (t % A_TPK) * 4) * 2;
            *reinterpret_cast<u32x2*>(sA + off) = hi;
            if (PARTS == 2) *reinterpret_cast<u32x2*>(sA + A_BYTES + off) = lo;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            u32x2 hi, lo;
            if (BSPLIT) {
                hi[0] = __float_as_uint(rb[i].x); hi[1] = __float_as_uint(rb[i].y);
                lo[0] = __float_as_uint(rb[i].z); lo[1] = __float_as_uint(rb[i].w);
            } else {
                split4<NSPLIT>(rb[i], hi, lo);
            }
            const int off = B_KC ? kc_off((t >> 3) + KC_RPP * i, (t & 7) * 4) : ((t / B_TPK + B_KPP * i) * B_RCS + (t % B_TPK) * 4) * 2;
            *reinterpret_cast<u32x2*>(sB + off) = hi;
            if (PARTS == 2) *reinterpret_cast<u32x2*>(sB + B_BYTES + off) = lo;
        }
    };
    // MFMA operand fragment for the 16-row sub-tile starting at tile row `rbase`: lane holds [row l15][k 8g..8g+7]
    auto frag = [&](const unsigned char* img, bool kc, int rcs, int rbase) -> bf16x8_t {
        if (kc) {
            int row = rbase + l15;
            return *reinterpret_cast<const bf16x8_t*>(img + row * 64 + ((g ^ swz_h(row)) << 4));
        } else {
            const int q = l15 >> 2, pp = l15 & 3;
            s16x4 v0 = lds_read_tr16(img + ((8 * g + q) * rcs + rbase + 4 * pp) * 2);
            s16x4 v1 = lds_read_tr16(img + ((8 * g + 4 + q) * rcs + rbase + 4 * pp) * 2);
            s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            return __builtin_bit_cast(bf16x8_t, v);
        }
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const unsigned char* sA, const unsigned char* sB) {
        bf16x8_t af[MI][PARTS], bfr[4][PARTS];
#pragma unroll
        for (int s = 0; s < PARTS; ++s) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i][s] = frag(sA + s * A_BYTES, A_KC, A_RCS, wm * (16 * MI) + i * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) bfr[i][s] = frag(sB + s * B_BYTES, B_KC, B_RCS, wn * 64 + i * 16);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // swapped operands: the B tile is the MFMA "A" (rows = n), so each lane ends up with 4 consecutive n
                if (NSPLIT == 3) {
                    acc[i][j] = mfma16(bfr[j][PARTS - 1], af[i][0], acc[i][j]);
                    acc[i][j] = mfma16(bfr[j][0], af[i][PARTS - 1], acc[i][j]);
                }
                acc[i][j] = mfma16(bfr[j][0], af[i][0], acc[i][j]);
            }
    };

    // Two-stage software pipeline, one barrier per k-step.  In iteration kt a wave (1) issues the global loads of tile
    // kt+2 into the register set that was drained one iteration ago, (2) multiplies tile kt from LDS stage kt%2 and, in
    // the shadow of those MFMAs, converts tile kt+1 (whose loads were issued an iteration ago) and writes it to the
    // other LDS stage.  Loads are issued unconditionally (clamped addresses) so the compiler's vmcnt counts are static.
    float4 ra0[A_PASS], rb0[B_PASS], ra1[A_PASS], rb1[B_PASS];
    unsigned char* const s0A = smem;
    unsigned char* const s0B = smem + A_BYTES * PARTS;
    unsigned char* const s1A = smem + STAGE;
    unsigned char* const s1B = smem + STAGE + A_BYTES * PARTS;
    load_tiles(0, ra0, rb0);
    load_tiles(1, ra1, rb1);
    store_tiles(ra0, rb0, s0A, s0B, true);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        load_tiles(kt + 2, ra0, rb0);
        compute(s0A, s0B);
        store_tiles(ra1, rb1, s1A, s1B, kt + 1 < nk);
        __syncthreads();
        if (kt + 1 < nk) {
            load_tiles(kt + 3, ra1, rb1);
            compute(s1A, s1B);
            store_tiles(ra0, rb0, s0A, s0B, kt + 2 < nk);
            __syncthreads();
        }
    }

    if (AM == OP_RC && do_rowsum) {          // threads t, t+32, ... hold partial sums of rows m0 + 4*(t&31) .. +3
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (t < 128) red[t] = 0.f;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(&red[(t % A_TPK) * 4 + c], bsum[c]);
        __syncthreads();
        if (t < 128 && m0 + t < p.M) atomicAdd(p.rowsum_a + m0 + t, red[t]);
    }
    if constexpr (AM == OP_KC_CONV && BMODE == OP_KC) {        // (only the conv forward instantiations carry this code)
        if (p.colstats) gemm_colstats<MI, GBN, WM>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, l15, g, t, NT);
    }
    gemm_epilogue<MI>(p, acc, m0, n0, wm, wn, l15, g, zsplit);
}

template <int AM, int BMODE, int NSPLIT, int WN, int WM, int FLAGS>
__global__ __launch_bounds__(64 * WM * WN, (WM == 4 ? 4 : 2)) void gemm_kernel(const GemmParams p) {
    gemm_body<AM, BMODE, NSPLIT, WN, WM, FLAGS>(p, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Grouped weight gradients: up to GROUP_MAX problems dW_i[M_i,N_i] += dY_i^T X_i of one backward closure (the out-proj and
// in-proj gradients of an attention sub-layer, the two FFN gradients, ...) in ONE launch.  A 256x256 gradient over 25 600
// tokens is 4 output tiles: launched alone it needs 64 K-slices of 12 k-steps to fill the chip and three quarters of its time
// is fixed cost (prologue, epilogue, 16.8 MB of partial slabs written and re-read).  Grouped, the problems share the
// chip: each workgroup runs 50-100 k-steps and a layer's slab traffic falls from 118 MB to ~25 MB.  Every workgroup maps its block
// id to (problem, tile, K-slice), builds that problem's GemmParams in scalar registers and runs the same body as gemm_kernel;
// partial sums go to slabs, a grouped reduction adds them into the gradient buffers.
// ---------------------------------------------------------------------------------------------
#define GROUP_MAX 8
struct GroupItem {
    const float* A; const float* B; float* C; float* rowsum_a; float* slab;
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tiles_n, nsplitk, kchunk, ld_slab;
};
struct GroupParams {
    int count;
    int base[GROUP_MAX + 1];            // first block id of each problem (GEMM launch)
    int rbase[GROUP_MAX + 1];           // first block id of each problem (reduction launch)
    GroupItem it[GROUP_MAX];
};

template <int NSPLIT, int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_group_kernel(const GroupParams gp) {
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int j = 1; j < GROUP_MAX; ++j)
        if (j < gp.count && b >= gp.base[j]) i = j;
    const GroupItem& it = gp.it[i];
    GemmParams p;
    p.A = it.A; p.B = it.B; p.C = it.C; p.M = it.M; p.N = it.N; p.K = it.K; p.lda = it.lda; p.ldb = it.ldb; p.ldc = it.ldc;
    p.T = 1; p.ca = 1; p.cb = 1; p.shift = 0; p.KS = 5;
    p.bias = nullptr; p.R = nullptr; p.ldr = 0; p.G = nullptr; p.ldg = 0; p.gate_scale = 1.f;
    p.alpha = 1.f; p.beta = 1; p.act = 0; p.drop_thresh = 0u; p.drop_scale = 1.f; p.seed = 0u; p.stream = 0u;
    p.kchunk = it.kchunk; p.atomic = 0; p.tiles_m = it.tiles_m; p.tiles_n = it.tiles_n; p.nsplitk = it.nsplitk;
    p.slab = it.slab; p.ld_slab = it.ld_slab; p.slab_stride = (size_t)it.M * it.ld_slab;
    p.kb_valid = it.K; p.rowsum_a = it.rowsum_a; p.group = 1; p.out_split = 0; p.colstats = nullptr;
    gemm_body<OP_RC, OP_RC, NSPLIT, 2, 2, FLAGS>(p, b - gp.base[i]);
}

// C_i[m][n] += sum_z slab_i[z][m][n] for every problem of a group: 64 float4 elements x 4 slab groups per workgroup, as
// splitk_reduce_kernel below.
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const GroupParams gp) {
    __shared__ float4 part[4][64];
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int j = 1; j < GROUP_MAX; ++j)
        if (j < gp.count && b >= gp.rbase[j]) i = j;
    const GroupItem& it = gp.it[i];
    const int nq = it.ld_slab >> 2;
    const size_t total = (size_t)it.M * nq, slab_stride = (size_t)it.M * it.ld_slab;
    const int e = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const size_t idx = (size_t)(b - gp.rbase[i]) * 64 + e;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int m = 0, n = 0;
    if (idx < total) {
        m = (int)(idx / nq); n = (int)(idx - (size_t)m * nq) * 4;
        const float* sp = it.slab + (size_t)m * it.ld_slab + n;
        for (int z = zg; z < it.nsplitk; z += 4) {
            const float4 v = *reinterpret_cast<const float4*>(sp + (size_t)z * slab_stride);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    part[zg][e] = a;
    __syncthreads();
    if (zg == 0 && idx < total) {
        const float4 b1 = part[1][e], b2 = part[2][e], b3 = part[3][e];
        const float av[4] = {(a.x + b1.x) + (b2.x + b3.x), (a.y + b1.y) + (b2.y + b3.y), (a.z + b1.z) + (b2.z + b3.z), (a.w + b1.w) + (b2.w + b3.w)};
        float* cp = it.C + (size_t)m * it.ldc + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < it.N) cp[r] += av[r];
    }
}

// ---------------------------------------------------------------------------------------------
// What bounds this kernel (MI355X, round-1 measurements; tools/make_gemm_variants.py, tools/make_gemm_stamps.py,
// tools/mfma_peak.cpp, tools/pmc_gemm.sh; numbers in DESIGN.md section 4):
//   * no compute unit is saturated: MFMA pipe 40 % busy by SQ_VALU_MFMA_BUSY_CYCLES (54 % of the 1.7-1.8 PFLOP/s this
//     chip sustains on random bf16 data -- it holds ~1.7 GHz under MFMA load, not 2.4), LDS 30 %, TCP 30 %, zero LDS
//     bank conflicts, vector issue < 50 %;
//   * a "skeleton" build that only ISSUES the tile loads (6 of 24 MFMAs, no LDS traffic, no barrier, loaded data never
//     waited for) still needs 0.55 us per k-step per workgroup and 1.13 us with two workgroups per CU, and 36 of the full
//     kernel's 53 us at 25600x256x1024: the floor is the rate at which a CU pulls the fp32 operand slabs through its
//     vector-memory path (~58 GB/s per CU, ~15 TB/s chip-wide from L2) on top of the HBM/MALL stream of A and C
//     (131 MB per launch = 2.5 TB/s at 52 us).  With fp32 activations and 3 MFMAs per product these d=256 contractions
//     sit below the ridge point (~300 FLOP/B vs ~450), i.e. they are bandwidth-bound, not MFMA-bound;
//   * three restructurings of the main loop were built, verified bit-identical and measured SLOWER than this kernel at
//     25600x256x1024 (52 us): an explicit ping-pong split of the workgroup (one half multiplies while the other stages;
//     85 us), an "A-direct" kernel whose waves load their A fragments straight from global memory into registers
//     (32x128 wave tiles, B alone in LDS; 67 us) and a fully software-pipelined 3-stage version of this kernel (fragments
//     of tile kt+1 read during the MFMAs of tile kt; 72 us).  Padding the row strides away from powers of two changes
//     nothing (no channel camping).  Issuing the tile loads through inline asm with hand-placed s_waitcnt vmcnt(N) -- so that
//     the compiler's vmcnt(0) at the loop top no longer drains the second prefetch set -- kept two sets in flight as designed
//     (bit-identical results) and changed nothing either (52.5 us): bytes in flight per CU are not the limit.  What helped: the interior fast path and pre-split weights below (-7 % of GEMM time).
// The remaining lever is fewer operand bytes per FLOP (wider tiles for N >= 512, fusing producer epilogues), not more
// MFMA overlap.

// C[m][n] (+)= sum_z slab[z][m][n]: reduction of the split-K partial slabs (deterministic order).  64 float4 elements x 4 slab
// groups per workgroup: a thread sums every 4th slab of its element with 8 loads in flight, the four partial sums meet in
// LDS.  (One thread per element walking all slabs in a loop took 14 us for 64 slabs of 256 KB -- a chain of dependent
// round trips from 64 workgroups; this form takes ~5.)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, size_t slab_stride, int ld_slab, int nsplit,
                                                            float* __restrict__ C, int ldc, int M, int N, int beta) {
    __shared__ float4 part[4][64];
    const int nq = ld_slab >> 2;
    const size_t total = (size_t)M * nq;
    const int e = threadIdx.x & 63, zg = threadIdx.x >> 6;
    for (size_t i0 = (size_t)blockIdx.x * 64; i0 < total; i0 += (size_t)gridDim.x * 64) {
        const size_t i = i0 + e;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int m = 0, n = 0;
        if (i < total) {
            m = (int)(i / nq); n = (int)(i - (size_t)m * nq) * 4;
            const float* sp = slab + (size_t)m * ld_slab + n;
            int z = zg;
            for (; z + 28 < nsplit; z += 32) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(sp + (size_t)(z + 4 * u) * slab_stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
            }
            for (; z < nsplit; z += 4) {
                const float4 v = *reinterpret_cast<const float4*>(sp + (size_t)z * slab_stride);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        part[zg][e] = a;
        __syncthreads();
        if (zg == 0 && i < total) {
            const float4 b1 = part[1][e], b2 = part[2][e], b3 = part[3][e];
            const float av[4] = {(a.x + b1.x) + (b2.x + b3.x), (a.y + b1.y) + (b2.y + b3.y), (a.z + b1.z) + (b2.z + b3.z), (a.w + b1.w) + (b2.w + b3.w)};
            float* cp = C + (size_t)m * ldc + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < N) cp[r] = beta ? cp[r] + av[r] : av[r];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Host launcher / C ABI
// ---------------------------------------------------------------------------------------------
template <int AM, int BMODE, int FLAGS>
static void launch_tile(const GemmParams& p, int nsplit, int wn, dim3 grid, hipStream_t s) {
    if (wn == 4) {
        if (nsplit == 3) hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 3, 4, 2, FLAGS>), grid, dim3(512), 0, s, p);
        else             hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 1, 4, 2, FLAGS>), grid, dim3(512), 0, s, p);
    } else if (wn == 1) {      // 128x64 tile, 4 waves (4 along M): twice the workgroups of the 128x128 tile for outputs that leave the chip underfilled
        if (nsplit == 3) hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 3, 1, 4, FLAGS>), grid, dim3(256), 0, s, p);
        else             hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 1, 1, 4, FLAGS>), grid, dim3(256), 0, s, p);
    } else if (wn == 8) {      // 128x128 tile, 8 waves (4 along M x 2 along N): <=128 VGPRs, 4 waves/SIMD
        if (nsplit == 3) hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 3, 2, 4, FLAGS>), grid, dim3(512), 0, s, p);
        else             hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 1, 2, 4, FLAGS>), grid, dim3(512), 0, s, p);
    } else {
        if (nsplit == 3) hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 3, 2, 2, FLAGS>), grid, dim3(256), 0, s, p);
        else             hipLaunchKernelGGL((gemm_kernel<AM, BMODE, 1, 2, 2, FLAGS>), grid, dim3(256), 0, s, p);
    }
}
// The fast-path variants exist for the tile each operand form runs with by default (8 waves for forward / dgrad, 4 waves
// for weight gradients); any other request falls back to the general kernel (flags & ~1) -- never to a wrong one.
template <int AM, int BMODE, bool WEIGHT_B, bool PLAIN>
static void launch_split(const GemmParams& p, int nsplit, int wn, int flags, dim3 grid, hipStream_t s) {
    const bool fast_tile = (AM == OP_RC) ? (wn == 2) : (wn == 8 || wn == 1);
    if constexpr (PLAIN) {
        if (fast_tile && (flags & 1)) {
            if constexpr (WEIGHT_B) { if (flags & 2) { launch_tile<AM, BMODE, 3>(p, nsplit, wn, grid, s); return; } }
            launch_tile<AM, BMODE, 1>(p, nsplit, wn, grid, s);
            return;
        }
    }
    if constexpr (WEIGHT_B) { if (flags & 2) { launch_tile<AM, BMODE, 2>(p, nsplit, wn, grid, s); return; } }
    launch_tile<AM, BMODE, 0>(p, nsplit, wn, grid, s);
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int unast_gemm(int a_mode, int b_mode, int nsplit,
                          const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          int M, int N, int K, int kb_valid,
                          int conv_T, int conv_ca, int conv_cb, int conv_shift,
                          const float* bias, const float* R, int ldr, const float* G, int ldg, float gate_scale,
                          float alpha, int beta, int act,
                          float drop_p, unsigned int seed, unsigned int stream_id,
                          int splitk, float* splitk_ws, int64_t splitk_ws_floats, float* rowsum_a, int tile_wn, int b_presplit,
                          int out_split, double* colstats, hipStream_t stream) {
    UNAST_REQUIRE(A && B && C, "unast_gemm: null operand");
    UNAST_REQUIRE(!b_presplit || (a_mode == OP_KC || a_mode == OP_KC_CONV), "unast_gemm: a pre-split B is a weight (forward / dgrad forms only)");
    UNAST_REQUIRE(M > 0 && N > 0 && K > 0, "unast_gemm: bad dims M=%d N=%d K=%d", M, N, K);
    UNAST_REQUIRE(nsplit == 1 || nsplit == 3, "unast_gemm: nsplit must be 1 or 3");
    UNAST_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C), "unast_gemm: operands must be 16-byte aligned");
    UNAST_REQUIRE((lda & 3) == 0 && (ldb & 3) == 0, "unast_gemm: lda/ldb must be multiples of 4 (got %d, %d)", lda, ldb);
    UNAST_REQUIRE(splitk >= 1, "unast_gemm: splitk >= 1");
    const bool a_kc = (a_mode == OP_KC || a_mode == OP_KC_CONV), b_kc = (b_mode == OP_KC);
    UNAST_REQUIRE(!(a_kc || b_kc) || ((K & 3) == 0 && K >= 4), "unast_gemm: K-contiguous operands need K %% 4 == 0 (zero-pad the operand; K=%d)", K);
    UNAST_REQUIRE(a_kc || lda >= ((M + 3) & ~3), "unast_gemm: row-contiguous A needs lda >= ceil4(M)");
    UNAST_REQUIRE(b_kc || b_mode != OP_RC || ldb >= ((N + 3) & ~3), "unast_gemm: row-contiguous B needs ldb >= ceil4(N)");
    if (kb_valid <= 0 || kb_valid > K) kb_valid = K;
    UNAST_REQUIRE(!(splitk > 1 && (act || drop_p > 0.f || G || bias || R)),
                  "unast_gemm: split-K supports only the plain alpha/beta epilogue");
    UNAST_REQUIRE(!(splitk > 1 && !splitk_ws && beta == 0), "unast_gemm: atomic split-K (no workspace) accumulates into C: needs beta=1");
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.T = conv_T > 0 ? conv_T : 1; p.ca = conv_ca > 0 ? conv_ca : 1; p.cb = conv_cb > 0 ? conv_cb : 1;
    p.shift = conv_shift; p.KS = 5;
    p.bias = bias; p.R = R; p.ldr = ldr; p.G = G; p.ldg = ldg; p.gate_scale = gate_scale;
    p.alpha = alpha; p.beta = beta; p.act = act;
    p.drop_thresh = drop_threshold(drop_p); p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    p.seed = seed; p.stream = stream_id; p.group = 0; p.out_split = out_split; p.colstats = colstats;
    UNAST_REQUIRE(!colstats || (a_mode == OP_KC_CONV && b_mode == OP_KC && splitk == 1 && !act && drop_p <= 0.f && !R && !G && beta == 0),
                  "unast_gemm: colstats is for the plain conv forward (bias only)");
    UNAST_REQUIRE(!out_split || ((N & 3) == 0 && (ldc & 3) == 0 && beta == 0 && splitk == 1), "unast_gemm: out_split needs N %% 4 == 0, ldc %% 4 == 0, beta = 0, no split-K");
    // tile width: 128x256 when N is wide enough and the grid still fills the chip; `tile_wn` (2/4) overrides, 0 = auto
    int wn = tile_wn;
    if (wn != 1 && wn != 2 && wn != 4 && wn != 8) {
        // measured on MI355X (tools/bench_gemm.py): for forward / dgrad shapes the 128x128 tile shared by 8 waves
        // (<=128 VGPRs, 4 waves/SIMD) is 10-25 % faster than 4 waves (more waves to cover LDS/barrier/global latencies);
        // weight gradients (long K, split-K) are equal, the 128x256 tile only helps isolated large shapes.
        wn = (a_mode == OP_RC) ? 2 : 8;
        // few row panels and a narrow output (the text side: 5 760 rows into 256 columns = 90 tiles of 128x128 on 256 CUs): 128x64 tiles
        if (a_mode == OP_KC && splitk == 1 && (M + GBM - 1) / GBM * ((N + 127) / 128) <= 128 && N >= 128) wn = 1;
    }
    const int gbn = (wn == 4) ? 256 : (wn == 1) ? 64 : 128;
    p.tiles_m = (M + GBM - 1) / GBM; p.tiles_n = (N + gbn - 1) / gbn;
    int ksteps = (K + GBK - 1) / GBK;
    if (splitk > ksteps) splitk = ksteps;
    int steps_per = (ksteps + splitk - 1) / splitk;
    p.kchunk = steps_per * GBK;
    splitk = (ksteps + steps_per - 1) / steps_per;
    p.slab = nullptr; p.ld_slab = 0; p.slab_stride = 0;
    p.rowsum_a = rowsum_a;
    p.kb_valid = kb_valid;
    UNAST_REQUIRE(!rowsum_a || a_mode == OP_RC, "unast_gemm: rowsum_a needs a row-contiguous A operand (weight-gradient form)");
    p.atomic = 0;
    if (splitk > 1) {
        if (splitk_ws) {
            p.ld_slab = (N + 3) & ~3;
            p.slab_stride = (size_t)M * p.ld_slab;
            UNAST_REQUIRE((int64_t)(p.slab_stride * splitk) <= splitk_ws_floats && ((((uintptr_t)splitk_ws) & 15) == 0),
                          "unast_gemm: split-K workspace too small (need %lld floats) or misaligned", (long long)(p.slab_stride * splitk));
            p.slab = splitk_ws;
        } else {
            p.atomic = 1;
        }
    }
    if (a_mode == OP_KC_CONV) UNAST_REQUIRE((conv_ca & 3) == 0 && K == 5 * conv_ca && M % p.T == 0, "unast_gemm: bad conv A geometry");
    if (b_mode == OP_RC_CONV_DGRAD) UNAST_REQUIRE(K == 5 * conv_cb, "unast_gemm: bad conv dgrad geometry");
    if (b_mode == OP_RC_CONV_WGRAD) UNAST_REQUIRE((conv_cb & 3) == 0 && N == 5 * conv_cb && K % p.T == 0, "unast_gemm: bad conv wgrad geometry");
    p.nsplitk = splitk;
    // interior-only fast path (FLAGS bit 0): see gemm_kernel
    int flags = b_presplit ? 2 : 0;
    {
        const bool plain = (a_mode == OP_KC || a_mode == OP_RC) && (b_mode == OP_KC || b_mode == OP_RC);
        const uint64_t a_span = a_kc ? (uint64_t)M * lda * 4 : (uint64_t)(GBK + 1) * lda * 4 + (uint64_t)M * 4;
        const uint64_t b_span = b_kc ? (uint64_t)N * ldb * 4 : (uint64_t)(GBK + 1) * ldb * 4 + (uint64_t)N * 4;
        if (plain && M % GBM == 0 && N % gbn == 0 && K % GBK == 0 && kb_valid == K && a_span < (1ull << 32) && b_span < (1ull << 32))
            flags |= 1;
    }
    dim3 grid((p.tiles_m >= 8 && splitk == 1) ? ((p.tiles_m + 7) / 8) * 8 * p.tiles_n : ((splitk + 7) / 8) * 8 * p.tiles_m * p.tiles_n, 1, 1);
    if (a_mode == OP_KC && b_mode == OP_KC) launch_split<OP_KC, OP_KC, true, true>(p, nsplit, wn, flags, grid, stream);
    else if (a_mode == OP_KC && b_mode == OP_RC) launch_split<OP_KC, OP_RC, true, true>(p, nsplit, wn, flags, grid, stream);
    else if (a_mode == OP_RC && b_mode == OP_RC) launch_split<OP_RC, OP_RC, false, true>(p, nsplit, wn, flags, grid, stream);
    else if (a_mode == OP_KC_CONV && b_mode == OP_KC) launch_split<OP_KC_CONV, OP_KC, true, false>(p, nsplit, wn, flags, grid, stream);
    else if (a_mode == OP_KC_CONV && b_mode == OP_RC_CONV_DGRAD) launch_split<OP_KC_CONV, OP_RC_CONV_DGRAD, true, false>(p, nsplit, wn, flags, grid, stream);
    else if (a_mode == OP_RC && b_mode == OP_RC_CONV_WGRAD) launch_split<OP_RC, OP_RC_CONV_WGRAD, false, false>(p, nsplit, wn, flags, grid, stream);
    else return unast_set_error(UNAST_ERR_ARG, "unast_gemm: unsupported operand mode pair (%d,%d)", a_mode, b_mode);
    if (p.slab) {
        size_t work = (size_t)M * (p.ld_slab / 4);
        size_t blocks = (work + 63) / 64;                  // 64 float4 elements per workgroup
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p.slab, p.slab_stride, p.ld_slab, splitk, C, ldc, M, N, beta);
    }
    return unast_check_launch("unast_gemm");
}

// Host side of the grouped weight-gradient launch.  items[i] = {A = dY_i [tokens][M_i] (row stride lda), B = X_i [tokens][N_i],
// C = dW_i [M_i][N_i] (accumulated into), rowsum_a = db_i or NULL, M, N, K = tokens, lda, ldb, ldc}.
extern "C" int unast_wgrad_group(int nsplit, int count, const unast_wgrad_item* items, float* ws, int64_t ws_floats, int target_blocks,
                                 hipStream_t stream) {
    UNAST_REQUIRE(items && count >= 1 && count <= GROUP_MAX, "unast_wgrad_group: 1..%d problems per launch (got %d)", GROUP_MAX, count);
    UNAST_REQUIRE(nsplit == 1 || nsplit == 3, "unast_wgrad_group: nsplit must be 1 or 3");
    UNAST_REQUIRE(ws && ((((uintptr_t)ws) & 15) == 0), "unast_wgrad_group: needs a 16-byte aligned slab workspace");
    if (target_blocks <= 0) target_blocks = 512;
    GroupParams gp;
    gp.count = count;
    bool full = true;
    long long tile_ksteps = 0;
    for (int i = 0; i < count; ++i) {
        const unast_wgrad_item& s = items[i];
        UNAST_REQUIRE(s.A && s.B && s.C && s.M > 0 && s.N > 0 && s.K > 0, "unast_wgrad_group: bad problem %d", i);
        UNAST_REQUIRE(aligned16(s.A) && aligned16(s.B) && aligned16(s.C) && (s.lda & 3) == 0 && (s.ldb & 3) == 0, "unast_wgrad_group: problem %d: operands must be 16-byte aligned with ld %% 4 == 0", i);
        UNAST_REQUIRE(s.lda >= ((s.M + 3) & ~3) && s.ldb >= ((s.N + 3) & ~3), "unast_wgrad_group: problem %d: row-contiguous operands need ld >= ceil4(rows)", i);
        GroupItem& it = gp.it[i];
        it.A = s.A; it.B = s.B; it.C = s.C; it.rowsum_a = s.rowsum_a; it.M = s.M; it.N = s.N; it.K = s.K; it.lda = s.lda; it.ldb = s.ldb; it.ldc = s.ldc;
        it.tiles_m = (s.M + GBM - 1) / GBM; it.tiles_n = (s.N + 127) / 128;
        it.ld_slab = (s.N + 3) & ~3;
        const uint64_t a_span = (uint64_t)(GBK + 1) * s.lda * 4 + (uint64_t)s.M * 4, b_span = (uint64_t)(GBK + 1) * s.ldb * 4 + (uint64_t)s.N * 4;
        if (!(s.M % GBM == 0 && s.N % 128 == 0 && s.K % GBK == 0 && a_span < (1ull << 32) && b_span < (1ull << 32))) full = false;
        tile_ksteps += (long long)it.tiles_m * it.tiles_n * ((s.K + GBK - 1) / GBK);
    }
    // K-slices: every workgroup gets about the same number of k-steps (at least 10), ~target_blocks workgroups in all
    long long per_wg = (tile_ksteps + target_blocks - 1) / target_blocks;
    if (per_wg < 10) per_wg = 10;
    size_t ws_need = 0;
    int blocks = 0, rblocks = 0;
    for (int i = 0; i < count; ++i) {
        GroupItem& it = gp.it[i];
        const int ksteps = (it.K + GBK - 1) / GBK;
        int splits = (int)((ksteps + per_wg - 1) / per_wg);
        if (splits < 1) splits = 1;
        const int steps_per = (ksteps + splits - 1) / splits;
        it.kchunk = steps_per * GBK;
        it.nsplitk = (ksteps + steps_per - 1) / steps_per;
        it.slab = ws + ws_need;
        ws_need += (size_t)it.nsplitk * it.M * it.ld_slab;
        gp.base[i] = blocks;
        blocks += ((it.nsplitk + 7) / 8) * 8 * it.tiles_m * it.tiles_n;
        gp.rbase[i] = rblocks;
        rblocks += (int)(((size_t)it.M * (it.ld_slab / 4) + 63) / 64);
    }
    for (int i = count; i <= GROUP_MAX; ++i) { gp.base[i] = blocks; gp.rbase[i] = rblocks; }
    UNAST_REQUIRE((int64_t)ws_need <= ws_floats, "unast_wgrad_group: workspace too small (need %lld floats, got %lld)", (long long)ws_need, (long long)ws_floats);
    if (nsplit == 3) {
        if (full) hipLaunchKernelGGL((gemm_group_kernel<3, 1>), dim3(blocks), dim3(256), 0, stream, gp);
        else      hipLaunchKernelGGL((gemm_group_kernel<3, 0>), dim3(blocks), dim3(256), 0, stream, gp);
    } else {
        if (full) hipLaunchKernelGGL((gemm_group_kernel<1, 1>), dim3(blocks), dim3(256), 0, stream, gp);
        else      hipLaunchKernelGGL((gemm_group_kernel<1, 0>), dim3(blocks), dim3(256), 0, stream, gp);
    }
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(rblocks), dim3(256), 0, stream, gp);
    return unast_check_launch("unast_wgrad_group");
}

// Floats of slab workspace unast_wgrad_group needs for these problems (same split policy).
extern "C" int64_t unast_wgrad_group_ws_floats(int count, const unast_wgrad_item* items, int target_blocks) {
    if (!items || count < 1 || count > GROUP_MAX) return -1;
    if (target_blocks <= 0) target_blocks = 512;
    long long tile_ksteps = 0;
    for (int i = 0; i < count; ++i)
        tile_ksteps += (long long)((items[i].M + GBM - 1) / GBM) * ((items[i].N + 127) / 128) * ((items[i].K + GBK - 1) / GBK);
    long long per_wg = (tile_ksteps + target_blocks - 1) / target_blocks;
    if (per_wg < 10) per_wg = 10;
    int64_t need = 0;
    for (int i = 0; i < count; ++i) {
        const int ksteps = (items[i].K + GBK - 1) / GBK;
        int splits = (int)((ksteps + per_wg - 1) / per_wg);
        if (splits < 1) splits = 1;
        const int steps_per = (ksteps + splits - 1) / splits;
        const int ns = (ksteps + steps_per - 1) / steps_per;
        need += (int64_t)ns * items[i].M * ((items[i].N + 3) & ~3);
    }
    return need;
}

UNAST_DEFINE_RNG_EPOCH_SETTER(gemm)

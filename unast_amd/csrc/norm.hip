// LayerNorm (post-LN transformer blocks) and train-mode BatchNorm1d (conv stacks) for gfx950.
// All kernels are HBM-bound: one pass over each operand with 16-byte loads per lane, statistics in registers,
// cross-lane reductions with wave shuffles.  Reference semantics: torch.nn.LayerNorm(eps=1e-5) inside
// torch.nn.TransformerEncoderLayer/DecoderLayer (src/module.py:273-274, 286-287) and nn.BatchNorm1d in training
// mode over all B*T positions incl. padding (src/module.py:145-147, 208-210).
#include "common.h"

// ------------------------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row, C <= 1024, C % 4 == 0.  y = (z-mean)*rstd*gamma + beta
// ------------------------------------------------------------------------------------------------------------
#define LN_MAXV 4
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* zr = z + (size_t)row * C;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int c = (lane + 64 * i) * 4;
        if (c < C) {
            v[i] = *reinterpret_cast<const float4*>(zr + c);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int c = (lane + 64 * i) * 4;
        if (c < C) {
            float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    float* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int c = (lane + 64 * i) * 4;
        if (c < C) {
            float4 g = *reinterpret_cast<const float4*>(gamma + c);
            float4 b = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = (v[i].x - mu) * rs * g.x + b.x;
            o.y = (v[i].y - mu) * rs * g.y + b.y;
            o.z = (v[i].z - mu) * rs * g.z + b.z;
            o.w = (v[i].w - mu) * rs * g.w + b.w;
            *reinterpret_cast<float4*>(yr + c) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  dz = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma;  dgamma += sum dy*xhat,
// dbeta += sum dy (block partials, then fp32 atomics into the gradient buffer).  Optional second output
// dz_drop = dz * dropout-mask(seed,stream,row,col) / (1-p): the gradient of the sub-layer output that was
// dropped before the residual add in the forward pass.
// ------------------------------------------------------------------------------------------------------------
// NV = 256-column groups per row (C <= 256 * NV), RU = rows in flight per wave.  With one row in flight a wave has 2 KB of
// loads outstanding and the 3 200 waves of a config-3 launch keep ~6 MB in flight -- the kernel then runs at ~3.2 TB/s, the
// rate that latency allows; RU = 4 for the d = 256 model quadruples that.
template <int NV, int RU>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dz,
                                                            float* __restrict__ dz_drop, float* __restrict__ part,
                                                            int rows, int C, int rows_per_block,
                                                            uint32_t drop_thresh, float drop_scale, uint32_t seed, uint32_t stream) {
    __shared__ float red[2][4][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 ag[NV], ab[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = make_float4(0, 0, 0, 0);
        ab[i] = make_float4(0, 0, 0, 0);
        int c = (lane + 64 * i) * 4;
        gm[i] = (c < C) ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0, 0, 0, 0);
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    for (int row0 = r0 + wave; row0 < r1; row0 += 4 * RU) {
        float4 dv[RU][NV], zv[RU][NV];
        float mus[RU], rss[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {                     // all loads of the RU rows first (rows past the block re-read row0; not used)
            const int row = (row0 + 4 * u < r1) ? row0 + 4 * u : row0;
            mus[u] = mean[row]; rss[u] = rstd[row];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int c = (lane + 64 * i) * 4;
                if (c < C) {
                    dv[u][i] = *reinterpret_cast<const float4*>(dy + (size_t)row * C + c);
                    zv[u][i] = *reinterpret_cast<const float4*>(z + (size_t)row * C + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = row0 + 4 * u;
            if (row >= r1) break;                            // wave-uniform
            const float mu = mus[u], rs = rss[u];
            float4 xh[NV], g[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int c = (lane + 64 * i) * 4;
                if (c < C) {
                    const float4 d = dv[u][i], zz = zv[u][i];
                    xh[i] = make_float4((zz.x - mu) * rs, (zz.y - mu) * rs, (zz.z - mu) * rs, (zz.w - mu) * rs);
                    g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
                    s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                    s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                    ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
                    ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
                }
            }
            const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
            uint32_t rkey = drop_thresh ? rng_row_key(seed, stream, (uint32_t)row) : 0u;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int c = (lane + 64 * i) * 4;
                if (c < C) {
                    float4 o;
                    o.x = rs * (g[i].x - c1 - xh[i].x * c2);
                    o.y = rs * (g[i].y - c1 - xh[i].y * c2);
                    o.z = rs * (g[i].z - c1 - xh[i].z * c2);
                    o.w = rs * (g[i].w - c1 - xh[i].w * c2);
                    *reinterpret_cast<float4*>(dz + (size_t)row * C + c) = o;
                    if (dz_drop) {
                        float4 od;
                        od.x = rng_keep(rkey, c + 0, drop_thresh) ? o.x * drop_scale : 0.f;
                        od.y = rng_keep(rkey, c + 1, drop_thresh) ? o.y * drop_scale : 0.f;
                        od.z = rng_keep(rkey, c + 2, drop_thresh) ? o.z * drop_scale : 0.f;
                        od.w = rng_keep(rkey, c + 3, drop_thresh) ? o.w * drop_scale : 0.f;
                        *reinterpret_cast<float4*>(dz_drop + (size_t)row * C + c) = od;
                    }
                }
            }
        }
    }
    if (!part) return;
    // block reduction of the column partials (4 waves); one row of [2][C] partials per workgroup, summed by ln_bwd_finalize
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (lane + 64 * i) * 4;
        if (c < C) {
            *reinterpret_cast<float4*>(&red[0][wave][c]) = ag[i];
            *reinterpret_cast<float4*>(&red[1][wave][c]) = ab[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float sg = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        float sb = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = sg;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = sb;
    }
}

__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // 64 columns x 16 partial-row lanes per workgroup; columns index the concatenated [dgamma | dbeta] vector of 2C entries
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float a = 0.f;
    if (c < 2 * C) {
        const int which = c / C, col = c - which * C;
        const float* pp = part + (size_t)which * C + col;
        const size_t st = (size_t)2 * C;
        int b = ty;
        for (; b + 112 < nblk; b += 128) {                  // 8 loads in flight per thread (a plain loop is a chain of round trips)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(b + 16 * u) * st];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; b < nblk; b += 16) a += pp[(size_t)b * st];
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && c < 2 * C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][tx];
        const int which = c / C, col = c - which * C;
        (which ? dbeta : dgamma)[col] += s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Column sums over rows: s1[c] += sum_r x[r,c] ; s2[c] += sum_r x[r,c]^2 (s2 optional).
// Used for BatchNorm batch statistics (double accumulators) and bias gradients (float accumulators).
// Thread (tx, ty): tx = column quad, ty = row lane; per-thread fp32 partials over <= rows_per_block/nry rows.
// ------------------------------------------------------------------------------------------------------------
template <typename ACC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ldx, int rows, int C, int rows_per_block,
                                                     ACC* __restrict__ s1, ACC* __restrict__ s2) {
    __shared__ float red[2][256][4];
    const int cq = C >> 2;
    const int nry = 256 / cq;
    const int tx = threadIdx.x % cq, ty = threadIdx.x / cq;
    float4 a = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if (ty < nry) {
        int r = r0 + ty;
        for (; r + 3 * nry < r1; r += 4 * nry) {            // four rows in flight per thread
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (size_t)(r + u * nry) * ldx + tx * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
                q.x += v[u].x * v[u].x; q.y += v[u].y * v[u].y; q.z += v[u].z * v[u].z; q.w += v[u].w * v[u].w;
            }
        }
        for (; r < r1; r += nry) {
            float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + tx * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
        }
    }
    *reinterpret_cast<float4*>(red[0][threadIdx.x]) = a;
    *reinterpret_cast<float4*>(red[1][threadIdx.x]) = q;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int qd = c >> 2, e = c & 3;
        float sa = 0.f, sq = 0.f;
        for (int yy = 0; yy < nry; ++yy) { sa += red[0][yy * cq + qd][e]; sq += red[1][yy * cq + qd][e]; }
        atomicAdd(s1 + c, (ACC)sa);
        if (s2) atomicAdd(s2 + c, (ACC)sq);
    }
}

// Scalar variant for column counts that are not multiples of 4 (46 phoneme logits, 81 = mel+stop head, 1 = fc2).
__global__ __launch_bounds__(256) void colsum_scalar_kernel(const float* __restrict__ x, int ldx, int rows, int C, int rows_per_block, float* __restrict__ s1) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int r = r0; r < r1; ++r) a += x[(size_t)r * ldx + c];
        atomicAdd(s1 + c, a);
    }
}

__device__ __forceinline__ float act_fwd(float x, int act) { return act == 1 ? fmaxf(x, 0.f) : (act == 2 ? tanhf(x) : x); }

// Train-mode BatchNorm forward in ONE pass over x: y = dropout(act((x-mean)*rstd*gamma + beta)); act: 0 none, 1 relu, 2 tanh.
// The batch statistics are finalized HERE from the fp64 column sums s1 = sum x, s2 = sum x^2 (the conv GEMM's epilogue or colsum_kernel
// produced them): every thread turns the sums of its 4 channels into mean / rstd (biased variance) before its first row -- its channel
// quad never changes when the grid's thread count is a multiple of C/4 -- so the former bn_finalize launch is gone; workgroup 0 also
// stores mean / rstd for the backward, updates the running statistics (momentum, unbiased variance) and counts the batch
// (num_batches_tracked += 1, formerly a torch add per layer and step).  stats_given: mean / rstd are read (eval mode) instead.
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const double* __restrict__ s1, const double* __restrict__ s2,
                                                           double n, float eps, float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           long long* __restrict__ num_batches_tracked, int stats_given,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, int rows, int C,
                                                           int act, uint32_t drop_thresh, float drop_scale, uint32_t seed, uint32_t stream) {
    const int cq = C >> 2;
    const size_t total = (size_t)rows * cq;
    const bool fixed = ((size_t)gridDim.x * 256) % (size_t)cq == 0;           // this thread's channel quad is the same in every iteration
    float m[4], rs[4];
    auto stats = [&](int c) {
        if (stats_given) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { m[e] = mean[c + e]; rs[e] = rstd[c + e]; }
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double mu = s1[c + e] / n;
            double var = s2[c + e] / n - mu * mu;
            if (var < 0.0) var = 0.0;
            m[e] = (float)mu;
            rs[e] = (float)(1.0 / sqrt(var + (double)eps));
        }
    };
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (fixed) stats((int)(i0 % cq) * 4);
    if (!stats_given && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            const double mu = s1[c] / n;
            double var = s2[c] / n - mu * mu;
            if (var < 0.0) var = 0.0;
            mean[c] = (float)mu;
            rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
        if (num_batches_tracked && threadIdx.x == 0) *num_batches_tracked += 1;
    }
    for (size_t i = i0; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cq), c = (int)(i - (size_t)r * cq) * 4;
        if (!fixed) stats(c);
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
        float o[4] = {(v.x - m[0]) * rs[0] * g.x + b.x, (v.y - m[1]) * rs[1] * g.y + b.y, (v.z - m[2]) * rs[2] * g.z + b.z,
                      (v.w - m[3]) * rs[3] * g.w + b.w};
        uint32_t rkey = drop_thresh ? rng_row_key(seed, stream, (uint32_t)r) : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = act_fwd(o[e], act);
            if (drop_thresh) a = rng_keep(rkey, c + e, drop_thresh) ? a * drop_scale : 0.f;
            o[e] = a;
        }
        *reinterpret_cast<float4*>(y + (size_t)r * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// BatchNorm backward, pass 1: dpre = dy * dropmask/(1-p) * act'(pre) written IN PLACE over dy, and the column sums
// S1 = sum dpre (= dbeta), S2 = sum dpre*xhat (= dgamma) accumulated in double.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int rows, int C, int rows_per_block, int act, uint32_t drop_thresh,
                                                            float drop_scale, uint32_t seed, uint32_t stream,
                                                            double* __restrict__ S1, double* __restrict__ S2) {
    __shared__ float red[2][256][4];
    const int cq = C >> 2;
    const int nry = 256 / cq;
    const int tx = threadIdx.x % cq, ty = threadIdx.x / cq;
    const int c = tx * 4;
    float a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if (ty < nry) {
        float4 m = *reinterpret_cast<const float4*>(mean + c), s = *reinterpret_cast<const float4*>(rstd + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
        const float mm[4] = {m.x, m.y, m.z, m.w}, ss[4] = {s.x, s.y, s.z, s.w}, gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
        for (int r = r0 + ty; r < r1; r += nry) {
            float4 d4 = *reinterpret_cast<const float4*>(dy + (size_t)r * C + c);
            float4 x4 = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
            float d[4] = {d4.x, d4.y, d4.z, d4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
            uint32_t rkey = drop_thresh ? rng_row_key(seed, stream, (uint32_t)r) : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[e] - mm[e]) * ss[e];
                const float pre = xh * gg[e] + bb[e];
                float dd = d[e];
                if (drop_thresh) dd = rng_keep(rkey, c + e, drop_thresh) ? dd * drop_scale : 0.f;
                if (act == 1) dd = pre > 0.f ? dd : 0.f;
                else if (act == 2) { const float th = tanhf(pre); dd *= (1.f - th * th); }
                d[e] = dd;
                a[e] += dd;
                q[e] += dd * xh;
            }
            *reinterpret_cast<float4*>(dy + (size_t)r * C + c) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
    *reinterpret_cast<float4*>(red[0][threadIdx.x]) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(red[1][threadIdx.x]) = make_float4(q[0], q[1], q[2], q[3]);
    __syncthreads();
    for (int cc = threadIdx.x; cc < C; cc += 256) {
        const int qd = cc >> 2, e = cc & 3;
        float sa = 0.f, sq = 0.f;
        for (int yy = 0; yy < nry; ++yy) { sa += red[0][yy * cq + qd][e]; sq += red[1][yy * cq + qd][e]; }
        atomicAdd(S1 + cc, (double)sa);
        atomicAdd(S2 + cc, (double)sq);
    }
}

// BatchNorm backward, pass 2: dx = gamma*rstd*(dpre - S1/n - xhat*S2/n); also dgamma += S2, dbeta += S1 (block 0).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dpre, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ S1,
                                                           const double* __restrict__ S2, float* __restrict__ dx, int rows, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int cq = C >> 2;
    const size_t total = (size_t)rows * cq;
    const float invn = 1.f / (float)rows;
    if (blockIdx.x == 0 && dgamma) {
        for (int c = threadIdx.x; c < C; c += 256) { dgamma[c] += (float)S2[c]; dbeta[c] += (float)S1[c]; }
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cq), c = (int)(i - (size_t)r * cq) * 4;
        float4 d = *reinterpret_cast<const float4*>(dpre + (size_t)r * C + c);
        float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
        float4 m = *reinterpret_cast<const float4*>(mean + c), s = *reinterpret_cast<const float4*>(rstd + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float s1[4] = {(float)S1[c] * invn, (float)S1[c + 1] * invn, (float)S1[c + 2] * invn, (float)S1[c + 3] * invn};
        const float s2[4] = {(float)S2[c] * invn, (float)S2[c + 1] * invn, (float)S2[c + 2] * invn, (float)S2[c + 3] * invn};
        float4 o;
        o.x = g.x * s.x * (d.x - s1[0] - (xv.x - m.x) * s.x * s2[0]);
        o.y = g.y * s.y * (d.y - s1[1] - (xv.y - m.y) * s.y * s2[1]);
        o.z = g.z * s.z * (d.z - s1[2] - (xv.z - m.z) * s.z * s2[2]);
        o.w = g.w * s.w * (d.w - s1[3] - (xv.w - m.w) * s.w * s2[3]);
        *reinterpret_cast<float4*>(dx + (size_t)r * C + c) = o;
    }
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
static int grid_for(size_t work, int per_block, int cap = 2048) {
    size_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

extern "C" int unast_layernorm_fwd(const float* z, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int rows, int C, float eps, hipStream_t stream) {
    UNAST_REQUIRE(z && gamma && beta && y && mean && rstd, "unast_layernorm_fwd: null pointer");
    UNAST_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && C <= 1024, "unast_layernorm_fwd: need C%%4==0, C<=1024 (C=%d)", C);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, z, gamma, beta, y, mean, rstd, rows, C, eps);
    return unast_check_launch("unast_layernorm_fwd");
}

static void ln_bwd_geometry(int rows, int* blocks, int* rpb) {
    int b = grid_for(rows, 32, 1024);
    *rpb = (rows + b - 1) / b;
    *blocks = (rows + *rpb - 1) / *rpb;
}

extern "C" int64_t unast_layernorm_bwd_ws_floats(int rows, int C) {
    int blocks, rpb;
    ln_bwd_geometry(rows, &blocks, &rpb);
    return (int64_t)blocks * 2 * C;
}

extern "C" int unast_layernorm_bwd(const float* dy, const float* z, const float* gamma, const float* mean, const float* rstd,
                                   float* dz, float* dz_drop, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, int rows, int C,
                                   float drop_p, unsigned int seed, unsigned int stream_id, int finalize, hipStream_t stream) {
    UNAST_REQUIRE(dy && z && gamma && mean && rstd && dz, "unast_layernorm_bwd: null pointer");
    UNAST_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "unast_layernorm_bwd: dgamma/dbeta must both be given or both null");
    UNAST_REQUIRE(rows > 0 && (C & 3) == 0 && C <= 1024, "unast_layernorm_bwd: need C%%4==0, C<=1024 (C=%d)", C);
    int blocks, rpb;
    ln_bwd_geometry(rows, &blocks, &rpb);
    UNAST_REQUIRE(!dgamma || (ws && ws_floats >= (int64_t)blocks * 2 * C), "unast_layernorm_bwd: workspace too small (need %lld floats)", (long long)blocks * 2 * C);
    uint32_t th = dz_drop ? drop_threshold(drop_p) : 0u;
    float* const dzd = th ? dz_drop : (float*)nullptr;
    float* const prt = dgamma ? ws : (float*)nullptr;
    const float dsc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    if (C <= 256)      hipLaunchKernelGGL((layernorm_bwd_kernel<1, 4>), dim3(blocks), dim3(256), 0, stream, dy, z, gamma, mean, rstd, dz, dzd, prt, rows, C, rpb, th, dsc, seed, stream_id);
    else if (C <= 512) hipLaunchKernelGGL((layernorm_bwd_kernel<2, 2>), dim3(blocks), dim3(256), 0, stream, dy, z, gamma, mean, rstd, dz, dzd, prt, rows, C, rpb, th, dsc, seed, stream_id);
    else               hipLaunchKernelGGL((layernorm_bwd_kernel<4, 1>), dim3(blocks), dim3(256), 0, stream, dy, z, gamma, mean, rstd, dz, dzd, prt, rows, C, rpb, th, dsc, seed, stream_id);
    if (dgamma && finalize) hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, stream, ws, blocks, C, dgamma, dbeta);
    return unast_check_launch("unast_layernorm_bwd");
}

extern "C" int unast_layernorm_bwd_finalize(const float* ws, int64_t ws_floats, int rows, int C, float* dgamma, float* dbeta, hipStream_t stream) {
    UNAST_REQUIRE(ws && dgamma && dbeta && rows > 0 && (C & 3) == 0 && C <= 1024, "unast_layernorm_bwd_finalize: bad arguments");
    int blocks, rpb;
    ln_bwd_geometry(rows, &blocks, &rpb);
    UNAST_REQUIRE(ws_floats >= (int64_t)blocks * 2 * C, "unast_layernorm_bwd_finalize: workspace too small");
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, stream, ws, blocks, C, dgamma, dbeta);
    return unast_check_launch("unast_layernorm_bwd_finalize");
}

// The same reduction for partials written by another kernel (unast_panel_gemm_lnbwd: one [2][C] row per 128-row panel).
extern "C" int unast_layernorm_partials_finalize(const float* part, int nblk, int C, float* dgamma, float* dbeta, hipStream_t stream) {
    UNAST_REQUIRE(part && dgamma && dbeta && nblk > 0 && C > 0 && C <= 1024, "unast_layernorm_partials_finalize: bad arguments");
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, stream, part, nblk, C, dgamma, dbeta);
    return unast_check_launch("unast_layernorm_partials_finalize");
}

static int colsum_geometry(int rows, int C, int* blocks, int* rpb) {
    if ((C & 3) != 0 || C > 1024 || C <= 0) return -1;
    int b = grid_for(rows, 64, 1024);
    *rpb = (rows + b - 1) / b;
    *blocks = (rows + *rpb - 1) / *rpb;
    return 0;
}

// Column sums with a FIXED summation order (unast_colsum_det; parity / reproducibility mode): every workgroup writes the partial sums
// of its row chunk (rows in order, thread = column), a second launch adds the chunks in order.  No atomics.
__global__ __launch_bounds__(256) void colsum_det_part_kernel(const float* __restrict__ x, int ldx, int rows, int C, int rows_per_block, float* __restrict__ part) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int r = r0; r < r1; ++r) a += x[(size_t)r * ldx + c];
        part[(size_t)blockIdx.x * C + c] = a;
    }
}
__global__ __launch_bounds__(256) void colsum_det_final_kernel(const float* __restrict__ part, int blocks, int C, float* __restrict__ sum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = sum[c];
    for (int b = 0; b < blocks; ++b) a += part[(size_t)b * C + c];
    sum[c] = a;
}
static void colsum_det_geometry(int rows, int* blocks, int* rpb) {
    int b = grid_for(rows, 64, 512);
    *rpb = (rows + b - 1) / b;
    *blocks = (rows + *rpb - 1) / *rpb;
}
extern "C" int64_t unast_colsum_det_ws_floats(int rows, int C) {
    int blocks, rpb;
    colsum_det_geometry(rows, &blocks, &rpb);
    return (int64_t)blocks * C;
}
extern "C" int unast_colsum_det(const float* x, int ldx, int rows, int C, float* sum, float* ws, int64_t ws_floats, hipStream_t stream) {
    UNAST_REQUIRE(x && sum && ws && rows > 0 && C > 0, "unast_colsum_det: bad arguments");
    int blocks, rpb;
    colsum_det_geometry(rows, &blocks, &rpb);
    UNAST_REQUIRE(ws_floats >= (int64_t)blocks * C, "unast_colsum_det: workspace too small (need %lld floats)", (long long)blocks * C);
    hipLaunchKernelGGL(colsum_det_part_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, rows, C, rpb, ws);
    hipLaunchKernelGGL(colsum_det_final_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, ws, blocks, C, sum);
    return unast_check_launch("unast_colsum_det");
}

extern "C" int unast_colsum_f32(const float* x, int ldx, int rows, int C, float* sum, hipStream_t stream) {
    UNAST_REQUIRE(x && sum, "unast_colsum_f32: null pointer");
    int blocks, rpb;
    UNAST_REQUIRE(rows > 0 && C > 0 && C <= 1024, "unast_colsum_f32: need 0 < C <= 1024");
    if ((C & 3) != 0 || (ldx & 3) != 0 || (((uintptr_t)x) & 15) != 0) {
        blocks = grid_for(rows, 64, 1024);
        rpb = (rows + blocks - 1) / blocks;
        blocks = (rows + rpb - 1) / rpb;
        hipLaunchKernelGGL(colsum_scalar_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, rows, C, rpb, sum);
        return unast_check_launch("unast_colsum_f32");
    }
    colsum_geometry(rows, C, &blocks, &rpb);
    hipLaunchKernelGGL((colsum_kernel<float>), dim3(blocks), dim3(256), 0, stream, x, ldx, rows, C, rpb, sum, (float*)nullptr);
    return unast_check_launch("unast_colsum_f32");
}

extern "C" int unast_bn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            float* running_mean, float* running_var, double* ws /* 2*C doubles */, int rows, int C,
                            float eps, float momentum, int act, float drop_p, unsigned int seed, unsigned int stream_id,
                            int have_sums, int64_t* num_batches_tracked, hipStream_t stream) {
    UNAST_REQUIRE(x && gamma && beta && y && mean && rstd && ws, "unast_bn_fwd: null pointer");
    int blocks, rpb;
    UNAST_REQUIRE(colsum_geometry(rows, C, &blocks, &rpb) == 0, "unast_bn_fwd: need C%%4==0, C<=1024 (C=%d)", C);
    if (!have_sums) {       // have_sums: ws already holds sum x | sum x^2 per column (the producing conv GEMM's epilogue, unast_gemm colstats)
        hipMemsetAsync(ws, 0, sizeof(double) * 2 * C, stream);
        hipLaunchKernelGGL((colsum_kernel<double>), dim3(blocks), dim3(256), 0, stream, x, C, rows, C, rpb, ws, ws + C);
    }
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(grid_for((size_t)rows * (C / 4), 256)), dim3(256), 0, stream, x, ws, ws + C, (double)rows, eps, momentum,
                       mean, rstd, running_mean, running_var, (long long*)num_batches_tracked, 0, gamma, beta, y,
                       rows, C, act, drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id);
    return unast_check_launch("unast_bn_fwd");
}

// Eval-mode statistics: mean = running_mean, rstd = 1/sqrt(running_var + eps).
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    rstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

extern "C" int unast_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float* y, float* mean, float* rstd, int rows, int C, float eps, int act,
                                 hipStream_t stream) {
    UNAST_REQUIRE(x && gamma && beta && running_mean && running_var && y && mean && rstd, "unast_bn_eval_fwd: null pointer");
    UNAST_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "unast_bn_eval_fwd: need rows>0, C%%4==0 (rows=%d C=%d)", rows, C);
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, running_mean, running_var, C, eps, mean, rstd);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(grid_for((size_t)rows * (C / 4), 256)), dim3(256), 0, stream, x, (const double*)nullptr, (const double*)nullptr, 1.0, eps,
                       0.f, mean, rstd, (float*)nullptr, (float*)nullptr, (long long*)nullptr, 1, gamma, beta, y, rows, C, act, 0u, 1.f, 0u, 0u);
    return unast_check_launch("unast_bn_eval_fwd");
}

extern "C" int unast_bn_bwd(float* dy_inout, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            float* dx, float* dgamma, float* dbeta, double* ws /* 2*C doubles */, int rows, int C, int act,
                            float drop_p, unsigned int seed, unsigned int stream_id, int ws_zeroed, hipStream_t stream) {
    UNAST_REQUIRE(dy_inout && x && mean && rstd && gamma && beta && dx && ws, "unast_bn_bwd: null pointer");
    UNAST_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "unast_bn_bwd: dgamma/dbeta must both be given or both null");
    int blocks, rpb;
    UNAST_REQUIRE(colsum_geometry(rows, C, &blocks, &rpb) == 0, "unast_bn_bwd: need C%%4==0, C<=1024 (C=%d)", C);
    if (!ws_zeroed) hipMemsetAsync(ws, 0, sizeof(double) * 2 * C, stream);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, stream, dy_inout, x, mean, rstd, gamma, beta, rows, C, rpb, act,
                       drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id, ws, ws + C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((size_t)rows * (C / 4), 256)), dim3(256), 0, stream, dy_inout, x, mean, rstd, gamma,
                       ws, ws + C, dx, rows, C, dgamma, dbeta);
    return unast_check_launch("unast_bn_bwd");
}

UNAST_DEFINE_RNG_EPOCH_SETTER(norm)

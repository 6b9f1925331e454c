// Small HBM-bound kernels around the transformer stacks: embedding gather/scatter, positional encoding,
// timestep noise, SpecAugment, discriminator batch assembly, gradient accumulation.  16-byte accesses per lane.
#include "common.h"

static int ew_grid(size_t work_items, int cap = 2048) {
    size_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------------------
// Embedding forward (TextPrenet.embed + emb_dropout + noise_fn, src/module.py:189,226, src/network.py:429-432):
// out[b,t,:] = E[id] * dropmask/(1-p) * keep_row(noise).  shift_sos >= 0 builds the decoder input on the fly:
// id(b,t) = t==0 ? shift_sos : ids[b,t-1]  (src/network.py:483-487).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ E,
                                                        float* __restrict__ out, int rows, int T, int D, int shift_sos,
                                                        uint32_t drop_thresh, float drop_scale, uint32_t seed, uint32_t stream,
                                                        uint32_t noise_thresh, uint32_t noise_stream) {
    const int dq = D >> 2;
    const size_t total = (size_t)rows * dq;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / dq), c = (int)(i - (size_t)r * dq) * 4;
        int64_t id;
        if (shift_sos >= 0) { const int t = r % T; id = (t == 0) ? (int64_t)shift_sos : ids[r - 1]; }
        else id = ids[r];
        float4 v = *reinterpret_cast<const float4*>(E + (size_t)id * D + c);
        float o[4] = {v.x, v.y, v.z, v.w};
        if (drop_thresh) {
            const uint32_t rkey = rng_row_key(seed, stream, (uint32_t)r);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rng_keep(rkey, c + e, drop_thresh) ? o[e] * drop_scale : 0.f;
        }
        if (noise_thresh) {
            const bool keep = rng_keep(rng_row_key(seed, noise_stream, (uint32_t)r), 0u, noise_thresh);
            if (!keep) { o[0] = o[1] = o[2] = o[3] = 0.f; }
        }
        *reinterpret_cast<float4*>(out + (size_t)r * D + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Embedding backward: grid (vocab, token chunks); a workgroup scans its chunk, sums the rows that hit its vocabulary id
// in registers (thread = column) and adds them with one atomic per column.  Row `padding_idx` gets no gradient.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                        float* __restrict__ dE, int rows, int T, int D, int shift_sos, int padding_idx,
                                                        int rows_per_chunk, uint32_t drop_thresh, float drop_scale, uint32_t seed, uint32_t stream,
                                                        uint32_t noise_thresh, uint32_t noise_stream) {
    __shared__ int hit[512];
    __shared__ int nhit;
    const int v = blockIdx.x;
    if (v == padding_idx) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    if (threadIdx.x == 0) nhit = 0;
    __syncthreads();
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        int64_t id;
        if (shift_sos >= 0) { const int t = r % T; id = (t == 0) ? (int64_t)shift_sos : ids[r - 1]; }
        else id = ids[r];
        if (id == v) { const int k = atomicAdd(&nhit, 1); if (k < 512) hit[k] = r; }
    }
    __syncthreads();
    const int n = min(nhit, 512);
    if (n == 0) return;
    for (int c = threadIdx.x; c < D; c += 256) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) {
            const int r = hit[k];
            float g = dout[(size_t)r * D + c];
            if (drop_thresh) g = rng_keep(rng_row_key(seed, stream, (uint32_t)r), c, drop_thresh) ? g * drop_scale : 0.f;
            if (noise_thresh && !rng_keep(rng_row_key(seed, noise_stream, (uint32_t)r), 0u, noise_thresh)) g = 0.f;
            acc += g;
        }
        atomicAdd(dE + (size_t)v * D + c, acc);
    }
}

// ------------------------------------------------------------------------------------------------------------
// PositionalEncoding (src/module.py:249-267): y = dropout(x*scale + pe[t]).  Backward: dx = dy*mask/(1-p)*scale,
// optionally gated by gate>0 (ReLU of the layer that produced x, e.g. SpeechPrenet.fc2).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void posenc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pe, float* __restrict__ y,
                                                         int rows, int T, int D, float scale, uint32_t drop_thresh, float drop_scale,
                                                         uint32_t seed, uint32_t stream) {
    const int dq = D >> 2;
    const size_t total = (size_t)rows * dq;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / dq), c = (int)(i - (size_t)r * dq) * 4;
        const int t = r % T;
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * D + c);
        float4 p = *reinterpret_cast<const float4*>(pe + (size_t)t * D + c);
        float o[4] = {v.x * scale + p.x, v.y * scale + p.y, v.z * scale + p.z, v.w * scale + p.w};
        if (drop_thresh) {
            const uint32_t rkey = rng_row_key(seed, stream, (uint32_t)r);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rng_keep(rkey, c + e, drop_thresh) ? o[e] * drop_scale : 0.f;
        }
        *reinterpret_cast<float4*>(y + (size_t)r * D + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void posenc_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ gate, float* __restrict__ dx,
                                                         int rows, int D, float scale, uint32_t drop_thresh, float drop_scale,
                                                         uint32_t seed, uint32_t stream) {
    const int dq = D >> 2;
    const size_t total = (size_t)rows * dq;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / dq), c = (int)(i - (size_t)r * dq) * 4;
        float4 v = *reinterpret_cast<const float4*>(dy + (size_t)r * D + c);
        float o[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
        if (drop_thresh) {
            const uint32_t rkey = rng_row_key(seed, stream, (uint32_t)r);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rng_keep(rkey, c + e, drop_thresh) ? o[e] * drop_scale : 0.f;
        }
        if (gate) {
            float4 g = *reinterpret_cast<const float4*>(gate + (size_t)r * D + c);
            if (!(g.x > 0.f)) o[0] = 0.f;
            if (!(g.y > 0.f)) o[1] = 0.f;
            if (!(g.z > 0.f)) o[2] = 0.f;
            if (!(g.w > 0.f)) o[3] = 0.f;
        }
        *reinterpret_cast<float4*>(dx + (size_t)r * D + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// noise_fn (src/utils.py:40-49): zero whole timesteps with probability p, NO rescale.  Also serves as its own backward.
__global__ __launch_bounds__(256) void rowmask_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int D,
                                                      uint32_t thresh, uint32_t seed, uint32_t stream) {
    const int dq = D >> 2;
    const size_t total = (size_t)rows * dq;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / dq), c = (int)(i - (size_t)r * dq) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * D + c);
        if (!rng_keep(rng_row_key(seed, stream, (uint32_t)r), 0u, thresh)) v = make_float4(0, 0, 0, 0);
        *reinterpret_cast<float4*>(y + (size_t)r * D + c) = v;
    }
}

// a += b  (gradient accumulation for tensors consumed by several ops)
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n4, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 x = reinterpret_cast<float4*>(a)[i];
        float4 y = reinterpret_cast<const float4*>(b)[i];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        reinterpret_cast<float4*>(a)[i] = x;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) a[i] += b[i];
}

// dst = a + b (b may be null: dst = a): two gradient contributions of one tensor gathered into a row block of a larger buffer in one pass
__global__ __launch_bounds__(256) void sum2_kernel(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b, size_t n4, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 x = reinterpret_cast<const float4*>(a)[i];
        if (b) { const float4 y = reinterpret_cast<const float4*>(b)[i]; x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; }
        reinterpret_cast<float4*>(dst)[i] = x;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = a[i] + (b ? b[i] : 0.f);
}

// a *= alpha (gradient averaging after the data-parallel all-reduce)
__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ a, float alpha, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] *= alpha;
}

// dst[r, c] += src[r, c] for c < cols with independent row strides (merging gradient pieces of a padded buffer)
__global__ __launch_bounds__(256) void add_strided_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
        dst[(size_t)r * ldd + c] += src[(size_t)r * lds + c];
    }
}

// ------------------------------------------------------------------------------------------------------------
// SpecAugment as the reference implements it (src/utils.py:51-75): per sample, two random TIME spans (widths
// f ~ U{0..19}, t ~ U{0..99}) are overwritten with that sample's mean over the whole padded [T,M] slab.
// One block per sample; randomness from the counter RNG (seed, stream, sample).
// ------------------------------------------------------------------------------------------------------------
// Pass 1: per-sample sum over the padded [T,M] slab (several workgroups per sample, one atomic each).
__global__ __launch_bounds__(256) void specaugment_sum_kernel(const float* __restrict__ mel, float* __restrict__ sums, int n) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* src = mel + (size_t)b * n;
    float s = 0.f;
    if ((n & 3) == 0) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += gridDim.x * 256) {
            const float4 v = reinterpret_cast<const float4*>(src)[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += src[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + b, (red[0] + red[1]) + (red[2] + red[3]));
}

// Pass 2: copy with the two row spans replaced by the sample mean.
__global__ __launch_bounds__(256) void specaugment_kernel(const float* __restrict__ mel, const int* __restrict__ lens, const float* __restrict__ sums,
                                                          float* __restrict__ out, int T, int M, int freq_mask, int time_mask, uint32_t seed, uint32_t stream) {
    const int b = blockIdx.y;
    const float* src = mel + (size_t)b * T * M;
    float* dst = out + (size_t)b * T * M;
    const int n = T * M;
    const float mean = sums[b] / (float)n;
    const int len = lens[b];
    const uint32_t key = rng_row_key(seed, stream, (uint32_t)b);
    int f = (int)(rng_u32(key, 0) % (uint32_t)freq_mask);
    int t = (int)(rng_u32(key, 1) % (uint32_t)time_mask);
    if (len - t <= 0) t = (len / 2 > 0) ? (int)(rng_u32(key, 2) % (uint32_t)(len / 2)) : 0;
    if (f >= len) f = len - 1 > 0 ? len - 1 : 0;
    const int f0 = (int)(rng_u32(key, 3) % (uint32_t)max(len - f, 1));
    const int t0 = (int)(rng_u32(key, 4) % (uint32_t)max(len - t, 1));
    if ((M & 3) == 0) {
        const int mq = M >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += gridDim.x * 256) {
            const int row = i / mq;
            const bool masked = (row >= f0 && row < f0 + f) || (row >= t0 && row < t0 + t);
            reinterpret_cast<float4*>(dst)[i] = masked ? make_float4(mean, mean, mean, mean) : reinterpret_cast<const float4*>(src)[i];
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const int row = i / M;
            const bool masked = (row >= f0 && row < f0 + f) || (row >= t0 && row < t0 + t);
            dst[i] = masked ? mean : src[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// discriminator_shuffle_batch (src/train.py:296-329): rows of [text hidden (B,Tt,D); speech hidden (B,Tm,D)] are
// zero-padded to Tmax, concatenated on batch and permuted: out[i] = src[perm[i]]; out_len[i] = len[perm[i]].
// Backward scatters the gradient of row i back to source row perm[i] (only the un-padded part exists).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void disc_gather_kernel(const float* __restrict__ th, const float* __restrict__ sh, const int* __restrict__ tlen,
                                                          const int* __restrict__ slen, const int64_t* __restrict__ perm, float* __restrict__ out,
                                                          int* __restrict__ out_len, int B, int Tt, int Ts, int Tmax, int D) {
    const int i = blockIdx.y;                    // output row
    const int src = (int)perm[i];
    const bool is_text = src < B;
    const int sb = is_text ? src : src - B;
    const int Tsrc = is_text ? Tt : Ts;
    const float* base = (is_text ? th : sh) + (size_t)sb * Tsrc * D;
    if (blockIdx.x == 0 && threadIdx.x == 0) out_len[i] = is_text ? tlen[sb] : slen[sb];
    const int dq = D >> 2;
    const size_t total = (size_t)Tmax * dq;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (size_t)gridDim.x * 256) {
        const int t = (int)(k / dq), c = (int)(k - (size_t)t * dq) * 4;
        float4 v = make_float4(0, 0, 0, 0);
        if (t < Tsrc) v = *reinterpret_cast<const float4*>(base + (size_t)t * D + c);
        *reinterpret_cast<float4*>(out + ((size_t)i * Tmax + t) * D + c) = v;
    }
}

__global__ __launch_bounds__(256) void disc_scatter_kernel(const float* __restrict__ dout, const int64_t* __restrict__ perm, float* __restrict__ dth,
                                                           float* __restrict__ dsh, int B, int Tt, int Ts, int Tmax, int D) {
    const int i = blockIdx.y;
    const int src = (int)perm[i];
    const bool is_text = src < B;
    const int sb = is_text ? src : src - B;
    const int Tsrc = is_text ? Tt : Ts;
    float* base = (is_text ? dth : dsh) + (size_t)sb * Tsrc * D;
    const int dq = D >> 2;
    const size_t total = (size_t)Tsrc * dq;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (size_t)gridDim.x * 256) {
        const int t = (int)(k / dq), c = (int)(k - (size_t)t * dq) * 4;
        *reinterpret_cast<float4*>(base + (size_t)t * D + c) = *reinterpret_cast<const float4*>(dout + ((size_t)i * Tmax + t) * D + c);
    }
}

// Row argmax (first maximum, as torch.argmax) for the autoregressive text decoder (src/network.py:466); one thread per row.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ld, int rows, int cols, int64_t* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * ld;
    float best = xr[0];
    int bi = 0;
    for (int c = 1; c < cols; ++c) {
        const float v = xr[c];
        if (v > best) { best = v; bi = c; }
    }
    out[r] = bi;
}

// x[b, t, :] = 0 for t >= lens[b]  (pad masking of generated sequences, src/network.py:245-251, 476-480)
__global__ __launch_bounds__(256) void mask_by_len_kernel(float* __restrict__ x, const int64_t* __restrict__ lens, int T, int D, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / D;
        const int b = (int)(row / T), t = (int)(row - (size_t)b * T);
        if (t >= lens[b]) x[i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" int unast_argmax_rows(const float* x, int ld, int rows, int cols, int64_t* out, hipStream_t stream) {
    UNAST_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "unast_argmax_rows: bad arguments");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, x, ld, rows, cols, out);
    return unast_check_launch("unast_argmax_rows");
}

extern "C" int unast_mask_by_len(float* x, const int64_t* lens, int B, int T, int D, hipStream_t stream) {
    UNAST_REQUIRE(x && lens && B > 0 && T > 0 && D > 0, "unast_mask_by_len: bad arguments");
    const size_t total = (size_t)B * T * D;
    hipLaunchKernelGGL(mask_by_len_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, x, lens, T, D, total);
    return unast_check_launch("unast_mask_by_len");
}

extern "C" int unast_embed_fwd(const int64_t* ids, const float* E, float* out, int rows, int T, int D, int shift_sos,
                               float drop_p, unsigned int seed, unsigned int stream_id, float noise_p, unsigned int noise_stream,
                               hipStream_t stream) {
    UNAST_REQUIRE(ids && E && out && rows > 0 && T > 0 && (D & 3) == 0, "unast_embed_fwd: bad arguments");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(ew_grid((size_t)rows * (D / 4))), dim3(256), 0, stream, ids, E, out, rows, T, D, shift_sos,
                       drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id, drop_threshold(noise_p), noise_stream);
    return unast_check_launch("unast_embed_fwd");
}

// The same gradient with a FIXED summation order (unast_embed_bwd_det; parity / reproducibility mode): one workgroup per vocabulary id
// walks all rows in order, thread = column, no atomics.  (The kernel above collects its hits in arrival order and adds chunk sums with
// atomics: the result depends on scheduling in its last bits.)
__global__ __launch_bounds__(256) void embed_bwd_det_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                            float* __restrict__ dE, int rows, int T, int D, int shift_sos, int padding_idx,
                                                            uint32_t drop_thresh, float drop_scale, uint32_t seed, uint32_t stream,
                                                            uint32_t noise_thresh, uint32_t noise_stream) {
    const int v = blockIdx.x;
    if (v == padding_idx) return;
    for (int c0 = 0; c0 < D; c0 += 256) {
        const int c = c0 + threadIdx.x;
        float acc = 0.f;
        for (int r = 0; r < rows; ++r) {
            int64_t id;
            if (shift_sos >= 0) { const int t = r % T; id = (t == 0) ? (int64_t)shift_sos : ids[r - 1]; }
            else id = ids[r];
            if (id != v || c >= D) continue;                       // (the first test is uniform)
            float g = dout[(size_t)r * D + c];
            if (drop_thresh) g = rng_keep(rng_row_key(seed, stream, (uint32_t)r), c, drop_thresh) ? g * drop_scale : 0.f;
            if (noise_thresh && !rng_keep(rng_row_key(seed, noise_stream, (uint32_t)r), 0u, noise_thresh)) g = 0.f;
            acc += g;
        }
        if (c < D) dE[(size_t)v * D + c] += acc;
    }
}

extern "C" int unast_embed_bwd_det(const int64_t* ids, const float* dout, float* dE, int rows, int T, int D, int vocab, int shift_sos,
                                   int padding_idx, float drop_p, unsigned int seed, unsigned int stream_id, float noise_p,
                                   unsigned int noise_stream, hipStream_t stream) {
    UNAST_REQUIRE(ids && dout && dE && rows > 0 && T > 0 && vocab > 0, "unast_embed_bwd_det: bad arguments");
    hipLaunchKernelGGL(embed_bwd_det_kernel, dim3(vocab), dim3(256), 0, stream, ids, dout, dE, rows, T, D, shift_sos, padding_idx,
                       drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id, drop_threshold(noise_p), noise_stream);
    return unast_check_launch("unast_embed_bwd_det");
}

extern "C" int unast_embed_bwd(const int64_t* ids, const float* dout, float* dE, int rows, int T, int D, int vocab, int shift_sos,
                               int padding_idx, float drop_p, unsigned int seed, unsigned int stream_id, float noise_p,
                               unsigned int noise_stream, hipStream_t stream) {
    UNAST_REQUIRE(ids && dout && dE && rows > 0 && T > 0 && vocab > 0, "unast_embed_bwd: bad arguments");
    const int rpc = 256;                       // <= 512 hits per (vocab row, chunk) by construction
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(vocab, (rows + rpc - 1) / rpc), dim3(256), 0, stream, ids, dout, dE, rows, T, D, shift_sos, padding_idx, rpc,
                       drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id, drop_threshold(noise_p), noise_stream);
    return unast_check_launch("unast_embed_bwd");
}

extern "C" int unast_posenc_fwd(const float* x, const float* pe, float* y, int rows, int T, int D, float scale, float drop_p,
                                unsigned int seed, unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(x && pe && y && rows > 0 && T > 0 && (D & 3) == 0, "unast_posenc_fwd: bad arguments");
    hipLaunchKernelGGL(posenc_fwd_kernel, dim3(ew_grid((size_t)rows * (D / 4))), dim3(256), 0, stream, x, pe, y, rows, T, D, scale,
                       drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id);
    return unast_check_launch("unast_posenc_fwd");
}

extern "C" int unast_posenc_bwd(const float* dy, const float* gate, float* dx, int rows, int D, float scale, float drop_p,
                                unsigned int seed, unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(dy && dx && rows > 0 && (D & 3) == 0, "unast_posenc_bwd: bad arguments");
    hipLaunchKernelGGL(posenc_bwd_kernel, dim3(ew_grid((size_t)rows * (D / 4))), dim3(256), 0, stream, dy, gate, dx, rows, D, scale,
                       drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, stream_id);
    return unast_check_launch("unast_posenc_bwd");
}

extern "C" int unast_rowmask(const float* x, float* y, int rows, int D, float p, unsigned int seed, unsigned int stream_id,
                             hipStream_t stream) {
    UNAST_REQUIRE(x && y && rows > 0 && (D & 3) == 0, "unast_rowmask: bad arguments");
    hipLaunchKernelGGL(rowmask_kernel, dim3(ew_grid((size_t)rows * (D / 4))), dim3(256), 0, stream, x, y, rows, D, drop_threshold(p), seed, stream_id);
    return unast_check_launch("unast_rowmask");
}

extern "C" int unast_add_inplace(float* a, const float* b, int64_t n, hipStream_t stream) {
    UNAST_REQUIRE(a && b && n > 0, "unast_add_inplace: bad arguments");
    UNAST_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "unast_add_inplace: operands must be 16-byte aligned");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(ew_grid((size_t)n / 4 + 1)), dim3(256), 0, stream, a, b, (size_t)n / 4, (size_t)n);
    return unast_check_launch("unast_add_inplace");
}

extern "C" int unast_sum2(float* dst, const float* a, const float* b, int64_t n, hipStream_t stream) {
    UNAST_REQUIRE(dst && a && n > 0, "unast_sum2: bad arguments");
    UNAST_REQUIRE((((uintptr_t)dst | (uintptr_t)a | (uintptr_t)b) & 15) == 0, "unast_sum2: operands must be 16-byte aligned");
    hipLaunchKernelGGL(sum2_kernel, dim3(ew_grid((size_t)n / 4 + 1)), dim3(256), 0, stream, dst, a, b, (size_t)n / 4, (size_t)n);
    return unast_check_launch("unast_sum2");
}

extern "C" int unast_scale_inplace(float* a, float alpha, int64_t n, hipStream_t stream) {
    UNAST_REQUIRE(a && n > 0, "unast_scale_inplace: bad arguments");
    hipLaunchKernelGGL(scale_inplace_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, stream, a, alpha, (size_t)n);
    return unast_check_launch("unast_scale_inplace");
}

// The teacher-forced decoder input of SpeechTransformer.decode_sequence (src/network.py:254-262): dst[b, 0, :] = 0 (the "go" frame),
// dst[b, t, :] = src[b, t - 1, :].  n4 = T * M / 4 float4 per sequence.
__global__ __launch_bounds__(256) void shift_frames_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int B, size_t n4, int m4) {
    const size_t total = (size_t)B * n4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t in = i % n4;
        dst[i] = in < (size_t)m4 ? make_float4(0.f, 0.f, 0.f, 0.f) : src[i - m4];
    }
}

extern "C" int unast_shift_frames(const float* src, float* dst, int B, int T, int M, hipStream_t stream) {
    UNAST_REQUIRE(src && dst && B > 0 && T > 0 && M > 0 && (M & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0,
                  "unast_shift_frames: needs M %% 4 == 0 and 16-byte aligned buffers");
    const size_t n4 = (size_t)T * (M / 4);
    hipLaunchKernelGGL(shift_frames_kernel, dim3(ew_grid((size_t)B * n4)), dim3(256), 0, stream, (const float4*)src, (float4*)dst, B, n4, M / 4);
    return unast_check_launch("unast_shift_frames");
}

// Test infrastructure (unast_amd.config.STREAM_JITTER): one wave that keeps its stream busy for `us` microseconds of the constant
// 100 MHz clock.  Placed at the head of every side-stream call it shifts the streams against each other, so a missing dependency
// (a buffer filled on one stream and read on another without an event in between) changes results instead of hiding behind timing.
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); }
}

extern "C" int unast_spin(int us, hipStream_t stream) {
    UNAST_REQUIRE(us >= 0 && us <= 100000, "unast_spin: 0..100 000 us");
    if (us == 0) return UNAST_OK;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long)us * 100ull);
    return unast_check_launch("unast_spin");
}

extern "C" int unast_add_strided(float* dst, int ldd, const float* src, int lds, int rows, int cols, hipStream_t stream) {
    UNAST_REQUIRE(dst && src && rows > 0 && cols > 0 && ldd >= cols && lds >= cols, "unast_add_strided: bad arguments");
    hipLaunchKernelGGL(add_strided_kernel, dim3(ew_grid((size_t)rows * cols)), dim3(256), 0, stream, dst, ldd, src, lds, rows, cols);
    return unast_check_launch("unast_add_strided");
}

extern "C" int unast_specaugment(const float* mel, const int* lens, float* out, float* ws, int B, int T, int M, int freq_mask, int time_mask,
                                 unsigned int seed, unsigned int stream_id, hipStream_t stream) {
    UNAST_REQUIRE(mel && lens && out && ws && B > 0 && T > 0 && M > 0 && freq_mask > 0 && time_mask > 0, "unast_specaugment: bad arguments");
    UNAST_REQUIRE(((((uintptr_t)mel) | ((uintptr_t)out)) & 15) == 0, "unast_specaugment: buffers must be 16-byte aligned");
    const int n = T * M;
    int chunks = (n / 4 + 255) / 256;
    if (chunks > 16) chunks = 16;
    if (chunks < 1) chunks = 1;
    hipMemsetAsync(ws, 0, sizeof(float) * B, stream);
    hipLaunchKernelGGL(specaugment_sum_kernel, dim3(chunks, B), dim3(256), 0, stream, mel, ws, n);
    hipLaunchKernelGGL(specaugment_kernel, dim3(chunks, B), dim3(256), 0, stream, mel, lens, ws, out, T, M, freq_mask, time_mask, seed, stream_id);
    return unast_check_launch("unast_specaugment");
}

extern "C" int unast_disc_gather(const float* t_hid, const float* s_hid, const int* t_len, const int* s_len, const int64_t* perm,
                                 float* out, int* out_len, int B, int Tt, int Ts, int D, hipStream_t stream) {
    UNAST_REQUIRE(t_hid && s_hid && t_len && s_len && perm && out && out_len && B > 0 && (D & 3) == 0, "unast_disc_gather: bad arguments");
    const int Tmax = Tt > Ts ? Tt : Ts;
    hipLaunchKernelGGL(disc_gather_kernel, dim3(ew_grid((size_t)Tmax * (D / 4), 64), 2 * B), dim3(256), 0, stream, t_hid, s_hid, t_len, s_len,
                       perm, out, out_len, B, Tt, Ts, Tmax, D);
    return unast_check_launch("unast_disc_gather");
}

extern "C" int unast_disc_scatter(const float* dout, const int64_t* perm, float* dt_hid, float* ds_hid, int B, int Tt, int Ts, int D,
                                  hipStream_t stream) {
    UNAST_REQUIRE(dout && perm && dt_hid && ds_hid && B > 0 && (D & 3) == 0, "unast_disc_scatter: bad arguments");
    const int Tmax = Tt > Ts ? Tt : Ts;
    hipLaunchKernelGGL(disc_scatter_kernel, dim3(ew_grid((size_t)Tmax * (D / 4), 64), 2 * B), dim3(256), 0, stream, dout, perm, dt_hid, ds_hid,
                       B, Tt, Ts, Tmax, D);
    return unast_check_launch("unast_disc_scatter");
}

UNAST_DEFINE_RNG_EPOCH_SETTER(elementwise)

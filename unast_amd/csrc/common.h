// Shared device helpers for the UNAST gfx950 kernels (CDNA4 only: wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define UNAST_WAVE 64

// ---------------------------------------------------------------------------------------------
// Status / error reporting (host side, defined in api.cpp)
// ---------------------------------------------------------------------------------------------
enum { UNAST_OK = 0, UNAST_ERR_ARG = -1, UNAST_ERR_LAUNCH = -2, UNAST_ERR_ALIGN = -3 };
int unast_set_error(int code, const char* fmt, ...);
int unast_check_launch(const char* what);

#define UNAST_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return unast_set_error(UNAST_ERR_ARG, __VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Counter-based RNG for dropout / noise masks: a function of (seed, stream, row, col) only, so the
// backward pass regenerates the forward mask from the same four integers in any thread layout.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
// Launches replayed from a captured graph carry the same (seed, stream) every time; a counter in device memory is therefore
// mixed into every stream id.  It is 0 except while a generation graph is being replayed (unast_amd/inference.py advances it
// once per decoded position and zeroes it afterwards), so training kernels -- whose backward must regenerate the forward's
// masks -- always see 0.  One copy of the pointer per translation unit, set together by unast_set_rng_epoch (api.cpp).
static __device__ const uint32_t* g_unast_rng_epoch = nullptr;
__device__ __forceinline__ uint32_t rng_epoch() {
    const uint32_t* p = g_unast_rng_epoch;
    return p ? *p : 0u;
}
// The epoch is hashed on its own before the stream id is added: with `stream + epoch` under one hash, site s at replay e + 1 drew
// exactly the mask site s + 1 had drawn at replay e (sites of a call use consecutive stream ids, replays consecutive epochs).
__device__ __forceinline__ uint32_t rng_stream_base(uint32_t seed, uint32_t stream) {
    return pcg_hash(stream + pcg_hash(rng_epoch() + pcg_hash(seed)));
}
__device__ __forceinline__ uint32_t rng_row_key(uint32_t seed, uint32_t stream, uint32_t row) {
    return pcg_hash(row + rng_stream_base(seed, stream));
}
#define UNAST_DEFINE_RNG_EPOCH_SETTER(tu)                                                                                 \
    extern "C" int unast_tu_##tu##_set_rng_epoch(const unsigned int* p) {                                                 \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_unast_rng_epoch), &p, sizeof(p), 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; \
    }
__device__ __forceinline__ uint32_t rng_u32(uint32_t row_key, uint32_t col) { return pcg_hash(col ^ row_key) ; }
// Keep decisions for dropout: ONE multiply-xorshift hash serves a pair of adjacent columns (16 bits each), keyed by the
// already well-mixed per-row key.  thresh = p * 2^16 (0 => keep everything).  All kernels (GEMM epilogue, LayerNorm
// backward, attention forward/backward, element-wise) use these two helpers, so masks agree across passes.
__device__ __forceinline__ uint32_t rng_pair(uint32_t row_key, uint32_t col) {
    uint32_t h = ((col >> 1) ^ row_key) * 0x9E3779B1u;
    return h ^ (h >> 15);
}
__device__ __forceinline__ bool rng_keep_lo(uint32_t pair_hash, uint32_t thresh) { return (pair_hash & 0xFFFFu) >= thresh; }
__device__ __forceinline__ bool rng_keep_hi(uint32_t pair_hash, uint32_t thresh) { return (pair_hash >> 16) >= thresh; }
__device__ __forceinline__ bool rng_keep(uint32_t row_key, uint32_t col, uint32_t thresh) {
    const uint32_t h = rng_pair(row_key, col);
    return ((col & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thresh;
}
static inline uint32_t drop_threshold(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}

// ---------------------------------------------------------------------------------------------
// fp32 -> bf16 operand preparation for MFMA.
//   NSPLIT==1: one bf16 (round to nearest even).
//   NSPLIT==3: x = hi + lo with hi = RNE_bf16(x), lo = RNE_bf16(x - hi); the product
//              a*b is then formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (fp32 accumulate), which
//              keeps ~16 mantissa bits per operand.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_bf16_rne(float a, float b) {
    f32x2 v = {a, b};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pack_bf16_trunc(float a, float b) {
    return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
}
template <int NSPLIT>
__device__ __forceinline__ void split4(const float4& v, u32x2& hi, u32x2& lo) {
#ifdef UNAST_EXP_FAKE_SPLIT      // timing experiment only: no conversion work (wrong numerics)
    hi[0] = __float_as_uint(v.x); hi[1] = __float_as_uint(v.y); lo[0] = __float_as_uint(v.z); lo[1] = __float_as_uint(v.w);
    return;
#endif
    hi[0] = pack_bf16_rne(v.x, v.y);
    hi[1] = pack_bf16_rne(v.z, v.w);
    if (NSPLIT == 1) {
        lo[0] = 0; lo[1] = 0;
    } else {        // residuals against the ROUNDED hi parts (zero-mean error; exact in fp32), rounded to bf16 in turn
        const float rx = v.x - __uint_as_float(hi[0] << 16);
        const float ry = v.y - __uint_as_float(hi[0] & 0xFFFF0000u);
        const float rz = v.z - __uint_as_float(hi[1] << 16);
        const float rw = v.w - __uint_as_float(hi[1] & 0xFFFF0000u);
        lo[0] = pack_bf16_rne(rx, ry);
        lo[1] = pack_bf16_rne(rz, rw);
    }
}
// Pre-split operand chunk of 4 consecutive fp32 values: [hi0 hi1 | hi2 hi3 | lo0 lo1 | lo2 lo3] (bf16 pairs).  A tensor kept
// in this format has the byte offsets of its fp32 original, so a GEMM loader reads it with the same addressing.
__device__ __forceinline__ uint4 split_chunk(const float4& v) {
    u32x2 hi, lo;
    split4<3>(v, hi, lo);
    return make_uint4(hi[0], hi[1], lo[0], lo[1]);
}
__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float(((uint32_t)b) << 16); }

__device__ __forceinline__ f32x4 mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Transposed LDS read (gfx950 ds_read_b64_tr_b16): within each group of 16 lanes, lane 4q+p supplies the
// address of row q, columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i of the
// four rows (row q in element q).  EXEC must be all ones.
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds_addr));
}

// ---------------------------------------------------------------------------------------------
// Wave / block reductions
// ---------------------------------------------------------------------------------------------
// Reductions over the four 16-lane rows of a wave (lanes that differ in bits 4 and 5) without the LDS crossbar: gfx950's
// v_permlane16_swap exchanges the odd rows of one register with the even rows of another, v_permlane32_swap the upper half of one with
// the lower half of the other -- fed the same value twice they return {partner's value, own value} pairs, i.e. an xor-16 / xor-32
// butterfly in two vector instructions instead of a ds_bpermute round trip (~100 cycles of latency on the softmax's dependent chain).
__device__ __forceinline__ float rows4_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float x = __uint_as_float(a[0]), y = __uint_as_float(a[1]);
    asm("v_max_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));                 // (fmaxf would add a canonicalising v_max per operand)
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float z = __uint_as_float(b[0]), w = __uint_as_float(b[1]);
    asm("v_max_f32 %0, %1, %2" : "=v"(z) : "v"(z), "v"(w));
    return z;
}
__device__ __forceinline__ float rows4_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

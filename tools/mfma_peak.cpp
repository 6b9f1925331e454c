// Sustained v_mfma_f32_16x16x32_bf16 rate of this MI355X, with the data pattern and occupancy of the GEMM kernels:
// hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.cpp -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_loop(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
    bf16x8_t a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8_t, in[(threadIdx.x * 8 + i) % 4096]);
        b[i] = __builtin_bit_cast(bf16x8_t, in[(threadIdx.x * 8 + 4 + i) % 4096]);
    }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + r) & 3], b[i & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<unsigned> h(4096 * 4);
    srand(1);
    for (auto& v : h) {   // random bf16 pairs with moderate exponents (like activations), not zeros
        unsigned short x = (unsigned short)(0x3C00 + (rand() % 0x0600)) | ((rand() & 1) << 15);
        unsigned short y = (unsigned short)(0x3C00 + (rand() % 0x0600)) | ((rand() & 1) << 15);
        v = x | ((unsigned)y << 16);
    }
    u32x4* din; float* dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 1024 * 512 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks : {256, 512}) {
        for (int threads : {256, 512}) {
            hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(threads), 0, 0, din, dout, 2000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mfmas_per_wave = (double)iters * 24;
            double waves_per_simd = (double)blocks * (threads / 64) / 1024.0;
            double ns_per_mfma_simd = ms * 1e6 / (mfmas_per_wave * waves_per_simd);
            double tflops = (double)blocks * (threads / 64) * mfmas_per_wave * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12;
            printf("blocks %d x %d threads (%.1f waves/SIMD): %.2f ms  %.2f ns per MFMA per SIMD  %.0f TFLOP/s (MFMA rate)\n", blocks, threads, waves_per_simd, ms, ns_per_mfma_simd, tflops);
        }
    }
    return 0;
}

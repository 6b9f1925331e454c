// Does gfx950 drop raw-buffer stores that lie beyond the descriptor's num_records?  (DESIGN 5c-8 claimed "8-byte stores are not dropped,
// 16-byte ones are" after the keep-bit overrun of round 3; ADVICE round 3 asked for the real cause.)  One wave stores 4-, 8- and 16-byte
// values at offsets straddling num_records of a descriptor that covers the first half of a buffer; the host prints which bytes changed.
// Build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 tools/buffer_oob_probe.cpp -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int W>
__global__ void probe(unsigned char* buf, int num_records, int voff0, int use_soffset) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, num_records, 0x00020000);
    const int lane = threadIdx.x;
    int voff = voff0 + lane * W, soff = 0;
    if (use_soffset) { soff = voff0; voff = lane * W; }          // the wave-uniform part in the scalar offset (not range-checked on gfx9)
    if (W == 4) __builtin_amdgcn_raw_buffer_store_b32(0xA5A5A5A5u, r, voff, soff, 0);
    if (W == 8) __builtin_amdgcn_raw_buffer_store_b64((u32x2){0xA5A5A5A5u, 0xA5A5A5A5u}, r, voff, soff, 0);
    if (W == 16) __builtin_amdgcn_raw_buffer_store_b128((u32x4){0xA5A5A5A5u, 0xA5A5A5A5u, 0xA5A5A5A5u, 0xA5A5A5A5u}, r, voff, soff, 0);
}

template <int W>
static void run(unsigned char* d, int bytes, int num_records, int voff0, int use_soffset) {
    hipMemset(d, 0, bytes);
    hipLaunchKernelGGL(probe<W>, dim3(1), dim3(64), 0, 0, d, num_records, voff0, use_soffset);
    std::vector<unsigned char> h(bytes);
    hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
    int first = -1, last = -1, n = 0, beyond = 0;
    for (int i = 0; i < bytes; ++i) if (h[i]) { if (first < 0) first = i; last = i; ++n; if (i >= num_records) ++beyond; }
    printf("  %2d-byte stores, lanes cover [%d, %d), num_records %d, %s: bytes written %d (first %d, last %d), of them beyond num_records: %d\n",
           W, voff0, voff0 + 64 * W, num_records, use_soffset ? "base in soffset" : "base in voffset", n, first, last, beyond);
}

int main() {
    const int bytes = 8192, nr = 2048;
    unsigned char* d;
    hipMalloc(&d, bytes);
    for (int use_s = 0; use_s < 2; ++use_s) {
        printf("%s\n", use_s ? "wave-uniform base in the scalar offset:" : "whole offset in the vector offset:");
        run<4>(d, bytes, nr, nr - 128, use_s);          // half of the lanes in range, half beyond
        run<8>(d, bytes, nr, nr - 256, use_s);
        run<16>(d, bytes, nr, nr - 512, use_s);
        run<8>(d, bytes, nr - 4, nr - 256, use_s);      // num_records cuts an 8-byte store in the middle
        run<16>(d, bytes, nr - 8, nr - 512, use_s);     // ... and a 16-byte one
    }
    hipFree(d);
    return 0;
}

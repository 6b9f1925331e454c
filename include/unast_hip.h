/* unast_hip.h — C ABI of libunast_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the UNAST
 * adversarial speech/text train-step hot path.
 *
 * The reference (Lucaskabela/UNAST) has no FFI: its arithmetic is reached through torch.nn modules and
 * torch.nn.functional calls.  Each entry point below therefore cites the reference call site (file:line in
 * /root/reference) whose arithmetic it replaces.  Conventions:
 *   - every function returns 0 on success or a negative status; unast_last_error() gives the message;
 *   - the library never allocates, frees or synchronises: all pointers are caller-owned DEVICE pointers,
 *     borrowed for the duration of the work enqueued on `stream`;
 *   - activations are fp32, batch-first, row-major [B*T, C]; `lens` are int32 device arrays [B];
 *   - nsplit selects the MFMA operand precision: 1 = bf16 operands, 3 = split-bf16 (hi/lo) operands,
 *     both with fp32 accumulation;
 *   - dropout/noise masks are a pure function of (seed, stream_id, row, col) so forward and backward agree.
 */
#ifndef UNAST_HIP_H
#define UNAST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

const char* unast_last_error(void);
int unast_version(void);
const char* unast_arch(void);

/* Dense contraction C[M,N] = epi(alpha * A*B^T) on the MFMA pipe.  Replaces torch.nn.Linear / nn.Conv1d /
 * in-/out-projection GEMMs and their autograd backward: src/module.py:31,63 (Linear/Conv wrappers),
 * src/module.py:273-274,286-287 (torch Transformer layers), src/module.py:152-153,240 (heads),
 * src/module.py:306-310 (LSTM input projections, reduce_h_W), src/network.py:182 (fc2).
 * a_mode/b_mode: 0 K-contiguous, 1 K-contiguous conv gather (A only), 2 row-contiguous,
 *                3 conv-dgrad weights (B only), 4 conv-wgrad gather (B only).
 * Epilogue: x = alpha*acc (+bias[n]) -> relu if act==1 -> dropout(drop_p) -> gate (G>0 ? x*gate_scale : 0)
 *           -> + R[m,n] -> (+C if beta) ; split-K (splitk>1) accumulates into C with fp32 atomics. */
int unast_gemm(int a_mode, int b_mode, int nsplit,
               const float* A, int lda, const float* B, int ldb, float* C, int ldc,
               int M, int N, int K,
               int conv_T, int conv_ca, int conv_cb, int conv_shift,
               const float* bias, const float* R, int ldr, const float* G, int ldg, float gate_scale,
               float alpha, int beta, int act,
               float drop_p, unsigned int seed, unsigned int stream_id,
               int splitk, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif

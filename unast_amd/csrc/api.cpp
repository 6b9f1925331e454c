// Host-side glue of the C ABI: error reporting, version/arch query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "common.h"
#include "../../include/unast_hip.h"

static thread_local char g_err[512] = "";

int unast_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int unast_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return unast_set_error(UNAST_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return UNAST_OK;
}

extern "C" const char* unast_last_error(void) { return g_err; }
extern "C" int unast_version(void) { return 100; }
extern "C" const char* unast_arch(void) { return "gfx950"; }

// One pointer copy per translation unit that draws dropout / noise masks (common.h, rng_epoch).
extern "C" int unast_tu_attention_set_rng_epoch(const unsigned int*);
extern "C" int unast_tu_decode_set_rng_epoch(const unsigned int*);
extern "C" int unast_tu_elementwise_set_rng_epoch(const unsigned int*);
extern "C" int unast_tu_gemm_set_rng_epoch(const unsigned int*);
extern "C" int unast_tu_loss_set_rng_epoch(const unsigned int*);
extern "C" int unast_tu_lstm_set_rng_epoch(const unsigned int*);
extern "C" int unast_tu_norm_set_rng_epoch(const unsigned int*);
extern "C" int unast_tu_panel_set_rng_epoch(const unsigned int*);
extern "C" int unast_set_rng_epoch(const unsigned int* counter) {
    int rc = unast_tu_attention_set_rng_epoch(counter) | unast_tu_decode_set_rng_epoch(counter) | unast_tu_elementwise_set_rng_epoch(counter) | unast_tu_gemm_set_rng_epoch(counter) |
             unast_tu_loss_set_rng_epoch(counter) | unast_tu_lstm_set_rng_epoch(counter) | unast_tu_norm_set_rng_epoch(counter) | unast_tu_panel_set_rng_epoch(counter);
    return rc ? unast_set_error(UNAST_ERR_LAUNCH, "unast_set_rng_epoch: hipMemcpyToSymbol failed") : UNAST_OK;
}

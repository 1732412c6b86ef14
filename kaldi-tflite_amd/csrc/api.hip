// Error string + version of libktf_hip (the only state the library keeps: thread-local).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ktf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int32_t ktf_version(void) { return 114; /* 0.1.1 + KTF_TDNN_MX_SLAB, KTF_ACT_ELU .. KTF_ACT_SOFTMAX, ktf_activation_f32, KTF_BF16P / KTF_GEMM_BF16X4, ktf_tdnn_split_flat, ktf_tdnn_out_lens */ }

extern "C" size_t ktf_last_error(char* buf, size_t cap) {
    const size_t n = strlen(g_err);
    if (buf && cap > 0) {
        const size_t c = n < cap - 1 ? n : cap - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return n;
}

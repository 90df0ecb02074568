// api.hip -- version / error reporting entry points of libhsp.so.
#include "common.h"

namespace hsp {
static thread_local char g_last_hip_error[256] = "";
void set_last_hip_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    int i = 0;
    for (; s && s[i] && i < 255; ++i) g_last_hip_error[i] = s[i];
    g_last_hip_error[i] = 0;
}
}  // namespace hsp

extern "C" int hsp_version(void) { return 100; }   // 0.1.0

extern "C" const char* hsp_error_string(int code) {
    switch (code) {
        case HSP_OK: return "ok";
        case HSP_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or k out of range)";
        case HSP_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case HSP_ERR_WORKSPACE: return "workspace missing or too small";
        case HSP_ERR_LAUNCH: return "HIP launch failed (see hsp_last_hip_error)";
        default: return "unknown error";
    }
}

extern "C" const char* hsp_last_hip_error(void) { return hsp::g_last_hip_error; }

// Library-level entry points of libatx: version, error strings, device probe.
#include "atx_common.hpp"

#include <string>

namespace atx {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return ATX_OK;
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return ATX_EHIP;
}

}  // namespace atx

extern "C" int atx_version(void) { return ATX_VERSION; }

extern "C" const char* atx_last_error(void) { return atx::g_last_error.c_str(); }

extern "C" const char* atx_strerror(int code) {
    switch (code) {
        case ATX_OK: return "ok";
        case ATX_EINVAL: return "invalid argument";
        case ATX_ESHAPE: return "shape mismatch";
        case ATX_ENOTIMPL: return "not implemented";
        case ATX_EHIP: return "HIP runtime error";
        case ATX_EALIGN: return "alignment requirement not met";
        case ATX_EWORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

extern "C" int atx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        // no device / no driver is an answer, not a failure of the probe
        (void)hipGetLastError();
        if (e == hipErrorNoDevice) return 0;
        return atx::hip_status(e, "hipGetDeviceCount");
    }
    return n;
}

// Library-level entry points of libatx: version, error strings, device probe.
#include "atx_common.hpp"

#include <cstring>
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

extern "C" int64_t atx_vector_program(const atx_level_op* prog, int32_t n_stage, int64_t n_lev, int dtype, atx_level_op* out) {
    if (!prog || n_stage < 1 || n_stage > 8 || n_lev < 1 || (dtype != ATX_F32 && dtype != ATX_F64)) {
        atx::set_error("atx_vector_program: bad arguments (n_stage=%d, n_lev=%lld, dtype=%d)", n_stage, (long long)n_lev, dtype);
        return ATX_EINVAL;
    }
    const int64_t V = dtype == ATX_F32 ? 4 : 2;
    const int64_t C = (n_lev + V - 1) / V;
    if (!out) return (int64_t)n_stage * C;
    // the same rule as the kernels' own table build (build_vector_ops): equality of op, use_mask and of the parameters AS
    // THE KERNEL WILL SEE THEM, i.e. after rounding to the stack's arithmetic type
    auto same_param = [dtype](double a, double b) {
        if (dtype == ATX_F32) {
            const float fa = (float)a, fb = (float)b;
            return std::memcmp(&fa, &fb, sizeof(float)) == 0;
        }
        return std::memcmp(&a, &b, sizeof(double)) == 0;
    };
    for (int32_t s = 0; s < n_stage; ++s) {
        for (int64_t c = 0; c < C; ++c) {
            const atx_level_op* first = prog + (int64_t)s * n_lev + c * V;
            atx_level_op o = *first;
            bool mixed = false;
            int any_mask = first->use_mask;
            for (int64_t e = 1; e < V && c * V + e < n_lev; ++e) {
                const atx_level_op& q = first[e];
                mixed = mixed || q.op != o.op || q.use_mask != o.use_mask || !same_param(q.p0, o.p0) || !same_param(q.p1, o.p1);
                any_mask |= q.use_mask;
            }
            if (mixed) {  // the kernel goes level by level through `prog`; it only needs to know whether any level wants the mask
                o.op = ATX_OP_MIXED;
                o.use_mask = any_mask;
            }
            out[(int64_t)s * C + c] = o;
        }
    }
    return (int64_t)n_stage * C;
}

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
        case ATX_ECOMM: return "RCCL unavailable or collective failed";
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

extern "C" int64_t atx_vector_program(const atx_level_op* prog, int32_t n_stage, int64_t n_lev, int dtype, atx_level_op* out, int64_t out_entries) {
    if (!prog || n_stage < 1 || n_stage > 8 || n_lev < 1 || (dtype != ATX_F32 && dtype != ATX_F64)) {
        atx::set_error("atx_vector_program: bad arguments (n_stage=%d, n_lev=%lld, dtype=%d)", n_stage, (long long)n_lev, dtype);
        return ATX_EINVAL;
    }
    const int64_t V = dtype == ATX_F32 ? 4 : 2;
    const int64_t C = (n_lev + V - 1) / V;
    const atx::LevelTables lay = atx::level_tables_layout(n_stage, n_lev, dtype);
    const int64_t n_entries = (lay.total_bytes + (int64_t)sizeof(atx_level_op) - 1) / (int64_t)sizeof(atx_level_op);
    if (!out) return n_entries;
    if (out_entries < n_entries) {  // never write past what the caller says it allocated (0.3 had no capacity argument)
        atx::set_error("atx_vector_program: out holds %lld entries, the table needs %lld (size it with the out == NULL query)",
                       (long long)out_entries, (long long)n_entries);
        return ATX_EWORKSPACE;
    }
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
    // second part: every level's operator in the stack's type (atx_common.hpp: level_tables_layout)
    unsigned char* base = reinterpret_cast<unsigned char*>(out);
    std::memset(base + (int64_t)n_stage * C * (int64_t)sizeof(atx_level_op), 0,
                (size_t)(n_entries * (int64_t)sizeof(atx_level_op) - (int64_t)n_stage * C * (int64_t)sizeof(atx_level_op)));
    const int64_t B = dtype == ATX_F32 ? 4 : 8;
    unsigned char* p0 = base + lay.levels_offset;
    unsigned char* p1 = p0 + (int64_t)n_stage * lay.Lp * B;
    unsigned char* code = p1 + (int64_t)n_stage * lay.Lp * B;
    for (int32_t s = 0; s < n_stage; ++s) {
        for (int64_t l = 0; l < lay.Lp; ++l) {
            const atx_level_op& o = prog[(int64_t)s * n_lev + (l < n_lev ? l : n_lev - 1)];
            const int64_t i = (int64_t)s * lay.Lp + l;
            if (dtype == ATX_F32) {
                const float a = (float)o.p0, b = (float)o.p1;
                std::memcpy(p0 + i * B, &a, sizeof a);
                std::memcpy(p1 + i * B, &b, sizeof b);
            } else {
                std::memcpy(p0 + i * B, &o.p0, sizeof(double));
                std::memcpy(p1 + i * B, &o.p1, sizeof(double));
            }
            code[i] = (unsigned char)((o.op & 0x7f) | (o.use_mask ? 0x80 : 0));
        }
    }
    return n_entries;
}

namespace atx {
struct alignas(16) Bytes16 {
    uint32_t w[4];
};
__global__ void __launch_bounds__(kBlock) stream_copy_kernel(const Bytes16* __restrict__ src, Bytes16* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
}  // namespace atx

extern "C" int atx_stream_copy(const void* src, void* dst, int64_t n_bytes, void* stream) {
    using namespace atx;
    ATX_REQUIRE((src && dst) || n_bytes == 0, ATX_EINVAL, "atx_stream_copy: null pointer");
    ATX_REQUIRE(n_bytes >= 0 && n_bytes % 16 == 0, ATX_EINVAL, "atx_stream_copy: n_bytes=%lld is not a multiple of 16", (long long)n_bytes);
    ATX_REQUIRE(aligned16(src) && aligned16(dst), ATX_EALIGN, "atx_stream_copy: pointers must be 16-byte aligned");
    const int64_t n = n_bytes / 16;
    if (n == 0) return ATX_OK;
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    ATX_REQUIRE(blocks <= 0x7fffffffll, ATX_ENOTIMPL, "atx_stream_copy: %lld bytes exceed one launch", (long long)n_bytes);
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const Bytes16*>(src), static_cast<Bytes16*>(dst), n);
    ATX_LAUNCH_CHECK("stream_copy");
    return ATX_OK;
}

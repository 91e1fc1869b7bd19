// Shared host/device helpers of libatx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "../../include/atx.h"

namespace atx {

// ---- host-side error plumbing ------------------------------------------------
void set_error(const char* fmt, ...);  // defined in atx_api.hip
int hip_status(hipError_t e, const char* what);

#define ATX_REQUIRE(cond, code, ...)  \
    do {                              \
        if (!(cond)) {                \
            atx::set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

#define ATX_LAUNCH_CHECK(what)                                   \
    do {                                                         \
        int _st = atx::hip_status(hipGetLastError(), what);      \
        if (_st != ATX_OK) return _st;                           \
    } while (0)

constexpr int kWave = 64;    // CDNA4 wavefront
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU
constexpr int kXcds = 8;     // MI355X: 8 XCDs, private 4 MiB L2 each

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- 16-byte vectors ----------------------------------------------------------
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    static constexpr int N = 4;
    using type = float4;
};
template <>
struct Vec16<double> {
    static constexpr int N = 2;
    using type = double2;
};

template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
    T v[N];
};

// Workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share an L2).
// Give every XCD one contiguous range of tiles so that neighbouring targets —
// which read the same or adjacent source columns — meet in the same L2.
// Bijective for any tile count (speed only, never correctness).
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned n_tiles) {
    const unsigned x = b % kXcds;
    const unsigned r = b / kXcds;
    const unsigned per = n_tiles / kXcds;
    const unsigned rem = n_tiles % kXcds;
    // XCD x owns `per` tiles, the first `rem` XCDs one more
    const unsigned start = x * per + (x < rem ? x : rem);
    return start + r;
}

template <typename T>
__device__ __forceinline__ T quiet_nan();
template <>
__device__ __forceinline__ float quiet_nan<float>() {
    return __uint_as_float(0x7fc00000u);  // np.float32(np.nan)
}
template <>
__device__ __forceinline__ double quiet_nan<double>() {
    return __longlong_as_double(0x7ff8000000000000ll);  // np.nan
}

// One per-level operator in the arithmetic type of the stack.
template <typename T>
struct LevelOp {
    int op;
    int use_mask;
    T p0;
    T p1;
};

template <typename T>
__device__ __forceinline__ LevelOp<T> load_level_op(const atx_level_op* prog, int64_t i) {
    LevelOp<T> o;
    o.op = prog[i].op;
    o.use_mask = prog[i].use_mask;
    o.p0 = static_cast<T>(prog[i].p0);
    o.p1 = static_cast<T>(prog[i].p1);
    return o;
}

// The reference statements, one rounding per numpy ufunc call (file compiled with
// -ffp-contract=off so x*p0+p1 stays a multiply and an add).
template <typename T>
__device__ __forceinline__ T apply_level_op(const LevelOp<T>& o, T x, bool masked) {
    T y = x;
    switch (o.op) {
        case ATX_OP_COPY: break;
        case ATX_OP_AFFINE: y = x * o.p0 + o.p1; break;
        case ATX_OP_AFFINE_INV: y = (x - o.p1) / o.p0; break;
        case ATX_OP_MUL: y = x * o.p0; break;
        case ATX_OP_DIV: y = x / o.p0; break;
        case ATX_OP_CLIP:
            if (o.p0 == o.p0) y = (y < o.p0) ? o.p0 : y;  // np.maximum keeps a NaN x
            if (o.p1 == o.p1) y = (y > o.p1) ? o.p1 : y;
            break;
        case ATX_OP_IMPUTE_NAN: y = (x != x) ? o.p0 : x; break;
        case ATX_OP_EXP: y = exp(x); break;
        case ATX_OP_LOG: y = log(x); break;
        case ATX_OP_SET_NAN: y = quiet_nan<T>(); break;
        default: break;
    }
    if (o.use_mask && masked) y = quiet_nan<T>();
    return y;
}

}  // namespace atx

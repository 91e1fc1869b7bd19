// Shared host/device helpers of libatx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/atx.h"

namespace atx {

// ---- host-side error plumbing ------------------------------------------------
void set_error(const char* fmt, ...);  // defined in atx_api.hip
int hip_status(hipError_t e, const char* what);

// Internal status, never returned across the C ABI: the per-level operator tables of this call do not fit the 64 KB a workgroup
// may stage in LDS (tall stacks with long programs: float32, ~1000 levels, 8 stages = 72 KB).  The entry points answer it themselves:
// atx_pointwise_stack runs the stages in two halves (stages compose, so the bits are the same), the regrid entry points run the gather
// without the program and apply it to the output in place.  (Round 3 returned ATX_ENOTIMPL for shapes round 2 had served.)
#define ATX_SPLIT_PROGRAM 1

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

// Grid cap of the grid-stride streaming kernels.  Measured on MI355X (per-point kernel over a 3.7 GB stack): 2048 workgroups
// (8 per CU) 1.42 ms, 8192 1.26, 65536 1.20-1.27, 131072 the same, one workgroup per 16 KB chunk (225 k) 1.22-1.26 —
// many short workgroups keep more requests in different phases in flight than a few long-lived ones.
#ifndef ATX_MAX_GRID
#define ATX_MAX_GRID 65536
#endif
constexpr int64_t kStreamGrid = ATX_MAX_GRID;

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

// The same deal in STRIPES: XCD x takes tiles x*G .. x*G+G-1 of every group of 8*G consecutive tiles.  Neighbouring tiles still meet
// in one L2 (within a stripe), and rows whose cost varies slowly along the tile order — a conservative matrix whose boxes hold ~20
// source points at the poles and ~200 at the equator — spread over all XCDs instead of loading the equatorial ones 2.5x (xcd_tile
// gives every XCD one latitude band).  Bijective for any tile count (the tail beyond the last whole group maps to itself).
__device__ __forceinline__ unsigned xcd_stripe(unsigned b, unsigned n_tiles, unsigned G) {
    const unsigned group = kXcds * G;
    const unsigned n_full = n_tiles / group * group;
    if (b >= n_full) return b;
    const unsigned x = b % kXcds;
    const unsigned r = b / kXcds;
    return (r / G) * group + x * G + (r % G);
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

// fma(a, b, c) with c a compile-time constant held in SCALAR registers (v_fma_f64 v, v, v, s[..]).  Left to itself the compiler keeps the
// coefficients of a float64 polynomial in vector registers and evaluates Horner steps with the two-address v_fmac_f64, which overwrites
// its addend — so every step of every element first COPIES its coefficient (v_mov_b64): 12 extra instructions in a degree-13 Horner chain,
// a third of it.  Used with literal coefficients; a `c` that is not a compile-time constant falls back to __builtin_fma.
// ATX_FMA_SGPR=0 restores __builtin_fma.
#ifndef ATX_FMA_SGPR
#define ATX_FMA_SGPR 1
#endif
__device__ __forceinline__ double fma_k(double a, double b, double c) {
#if ATX_FMA_SGPR
    // the scalar-register form only for an addend the compiler KNOWS to be a constant once this is inlined; anything per-lane takes the
    // ordinary fma (without this check the "s" constraint would silently broadcast lane 0's addend through v_readfirstlane)
    if (__builtin_constant_p(c)) {
        double d;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
        return d;
    }
#endif
    return __builtin_fma(a, b, c);
}

// a / b for float64 WITHOUT the IEEE division sequence (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup = 13 VALU instructions):
// hardware reciprocal (2^-23), one Newton step, the quotient and one correction with its exact residual — 7 instructions, <= 1 ulp.
// For quotients INSIDE a library function (log's s = f / (2 + f), tanh's e / (e + 2)), whose own error budget is wider; the
// reference's statements (`x / g`, `(x - b) / a`, `1000 sd / rsn`) keep the IEEE division and numpy's bits.  Operands must be
// normal and finite (no scaling, no fix-up).
__device__ __forceinline__ double quotient_1ulp(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// The same quotient in 6 instructions: the raw reciprocal (2^-23 relative: "2^29 ulp" in the ISA manual) and TWO corrections of the quotient with
// its exact residual — relative error 2^-23 -> 2^-46 -> below one rounding.  (quotient_1ulp refines the reciprocal first, which tanh's
// e / (e + 2) keeps: there the same reciprocal error would enter a result that has no other slack.)
__device__ __forceinline__ double quotient_2fix(double a, double b) {
    const double r = __builtin_amdgcn_rcp(b);
    double q = a * r;
    q = __builtin_fma(__builtin_fma(-b, q, a), r, q);
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// Natural logarithm for the per-level operator ATX_OP_LOG (sp_to_lnsp, R: filters/fields/lnsp_to_sp.py:65).  float32: the device
// library's.  float64: the classic argument reduction x = 2^k * m, m in [sqrt(2)/2, sqrt(2)), log(m) = f - (f^2/2 - s*(f^2/2 + R(s^2)))
// with s = f / (2 + f), f = m - 1 and a degree-14 minimax R (the published fdlibm coefficients) — the device library's double log
// measured 3.18 ms over 137 levels of O1280 (0.57 of the HBM peak, ALU-bound: 102 VALU instructions per element, 43 of them plain
// additions of its double-double arithmetic) where exp takes 2.34 ms.
//   ATX_FAST_LOG=1 (round 3): fdlibm's integer bit manipulation and IEEE division, 59 VALU instructions per element, 2.50 ms (0.72).
//   ATX_FAST_LOG=2 (round 5, default): the hardware's v_frexp_mant / v_frexp_exp (subnormals included) instead of the integer
//     sequence, quotient_1ulp instead of the IEEE division, R by Horner in s^2 and the combination with fused multiply-adds,
//     the special operands (0, negative, inf, NaN) behind ONE v_cmp_class and a branch no wave of real data takes:
//     31 instructions per element (tools/kernel_isa.py).  Measured <= 1 ulp from numpy's (true error <= 0.73 ulp against
//     200-bit arithmetic on the host prototype), exact at 1; zero, negatives, infinities and NaN follow IEEE / numpy: -inf, NaN, +inf, NaN.
//   ATX_FAST_LOG=0: the device library.
#ifndef ATX_FAST_LOG
#define ATX_FAST_LOG 2
#endif
__device__ __forceinline__ float atx_log(float x) { return log(x); }
__device__ __forceinline__ double atx_log(double x) {
    [[maybe_unused]] constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    [[maybe_unused]] constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                                      Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                                      Lg7 = 1.479819860511658591e-01;
#if ATX_FAST_LOG == 2
    const double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1), subnormals normalised by the instruction
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;  // bring m into [sqrt(2)/2, sqrt(2)): double it, k - 1
    k -= low ? 1 : 0;
    // f = m' - 1 in ONE instruction: 2.0 and 1.0 differ in their high word only, so the factor is one 32-bit select; the fma is exact
    const double f = __builtin_fma(m, __hiloint2double(low ? 0x40000000 : 0x3ff00000, 0), -1.0);
    const double s = quotient_2fix(f, 2.0 + f);
    const double z = s * s;
    const double R = z * fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, Lg7, Lg6), Lg5), Lg4), Lg3), Lg2), Lg1);
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    double y = __builtin_fma(dk, ln2_hi, f - (hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)));
    // everything but positive normal / subnormal numbers: +-0, negatives, +-inf, NaN (class bits 0-6 and 9)
    if (__builtin_amdgcn_class(x, 0x27f)) {
        y = (x == 0.0) ? -__longlong_as_double(0x7ff0000000000000ll) : ((x < 0.0 || x != x) ? __longlong_as_double(0x7ff8000000000000ll) : x);
    }
    return y;
#elif ATX_FAST_LOG == 1
    const double x0 = x;
    int k = 0;
    int hx = __double2hiint(x);
    if (hx < 0x00100000) {  // subnormal (or zero / negative: settled at the end): scale into the normal range, exactly
        x *= 18014398509481984.0;  // 2^54
        k = -54;
        hx = __double2hiint(x);
    }
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;  // m >= sqrt(2): halve it, k + 1
    x = __hiloint2double(hx | (i ^ 0x3ff00000), __double2loint(x));
    k += i >> 20;
    const double f = x - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    double y = s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;  // (the one-form evaluation of FreeBSD / musl)
    if (x0 == 0.0) y = -__longlong_as_double(0x7ff0000000000000ll);
    if (x0 < 0.0 || x0 != x0) y = __longlong_as_double(0x7ff8000000000000ll);
    if (x0 == __longlong_as_double(0x7ff0000000000000ll)) y = x0;
    return y;
#else
    return log(x);
#endif
}

// exp for the per-level operator ATX_OP_EXP (lnsp_to_sp, R: filters/fields/lnsp_to_sp.py:47).  float32: the device library's.  float64
// (ATX_FAST_EXP=1, round 5): x = k ln2 + r with two fused multiply-adds (|r| <= ln2 / 2), exp(r) = 1 + r (1 + r q(r)) with q of degree 9 —
// a Chebyshev fit of (e^r - 1 - r) / r^2 on the interval, relative error 0.15 x 2^-53 before rounding (derived with 300-bit arithmetic,
// tools/experiments/exp_polynomial.py) — exact scaling by v_ldexp_f64, which also delivers +inf beyond 709.78 and the subnormal
// results and 0 below -708.4 with one rounding.  The argument is clamped to +-1100 first (see below) and a NaN passed through at the
// end: no branch.  18 VALU instructions per element against the device library's 22 (it spends 5 on range checks that ldexp makes
// unnecessary); <= 1 ulp from numpy's on every range of tests/test_gpu_kernels.py::test_float64_exp_within_one_ulp_of_numpy (true error
// <= 0.9 ulp on a host prototype against 200-bit arithmetic).  ATX_FAST_EXP=0: the device library.
#ifndef ATX_FAST_EXP
#define ATX_FAST_EXP 1
#endif
__device__ __forceinline__ float atx_exp(float x) { return exp(x); }
__device__ __forceinline__ double atx_exp(double x) {
#if ATX_FAST_EXP
    constexpr double log2e = 1.4426950408889634, ln2hi = 6.93147180559945286227e-01, ln2lo = 2.31904681384629955842e-17;
    // Everything beyond +-1100 already has its answer (+inf above 709.78, 0 below -745.1): clamping there keeps k = rint(x log2e)
    // within +-1587, so the conversion to int below is always in range (an out-of-range double -> int cast is undefined in C++ and
    // poison in LLVM, whatever v_cvt_i32_f64 does), and ldexp delivers +inf / 0 for the clamped operands, +-inf included.  A NaN
    // comes out of the clamp as a number (fmax / fmin return the other operand) and is put back by the last line.
    const double xin = x;
    x = __builtin_fmin(__builtin_fmax(x, -1100.0), 1100.0);
    const double k = __builtin_rint(x * log2e);
    double r = __builtin_fma(-k, ln2hi, x);
    r = __builtin_fma(-k, ln2lo, r);
    double q = 0x1.af38d53857513p-26;
    q = fma_k(q, r, 0x1.2891a8c1d838dp-22);
    q = fma_k(q, r, 0x1.71de0d9c145d0p-19);
    q = fma_k(q, r, 0x1.a019b8ef67c6cp-16);
    q = fma_k(q, r, 0x1.a01a01a7c8d47p-13);
    q = fma_k(q, r, 0x1.6c16c17893833p-10);
    q = fma_k(q, r, 0x1.11111111109adp-7);
    q = fma_k(q, r, 0x1.5555555553d4fp-5);
    q = fma_k(q, r, 0x1.5555555555556p-3);
    q = fma_k(q, r, 0x1.0000000000001p-1);
    const double p = __builtin_fma(r, __builtin_fma(r, q, 1.0), 1.0);
    const double y = __builtin_amdgcn_ldexp(p, (int)k);  // |k| <= 1587
    return (xin != xin) ? xin : y;
#else
    return exp(x);
#endif
}

// expm1(y) in float64 for moderate arguments (|y| <= 40; beyond, or for NaN, the result is garbage or NaN but never a trap) and, on it,
// tanh — what `snow_cover` needs between bare ground and deep snow (R: filters/fields/snow_cover.py:34-39; atx_combine.hip).
// y = k ln2 + r, |r| <= ln2 / 2; expm1(r) = r + r^2 q(r) with q the Taylor series to r^13 (truncation 1e-17 relative);
// 2^k (r + r^2 q) + (2^k - 1) evaluated as fma(2^k r^2, q, fma(2^k, r, 2^k - 1)): the dominant part takes ONE rounding (k = 1, r < 0 would
// otherwise cancel a rounded expm1(r) against 1).  24 VALU instructions; <= 1 ulp from numpy's expm1 on (0, 40] (true error <= 1.2 ulp on
// the host prototype).  The device library's expm1: 57.
__device__ __forceinline__ double atx_expm1_moderate(double y) {
    constexpr double log2e = 1.4426950408889634, ln2hi = 6.93147180559945286227e-01, ln2lo = 2.31904681384629955842e-17;
    const double k = __builtin_rint(y * log2e);
    double r = __builtin_fma(-k, ln2hi, y);
    r = __builtin_fma(-k, ln2lo, r);
    double q = 1.0 / 6227020800.0;  // 1 / 13!
    q = fma_k(q, r, 1.0 / 479001600.0);
    q = fma_k(q, r, 1.0 / 39916800.0);
    q = fma_k(q, r, 1.0 / 3628800.0);
    q = fma_k(q, r, 1.0 / 362880.0);
    q = fma_k(q, r, 1.0 / 40320.0);
    q = fma_k(q, r, 1.0 / 5040.0);
    q = fma_k(q, r, 1.0 / 720.0);
    q = fma_k(q, r, 1.0 / 120.0);
    q = fma_k(q, r, 1.0 / 24.0);
    q = fma_k(q, r, 1.0 / 6.0);
    q = fma_k(q, r, 0.5);
    // 2^k for |k| <= 58 (|y| <= 40).  Callers may hand over anything and discard the result (a negative or NaN snow-cover argument):
    // the exponent is built from a CLAMPED k, so the double -> int conversion is always in range and the shifted value never negative —
    // r keeps the unclamped k, so a NaN still comes out as NaN and an oversized |y| as (defined) garbage.
    const int ki = (int)__builtin_fmin(__builtin_fmax(k, -60.0), 60.0);
    const double t = __hiloint2double((int)((unsigned)(ki + 1023) << 20), 0);
    const double a = __builtin_fma(t, r, t - 1.0);
    return __builtin_fma(t * (r * r), q, a);
}
// tanh(x) for 0 < x <= 20 (NaN passes through; a negative x of moderate size gives tanh(x) too): e / (e + 2) with e = expm1(2x) — no
// cancellation anywhere, 2-3 ulp from numpy's.
__device__ __forceinline__ double atx_tanh_moderate(double x) {
    const double e = atx_expm1_moderate(2.0 * x);
    return quotient_1ulp(e, e + 2.0);
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
// TRANS = false: the instantiation for programs WITHOUT exp / log (the caller has checked the host copy of the program).  The
// inlined float64 exp / log bodies cost every kernel that can reach them ~40 VGPRs — the chunked per-point kernel sat at 110
// VGPRs = 4 waves per SIMD with them, on the edge of what hides HBM latency (its run time moved 2.34 <-> 2.65 ms from box to box
// while the plain copy stayed at 2.34 ms, profiles/r03_pointwise_ab*.log) — and rescale / convert / orog_to_z / clip /
// impute_nans / apply_mask never call them.
template <typename T, bool TRANS = true>
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
        case ATX_OP_EXP:
            if constexpr (TRANS) y = atx_exp(x);
            break;
        case ATX_OP_LOG:
            if constexpr (TRANS) y = atx_log(x);
            break;
        case ATX_OP_SET_NAN: y = quiet_nan<T>(); break;
        default: break;
    }
    if (o.use_mask && masked) y = quiet_nan<T>();
    return y;
}


// The same operator applied to a whole 16-byte vector: ONE dispatch on the operator instead of
// one per element (the per-level programs of real pipelines are uniform over the levels a vector
// spans almost always; kernels fall back to apply_level_op per element when they are not).
template <typename T, int VEC, bool TRANS = true>
__device__ __forceinline__ void apply_level_op_vec(const LevelOp<T>& o, Pack<T, VEC>& v, bool masked) {
    switch (o.op) {
        case ATX_OP_COPY: break;
        case ATX_OP_AFFINE:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = v.v[e] * o.p0 + o.p1;
            break;
        case ATX_OP_AFFINE_INV:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = (v.v[e] - o.p1) / o.p0;
            break;
        case ATX_OP_MUL:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = v.v[e] * o.p0;
            break;
        case ATX_OP_DIV:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = v.v[e] / o.p0;
            break;
        case ATX_OP_CLIP:
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                T y = v.v[e];
                if (o.p0 == o.p0) y = (y < o.p0) ? o.p0 : y;
                if (o.p1 == o.p1) y = (y > o.p1) ? o.p1 : y;
                v.v[e] = y;
            }
            break;
        case ATX_OP_IMPUTE_NAN:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = (v.v[e] != v.v[e]) ? o.p0 : v.v[e];
            break;
        case ATX_OP_EXP:
            if constexpr (TRANS) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v.v[e] = atx_exp(v.v[e]);
            }
            break;
        case ATX_OP_LOG:
            if constexpr (TRANS) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v.v[e] = atx_log(v.v[e]);
            }
            break;
        case ATX_OP_SET_NAN:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = quiet_nan<T>();
            break;
        default: break;
    }
    if (o.use_mask && masked) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v.v[e] = quiet_nan<T>();
    }
}

// ONE operator kind over the levels of a vector, each level with its OWN parameters (a[e], b[e]): a scale / offset per level,
// the usual shape of a per-level program.  Same statements and roundings as apply_level_op.
template <typename T, int VEC, bool TRANS = true>
__device__ __forceinline__ void apply_level_op_params(int op, bool use_mask, const Pack<T, VEC>& a, const Pack<T, VEC>& b, Pack<T, VEC>& v,
                                                      bool masked) {
    switch (op) {
        case ATX_OP_COPY: break;
        case ATX_OP_AFFINE:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = v.v[e] * a.v[e] + b.v[e];
            break;
        case ATX_OP_AFFINE_INV:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = (v.v[e] - b.v[e]) / a.v[e];
            break;
        case ATX_OP_MUL:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = v.v[e] * a.v[e];
            break;
        case ATX_OP_DIV:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = v.v[e] / a.v[e];
            break;
        case ATX_OP_CLIP:
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                T y = v.v[e];
                if (a.v[e] == a.v[e]) y = (y < a.v[e]) ? a.v[e] : y;
                if (b.v[e] == b.v[e]) y = (y > b.v[e]) ? b.v[e] : y;
                v.v[e] = y;
            }
            break;
        case ATX_OP_IMPUTE_NAN:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = (v.v[e] != v.v[e]) ? a.v[e] : v.v[e];
            break;
        case ATX_OP_EXP:
            if constexpr (TRANS) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v.v[e] = atx_exp(v.v[e]);
            }
            break;
        case ATX_OP_LOG:
            if constexpr (TRANS) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v.v[e] = atx_log(v.v[e]);
            }
            break;
        case ATX_OP_SET_NAN:
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = quiet_nan<T>();
            break;
        default: break;
    }
    if (use_mask && masked) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v.v[e] = quiet_nan<T>();
    }
}

// ---- the operators of EVERY LEVEL of a program, staged in LDS -------------------------------------------------------------
// p0[stage][Lp], p1[stage][Lp] in the stack's type and one code byte (op | use_mask << 7) per level, Lp = C * VEC levels (the
// padding of the last vector repeats the last level): a lane reads the parameters of the VEC levels of its vector with two
// conflict-free LDS reads and their codes with one 1- / 2- / 4-byte read per stage.  (An array of LevelOp structs per level costs
// four-way bank conflicts; per-VECTOR operators — build_vector_ops below — send every vector of a program with a scale per level
// through the global program, VEC x 24 bytes per stage and vector.)
template <int VEC>
struct OpWordOf {
    using type = uint32_t;
};
template <>
struct OpWordOf<2> {
    using type = uint16_t;
};
template <>
struct OpWordOf<1> {
    using type = uint8_t;
};

template <typename T>
struct LevelTablesLds {
    const T* p0;
    const T* p1;
    const uint8_t* code;
    int Lp;
};

template <typename T>
__host__ __device__ static inline size_t level_tables_lds_bytes(int n_stage, int C, int vec) {
    return (((size_t)n_stage * C * vec * (2 * sizeof(T) + 1)) + 15) & ~size_t(15);
}

// Fill the tables at `smem` (16-byte aligned) from the global program; the caller's barrier follows.
template <typename T, int VEC>
__device__ __forceinline__ LevelTablesLds<T> build_level_tables(const atx_level_op* __restrict__ prog, unsigned char* smem, int n_stage,
                                                                int n_lev, int C, int tid, int n_threads) {
    const int Lp = C * VEC;
    T* p0 = reinterpret_cast<T*>(smem);
    T* p1 = p0 + (size_t)n_stage * Lp;
    uint8_t* code = reinterpret_cast<uint8_t*>(p1 + (size_t)n_stage * Lp);
    for (int i = tid; i < n_stage * Lp; i += n_threads) {
        const int s = i / Lp, l = i - s * Lp;
        const atx_level_op o = prog[(int64_t)s * n_lev + (l < n_lev ? l : n_lev - 1)];
        p0[i] = static_cast<T>(o.p0);
        p1[i] = static_cast<T>(o.p1);
        code[i] = (uint8_t)((o.op & 0x7f) | (o.use_mask ? 0x80 : 0));
    }
    LevelTablesLds<T> t;
    t.p0 = p0;
    t.p1 = p1;
    t.code = code;
    t.Lp = Lp;
    return t;
}

// Does any stage do something to vector column c?  (ATX_OP_COPY without the mask is the zero byte.)
template <typename T, int VEC>
__device__ __forceinline__ bool level_tables_active(const LevelTablesLds<T>& t, int n_stage, int c) {
    using OpWord = typename OpWordOf<VEC>::type;
    unsigned any = 0;
    for (int s = 0; s < n_stage; ++s) any |= *reinterpret_cast<const OpWord*>(t.code + (size_t)s * t.Lp + c * VEC);
    return any != 0;
}

// All stages applied to the vector of column c.
template <typename T, int VEC, bool TRANS = true>
__device__ __forceinline__ void apply_level_tables(const LevelTablesLds<T>& t, int n_stage, int c, Pack<T, VEC>& v, bool masked) {
    using V = Pack<T, VEC>;
    using OpWord = typename OpWordOf<VEC>::type;
    constexpr unsigned kRep = VEC == 4 ? 0x01010101u : (VEC == 2 ? 0x0101u : 0x01u);
    for (int s = 0; s < n_stage; ++s) {
        const unsigned wd = *reinterpret_cast<const OpWord*>(t.code + (size_t)s * t.Lp + c * VEC);
        if (wd == 0) continue;
        const V a = *reinterpret_cast<const V*>(t.p0 + (size_t)s * t.Lp + c * VEC);
        const V b = *reinterpret_cast<const V*>(t.p1 + (size_t)s * t.Lp + c * VEC);
        const unsigned first = wd & 0xffu;
        if (wd == first * kRep) {  // one operator kind over the vector's levels (their parameters may differ)
            apply_level_op_params<T, VEC, TRANS>((int)(first & 0x7fu), (first & 0x80u) != 0, a, b, v, masked);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const unsigned code = (wd >> (8 * e)) & 0xffu;
                LevelOp<T> o;
                o.op = (int)(code & 0x7fu);
                o.use_mask = (int)(code >> 7);
                o.p0 = a.v[e];
                o.p1 = b.v[e];
                v.v[e] = apply_level_op<T, TRANS>(o, v.v[e], masked);
            }
        }
    }
}

constexpr int kOpMixed = -1;  // marker in a per-vector operator table: the vector's levels differ

__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// Per-vector-column view of a per-level program, built straight from global memory into LDS in ONE
// pass (the workgroups that use it are short-lived, their prologue must stay cheap):
// vec_ops[s*C + c] = the operator shared by the levels c*VEC .. c*VEC+VEC-1 at stage s (padding
// levels >= n_lev join any operator), or op = kOpMixed when they differ.
template <typename T, int VEC>
__device__ __forceinline__ void build_vector_ops(const atx_level_op* __restrict__ prog, LevelOp<T>* __restrict__ vec_ops,
                                                 int n_stage, int n_lev, int C, int tid, int n_threads) {
    for (int i = tid; i < n_stage * C; i += n_threads) {
        const int s = i / C, c = i - s * C;
        const int64_t base = (int64_t)s * n_lev + (int64_t)c * VEC;
        LevelOp<T> o = load_level_op<T>(prog, base);
        bool same = true;
#pragma unroll
        for (int e = 1; e < VEC; ++e) {
            if (c * VEC + e >= n_lev) continue;
            const LevelOp<T> q = load_level_op<T>(prog, base + e);
            same = same && q.op == o.op && q.use_mask == o.use_mask && same_bits(q.p0, o.p0) && same_bits(q.p1, o.p1);
        }
        if (!same) o.op = kOpMixed;
        vec_ops[i] = o;
    }
}

// Apply all stages to one vector of column c: vector-uniform stages through ONE dispatch, mixed ones
// per element straight from the global program (rare).
template <typename T, int VEC, bool TRANS = true>
__device__ __forceinline__ void apply_program_vec(const LevelOp<T>* __restrict__ vec_ops, const atx_level_op* __restrict__ prog,
                                                  int n_stage, int n_lev, int C, int c, Pack<T, VEC>& v, bool masked) {
    for (int s = 0; s < n_stage; ++s) {
        const LevelOp<T> o = vec_ops[s * C + c];
        if (o.op != kOpMixed) {
            apply_level_op_vec<T, VEC, TRANS>(o, v, masked);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int l = c * VEC + e;
                if (l < n_lev) v.v[e] = apply_level_op<T, TRANS>(load_level_op<T>(prog, (int64_t)s * n_lev + l), v.v[e], masked);
            }
        }
    }
}

// Layout of the table atx_vector_program writes (units: bytes from the table's base, which must be 16-byte aligned for the second
// part to be used): [n_stage * C] per-vector atx_level_op entries, then — from the next 16-byte boundary — the operators of EVERY
// LEVEL in the stack's arithmetic type, parameters and codes apart: p0[n_stage][Lp], p1[n_stage][Lp] (T), code[n_stage][Lp]
// (one byte: op | use_mask << 7), Lp = C * V levels (the padding of the last vector repeats the last level).  A lane reads the
// parameters of its vector's levels with two 16-byte loads and their codes with one 2- / 4-byte load per stage.
struct LevelTables {
    int64_t levels_offset;  // bytes from the base to p0
    int64_t Lp;
    int64_t total_bytes;
};
static inline LevelTables level_tables_layout(int n_stage, int64_t n_lev, int dtype) {
    const int64_t V = dtype == ATX_F32 ? 4 : 2, B = dtype == ATX_F32 ? 4 : 8;
    const int64_t C = (n_lev + V - 1) / V;
    LevelTables t;
    t.Lp = C * V;
    t.levels_offset = (((int64_t)n_stage * C * (int64_t)sizeof(atx_level_op)) + 15) & ~int64_t(15);
    t.total_bytes = t.levels_offset + (int64_t)n_stage * t.Lp * (2 * B + 1);
    return t;
}

// A per-level program whose operators can travel BY VALUE in the kernel arguments (scalar registers, scalar branch on the operator):
// per stage the levels run ONE operator, or one operator up to a level that is a multiple of the vector width and another from
// there on ("136 levels of t, then orog"), at most kMaxUniform stages.  Used by the fused regrid epilogue (atx_regrid.hip) and by
// the per-point kernel (atx_pointwise.hip).
constexpr int kMaxUniform = 4;
template <typename T>
struct UniformOps {
    int n_stage;
    int split[kMaxUniform];          // first vector column of the second piece of stage s (>= C: the stage has one piece)
    LevelOp<T> stage[kMaxUniform];   // first piece
    LevelOp<T> second[kMaxUniform];  // second piece
};

template <typename T>
static inline bool uniform_level_program(const atx_level_op* host_prog, int n_stage, bool have_mask, int n_lev, int vec, UniformOps<T>& out) {
    if (!host_prog || n_stage < 1 || n_stage > kMaxUniform) return false;
    auto typed = [](const atx_level_op& o) {
        LevelOp<T> r;
        r.op = o.op;
        r.use_mask = o.use_mask ? 1 : 0;
        r.p0 = static_cast<T>(o.p0);
        r.p1 = static_cast<T>(o.p1);
        return r;
    };
    auto same = [](const LevelOp<T>& a, const LevelOp<T>& b) {
        return a.op == b.op && a.use_mask == b.use_mask && std::memcmp(&a.p0, &b.p0, sizeof(T)) == 0 && std::memcmp(&a.p1, &b.p1, sizeof(T)) == 0;
    };
    const int C = (n_lev + vec - 1) / vec;
    out.n_stage = n_stage;
    for (int s = 0; s < kMaxUniform; ++s) out.split[s] = C;
    for (int s = 0; s < n_stage; ++s) {
        const atx_level_op* row = host_prog + (int64_t)s * n_lev;
        const LevelOp<T> first = typed(row[0]);
        int l = 1;
        while (l < n_lev && same(typed(row[l]), first)) ++l;
        out.stage[s] = out.second[s] = first;
        if (l == n_lev) continue;  // one piece
        if (l % vec != 0) return false;  // the change must fall on a vector boundary
        const LevelOp<T> second = typed(row[l]);
        for (int m = l + 1; m < n_lev; ++m)
            if (!same(typed(row[m]), second)) return false;  // a third piece
        out.second[s] = second;
        out.split[s] = l / vec;
    }
    for (int s = 0; s < n_stage; ++s)
        if ((out.stage[s].use_mask || out.second[s].use_mask) && !have_mask) return false;  // (rejected by validation anyway)
    return true;
}

// Round 4: the general piecewise-uniform form — per stage up to kMaxRuns RUNS of consecutive levels, each with its own operator, the
// boundaries ANYWHERE (not on the vector grid).  This is the program of a stack in which several variables share a column
// (DESIGN.md §2 "Tall stacks": 137 levels of t -> degC, 137 of orography x g, 137 left alone): boundaries at 137 and 274 fall
// inside 16-byte vectors, so uniform_level_program refuses it and it used to read the per-level table (+11 % on a 411-level
// float64 regrid) or go through the per-level LDS kernel (0.70-0.75 of the peak against 0.81).  By value in the kernel arguments:
// a lane evaluates every run's operator on its vector and keeps, element by element, the one whose levels it holds.
constexpr int kMaxRuns = 4;
template <typename T>
struct RunOps {
    int n_stage;
    int n_run[kMaxUniform];               // runs of stage s (1 .. kMaxRuns)
    int start[kMaxUniform][kMaxRuns];     // first LEVEL of run r (start[s][0] = 0)
    LevelOp<T> op[kMaxUniform][kMaxRuns];
};

template <typename T>
static inline bool runs_level_program(const atx_level_op* host_prog, int n_stage, bool have_mask, int n_lev, RunOps<T>& out) {
    if (!host_prog || n_stage < 1 || n_stage > kMaxUniform) return false;
    auto typed = [](const atx_level_op& o) {
        LevelOp<T> r;
        r.op = o.op;
        r.use_mask = o.use_mask ? 1 : 0;
        r.p0 = static_cast<T>(o.p0);
        r.p1 = static_cast<T>(o.p1);
        return r;
    };
    auto same = [](const LevelOp<T>& a, const LevelOp<T>& b) {
        return a.op == b.op && a.use_mask == b.use_mask && std::memcmp(&a.p0, &b.p0, sizeof(T)) == 0 && std::memcmp(&a.p1, &b.p1, sizeof(T)) == 0;
    };
    out.n_stage = n_stage;
    for (int s = 0; s < kMaxUniform; ++s) {
        out.n_run[s] = 1;
        for (int r = 0; r < kMaxRuns; ++r) {
            out.start[s][r] = r == 0 ? 0 : INT32_MAX;
            out.op[s][r] = LevelOp<T>{ATX_OP_COPY, 0, T(0), T(0)};
        }
    }
    for (int s = 0; s < n_stage; ++s) {
        const atx_level_op* row = host_prog + (int64_t)s * n_lev;
        int n = 0;
        for (int l = 0; l < n_lev; ++l) {
            const LevelOp<T> o = typed(row[l]);
            if (n > 0 && same(o, out.op[s][n - 1])) continue;
            if (n == kMaxRuns) return false;  // a fifth run
            out.start[s][n] = l;
            out.op[s][n] = o;
            ++n;
        }
        out.n_run[s] = n;
        for (int r = 0; r < n; ++r)
            if (out.op[s][r].use_mask && !have_mask) return false;  // (rejected by validation anyway)
    }
    return true;
}

// All stages of a RunOps program applied to the vector of levels c*VEC .. c*VEC+VEC-1 (scalar loop bounds and branches: the runs
// are uniform over the launch; the per-element choice is a select).  Every run is evaluated, COPY runs included: skipping them behind a
// scalar branch was tried and LOST — the fused regrid of a 411-level O2560 stack 2.89 -> 3.08 ms (float64), 1.45 -> 1.57 ms (float32), the
// per-point kernel unchanged: the branches keep the compiler from loading the operators ahead of the gather (profiles/r04_runs_probe.log).
template <typename T, int VEC, bool TRANS = true>
__device__ __forceinline__ void apply_run_ops(const RunOps<T>& runs, int c, Pack<T, VEC>& v, bool masked) {
    using V = Pack<T, VEC>;
#pragma unroll
    for (int s = 0; s < kMaxUniform; ++s) {
        if (s >= runs.n_stage) break;
        V res = v;
        apply_level_op_vec<T, VEC, TRANS>(runs.op[s][0], res, masked);
#pragma unroll
        for (int r = 1; r < kMaxRuns; ++r) {
            if (r >= runs.n_run[s]) break;
            V other = v;
            apply_level_op_vec<T, VEC, TRANS>(runs.op[s][r], other, masked);
            const int first = runs.start[s][r];
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (c * VEC + e >= first) res.v[e] = other.v[e];
        }
        v = res;
    }
}

// Does some 16-byte vector of levels hold two different operators at some stage (once the parameters are rounded to T)?  Such
// "mixed" vectors are evaluated level by level from the per-level program — fine for the odd boundary vector of a two-variable
// stack, slow when every vector is one (a different scale per level, float32: 2.20 ms on the per-vector-table kernel, 1.82 ms on
// the chunked one; staging the per-level operators in LDS as well made it 2.08 ms — bank conflicts — and was dropped).
template <typename T>
static inline bool program_has_mixed_vectors(const atx_level_op* host_prog, int n_stage, int n_lev, int vec) {
    if (!host_prog) return false;  // unknown
    for (int s = 0; s < n_stage; ++s) {
        const atx_level_op* row = host_prog + (int64_t)s * n_lev;
        for (int l0 = 0; l0 < n_lev; l0 += vec) {
            for (int l = l0 + 1; l < n_lev && l < l0 + vec; ++l) {
                const T a0 = static_cast<T>(row[l0].p0), a1 = static_cast<T>(row[l0].p1), b0 = static_cast<T>(row[l].p0), b1 = static_cast<T>(row[l].p1);
                if (row[l].op != row[l0].op || (row[l].use_mask != 0) != (row[l0].use_mask != 0) || std::memcmp(&a0, &b0, sizeof(T)) != 0 ||
                    std::memcmp(&a1, &b1, sizeof(T)) != 0)
                    return true;
            }
        }
    }
    return false;
}

// Does a host copy of a program (n_stage * n_lev entries) hold an operator that needs the TRANS = true instantiation?  Without a
// host copy the answer is "maybe".
static inline bool program_has_transcendental(const atx_level_op* host_prog, int n_stage, int n_lev) {
    if (!host_prog) return true;
    for (int64_t i = 0; i < (int64_t)n_stage * n_lev; ++i)
        if (host_prog[i].op == ATX_OP_EXP || host_prog[i].op == ATX_OP_LOG) return true;
    return false;
}

}  // namespace atx

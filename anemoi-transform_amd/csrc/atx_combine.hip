// Multi-input per-point transforms: one launch over all matching groups of a FieldList.
//
// The reference groups fields by MARS key and calls a numpy expression once per group
// (R: filters/fields/matching.py:223-246, grouping/__init__.py:93-137).  Here the host
// stacks the i-th operand of every group into one stack per operand, and this kernel
// evaluates the expression for every (point, level) in one streaming pass: HBM-bound,
// (n_in + n_out)*N*L*B algorithmic bytes, 16-byte vector accesses, grid-stride.
#include "atx_common.hpp"

namespace atx {

struct CombArgs {
    const void* in[ATX_COMB_MAX_INPUTS];
    void* out[2];
};

// cos and sin of a float64 angle of moderate size (|x| < 1e5: a wave direction, a phase in [-2 pi, 2 pi]) WITHOUT the general
// argument reduction of the device maths library.  SQ counters (profiles/r03_pmc_sq_combine.txt) show the float64 cos+sin operator
// VALU-bound with OCML's sincos: 537 VALU instructions per wave of 4 elements per lane, 66 % of wave time issuing or stalled on
// issue, against 7 % for a plain difference.  This is the classic fdlibm scheme — n = rint(x * 2/pi), a Cody-Waite subtraction of
// n * pi/2 in two (rarely three, four) pieces that is EXACT for |n| < 2^17, then the degree-13 / degree-14 kernels on [-pi/4, pi/4]
// with the reduction's tail — no fused multiply-add needed.  Checked on the host against numpy (glibc): at most 1 ulp apart on
// 4 M random angles in [-2 pi, 2 pi] and in [-1e5, 1e5], identical next to multiples of pi/2
// (tests/test_multi_filters.py::test_fast_sincos_*).  Anything else — larger, infinite, NaN — takes OCML's sincos.
__device__ __forceinline__ bool sincos_moderate(double x, double& sn, double& cs) {
    if (!(fabs(x) < 1.0e5)) return false;
    if (fabs(x) < 7.450580596923828125e-09) {  // |x| < 2^-27: sin x = x (keeps the sign of zero), cos x = 1, both correctly rounded
        sn = x;
        cs = 1.0;
        return true;
    }
    const double fn = rint(x * 6.36619772367581382433e-01);
    double r = x - fn * 1.57079632673412561417e+00;  // 33 bits of pi/2: the product is exact
    double w = fn * 6.07710050650619224932e-11;
    double y0 = r - w;
    const int ex = (__double2hiint(x) >> 20) & 0x7ff;
    if (ex - ((__double2hiint(y0) >> 20) & 0x7ff) > 16) {  // cancellation: x is close to a multiple of pi/2 — a second piece
        double t = r;
        w = fn * 6.07710050630396597660e-11;
        r = t - w;
        w = fn * 2.02226624879595063154e-21 - ((t - r) - w);
        y0 = r - w;
        if (ex - ((__double2hiint(y0) >> 20) & 0x7ff) > 49) {  // and a third
            t = r;
            w = fn * 2.02226624871116645580e-21;
            r = t - w;
            w = fn * 8.47842766036889956997e-32 - ((t - r) - w);
            y0 = r - w;
        }
    }
    const double y1 = (r - y0) - w;
    const double z = y0 * y0;
    // sin on [-pi/4, pi/4]
    const double v = z * y0;
    // (round 5: the two kernel polynomials by Horner steps with the coefficient in scalar registers — fma_k, atx_common.hpp; a fused step
    // rounds once where fdlibm's multiply and add round twice, so the results may differ from round 3's in the last bit and stay within
    // the same 1 ulp of the true value; ATX_FMA_SGPR=0 + ATX_SINCOS_FMA=0 give round 3's instruction stream)
#ifndef ATX_SINCOS_FMA
#define ATX_SINCOS_FMA 1
#endif
#if ATX_SINCOS_FMA
    const double rs = fma_k(z, fma_k(z, fma_k(z, fma_k(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
#else
    const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
#endif
    const double s = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
    // cos on [-pi/4, pi/4]
#if ATX_SINCOS_FMA
    const double rc = z * fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
#else
    const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
#endif
    const int iy = __double2hiint(y0) & 0x7fffffff;
    double c;
    if (iy < 0x3FD33333) {  // |y0| < 0.3
        c = 1.0 - (0.5 * z - (z * rc - y0 * y1));
    } else {
        const double qx = iy > 0x3fe90000 ? 0.28125 : __hiloint2double(iy - 0x00200000, 0);  // ~|y0| / 4
        const double hz = 0.5 * z - qx;
        c = (1.0 - qx) - (hz - (z * rc - y0 * y1));
    }
    const int q = ((int)fn) & 3;
    sn = (q & 1) ? c : s;
    cs = (q & 1) ? s : c;
    if (q == 2 || q == 3) sn = -sn;
    if (q == 1 || q == 2) cs = -cs;
    return true;
}

#ifndef ATX_FAST_SINCOS
#define ATX_FAST_SINCOS 1
#endif

// Saturation vapour pressure over water / ice and in the mixed phase, as earthkit-meteo (>= 0.4.1, absent here) publishes them
// (thermo.array.saturation_vapour_pressure: the IFS formulas, c1 = 611.21 Pa, T0 = 273.16 K, water 17.502 / 32.19, ice 22.587 / -0.7, the liquid
// fraction ((T - Ti) / (T0 - Ti))^2 between Ti = T0 - 23 and T0); restated in oracle.py, pinned by the reference's vectors at np.allclose.
// The quotients of these formulas through a refined reciprocal instead of the IEEE division sequence (float64: ~35 instructions
// each, four per element in q_to_r, which made the operator VALU-bound at 0.38 of the HBM peak): v_rcp + two Newton steps + one
// residual correction, within 1 ulp of the correctly rounded quotient for the finite, normal operands these formulas see
// (temperatures, pressures, vapour pressures; a zero or infinite divisor gives NaN where the division gives inf / 0 — no physical input).
// ATX_HUMIDITY_IEEE_DIV=1 restores the plain division.  Measured on 137-level O1280 stacks: q_to_r 0.37 -> 0.52 (f32), 0.38 -> 0.50 (f64) of
// 8 TB/s, r_to_d 0.56 -> 0.70 / 0.52 -> 0.57; an own float64 exp without the library's special cases and 1 or 4 vectors per lane instead of 2
// changed nothing (profiles/r04_humidity_variants.log).
#ifndef ATX_HUMIDITY_IEEE_DIV
#define ATX_HUMIDITY_IEEE_DIV 0
#endif
__device__ __forceinline__ double quotient(double a, double b) {
#if ATX_HUMIDITY_IEEE_DIV
    return a / b;
#else
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
#endif
}
__device__ __forceinline__ float quotient(float a, float b) {
#if ATX_HUMIDITY_IEEE_DIV
    return a / b;
#else
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
#endif
}
template <typename T>
__device__ __forceinline__ T es_water(T t) { return T(611.21) * atx_exp(quotient(T(17.502) * (t - T(273.16)), t - T(32.19))); }
template <typename T>
__device__ __forceinline__ T es_ice(T t) { return T(611.21) * atx_exp(quotient(T(22.587) * (t - T(273.16)), t - T(-0.7))); }
template <typename T>
__device__ __forceinline__ T es_mixed(T t) {
    const T t0 = T(273.16), ti = T(273.16 - 23.0);
    T c = t < t0 ? t : t0;  // min(t0, t), max(ti, .)
    c = c > ti ? c : ti;
    T a = (c - ti) * (T(1) / (t0 - ti));
    a = a * a;
    a = a < T(1) ? a : T(1);
    // each exponential only for the lanes it counts for (a column holds a whole profile: a wave usually has lanes on both sides, and
    // then pays for two exponentials, not for the three branches "water / ice / both" one after the other); a NaN temperature
    // takes neither branch and ends as NaN through `a`
    T ew = T(0), ei = T(0);
    if (a > T(0)) ew = es_water(t);
    if (a < T(1)) ei = es_ice(t);
    return a * ew + (T(1) - a) * ei;
}

// g = 9.80665 (R: constants.py:13, value pinned by filters/tabular/geopotential_to_height.py:51)
// OP is a template parameter: every operator gets its own kernel.  With a runtime switch the float64 kernels carried the
// inlined tanh, sincos and atan2 bodies for EVERY operator — 164-180 VGPRs, 2-3 waves per SIMD, even for `a - b` (round 2 called
// the result "ALU-bound"; it was occupancy).
template <typename T, int OP>
__device__ __forceinline__ void combine_one(int flags, const T* x, int n_in, T level, T& y0, T& y1) {
    const T g = T(9.80665);
    y1 = T(0);
    switch (OP) {
        case ATX_COMB_SNOW_DEPTH_M: y0 = T(1000.0) * x[0] / x[1]; break;
        case ATX_COMB_SNOW_COVER: {
            const T tmp1 = (T(1000) * x[0]) / x[1];
            // np.clip(rsn, 100, 400) as one max and one min: a NaN density — which np.clip would keep — has made tmp1, and with it arg, a
            // NaN already, so the hardware's "the other operand" for a NaN changes nothing (6 compare-and-select instructions -> 2-4)
            T tmp2;
            if constexpr (sizeof(T) == 4) tmp2 = __builtin_fminf(__builtin_fmaxf(x[1], 100.0f), 400.0f);  // (the unsuffixed builtins are the float64 ones)
            else tmp2 = __builtin_fmin(__builtin_fmax(x[1], 100.0), 400.0);
            const T arg = (T(4000) * tmp1) / tmp2;
            // The two common cases need no tanh and give the statement's bits exactly: deep snow — every arg > atanh(0.99) =
            // 2.6467 has tanh(arg) > 0.99 (at 2.65: 0.990066, four orders above any rounding of tanh), which the last line
            // sends to 1.0 — and no snow — tanh(+-0) = +-0, which the clip and the threshold leave alone.  float64 tanh
            // costs ~100 operations; on real fields most points are one or the other.  NaN fails both tests and takes the
            // full path.
            if (arg > T(2.65)) {
                y0 = T(1.0);
            } else if (arg == T(0)) {
                y0 = arg;
            } else {
                // tanh through expm1: e = expm1(2x), tanh x = e / (e + 2) — no cancellation anywhere on (0, 2.65], 2-3 ulp — instead
                // of the device library's tanh (float64: 169 VALU instructions per element).  Round 3 took the library's expm1 and an
                // IEEE division for it (907 -> ~350 instructions per wave of 4 elements per lane; profiles/r03_pmc_sq_combine.txt);
                // round 5 the library's own atx_tanh_moderate (atx_common.hpp: expm1 in 24 instructions, the quotient in 7) and no
                // clamping, sign or NaN handling around it: what reaches this branch is (0, 2.65], whose tanh lies in (0, 0.99007) and needs
                // only the statement's last line; a NaN, which the arithmetic carries through (every comparison below is false for it);
                // or a negative argument (a negative snow depth), whose tanh the clip to [0, 1] replaces by 0 whatever its value.
                // ATX_SNOW_TANH=1 restores the device library's tanh, =2 round 3's form.
#ifndef ATX_SNOW_TANH
#define ATX_SNOW_TANH 0
#endif
                if constexpr (ATX_SNOW_TANH == 0 && sizeof(T) == 8) {
                    const T t = (T)atx_tanh_moderate((double)arg);
                    const T sc = (t > T(0.99)) ? T(1.0) : t;
                    y0 = (arg < T(0)) ? T(0) : sc;
                } else {
                    T sc;
                    if constexpr (ATX_SNOW_TANH == 2 && sizeof(T) == 8) {
                        const T mag = fabs(arg);  // (only negative arguments get here beyond 2.65: they end up clipped to 0)
                        const T e = expm1(T(2) * (mag < T(20) ? mag : T(20)));
                        sc = copysign(mag < T(20) ? e / (e + T(2)) : T(1), arg);  // tanh(20) rounds to 1.0 in float64
                        if (arg != arg) sc = arg;  // NaN stays NaN
                    } else {
                        sc = tanh(arg);
                    }
                    sc = (sc < T(0)) ? T(0) : sc;
                    sc = (sc > T(1)) ? T(1) : sc;
                    y0 = (sc > T(0.99)) ? T(1.0) : sc;
                }
            }
            break;
        }
        case ATX_COMB_COS_SIN: {
            T a = x[0];
            if (flags & ATX_COMB_DEGREES) a = a * T(0.017453292519943295);  // np.deg2rad: x * (pi/180)
            if constexpr (ATX_FAST_SINCOS && sizeof(T) == 8) {
                double sn, cs;
                if (sincos_moderate((double)a, sn, cs)) {
                    y0 = (T)cs;
                    y1 = (T)sn;
                    break;
                }
            }
            y0 = cos(a);
            y1 = sin(a);
            break;
        }
        case ATX_COMB_ATAN2: {
            T d = atan2(x[1], x[0]);
            if (flags & ATX_COMB_DEGREES) {
                d = d * T(57.29577951308232);  // np.rad2deg: x * (180/pi)
                d = (d >= T(360)) ? d - T(360) : d;
                d = (d < T(0)) ? d + T(360) : d;
            }
            y0 = d;
            break;
        }
        case ATX_COMB_W_TO_WZ: {
            const T rho = (T(100) * level) / (T(287) * x[1] * (T(1) + T(0.61) * x[2]) + T(1e-8));
            y0 = (T(-1.0) / (rho * g + T(1e-8))) * x[0];
            break;
        }
        case ATX_COMB_WZ_TO_W: {
            const T rho = (T(100) * level) / (T(287) * x[1] * (T(1) + T(0.61) * x[2]) + T(1e-8));
            y0 = T(-1.0) * rho * g * x[0];
            break;
        }
        case ATX_COMB_SUM: {
            T s = x[0];
            for (int i = 1; i < n_in; ++i) s = s + x[i];
            y0 = s;
            break;
        }
        case ATX_COMB_SUB: y0 = x[0] - x[1]; break;
        case ATX_COMB_XY_TO_POLAR: {  // earthkit.meteo.wind.array.xy_to_polar(u, v, convention="meteo"): numpy's statements, one rounding each
            y0 = hypot(x[0], x[1]);
            T d = T(270.0) - atan2(x[1], x[0]) * T(57.29577951308232);  // constants.degree = 180 / pi
            // np.mod(d, 360): d lies in [90 - eps, 450 + eps] (|atan2| <= pi), where the remainder is d or, exactly, d - 360 — the device
            // library's fmod, a loop in float64, made this the one operator of the family far off the others (0.60 of the HBM peak)
            y1 = d >= T(360.0) ? d - T(360.0) : d;
            break;
        }
        case ATX_COMB_POLAR_TO_XY: {  // polar_to_xy(speed, direction, convention="meteo")
            const T a = (T(270.0) - x[1]) * T(0.017453292519943295);    // constants.radian = pi / 180
            if constexpr (ATX_FAST_SINCOS && sizeof(T) == 8) {
                double sn, cs;
                if (sincos_moderate((double)a, sn, cs)) {
                    y0 = x[0] * (T)cs;
                    y1 = x[0] * (T)sn;
                    break;
                }
            }
            y0 = x[0] * cos(a);
            y1 = x[0] * sin(a);
            break;
        }
        case ATX_COMB_OPERA_CLIP: {  // level = max_total_precipitation; R: rodeo_opera_preprocessing.py:34-37 (twice), rodeo_opera_clipping.py:98
            T tp = x[0], qi = x[1];
            tp = (tp < T(0)) ? T(0) : tp;  // `variable[variable < 0] = 0`: a NaN fails both tests and stays, -0.0 stays -0.0
            tp = (tp >= level) ? level : tp;
            qi = (qi < T(0)) ? T(0) : qi;
            qi = (qi >= T(1)) ? T(1) : qi;
            y0 = tp / T(1000);
            y1 = qi;
            break;
        }
        case ATX_COMB_OPERA_PREPROCESS: {  // (tp, qi, dm); R: rodeo_opera_preprocessing.py:83-87 then :34-37 on both
            T tp = x[0], qi = x[1];
            const T dm = x[2];
            if (dm == T(1)) tp = quiet_nan<T>();  // _NODATA
            if (dm == T(2)) tp = T(0);            // _UNDETECTED
            if (dm == T(3)) tp = quiet_nan<T>();  // _INF
            if (dm == T(2)) qi = T(0);
            tp = (tp < T(0)) ? T(0) : tp;
            tp = (tp >= level) ? level : tp;
            qi = (qi < T(0)) ? T(0) : qi;
            qi = (qi >= T(1)) ? T(1) : qi;
            y0 = tp;
            y1 = qi;
            break;
        }
        case ATX_COMB_ORAS6: {  // (x, siconc), level = what this field is (ATX_ORAS6_*); R: oras6_clipping.py:194-215
            const T puny = T(1e-5), tf = T(273.15), mintf = T(271.15 - 1e-5);
            const int kind = (int)level;
            const bool no_ice = x[1] <= puny;  // `mask = siconc_np <= PUNY`: false for a NaN concentration
            T y = x[0];
            if (kind == ATX_ORAS6_CELSIUS) y = y + tf;
            if (no_ice && (kind == ATX_ORAS6_ZERO || kind == ATX_ORAS6_HEAT)) y = T(0);
            if (no_ice && (kind == ATX_ORAS6_TEMPERATURE || kind == ATX_ORAS6_CELSIUS)) y = tf;
            if (kind == ATX_ORAS6_HEAT) y = (y >= -puny) ? T(0) : y;
            if (kind == ATX_ORAS6_SURFACE) y = (y <= mintf) ? mintf : y;
            y0 = y;
            break;
        }
        case ATX_COMB_R_TO_D: {  // (r, t) -> dewpoint; R: dewpoint.py:62-64 -> thermo.dewpoint_from_relative_humidity
            T r = x[0];
            r = (r == T(0)) ? T(1.0e-4) : r;  // `relative_humidity_values[relative_humidity_values == 0] = EPS`
            const T e = quotient(r * es_water(x[1]), T(100));
            const T ln = atx_log(quotient(e, T(611.21)));
            y0 = quotient(T(32.19) * ln - T(17.502) * T(273.16), ln - T(17.502));
            break;
        }
        case ATX_COMB_D_TO_R: y0 = quotient(T(100) * es_water(x[0]), es_water(x[1])); break;  // (td, t); R: dewpoint.py:71
        case ATX_COMB_Q_TO_R: {  // (q, t[, p]); level = levelist in hPa when there is no pressure operand; R: q_to_r.py:72-74, q_height.py:117-121
            const T eps = T(287.0597 / 461.5250);
            const T p = n_in > 2 ? x[2] : T(100) * level;
            const T e = quotient(p * x[0], eps + (eps * (T(1) / eps - T(1))) * x[0]);
            y0 = quotient(T(100) * e, es_mixed(x[1]));
            break;
        }
        case ATX_COMB_R_TO_Q: {  // (r, t[, p]); R: q_to_r.py:78-82, q_height.py:138-142
            const T eps = T(287.0597 / 461.5250);
            const T p = n_in > 2 ? x[2] : T(100) * level;
            const T e = quotient(x[0] * es_mixed(x[1]), T(100));
            T v = p - (T(1) - eps) * e;
            if (p - e < T(1.0e-4)) v = quiet_nan<T>();  // specific_humidity_from_vapour_pressure: no humidity where e reaches p
            y0 = quotient(eps * e, v);
            break;
        }
        default: y0 = x[0]; break;
    }
}

// The stacks are contiguous runs of vectors (pitch is a multiple of the vector): a workgroup sweeps line-aligned chunks
// of kBlock * U vectors, U independent loads per operand and lane in flight (comb_unroll below; the n-ary sum keeps one:
// 8 operands x 4 vectors would not fit the register file).
// Vectors per lane in flight and grid shape, measured on 137-level O1280 stacks (profiles/r01_kernel_bench.log):
//   f32: 2 vectors per lane, one workgroup per chunk, no grid cap — difference 2->1 1.81 ms (4 per lane under a 65536-workgroup
//        cap: 2.11 ms; 1 per lane uncapped: 1.78 ms but cos+sin 1->2 slower), snow_cover 2.66 -> 2.17 ms;
//   f64: 4 vectors per lane under the 65536-workgroup cap — difference 3.63 ms (uncapped 1 / 2 per lane: 4.30 / 3.60 ms, the
//        transcendental operators lose 10 % uncapped).
#ifndef ATX_COMB_NT
#define ATX_COMB_NT 1  // round 3: non-temporal stores AND loads (ATX_COMB_NT_LOAD) — nothing is touched twice; profiles/r03_combine_ab.log
#endif

template <typename T, int N>
__device__ __forceinline__ void comb_store(T* p, const Pack<T, N>& v) {
#if ATX_COMB_NT
    typedef T NV __attribute__((ext_vector_type(N)));
    __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&v), reinterpret_cast<NV*>(p));
#else
    *reinterpret_cast<Pack<T, N>*>(p) = v;
#endif
}
template <typename T>
__device__ __forceinline__ void comb_store(T* p, const Pack<T, 1>& v) {
    *p = v.v[0];
}

// Launch shape knobs (A/B builds; tools/experiments/combine_ab.py): vectors per lane for 4- and 8-byte elements, the workgroup cap
// for 8-byte elements (0 = none: one workgroup per chunk), non-temporal loads.
#ifndef ATX_COMB_U_F32
#define ATX_COMB_U_F32 2
#endif
// round 3: 2 per lane and NO cap for float64 too (was 4 under a 65536-workgroup cap): the no-loop shape runs the same on every
// box, the capped sweep moved 3.46 <-> 3.82 ms (difference 2 -> 1) between two boxes of one afternoon
#ifndef ATX_COMB_U_F64
#define ATX_COMB_U_F64 2
#endif
#ifndef ATX_COMB_CAP_F64
#define ATX_COMB_CAP_F64 0
#endif
#ifndef ATX_COMB_NT_LOAD
#define ATX_COMB_NT_LOAD 1
#endif
#ifndef ATX_COMB_U_OPERA
#define ATX_COMB_U_OPERA 1  // vectors per lane of the 4- and 5-stream OPERA operators (0: as the other operators).  Measured on 137-level O1280
                            // stacks, 1 / 2 / 4 per lane: clipping 2->2 f32 0.748 / 0.702 / 0.754 of 8 TB/s, f64 0.814 / 0.752 / 0.810; preprocessing
                            // 3->2 f32 0.764 / 0.699 / 0.694, f64 0.795 / 0.744 / 0.791 (profiles/r04_opera_unroll.log)
#endif
constexpr int comb_unroll(int nin, int elem_bytes, int op) {
    if (ATX_COMB_U_OPERA > 0 && (op == ATX_COMB_OPERA_CLIP || op == ATX_COMB_OPERA_PREPROCESS)) return ATX_COMB_U_OPERA;
    return nin > 3 ? 1 : (elem_bytes == 4 ? ATX_COMB_U_F32 : ATX_COMB_U_F64);
}
constexpr int64_t comb_grid_cap(int elem_bytes, int op) { return (elem_bytes == 4 || ATX_COMB_CAP_F64 == 0) ? 0x7fffffffll : (int64_t)ATX_COMB_CAP_F64; }

template <typename T, int N>
__device__ __forceinline__ Pack<T, N> comb_load(const T* p) {
#if ATX_COMB_NT_LOAD
    if constexpr (N > 1) {
        typedef T NV __attribute__((ext_vector_type(N)));
        NV v = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
        return *reinterpret_cast<Pack<T, N>*>(&v);
    }
#endif
    return *reinterpret_cast<const Pack<T, N>*>(p);
}

// NIN: compile-time bound of the operand count (1, 2, 3 or ATX_COMB_MAX_INPUTS) so the operand registers are exactly as many as needed
template <typename T, int VEC, int NIN, int OP>
__global__ void __launch_bounds__(kBlock)
combine_kernel(CombArgs a, int flags, int n_in, int n_out, int64_t n_rows, int64_t row_len, int64_t pitch,
               int layout, int n_lev, const double* __restrict__ level_param) {
    using V = Pack<T, VEC>;
    const int vec_per_row = (int)(pitch / VEC);  // pitch % VEC == 0 on this path (else VEC == 1)
    const int64_t total = n_rows * vec_per_row;
    constexpr int U = comb_unroll(NIN, (int)sizeof(T), OP);
    constexpr int64_t kChunk = (int64_t)kBlock * U;
    constexpr bool kLevels = OP == ATX_COMB_W_TO_WZ || OP == ATX_COMB_WZ_TO_W || OP == ATX_COMB_OPERA_CLIP ||
                             OP == ATX_COMB_OPERA_PREPROCESS || OP == ATX_COMB_ORAS6 || OP == ATX_COMB_Q_TO_R ||
                             OP == ATX_COMB_R_TO_Q;  // the operators that read level_param[level] (when given one)
    // every operator but the n-ary sum and the two humidity conversions that take the pressure as an optional third field has a fixed
    // operand count, NIN == n_in — validated on the host: no run-time test per operand and element for them
    constexpr bool kFixedOperands = OP != ATX_COMB_SUM && OP != ATX_COMB_Q_TO_R && OP != ATX_COMB_R_TO_Q;
    constexpr bool kShared1 = OP == ATX_COMB_ORAS6;  // operand 1 is ONE field [n_pts] shared by every level, not a stack
    const bool small_rows = vec_per_row < (1 << 20);  // columns layout: (row, col) from 32-bit arithmetic
    const float inv_vec_per_row = __builtin_amdgcn_rcpf((float)vec_per_row);
    // a workgroup takes a contiguous run of chunks, not every gridDim.x-th one: under the 65536-workgroup cap the grid stride is a
    // power of two (1 GiB for f64) and drifting workgroups alias onto the same HBM channels (atx_pointwise.hip, ATX_PW_ASSIGN)
#ifndef ATX_COMB_ASSIGN
#define ATX_COMB_ASSIGN 0
#endif
#if ATX_COMB_ASSIGN == 1
    const int64_t n_chunks = (total + kChunk - 1) / kChunk;
    const int64_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
    const int64_t first = (int64_t)blockIdx.x * per * kChunk;
    const int64_t last = first + per * kChunk < total ? first + per * kChunk : total;
    for (int64_t base = first; base < last; base += kChunk) {
#else
    for (int64_t base = (int64_t)blockIdx.x * kChunk; base < total; base += (int64_t)gridDim.x * kChunk) {
#endif
        const int64_t row_b = base / vec_per_row;  // uniform
        const int col_b = (int)(base - row_b * vec_per_row);
        V x[U][NIN];
        int64_t vi[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vi[u] = base + u * kBlock + threadIdx.x;
            ok[u] = vi[u] < total;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (k < n_in && ok[u] && !(kShared1 && k == 1)) x[u][k] = comb_load<T, VEC>(static_cast<const T*>(a.in[k]) + vi[u] * VEC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            int64_t row, col;
            if (small_rows) {
                // (row, column) of the lane's vector from the workgroup's: off < vec_per_row + kBlock * U <= 2^20 + 512.  The quotient through the
                // float reciprocal — floor((off + 0.5) * rcp(vec_per_row)) — is EXACT there: off + 0.5 lies at least 0.5 / vec_per_row from the
                // nearest multiple of vec_per_row, and three roundings of 2^-24 (the hardware reciprocal: 2^-23) move the product by at most
                // (1 + 512.5 / vec_per_row) x 1.8e-7 — smaller for every vec_per_row below 2.8e6 (tests/test_host_api.py checks the arithmetic
                // over 6 000 divisors with the reciprocal off by an ulp either way).  6 instructions instead of the ~16 of an integer division,
                // per vector, in every multi-input operator.  ATX_COMB_INT_DIV=1 restores the division.
#ifndef ATX_COMB_INT_DIV
#define ATX_COMB_INT_DIV 0
#endif
                const int off = col_b + u * kBlock + threadIdx.x;
                const int dr = ATX_COMB_INT_DIV ? off / vec_per_row : (int)(((float)off + 0.5f) * inv_vec_per_row);
                row = row_b + dr;
                col = (int64_t)(off - dr * vec_per_row) * VEC;
            } else {
                row = vi[u] / vec_per_row;
                col = (vi[u] - row * vec_per_row) * VEC;
            }
            V y0, y1;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                T xe[ATX_COMB_MAX_INPUTS];
#pragma unroll
                for (int k = 0; k < ATX_COMB_MAX_INPUTS; ++k)
                    xe[k] = (k < NIN && (kFixedOperands || k < n_in)) ? x[u][k < NIN ? k : 0].v[e] : T(0);
                const bool live = (col + e) < row_len;
                if constexpr (kShared1) {
                    const int64_t point = layout == ATX_COLUMNS ? row : col + e;
                    xe[1] = live ? static_cast<const T*>(a.in[1])[point] : T(0);
                }
                T lv = T(0);
                if constexpr (kLevels) {
                    const int64_t level = layout == ATX_COLUMNS ? col + e : row;
                    if (level_param && level < n_lev) lv = static_cast<T>(level_param[level]);
                }
                T r0, r1;
                if constexpr (OP == ATX_COMB_LOOKUP) {
                    // level_param = (n, value of class 0, ..., value of class n-1): `param_dic[x][key]` for a class x that is one of the
                    // keys 0 .. n-1 (R: land_parameters.py:71); anything else — a fraction, a class beyond the table, NaN — is a KeyError
                    // there and a NaN here, which the caller counts
                    const int n_cls = (int)level_param[0];
                    const T c = xe[0];
                    const bool known = c >= T(0) && c < T(n_cls) && c == floor(c);
                    r0 = known ? static_cast<T>(level_param[1 + (int)c]) : quiet_nan<T>();
                    r1 = T(0);
                } else {
                    combine_one<T, OP>(flags, xe, n_in, lv, r0, r1);
                }
                y0.v[e] = live ? r0 : T(0);  // padding stays zero
                y1.v[e] = live ? r1 : T(0);
            }
            comb_store<T>(static_cast<T*>(a.out[0]) + vi[u] * VEC, y0);
            if (n_out > 1) comb_store<T>(static_cast<T*>(a.out[1]) + vi[u] * VEC, y1);
        }
    }
}

template <typename T>
static int combine_typed(const CombArgs& a, int op, int flags, int n_in, int n_out, int64_t n_pts, int n_lev, int64_t pitch,
                         int layout, const double* level_param, hipStream_t st) {
    constexpr int VEC = Vec16<T>::N;
    bool vec_ok = pitch % VEC == 0;
    for (int k = 0; k < n_in; ++k) vec_ok = vec_ok && aligned16(a.in[k]);
    for (int k = 0; k < n_out; ++k) vec_ok = vec_ok && aligned16(a.out[k]);
    const int64_t n_rows = layout == ATX_COLUMNS ? n_pts : n_lev;
    const int64_t row_len = layout == ATX_COLUMNS ? n_lev : n_pts;
    const int per_block = kBlock * comb_unroll(n_in <= 3 ? n_in : ATX_COMB_MAX_INPUTS, (int)sizeof(T), op);
    int64_t blocks = (n_rows * (pitch / (vec_ok ? VEC : 1)) + per_block - 1) / per_block;
    if (blocks > comb_grid_cap((int)sizeof(T), op)) blocks = comb_grid_cap((int)sizeof(T), op);
    if (blocks < 1) blocks = 1;
#define ATX_COMB_LAUNCH(V_, N_, OP_)                                                                                            \
    hipLaunchKernelGGL((combine_kernel<T, V_, N_, OP_>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a, flags, n_in, n_out, n_rows, \
                       row_len, pitch, layout, n_lev, level_param)
#define ATX_COMB_CASE(OP_, N_)                 \
    case OP_:                                  \
        if (vec_ok) ATX_COMB_LAUNCH(VEC, N_, OP_); \
        else ATX_COMB_LAUNCH(1, N_, OP_);      \
        break
    switch (op) {  // the operand count of every operator but the n-ary sum is fixed (validated by the caller)
        ATX_COMB_CASE(ATX_COMB_SNOW_DEPTH_M, 2);
        ATX_COMB_CASE(ATX_COMB_SNOW_COVER, 2);
        ATX_COMB_CASE(ATX_COMB_COS_SIN, 1);
        ATX_COMB_CASE(ATX_COMB_ATAN2, 2);
        ATX_COMB_CASE(ATX_COMB_W_TO_WZ, 3);
        ATX_COMB_CASE(ATX_COMB_WZ_TO_W, 3);
        ATX_COMB_CASE(ATX_COMB_SUB, 2);
        ATX_COMB_CASE(ATX_COMB_XY_TO_POLAR, 2);
        ATX_COMB_CASE(ATX_COMB_POLAR_TO_XY, 2);
        ATX_COMB_CASE(ATX_COMB_OPERA_CLIP, 2);
        ATX_COMB_CASE(ATX_COMB_OPERA_PREPROCESS, 3);
        ATX_COMB_CASE(ATX_COMB_ORAS6, 2);
        ATX_COMB_CASE(ATX_COMB_LOOKUP, 1);
        ATX_COMB_CASE(ATX_COMB_R_TO_D, 2);
        ATX_COMB_CASE(ATX_COMB_D_TO_R, 2);
        ATX_COMB_CASE(ATX_COMB_Q_TO_R, 3);
        ATX_COMB_CASE(ATX_COMB_R_TO_Q, 3);
        default:  // ATX_COMB_SUM
            if (vec_ok) {
                if (n_in <= 1) ATX_COMB_LAUNCH(VEC, 1, ATX_COMB_SUM);
                else if (n_in == 2) ATX_COMB_LAUNCH(VEC, 2, ATX_COMB_SUM);
                else if (n_in == 3) ATX_COMB_LAUNCH(VEC, 3, ATX_COMB_SUM);
                else ATX_COMB_LAUNCH(VEC, ATX_COMB_MAX_INPUTS, ATX_COMB_SUM);
            } else {
                if (n_in <= 1) ATX_COMB_LAUNCH(1, 1, ATX_COMB_SUM);
                else if (n_in == 2) ATX_COMB_LAUNCH(1, 2, ATX_COMB_SUM);
                else if (n_in == 3) ATX_COMB_LAUNCH(1, 3, ATX_COMB_SUM);
                else ATX_COMB_LAUNCH(1, ATX_COMB_MAX_INPUTS, ATX_COMB_SUM);
            }
            break;
    }
#undef ATX_COMB_CASE
#undef ATX_COMB_LAUNCH
    ATX_LAUNCH_CHECK("combine_stack");
    return ATX_OK;
}

}  // namespace atx

using namespace atx;

extern "C" int atx_combine_stack(int op, const void* const* inputs, int32_t n_in, void* const* outputs, int32_t n_out,
                                 int64_t n_pts, int64_t n_lev, int64_t pitch, int dtype, int layout,
                                 const double* level_param, int32_t flags, void* stream) {
    static const int kIn[ATX_COMB_COUNT_] = {2, 2, 1, 2, 3, 3, -1, 2, 2, 2, 2, 3, 2, 1, 2, 2, -2, -2};  // -1: any number, -2: two or three
    static const int kOut[ATX_COMB_COUNT_] = {1, 1, 2, 1, 1, 1, 1, 1, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1};
    ATX_REQUIRE(op >= 0 && op < ATX_COMB_COUNT_, ATX_EINVAL, "atx_combine_stack: bad operator %d", op);
    ATX_REQUIRE(inputs && outputs, ATX_EINVAL, "atx_combine_stack: null pointer table");
    ATX_REQUIRE(n_in >= 1 && n_in <= ATX_COMB_MAX_INPUTS, ATX_EINVAL, "atx_combine_stack: n_in=%d outside [1, %d]", n_in, ATX_COMB_MAX_INPUTS);
    ATX_REQUIRE(kIn[op] < 0 || kIn[op] == n_in, ATX_EINVAL, "atx_combine_stack: operator %d takes %d inputs, got %d", op, kIn[op], n_in);
    ATX_REQUIRE(kIn[op] != -2 || n_in == 2 || n_in == 3, ATX_EINVAL, "atx_combine_stack: operator %d takes 2 or 3 inputs, got %d", op, n_in);
    ATX_REQUIRE(kOut[op] == n_out, ATX_EINVAL, "atx_combine_stack: operator %d gives %d outputs, got %d", op, kOut[op], n_out);
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "atx_combine_stack: bad dtype %d", dtype);
    ATX_REQUIRE(layout == ATX_COLUMNS || layout == ATX_FIELDS, ATX_EINVAL, "atx_combine_stack: bad layout %d", layout);
    ATX_REQUIRE(n_pts >= 0 && n_lev > 0 && n_lev <= INT32_MAX, ATX_EINVAL, "atx_combine_stack: bad sizes");
    ATX_REQUIRE(pitch >= (layout == ATX_COLUMNS ? n_lev : n_pts), ATX_ESHAPE, "atx_combine_stack: pitch %lld too small", (long long)pitch);
    const bool reads_param = op == ATX_COMB_W_TO_WZ || op == ATX_COMB_WZ_TO_W || (op >= ATX_COMB_OPERA_CLIP && op <= ATX_COMB_LOOKUP) ||
                             ((op == ATX_COMB_Q_TO_R || op == ATX_COMB_R_TO_Q) && n_in == 2);
    ATX_REQUIRE(level_param || !reads_param, ATX_EINVAL, "atx_combine_stack: operator %d needs level_param", op);
    CombArgs a{};
    for (int k = 0; k < n_in; ++k) {
        ATX_REQUIRE(inputs[k] || n_pts == 0, ATX_EINVAL, "atx_combine_stack: null input %d", k);  // (an empty stack may have no storage)
        a.in[k] = inputs[k];
    }
    for (int k = 0; k < n_out; ++k) {
        ATX_REQUIRE(outputs[k] || n_pts == 0, ATX_EINVAL, "atx_combine_stack: null output %d", k);
        a.out[k] = outputs[k];
    }
    if (n_pts == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ATX_F32) return combine_typed<float>(a, op, flags, n_in, n_out, n_pts, (int)n_lev, pitch, layout, level_param, s);
    return combine_typed<double>(a, op, flags, n_in, n_out, n_pts, (int)n_lev, pitch, layout, level_param, s);
}

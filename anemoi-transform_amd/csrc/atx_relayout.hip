// Layout conversion between field-major (ATX_FIELDS, the reference's unit:
// `field.to_numpy()`, R: fields.py:178-202) and column stacks (ATX_COLUMNS, the
// engine's native HBM layout).
//
// A workgroup moves TP points x LC levels through LDS.  On the FIELDS side a wave
// touches 256 contiguous bytes of one level (35 KB LDS tiles, 4 workgroups per CU: measured best;
// 128-B tiles -5 %, 512-B tiles -30 %); on the COLUMNS side the TP columns of
// the tile form one contiguous TP*pitch run, swept by consecutive lanes.  The LDS
// tile is [point][level] with an odd row length (in 4-byte words for f32) so both
// phases are bank-conflict free for f32 and at most 2-way for f64.
#include "atx_common.hpp"

namespace atx {

// global accesses of the transposes: every element is read once and written once.  Non-temporal accesses measured in round 3
// (profiles/r03_relayout_nt_experiment.log): towards columns +6 % f32 / +3 % f64 (1.32 -> 1.25 ms, 2.64 -> 2.55 ms), towards fields
// -10 % / -2 % (that direction runs on 4-byte accesses) — so they are used towards columns only.
#ifndef ATX_TP_NT
#define ATX_TP_NT 1
#endif
template <bool NT, typename X>
__device__ __forceinline__ X tp_load(const X* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename X>
__device__ __forceinline__ void tp_store(X* p, X v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT, typename T, int N>
__device__ __forceinline__ Pack<T, N> tp_load_vec(const T* p) {
    if constexpr (NT && N > 1) {
        typedef T NV __attribute__((ext_vector_type(N)));
        NV v = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
        return *reinterpret_cast<Pack<T, N>*>(&v);
    } else {
        return *reinterpret_cast<const Pack<T, N>*>(p);
    }
}
template <bool NT, typename T, int N>
__device__ __forceinline__ void tp_store_vec(T* p, const Pack<T, N>& v) {
    if constexpr (NT && N > 1) {
        typedef T NV __attribute__((ext_vector_type(N)));
        __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&v), reinterpret_cast<NV*>(p));
    } else {
        *reinterpret_cast<Pack<T, N>*>(p) = v;
    }
}

// (p, l) of item i = p * nl + l advanced by kBlock items without a division per element (nl is a run-time 137: the quotient and
// remainder of every element cost more issue slots than the 4-byte load they address — round 4: columns -> fields f32 1.60 -> see
// profiles/r04_relayout_index_math.log).  TPC is the tile's point count when it is a compile-time power of two (0: run-time TP).
template <typename T, bool TO_COLUMNS, int TPC>
__global__ void __launch_bounds__(kBlock)
transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t n_pts, int n_lev,
                 int64_t src_pitch, int64_t dst_pitch, int TP_, int LC, int LCpad) {
    extern __shared__ __align__(16) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);
    const int TP = TPC > 0 ? TPC : TP_;
    const int64_t p0 = (int64_t)blockIdx.x * TP;
    const int l0 = blockIdx.y * LC;
    const int np = (int)min((int64_t)TP, n_pts - p0);
    const int nl = min(LC, n_lev - l0);
    const int tid = threadIdx.x;
    // steps of kBlock items in the (point, level) order of the columns side
    const int dq = kBlock / nl, dr = kBlock - dq * nl;

    // fields side: element (p, l) at base[l*pitch + p]; columns side: base[p*pitch + l]
    if (TO_COLUMNS) {
#pragma unroll 4
        for (int i = tid; i < nl * TP; i += kBlock) {
            const int l = i / TP, p = i - l * TP;  // (TPC > 0: a shift and a mask)
            if (p < np) tile[p * LCpad + l] = tp_load<TO_COLUMNS && ATX_TP_NT>(src + (int64_t)(l0 + l) * src_pitch + p0 + p);
        }
        __syncthreads();
        int p = tid / nl, l = tid - p * nl;
        for (int i = tid; i < np * nl; i += kBlock) {
            tp_store<TO_COLUMNS && ATX_TP_NT>(dst + (p0 + p) * dst_pitch + l0 + l, tile[p * LCpad + l]);
            p += dq;
            l += dr;
            if (l >= nl) {
                l -= nl;
                ++p;
            }
        }
    } else {
        int p = tid / nl, l = tid - p * nl;
#pragma unroll 4
        for (int i = tid; i < np * nl; i += kBlock) {
            tile[p * LCpad + l] = tp_load<TO_COLUMNS && ATX_TP_NT>(src + (p0 + p) * src_pitch + l0 + l);
            p += dq;
            l += dr;
            if (l >= nl) {
                l -= nl;
                ++p;
            }
        }
        __syncthreads();
        for (int i = tid; i < nl * TP; i += kBlock) {
            const int lv = i / TP, pp = i - lv * TP;
            if (pp < np) tp_store<TO_COLUMNS && ATX_TP_NT>(dst + (int64_t)(l0 + lv) * dst_pitch + p0 + pp, tile[pp * LCpad + lv]);
        }
    }
}

// The same tile with 16-byte accesses on BOTH sides (needs 16-byte aligned bases and pitches that are multiples of a
// vector on both sides, and whole vectors inside the column pitch): on the fields side a lane moves VEC consecutive
// points of one level, on the columns side VEC consecutive levels of one point; LDS accesses stay scalar.
template <typename T, bool TO_COLUMNS>
__global__ void __launch_bounds__(kBlock)
transpose_vec_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t n_pts, int n_lev,
                     int64_t src_pitch, int64_t dst_pitch, int TP, int LC, int LCpad) {
    constexpr int VEC = Vec16<T>::N;
    using V = Pack<T, VEC>;
    extern __shared__ __align__(16) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);
    const int64_t p0 = (int64_t)blockIdx.x * TP;
    const int l0 = blockIdx.y * LC;
    const int np = (int)min((int64_t)TP, n_pts - p0);
    const int nl = min(LC, n_lev - l0);
    const int tid = threadIdx.x;
    const int PV = TP / VEC;               // point vectors per level row of the tile
    const int CV = (nl + VEC - 1) / VEC;   // level vectors per column of the tile (l0 is a multiple of VEC)
    const T* fields = TO_COLUMNS ? src : dst;
    const int64_t fields_pitch = TO_COLUMNS ? src_pitch : dst_pitch;
    const int64_t cols_pitch = TO_COLUMNS ? dst_pitch : src_pitch;
    (void)fields;

    if (TO_COLUMNS) {
        for (int i = tid; i < nl * PV; i += kBlock) {
            const int l = i / PV, pv = i - l * PV;
            const int p = pv * VEC;
            if (p + VEC <= np) {
                const V v = tp_load_vec<TO_COLUMNS && ATX_TP_NT, T, VEC>(src + (int64_t)(l0 + l) * fields_pitch + p0 + p);
#pragma unroll
                for (int e = 0; e < VEC; ++e) tile[(p + e) * LCpad + l] = v.v[e];
            } else {
                for (int e = 0; e < VEC; ++e)
                    if (p + e < np) tile[(p + e) * LCpad + l] = src[(int64_t)(l0 + l) * fields_pitch + p0 + p + e];
            }
        }
        __syncthreads();
        for (int i = tid; i < np * CV; i += kBlock) {
            const int p = i / CV, c = i - p * CV;
            V v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = (c * VEC + e < nl) ? tile[p * LCpad + c * VEC + e] : T(0);
            tp_store_vec<TO_COLUMNS && ATX_TP_NT, T, VEC>(dst + (p0 + p) * cols_pitch + l0 + c * VEC, v);
        }
    } else {
        for (int i = tid; i < np * CV; i += kBlock) {
            const int p = i / CV, c = i - p * CV;
            const V v = tp_load_vec<TO_COLUMNS && ATX_TP_NT, T, VEC>(src + (p0 + p) * cols_pitch + l0 + c * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (c * VEC + e < nl) tile[p * LCpad + c * VEC + e] = v.v[e];
        }
        __syncthreads();
        for (int i = tid; i < nl * PV; i += kBlock) {
            const int l = i / PV, pv = i - l * PV;
            const int p = pv * VEC;
            if (p + VEC <= np) {
                V v;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v.v[e] = tile[(p + e) * LCpad + l];
                tp_store_vec<TO_COLUMNS && ATX_TP_NT, T, VEC>(dst + (int64_t)(l0 + l) * fields_pitch + p0 + p, v);
            } else {
                for (int e = 0; e < VEC; ++e)
                    if (p + e < np) dst[(int64_t)(l0 + l) * fields_pitch + p0 + p + e] = tile[(p + e) * LCpad + l];
            }
        }
    }
}

// same layout on both sides: pitched copy of rows of `row_len` elements
template <typename T>
__global__ void __launch_bounds__(kBlock)
pitched_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t n_rows, int64_t row_len,
                    int64_t src_pitch, int64_t dst_pitch) {
    const int64_t total = n_rows * row_len;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / row_len, c = i - r * row_len;
        dst[r * dst_pitch + c] = src[r * src_pitch + c];
    }
}

template <typename T>
static int relayout_typed(const void* src_, void* dst_, int64_t n_pts, int n_lev, int64_t sp, int64_t dp,
                          int src_layout, int dst_layout, hipStream_t st) {
    const T* src = static_cast<const T*>(src_);
    T* dst = static_cast<T*>(dst_);
    if (src_layout == dst_layout) {
        const int64_t n_rows = src_layout == ATX_COLUMNS ? n_pts : n_lev;
        const int64_t row_len = src_layout == ATX_COLUMNS ? n_lev : n_pts;
        int64_t blocks = (n_rows * row_len + kBlock - 1) / kBlock;
        if (blocks > kStreamGrid) blocks = kStreamGrid;
        hipLaunchKernelGGL(pitched_copy_kernel<T>, dim3((unsigned)blocks), dim3(kBlock), 0, st, src, dst, n_rows, row_len, sp, dp);
        ATX_LAUNCH_CHECK("pitched_copy");
        return ATX_OK;
    }
#ifndef ATX_TP_BYTES
#define ATX_TP_BYTES 256  // re-measured with the vector kernel (137 levels of O1280, ms c->f / f->c): 128 B: 1.65 / 1.47 f32, 2.69 / 2.92 f64;
#endif                    // 256 B: 1.59-1.66 / 1.32-1.44 f32, 2.52 / 2.66 f64; 384 B and 512 B slower (LDS tiles cut the occupancy)
    const int TP = ATX_TP_BYTES / (int)sizeof(T);  // contiguous bytes per level on the fields side
#ifndef ATX_TP_LC
#define ATX_TP_LC 0  // levels per tile; 0: all of them up to 160, else 128
#endif
    int LC = ATX_TP_LC > 0 ? (n_lev < ATX_TP_LC ? n_lev : ATX_TP_LC) : (n_lev < 160 ? n_lev : 128);
    const int LCpad = LC | 1;
    const size_t lds = (size_t)TP * LCpad * sizeof(T);
    const unsigned gx = (unsigned)((n_pts + TP - 1) / TP);
    const unsigned gy = (unsigned)((n_lev + LC - 1) / LC);
    ATX_REQUIRE(gy <= 65535, ATX_ENOTIMPL, "atx_relayout: too many level chunks");
#ifndef ATX_TP_VEC
#define ATX_TP_VEC 1
#endif
    {
        constexpr int VEC = Vec16<T>::N;
        const int64_t cols_pitch = dst_layout == ATX_COLUMNS ? dp : sp, fields_pitch = dst_layout == ATX_COLUMNS ? sp : dp;
        const int64_t covered = ((int64_t)(n_lev + VEC - 1) / VEC) * VEC;
        const bool vec = ATX_TP_VEC && aligned16(src_) && aligned16(dst_) && cols_pitch % VEC == 0 && fields_pitch % VEC == 0 &&
                         covered <= cols_pitch && (gy == 1 || LC % VEC == 0) && TP % VEC == 0;
        // measured (137 levels of O1280): towards columns 2.14 -> 1.44 ms f32, 3.20 -> 2.89 ms f64; towards fields the scalar
        // kernel is as fast (f32) or faster (f64: 2.69 vs 3.16 ms), so only the columns direction takes the vector kernel
        if (vec && dst_layout == ATX_COLUMNS) {
            hipLaunchKernelGGL((transpose_vec_kernel<T, true>), dim3(gx, gy), dim3(kBlock), lds, st, src, dst, n_pts, n_lev, sp, dp, TP, LC, LCpad);
            ATX_LAUNCH_CHECK("transpose_vec");
            return ATX_OK;
        }
    }
    constexpr int kTileBytes = ATX_TP_BYTES;
    constexpr int TPC = (kTileBytes / (int)sizeof(T)) > 0 && ((kTileBytes / (int)sizeof(T)) & ((kTileBytes / (int)sizeof(T)) - 1)) == 0 ? kTileBytes / (int)sizeof(T) : 0;
    if (dst_layout == ATX_COLUMNS)
        hipLaunchKernelGGL((transpose_kernel<T, true, TPC>), dim3(gx, gy), dim3(kBlock), lds, st, src, dst, n_pts, n_lev, sp, dp, TP, LC, LCpad);
    else
        hipLaunchKernelGGL((transpose_kernel<T, false, TPC>), dim3(gx, gy), dim3(kBlock), lds, st, src, dst, n_pts, n_lev, sp, dp, TP, LC, LCpad);
    ATX_LAUNCH_CHECK("transpose");
    return ATX_OK;
}

// ---- level selection: dst level j = src level map[j] -------------------------------------------------
// The map of one launch travels by value in the kernel arguments (no device allocation, validated on the host).
constexpr int kSelChunk = 256;
struct LevelMap {
    int32_t src_level[kSelChunk];
};

// columns: a workgroup copies `tp` points x `nj` destination levels; consecutive lanes write consecutive levels
template <typename T>
__global__ void __launch_bounds__(kBlock)
select_cols_kernel(const T* __restrict__ src, T* __restrict__ dst, LevelMap map, int j0, int nj, int64_t n_pts,
                   int64_t src_pitch, int64_t dst_pitch, int tp) {
    __shared__ int32_t lm[kSelChunk];
    for (int i = threadIdx.x; i < nj; i += kBlock) lm[i] = map.src_level[i];
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * tp;
    const int np = (int)min((int64_t)tp, n_pts - p0);
    // (point, level) advanced by kBlock items without a division per element (nj is a run-time count)
    const int dq = kBlock / nj, dr = kBlock - dq * nj;
    int p = threadIdx.x / nj, j = threadIdx.x - p * nj;
#pragma unroll 4
    for (int i = threadIdx.x; i < np * nj; i += kBlock) {
        const int32_t l = lm[j];
        if (l >= 0) dst[(p0 + p) * dst_pitch + j0 + j] = src[(p0 + p) * src_pitch + l];
        p += dq;
        j += dr;
        if (j >= nj) {
            j -= nj;
            ++p;
        }
    }
}

// columns, 16-byte accesses on both sides (round 4): the levels [lo, lo + W) of the tile's columns — the range the map reads — go
// through LDS as they lie (16-byte loads of a strided slab), every destination vector is assembled from LDS through the map and
// stored with ONE 16-byte store.  The element-by-element kernel above issues a 4-byte load and a 4-byte store per element with a
// map look-up in between: a full permutation of 137 levels ran at 0.47 (f32) / 0.53 (f64) of the HBM peak where a copy runs at 0.8
// (profiles/r04_select_levels.log).  Vectors with an untouched (negative) or missing entry fall back to element stores.
template <typename T>
__global__ void __launch_bounds__(kBlock)
select_cols_slab_kernel(const T* __restrict__ src, T* __restrict__ dst, LevelMap map, int j0, int nj, int64_t n_pts,
                        int64_t src_pitch, int64_t dst_pitch, int tp, int lo, int W, int wpad) {
    constexpr int VEC = Vec16<T>::N;
    using V = Pack<T, VEC>;
    extern __shared__ __align__(16) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);                                  // [tp][wpad]
    int32_t* lm = reinterpret_cast<int32_t*>(tile + (size_t)tp * wpad);    // [nj], relative to lo
    for (int i = threadIdx.x; i < nj; i += kBlock) lm[i] = map.src_level[i] >= 0 ? map.src_level[i] - lo : -1;
    const int64_t p0 = (int64_t)blockIdx.x * tp;
    const int np = (int)min((int64_t)tp, n_pts - p0);
    {   // phase 1: the slab, WV vectors per column
        const int WV = W / VEC;
        const int dq = kBlock / WV, dr = kBlock - dq * WV;
        int p = threadIdx.x / WV, c = threadIdx.x - p * WV;
#pragma unroll 4
        for (int i = threadIdx.x; i < np * WV; i += kBlock) {
            *reinterpret_cast<V*>(tile + (size_t)p * wpad + c * VEC) = *reinterpret_cast<const V*>(src + (p0 + p) * src_pitch + lo + c * VEC);
            p += dq;
            c += dr;
            if (c >= WV) {
                c -= WV;
                ++p;
            }
        }
    }
    __syncthreads();
    {   // phase 2: destination vectors
        const int CV = (nj + VEC - 1) / VEC;
        const int dq = kBlock / CV, dr = kBlock - dq * CV;
        int p = threadIdx.x / CV, c = threadIdx.x - p * CV;
        for (int i = threadIdx.x; i < np * CV; i += kBlock) {
            V v;
            bool whole = true;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int j = c * VEC + e;
                const int32_t l = j < nj ? lm[j] : -1;
                whole = whole && l >= 0;
                v.v[e] = l >= 0 ? tile[(size_t)p * wpad + l] : T(0);
            }
            T* d = dst + (p0 + p) * dst_pitch + j0 + c * VEC;
            if (whole) {
                *reinterpret_cast<V*>(d) = v;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int j = c * VEC + e;
                    if (j < nj && lm[j] >= 0) d[e] = v.v[e];
                }
            }
            p += dq;
            c += dr;
            if (c >= CV) {
                c -= CV;
                ++p;
            }
        }
    }
}

// fields: grid.y = destination level, every level one contiguous row
template <typename T>
__global__ void __launch_bounds__(kBlock)
select_fields_kernel(const T* __restrict__ src, T* __restrict__ dst, LevelMap map, int j0, int64_t n_pts,
                     int64_t src_pitch, int64_t dst_pitch) {
    const int32_t l = map.src_level[blockIdx.y];
    if (l < 0) return;
    const T* s = src + (int64_t)l * src_pitch;
    T* d = dst + (int64_t)(j0 + blockIdx.y) * dst_pitch;
    // a row copy: 16 bytes per lane when both rows start on a 16-byte boundary (4-byte accesses ran a float32 row at 0.62 of the HBM
    // peak against 0.74 for float64; round 4), the <= VEC - 1 trailing points element by element
    constexpr int VEC = Vec16<T>::N;
    using V = Pack<T, VEC>;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
        const int64_t nv = n_pts / VEC;
        for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (int64_t)gridDim.x * kBlock)
            tp_store_vec<true, T, VEC>(d + v * VEC, tp_load_vec<true, T, VEC>(s + v * VEC));  // read once, written once: non-temporal
        for (int64_t p = nv * VEC + (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_pts; p += (int64_t)gridDim.x * kBlock) d[p] = s[p];
        return;
    }
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_pts; p += (int64_t)gridDim.x * kBlock) d[p] = s[p];
}

template <typename T>
static int select_typed(const void* src_, void* dst_, const int32_t* level_map, int n_map, int64_t n_pts, int64_t sp,
                        int64_t dp, int layout, hipStream_t st) {
    const T* src = static_cast<const T*>(src_);
    T* dst = static_cast<T*>(dst_);
    for (int j0 = 0; j0 < n_map; j0 += kSelChunk) {
        const int nj = n_map - j0 < kSelChunk ? n_map - j0 : kSelChunk;
        LevelMap map;
        bool any = false;
        for (int j = 0; j < kSelChunk; ++j) {
            map.src_level[j] = j < nj ? level_map[j0 + j] : -1;
            any = any || map.src_level[j] >= 0;
        }
        if (!any) continue;
        if (layout == ATX_COLUMNS) {
            // the slab kernel when both stacks take 16-byte accesses and the range of source levels the map reads is not much wider
            // than the 128-byte lines the selected levels occupy anyway (a column is fetched in whole lines whatever is used of them)
            constexpr int VEC = Vec16<T>::N;
            int lmin = INT32_MAX, lmax = -1;
            bool line_used[512] = {false};
            int lines = 0;
            for (int j = 0; j < nj; ++j) {
                const int l = map.src_level[j];
                if (l < 0) continue;
                lmin = l < lmin ? l : lmin;
                lmax = l > lmax ? l : lmax;
                const int slot = (int)(((int64_t)l * (int64_t)sizeof(T)) / 128);
                if (slot < 512 && !line_used[slot]) {
                    line_used[slot] = true;
                    ++lines;
                }
            }
            const int lo = lmin / VEC * VEC;
            const int W = (lmax + VEC) / VEC * VEC - lo;
            const bool aligned = aligned16(src) && aligned16(dst) && sp % VEC == 0 && dp % VEC == 0 && (int64_t)lo + W <= sp && j0 % VEC == 0 &&
                                 (int64_t)j0 + ((nj + VEC - 1) / VEC) * VEC <= dp;
#ifndef ATX_SELECT_SLAB
#define ATX_SELECT_SLAB 1
#endif
            if (ATX_SELECT_SLAB && aligned && lines > 0 && (int64_t)W * (int64_t)sizeof(T) * 10 <= (int64_t)lines * 128 * 13) {
                const int wpad = (W % 32 == 0) ? W + VEC : W;  // rows a multiple of 128 bytes apart would put a column of the tile into one LDS bank
                int tps = (int)((32 * 1024 - (size_t)nj * sizeof(int32_t)) / ((size_t)wpad * sizeof(T)));
                tps = tps > 64 ? 64 : (tps < 1 ? 1 : tps);
                const size_t lds = (size_t)tps * wpad * sizeof(T) + (size_t)nj * sizeof(int32_t);
                if (lds <= 48 * 1024) {
                    const unsigned gxs = (unsigned)((n_pts + tps - 1) / tps);
                    hipLaunchKernelGGL(select_cols_slab_kernel<T>, dim3(gxs), dim3(kBlock), lds, st, src, dst, map, j0, nj, n_pts, sp, dp, tps, lo, W, wpad);
                    ATX_LAUNCH_CHECK("select_levels_slab");
                    continue;
                }
            }
            int tp = 4096 / nj;  // ~16 items per lane
            tp = tp < 1 ? 1 : tp;
            const unsigned gx = (unsigned)((n_pts + tp - 1) / tp);
            hipLaunchKernelGGL(select_cols_kernel<T>, dim3(gx), dim3(kBlock), 0, st, src, dst, map, j0, nj, n_pts, sp, dp, tp);
        } else {
            const int64_t per_block = (int64_t)kBlock * 2 * Vec16<T>::N;  // two 16-byte vectors per lane: many short workgroups
            int64_t gx = (n_pts + per_block - 1) / per_block;
            gx = gx < 1 ? 1 : (gx > kStreamGrid ? kStreamGrid : gx);
            hipLaunchKernelGGL(select_fields_kernel<T>, dim3((unsigned)gx, (unsigned)nj), dim3(kBlock), 0, st, src, dst, map, j0, n_pts, sp, dp);
        }
        ATX_LAUNCH_CHECK("select_levels");
    }
    return ATX_OK;
}

}  // namespace atx

using namespace atx;

extern "C" int atx_select_levels(const void* src, void* dst, const int32_t* level_map, int32_t n_map, int64_t n_pts,
                                 int64_t n_src_lev, int64_t src_pitch, int64_t dst_pitch, int dtype, int layout, void* stream) {
    ATX_REQUIRE(((src && dst) || n_pts == 0 || n_map == 0) && (level_map || n_map == 0), ATX_EINVAL, "atx_select_levels: null pointer");  // (an empty stack may have no storage)
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "atx_select_levels: bad dtype %d", dtype);
    ATX_REQUIRE(layout == ATX_COLUMNS || layout == ATX_FIELDS, ATX_EINVAL, "atx_select_levels: bad layout %d", layout);
    ATX_REQUIRE(n_pts >= 0 && n_map >= 0 && n_src_lev > 0, ATX_EINVAL, "atx_select_levels: bad sizes");
    ATX_REQUIRE(src_pitch >= (layout == ATX_COLUMNS ? n_src_lev : n_pts), ATX_ESHAPE, "atx_select_levels: src pitch %lld too small", (long long)src_pitch);
    ATX_REQUIRE(dst_pitch >= (layout == ATX_COLUMNS ? (int64_t)n_map : n_pts), ATX_ESHAPE, "atx_select_levels: dst pitch %lld too small", (long long)dst_pitch);
    for (int32_t j = 0; j < n_map; ++j)
        ATX_REQUIRE(level_map[j] < n_src_lev, ATX_EINVAL, "atx_select_levels: level_map[%d] = %d but the source has %lld levels",
                    j, level_map[j], (long long)n_src_lev);
    if (n_pts == 0 || n_map == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ATX_F32) return select_typed<float>(src, dst, level_map, n_map, n_pts, src_pitch, dst_pitch, layout, s);
    return select_typed<double>(src, dst, level_map, n_map, n_pts, src_pitch, dst_pitch, layout, s);
}

extern "C" int atx_relayout(const void* src, void* dst, int64_t n_pts, int64_t n_lev, int64_t src_pitch,
                            int64_t dst_pitch, int src_layout, int dst_layout, int dtype, void* stream) {
    ATX_REQUIRE((src && dst) || n_pts == 0, ATX_EINVAL, "atx_relayout: null pointer");  // (an empty stack may have no storage)
    ATX_REQUIRE(src != dst || n_pts == 0, ATX_EINVAL, "atx_relayout: in-place relayout is not supported");
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "atx_relayout: bad dtype %d", dtype);
    ATX_REQUIRE((src_layout == ATX_COLUMNS || src_layout == ATX_FIELDS) && (dst_layout == ATX_COLUMNS || dst_layout == ATX_FIELDS),
                ATX_EINVAL, "atx_relayout: bad layout (%d, %d)", src_layout, dst_layout);
    ATX_REQUIRE(n_pts >= 0 && n_lev > 0 && n_lev <= INT32_MAX, ATX_EINVAL, "atx_relayout: bad sizes");
    ATX_REQUIRE(src_pitch >= (src_layout == ATX_COLUMNS ? n_lev : n_pts), ATX_ESHAPE, "atx_relayout: src pitch %lld too small", (long long)src_pitch);
    ATX_REQUIRE(dst_pitch >= (dst_layout == ATX_COLUMNS ? n_lev : n_pts), ATX_ESHAPE, "atx_relayout: dst pitch %lld too small", (long long)dst_pitch);
    if (n_pts == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ATX_F32) return relayout_typed<float>(src, dst, n_pts, (int)n_lev, src_pitch, dst_pitch, src_layout, dst_layout, s);
    return relayout_typed<double>(src, dst, n_pts, (int)n_lev, src_pitch, dst_pitch, src_layout, dst_layout, s);
}

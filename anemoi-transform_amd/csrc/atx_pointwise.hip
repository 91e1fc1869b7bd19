// Per-point transforms over a stack of levels: the per-level program evaluator.
//
// Replaces the per-field Python map of the reference
//   R: filter.py:188-196        SingleFieldFilter._map_transform / forward
//   R: filters/fields/rescale.py:25,28   x*scale+offset, (x-offset)/scale
//   R: filters/fields/orog_to_z.py:59,77 x*g, x/g
//   R: filters/fields/clipper.py:69, impute_nans.py:53-54, lnsp_to_sp.py:47,65
//   R: filters/fields/apply_mask.py:183-185, glacier_mask.py:33  values[mask] = nan
// by ONE streaming pass over the stack: level l gets prog[s][l] for each stage s.
// HBM-bound: 16-byte loads/stores, grid capped at 8 workgroups per CU and
// grid-strided; levels whose program is all-COPY are not touched when the
// operation is in place.
#include "atx_common.hpp"

namespace atx {

constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 workgroups

// ATX_COLUMNS: item = (point p, vector c); consecutive lanes = consecutive 16 B.
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_pts, int n_lev, int C,
                      int64_t x_pitch, int64_t y_pitch, const atx_level_op* __restrict__ prog, int n_stage,
                      const uint8_t* __restrict__ point_mask, int in_place) {
    using V = Pack<T, VEC>;
    extern __shared__ __align__(16) unsigned char smem[];
    LevelOp<T>* prog_s = reinterpret_cast<LevelOp<T>*>(smem);
    const int n_slots = C * VEC;
    uint8_t* active_s = reinterpret_cast<uint8_t*>(prog_s + (size_t)n_stage * n_slots);  // per vector column
    const int tid = threadIdx.x;
    for (int i = tid; i < n_stage * n_slots; i += kBlock) {
        const int s = i / n_slots, l = i - s * n_slots;
        LevelOp<T> o;
        if (l < n_lev) {
            o = load_level_op<T>(prog, (int64_t)s * n_lev + l);
        } else {
            o.op = ATX_OP_COPY; o.use_mask = 0; o.p0 = 0; o.p1 = 0;
        }
        prog_s[i] = o;
    }
    __syncthreads();
    for (int c = tid; c < C; c += kBlock) {
        bool act = false;
        for (int s = 0; s < n_stage; ++s)
            for (int e = 0; e < VEC; ++e) {
                const LevelOp<T>& o = prog_s[s * n_slots + c * VEC + e];
                act = act || (o.op != ATX_OP_COPY) || (o.use_mask != 0);
            }
        active_s[c] = act ? 1 : 0;
    }
    __syncthreads();

    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t dp = stride / C;
    const int dc = (int)(stride - dp * C);
    const int64_t q0 = (int64_t)blockIdx.x * kBlock + tid;
    int64_t p = q0 / C;
    int c = (int)(q0 - p * C);
    while (p < n_pts) {
        const bool act = active_s[c] != 0;
        if (act || !in_place) {
            V v = *reinterpret_cast<const V*>(x + p * x_pitch + (int64_t)c * VEC);
            if (act) {
                const bool masked = point_mask ? (point_mask[p] != 0) : false;
                for (int s = 0; s < n_stage; ++s) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        v.v[e] = apply_level_op(prog_s[s * n_slots + c * VEC + e], v.v[e], masked);
                }
            }
            *reinterpret_cast<V*>(y + p * y_pitch + (int64_t)c * VEC) = v;
        }
        p += dp;
        c += dc;
        if (c >= C) { c -= C; ++p; }
    }
}

// ATX_FIELDS: grid.y = level (operator uniform per workgroup), lanes along points.
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
pointwise_fields_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_pts, int n_lev,
                        int64_t x_pitch, int64_t y_pitch, const atx_level_op* __restrict__ prog, int n_stage,
                        const uint8_t* __restrict__ point_mask, int in_place) {
    using V = Pack<T, VEC>;
    const int l = blockIdx.y;
    LevelOp<T> ops[8];
    bool act = false;
    for (int s = 0; s < n_stage; ++s) {
        ops[s] = load_level_op<T>(prog, (int64_t)s * n_lev + l);
        act = act || ops[s].op != ATX_OP_COPY || ops[s].use_mask != 0;
    }
    if (!act && in_place) return;  // untouched field: identity (R: filter.py:193-194)
    const T* xs = x + (int64_t)l * x_pitch;
    T* ys = y + (int64_t)l * y_pitch;
    const int64_t n_vec = n_pts / VEC;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * kBlock) {
        V v = *reinterpret_cast<const V*>(xs + i * VEC);
        if (act) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const bool masked = point_mask ? (point_mask[i * VEC + e] != 0) : false;
                for (int s = 0; s < n_stage; ++s) v.v[e] = apply_level_op(ops[s], v.v[e], masked);
            }
        }
        *reinterpret_cast<V*>(ys + i * VEC) = v;
    }
    // tail points (n_pts % VEC) by the first lanes of block 0
    if (blockIdx.x == 0) {
        const int64_t i = n_vec * VEC + threadIdx.x;
        if (threadIdx.x < VEC && i < n_pts) {
            T v = xs[i];
            if (act) {
                const bool masked = point_mask ? (point_mask[i] != 0) : false;
                for (int s = 0; s < n_stage; ++s) v = apply_level_op(ops[s], v, masked);
            }
            ys[i] = v;
        }
    }
}

// ---- masks -----------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool compare(T m, T thr, int cmp) {
    switch (cmp) {
        case ATX_CMP_GT: return m > thr;
        case ATX_CMP_LT: return m < thr;
        case ATX_CMP_EQ: return m == thr;
        case ATX_CMP_NE: return m != thr;  // true for NaN, like np.not_equal
        case ATX_CMP_GE: return m >= thr;
        case ATX_CMP_LE: return m <= thr;
        case ATX_CMP_NOTNAN: return m == m;
        case ATX_CMP_ISNAN: return m != m;
        default: return false;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
mask_build_kernel(const T* __restrict__ m, int64_t m_stride, uint8_t* __restrict__ mask, int64_t n, int cmp, T thr) {
    // 4 points per lane -> one 32-bit store of 4 mask bytes
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) packed |= (compare<T>(m[(i * 4 + e) * m_stride], thr, cmp) ? 1u : 0u) << (8 * e);
        *reinterpret_cast<uint32_t*>(mask + i * 4) = packed;
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const int64_t i = n4 * 4 + threadIdx.x;
        if (i < n) mask[i] = compare<T>(m[i * m_stride], thr, cmp) ? 1 : 0;
    }
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

__global__ void __launch_bounds__(kBlock)
mask_count_kernel(const uint8_t* __restrict__ mask, int64_t n, unsigned long long* count) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        c += mask[i] != 0;
    c = wave_sum(c);
    if ((threadIdx.x & (kWave - 1)) == 0 && c) atomicAdd(count, c);
}

// ---- stable compaction: mask -> ascending index list ------------------------------
constexpr int kPerLane = 16;                     // mask bytes per lane
constexpr int kChunk = kBlock * kPerLane;        // mask bytes per workgroup

__device__ __forceinline__ int lane_count(const uint8_t* __restrict__ mask, int64_t base, int64_t n, uint32_t& bits) {
    bits = 0;
#pragma unroll
    for (int e = 0; e < kPerLane; ++e) {
        const int64_t i = base + e;
        if (i < n && mask[i] != 0) bits |= 1u << e;
    }
    return __popc(bits);
}

__global__ void __launch_bounds__(kBlock)
compact_count_kernel(const uint8_t* __restrict__ mask, int64_t n, int32_t* __restrict__ block_counts) {
    __shared__ unsigned long long wsum[kBlock / kWave];
    uint32_t bits;
    const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kPerLane;
    unsigned long long c = wave_sum((unsigned long long)lane_count(mask, base, n, bits));
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < kBlock / kWave; ++i) t += wsum[i];
        block_counts[blockIdx.x] = (int32_t)t;
    }
}

// single workgroup: exclusive scan of the per-block counts, in place
__global__ void __launch_bounds__(1024)
compact_scan_kernel(int32_t* __restrict__ block_counts, int n_blocks, long long* __restrict__ total) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int per = (n_blocks + 1023) / 1024;
    const int b0 = tid * per, b1 = min(n_blocks, b0 + per);
    long long s = 0;
    for (int b = b0; b < b1; ++b) s += block_counts[b];
    part[tid] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        long long v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long run = part[tid] - s;  // exclusive prefix of this lane's range
    for (int b = b0; b < b1; ++b) {
        const int32_t cnt = block_counts[b];
        block_counts[b] = (int32_t)run;
        run += cnt;
    }
    if (tid == 1023) *total = part[1023];
}

__global__ void __launch_bounds__(kBlock)
compact_scatter_kernel(const uint8_t* __restrict__ mask, int64_t n, const int32_t* __restrict__ block_offsets,
                       int32_t* __restrict__ index) {
    __shared__ int wsum[kBlock / kWave];
    uint32_t bits;
    const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kPerLane;
    const int cnt = lane_count(mask, base, n, bits);
    // inclusive wave scan by shuffles
    const int lane = threadIdx.x & (kWave - 1);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int v = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += v;
    }
    if (lane == kWave - 1) wsum[threadIdx.x / kWave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int i = 0; i < (int)(threadIdx.x / kWave); ++i) wave_base += wsum[i];
    int64_t o = (int64_t)block_offsets[blockIdx.x] + wave_base + (incl - cnt);
    while (bits) {
        const int e = __ffs(bits) - 1;
        bits &= bits - 1;
        index[o++] = (int32_t)(base + e);
    }
}

// ---- reductions -------------------------------------------------------------------
__device__ __forceinline__ void atomic_minmax(double* addr, double v, bool is_max) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *a;
    while (true) {
        const double cur = __longlong_as_double((long long)old);
        if (cur != cur) return;  // already NaN: np.min/np.max propagate it
        double nv;
        if (v != v) nv = v;
        else nv = is_max ? (v > cur ? v : cur) : (v < cur ? v : cur);
        const unsigned long long nb = (unsigned long long)__double_as_longlong(nv);
        if (nb == old) return;
        const unsigned long long prev = atomicCAS(a, old, nb);
        if (prev == old) return;
        old = prev;
    }
}

__global__ void reduce_init_kernel(double* result, int red) {
    *result = red == ATX_RED_MIN ? INFINITY : (red == ATX_RED_MAX ? -INFINITY : 0.0);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
reduce_kernel(const T* __restrict__ x, int64_t n, int red, double* result) {
    double acc = red == ATX_RED_MIN ? INFINITY : (red == ATX_RED_MAX ? -INFINITY : 0.0);
    bool saw_nan = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double v = (double)x[i];
        if (v != v) {
            saw_nan = true;
            if (red == ATX_RED_NANCOUNT) acc += 1.0;
        } else if (red == ATX_RED_MIN) {
            acc = v < acc ? v : acc;
        } else if (red == ATX_RED_MAX) {
            acc = v > acc ? v : acc;
        }
    }
    if (red != ATX_RED_NANCOUNT && saw_nan) acc = NAN;
    // 64-lane shuffle reduction
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const double o = __shfl_down(acc, off, kWave);
        if (red == ATX_RED_NANCOUNT) acc += o;
        else if (o != o || acc != acc) acc = NAN;
        else if (red == ATX_RED_MIN) acc = o < acc ? o : acc;
        else acc = o > acc ? o : acc;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (red == ATX_RED_NANCOUNT) {
            if (acc != 0.0) atomicAdd(result, acc);
        } else {
            atomic_minmax(result, acc, red == ATX_RED_MAX);
        }
    }
}

static unsigned grid_for(int64_t items) {
    int64_t b = (items + kBlock - 1) / kBlock;
    if (b > kMaxGrid) b = kMaxGrid;
    if (b < 1) b = 1;
    return (unsigned)b;
}

template <typename T>
static int pointwise_typed(const void* x_, void* y_, int64_t n_pts, int n_lev, int64_t xp, int64_t yp, int layout,
                           const atx_level_op* prog, int n_stage, const uint8_t* mask, hipStream_t st) {
    const T* x = static_cast<const T*>(x_);
    T* y = static_cast<T*>(y_);
    const int in_place = (x_ == y_ && xp == yp) ? 1 : 0;
    constexpr int VEC = Vec16<T>::N;
    const bool vec_ok = aligned16(x_) && aligned16(y_) && (xp % VEC == 0) && (yp % VEC == 0);
    if (layout == ATX_COLUMNS) {
        const int64_t covered = ((int64_t)(n_lev + VEC - 1) / VEC) * VEC;
        if (vec_ok && covered <= xp && covered <= yp) {
            const int C = (n_lev + VEC - 1) / VEC;
            const size_t lds = (size_t)n_stage * C * VEC * sizeof(LevelOp<T>) + (size_t)C;
            ATX_REQUIRE(lds <= 64 * 1024, ATX_ENOTIMPL, "pointwise: program needs %zu B of LDS", lds);
            hipLaunchKernelGGL((pointwise_cols_kernel<T, VEC>), dim3(grid_for(n_pts * C)), dim3(kBlock), lds, st, x, y, n_pts,
                               n_lev, C, xp, yp, prog, n_stage, mask, in_place);
        } else {
            const int C = n_lev;
            const size_t lds = (size_t)n_stage * C * sizeof(LevelOp<T>) + (size_t)C;
            ATX_REQUIRE(lds <= 64 * 1024, ATX_ENOTIMPL, "pointwise: program needs %zu B of LDS", lds);
            hipLaunchKernelGGL((pointwise_cols_kernel<T, 1>), dim3(grid_for(n_pts * C)), dim3(kBlock), lds, st, x, y, n_pts,
                               n_lev, C, xp, yp, prog, n_stage, mask, in_place);
        }
    } else {
        ATX_REQUIRE(n_lev <= 65535, ATX_ENOTIMPL, "pointwise: n_lev=%d exceeds grid.y", n_lev);
        unsigned gx = grid_for((n_pts + VEC - 1) / VEC);
        // keep the whole grid near 8 workgroups per CU
        const unsigned cap = (unsigned)((kMaxGrid + n_lev - 1) / n_lev);
        if (gx > cap) gx = cap < 1 ? 1 : cap;
        if (vec_ok) {
            hipLaunchKernelGGL((pointwise_fields_kernel<T, VEC>), dim3(gx, (unsigned)n_lev), dim3(kBlock), 0, st, x, y, n_pts,
                               n_lev, xp, yp, prog, n_stage, mask, in_place);
        } else {
            hipLaunchKernelGGL((pointwise_fields_kernel<T, 1>), dim3(gx, (unsigned)n_lev), dim3(kBlock), 0, st, x, y, n_pts,
                               n_lev, xp, yp, prog, n_stage, mask, in_place);
        }
    }
    ATX_LAUNCH_CHECK("pointwise_stack");
    return ATX_OK;
}

}  // namespace atx

using namespace atx;

extern "C" int atx_pointwise_stack(const void* x, void* y, int64_t n_pts, int64_t n_lev, int64_t x_pitch,
                                   int64_t y_pitch, int dtype, int layout, const atx_level_op* prog, int32_t n_stage,
                                   const uint8_t* point_mask, void* stream) {
    ATX_REQUIRE(x && y && prog, ATX_EINVAL, "atx_pointwise_stack: null pointer");
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "atx_pointwise_stack: bad dtype %d", dtype);
    ATX_REQUIRE(layout == ATX_COLUMNS || layout == ATX_FIELDS, ATX_EINVAL, "atx_pointwise_stack: bad layout %d", layout);
    ATX_REQUIRE(n_pts >= 0 && n_lev > 0 && n_lev <= 65535, ATX_EINVAL, "atx_pointwise_stack: bad sizes n_pts=%lld n_lev=%lld",
                (long long)n_pts, (long long)n_lev);
    ATX_REQUIRE(n_stage >= 1 && n_stage <= 8, ATX_EINVAL, "atx_pointwise_stack: n_stage=%d outside [1, 8]", n_stage);
    const int64_t need = layout == ATX_COLUMNS ? n_lev : n_pts;
    ATX_REQUIRE(x_pitch >= need && y_pitch >= need, ATX_ESHAPE, "atx_pointwise_stack: pitch (%lld, %lld) < %lld",
                (long long)x_pitch, (long long)y_pitch, (long long)need);
    if (n_pts == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ATX_F32) return pointwise_typed<float>(x, y, n_pts, (int)n_lev, x_pitch, y_pitch, layout, prog, n_stage, point_mask, s);
    return pointwise_typed<double>(x, y, n_pts, (int)n_lev, x_pitch, y_pitch, layout, prog, n_stage, point_mask, s);
}

extern "C" int atx_mask_build(const void* m, int64_t m_stride, uint8_t* mask, int64_t n, int cmp, double threshold,
                              int dtype, void* stream) {
    ATX_REQUIRE(m && mask, ATX_EINVAL, "atx_mask_build: null pointer");
    ATX_REQUIRE(n >= 0 && m_stride >= 1, ATX_EINVAL, "atx_mask_build: bad n=%lld / stride=%lld", (long long)n, (long long)m_stride);
    ATX_REQUIRE(cmp >= ATX_CMP_GT && cmp <= ATX_CMP_ISNAN, ATX_EINVAL, "atx_mask_build: bad comparison %d", cmp);
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "atx_mask_build: bad dtype %d", dtype);
    ATX_REQUIRE((reinterpret_cast<uintptr_t>(mask) & 3u) == 0, ATX_EALIGN, "atx_mask_build: mask must be 4-byte aligned");
    if (n == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned g = grid_for((n + 3) / 4);
    if (dtype == ATX_F32)
        hipLaunchKernelGGL(mask_build_kernel<float>, dim3(g), dim3(kBlock), 0, s, static_cast<const float*>(m), m_stride, mask, n, cmp, (float)threshold);
    else
        hipLaunchKernelGGL(mask_build_kernel<double>, dim3(g), dim3(kBlock), 0, s, static_cast<const double*>(m), m_stride, mask, n, cmp, threshold);
    ATX_LAUNCH_CHECK("mask_build");
    return ATX_OK;
}

extern "C" int atx_mask_count(const uint8_t* mask, int64_t n, int64_t* count, void* stream) {
    ATX_REQUIRE(mask && count, ATX_EINVAL, "atx_mask_count: null pointer");
    ATX_REQUIRE(n >= 0, ATX_EINVAL, "atx_mask_count: negative n");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int st = hip_status(hipMemsetAsync(count, 0, sizeof(int64_t), s), "atx_mask_count memset");
    if (st != ATX_OK) return st;
    if (n == 0) return ATX_OK;
    hipLaunchKernelGGL(mask_count_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, mask, n, reinterpret_cast<unsigned long long*>(count));
    ATX_LAUNCH_CHECK("mask_count");
    return ATX_OK;
}

extern "C" size_t atx_mask_to_index_workspace(int64_t n) {
    if (n < 0) return 0;
    const int64_t n_blocks = (n + kChunk - 1) / kChunk;
    return (size_t)((n_blocks + 1) * sizeof(int32_t) + 15) & ~size_t(15);
}

extern "C" int atx_mask_to_index(const uint8_t* mask, int64_t n, int32_t* index, int64_t* count, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    ATX_REQUIRE(mask && index && count && workspace, ATX_EINVAL, "atx_mask_to_index: null pointer");
    ATX_REQUIRE(n >= 0 && n <= INT32_MAX, ATX_EINVAL, "atx_mask_to_index: n=%lld outside int32", (long long)n);
    ATX_REQUIRE(workspace_bytes >= atx_mask_to_index_workspace(n), ATX_EWORKSPACE, "atx_mask_to_index: workspace %zu < %zu",
                workspace_bytes, atx_mask_to_index_workspace(n));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return hip_status(hipMemsetAsync(count, 0, sizeof(int64_t), s), "atx_mask_to_index memset");
    const int n_blocks = (int)((n + kChunk - 1) / kChunk);
    int32_t* block_counts = static_cast<int32_t*>(workspace);
    hipLaunchKernelGGL(compact_count_kernel, dim3(n_blocks), dim3(kBlock), 0, s, mask, n, block_counts);
    ATX_LAUNCH_CHECK("compact_count");
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, n_blocks, reinterpret_cast<long long*>(count));
    ATX_LAUNCH_CHECK("compact_scan");
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(n_blocks), dim3(kBlock), 0, s, mask, n, block_counts, index);
    ATX_LAUNCH_CHECK("compact_scatter");
    return ATX_OK;
}

extern "C" int atx_reduce(const void* x, int64_t n, int red, double* result, int dtype, void* stream) {
    ATX_REQUIRE(x && result, ATX_EINVAL, "atx_reduce: null pointer");
    ATX_REQUIRE(n >= 0, ATX_EINVAL, "atx_reduce: negative n");
    ATX_REQUIRE(red >= ATX_RED_MIN && red <= ATX_RED_NANCOUNT, ATX_EINVAL, "atx_reduce: bad reduction %d", red);
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "atx_reduce: bad dtype %d", dtype);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(reduce_init_kernel, dim3(1), dim3(1), 0, s, result, red);
    ATX_LAUNCH_CHECK("reduce_init");
    if (n == 0) return ATX_OK;
    if (dtype == ATX_F32)
        hipLaunchKernelGGL(reduce_kernel<float>, dim3(grid_for(n)), dim3(kBlock), 0, s, static_cast<const float*>(x), n, red, result);
    else
        hipLaunchKernelGGL(reduce_kernel<double>, dim3(grid_for(n)), dim3(kBlock), 0, s, static_cast<const double*>(x), n, red, result);
    ATX_LAUNCH_CHECK("reduce");
    return ATX_OK;
}

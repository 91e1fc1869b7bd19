// Per-point transforms over a stack of levels: the per-level program evaluator.
//
// Replaces the per-field Python map of the reference
//   R: filter.py:188-196        SingleFieldFilter._map_transform / forward
//   R: filters/fields/rescale.py:25,28   x*scale+offset, (x-offset)/scale
//   R: filters/fields/orog_to_z.py:59,77 x*g, x/g
//   R: filters/fields/clipper.py:69, impute_nans.py:53-54, lnsp_to_sp.py:47,65
//   R: filters/fields/apply_mask.py:183-185, glacier_mask.py:33  values[mask] = nan
// by ONE streaming pass over the stack: level l gets prog[s][l] for each stage s.
// HBM-bound: 16-byte loads/stores, grid-strided over up to kStreamGrid short workgroups; levels whose program is all-COPY are not touched when the
// operation is in place.
#include "atx_common.hpp"
#include <algorithm>
#include <type_traits>

namespace atx {

constexpr int64_t kMaxGrid = kStreamGrid;

// ATX_COLUMNS.  A workgroup sweeps CONTIGUOUS chunks of rows (points).  Its lanes are laid over
// (row-in-pass, vector column): lane = r*Cg + c with Cg = min(C, 256) columns per pass and
// 256/Cg rows per pass, so consecutive lanes touch consecutive 16 B, a lane keeps its column(s) — its
// operators are loop invariant — and the kPwUnroll loads a lane has in flight are adjacent passes of
// the same chunk (NOT megabytes apart: a fixed large power-of-two distance between a lane's
// concurrent streams aliases onto the same HBM channels and cost 25 % here).  HBM-bound; what the
// kernel needs is memory-level parallelism: 4 independent 16-byte loads per lane.
#ifndef ATX_PW_UNROLL
#define ATX_PW_UNROLL 4  // 2: -13 %, 8: +-1 % (f32 137 levels); non-temporal stores (ATX_PW_NT): -15 %
#endif
constexpr int kPwUnroll = ATX_PW_UNROLL;
#ifndef ATX_PW_NT
#define ATX_PW_NT 0  // 0: plain, 1: nt stores, 2: nt loads + nt stores
#endif
template <typename T, int N>
struct PwNative {
    typedef T type __attribute__((ext_vector_type(N)));
};
template <typename T>
struct PwNative<T, 1> {
    typedef T type;
};
template <typename T, int N>
__device__ __forceinline__ Pack<T, N> pw_load(const T* p) {
#if ATX_PW_NT >= 2
    using NV = typename PwNative<T, N>::type;
    NV v = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
    return *reinterpret_cast<Pack<T, N>*>(&v);
#else
    return *reinterpret_cast<const Pack<T, N>*>(p);
#endif
}
template <typename T, int N>
__device__ __forceinline__ void pw_store(T* p, const Pack<T, N>& v) {
#if ATX_PW_NT >= 1
    using NV = typename PwNative<T, N>::type;
    __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&v), reinterpret_cast<NV*>(p));
#else
    *reinterpret_cast<Pack<T, N>*>(p) = v;
#endif
}

// explicit non-temporal forms (whatever ATX_PW_NT says) for kernels that pick per launch
template <typename T, int N>
__device__ __forceinline__ Pack<T, N> pw_load_nt(const T* p) {
    using NV = typename PwNative<T, N>::type;
    NV v = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
    return *reinterpret_cast<Pack<T, N>*>(&v);
}
template <typename T, int N>
__device__ __forceinline__ void pw_store_nt(T* p, const Pack<T, N>& v) {
    using NV = typename PwNative<T, N>::type;
    __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&v), reinterpret_cast<NV*>(p));
}

template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_pts, int n_lev, int C,
                      int64_t x_pitch, int64_t y_pitch, const atx_level_op* __restrict__ prog, int n_stage,
                      const uint8_t* __restrict__ point_mask, int in_place) {
    using V = Pack<T, VEC>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const LevelTablesLds<T> tab = build_level_tables<T, VEC>(prog, smem, n_stage, n_lev, C, tid, kBlock);
    __syncthreads();

    const int Cg = C < kBlock ? C : kBlock;
    const int rows_per_pass = kBlock / Cg;
    const int r = tid / Cg;
    const int cl = tid - r * Cg;
    if (r >= rows_per_pass) return;  // lanes beyond the last whole row of a pass idle (after the barrier)
    const int64_t chunk = (int64_t)rows_per_pass * kPwUnroll;

    for (int c = cl; c < C; c += Cg) {
        const bool act = level_tables_active<T, VEC>(tab, n_stage, c);
        if (!act && in_place) continue;  // untouched levels of an in-place call: nothing to move
        for (int64_t row0 = (int64_t)blockIdx.x * chunk; row0 < n_pts; row0 += (int64_t)gridDim.x * chunk) {
            V v[kPwUnroll];
            int64_t pp[kPwUnroll];
            bool ok[kPwUnroll];
#pragma unroll
            for (int u = 0; u < kPwUnroll; ++u) {
                pp[u] = row0 + (int64_t)u * rows_per_pass + r;
                ok[u] = pp[u] < n_pts;
                if (!ok[u]) pp[u] = row0;
                v[u] = pw_load<T, VEC>(x + pp[u] * x_pitch + (int64_t)c * VEC);
            }
#pragma unroll
            for (int u = 0; u < kPwUnroll; ++u) {
                if (!ok[u]) continue;
                if (act) {
                    const bool masked = point_mask ? (point_mask[pp[u]] != 0) : false;
                    apply_level_tables<T, VEC>(tab, n_stage, c, v[u], masked);
                }
                pw_store<T, VEC>(y + pp[u] * y_pitch + (int64_t)c * VEC, v[u]);
            }
        }
    }
}

// ATX_COLUMNS with tight, equal pitches (pitch == C*VEC: what Stack.empty allocates): the stack is ONE contiguous run of
// n_pts*C vectors.  A workgroup sweeps 16 KB chunks of it (kBlock * kPwUnroll vectors = 128 whole cache lines when the
// base is line aligned), so no line is shared between workgroups — the row-chunk kernel above fetched 4.4 % and wrote
// 1.8 % more than the stack holds (boundary lines of its chunks, PMC counters in profiles/traffic.json) and idles
// kBlock % C lanes.  A lane's column changes from pass to pass; its operators come from the LDS table either way.
template <typename T, int VEC, bool TRANS>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_flat_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_pts, int n_lev, int C,
                           const atx_level_op* __restrict__ prog, int n_stage,
                           const uint8_t* __restrict__ point_mask, int in_place) {
    using V = Pack<T, VEC>;
    extern __shared__ __align__(16) unsigned char smem[];
    LevelOp<T>* vec_ops = reinterpret_cast<LevelOp<T>*>(smem);  // [n_stage][C]
    uint8_t* active = reinterpret_cast<uint8_t*>(vec_ops + (size_t)n_stage * C);  // [C]: column has a non-COPY stage
    const int tid = threadIdx.x;
    build_vector_ops<T, VEC>(prog, vec_ops, n_stage, n_lev, C, tid, kBlock);
    __syncthreads();
    for (int c = tid; c < C; c += kBlock) {
        bool act = false;
        for (int s = 0; s < n_stage; ++s) {
            const LevelOp<T> o = vec_ops[s * C + c];
            act = act || o.op != ATX_OP_COPY || o.use_mask != 0;
        }
        active[c] = act ? 1 : 0;
    }
    __syncthreads();

    const int64_t n_vec = n_pts * C;
    constexpr int64_t kChunk = (int64_t)kBlock * kPwUnroll;
    // A workgroup takes a CONTIGUOUS run of chunks (ATX_PW_ASSIGN 1), not every gridDim.x-th one: under the 65536-workgroup cap the
    // grid stride is 65536 * 16 KB = exactly 1 GiB, and workgroups that drift apart then stream from addresses a power of two
    // apart — on boxes whose allocations are physically contiguous these alias onto the same HBM channels / banks (the same kernel
    // binary measured 2.37 ms on one box and 2.63 ms on another for 137 float64 levels of O1280 while atx_stream_copy stayed at
    // 2.33 ms on both; profiles/r03_pointwise_ab.log, r03_pointwise_placement.log).
#ifndef ATX_PW_ASSIGN
#define ATX_PW_ASSIGN 0
#endif
#if ATX_PW_ASSIGN == 1
    const int64_t n_chunks = (n_vec + kChunk - 1) / kChunk;
    const int64_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
    const int64_t first = (int64_t)blockIdx.x * per * kChunk;
    const int64_t last = first + per * kChunk < n_vec ? first + per * kChunk : n_vec;
    for (int64_t base = first; base < last; base += kChunk) {
#else
    for (int64_t base = (int64_t)blockIdx.x * kChunk; base < n_vec; base += (int64_t)gridDim.x * kChunk) {
#endif
        const int64_t row_b = base / C;  // uniform: scalar unit
        const int col_b = (int)(base - row_b * C);
        V v[kPwUnroll];
        int64_t row[kPwUnroll];
        int col[kPwUnroll];
        bool ok[kPwUnroll], act[kPwUnroll];
#pragma unroll
        for (int u = 0; u < kPwUnroll; ++u) {
            const int off = col_b + u * kBlock + tid;  // < C + kChunk: 32-bit arithmetic
            const int dr = off / C;
            col[u] = off - dr * C;
            row[u] = row_b + dr;
            const int64_t vi = base + u * kBlock + tid;
            ok[u] = vi < n_vec;
            act[u] = active[col[u]] != 0;
            if (in_place && !act[u]) ok[u] = false;  // untouched levels of an in-place call: nothing to move
            if (ok[u]) v[u] = pw_load<T, VEC>(x + vi * VEC);
        }
#pragma unroll
        for (int u = 0; u < kPwUnroll; ++u) {
            if (!ok[u]) continue;
            if (act[u]) {
                const bool masked = point_mask ? (point_mask[row[u]] != 0) : false;
                apply_program_vec<T, VEC, TRANS>(vec_ops, prog, n_stage, n_lev, C, col[u], v[u], masked);
            }
            pw_store<T, VEC>(y + (base + u * kBlock + tid) * VEC, v[u]);
        }
    }
}

// The chunked sweep above with the operators of EVERY LEVEL staged in LDS, parameters and operator codes in separate arrays
// (p0[stage][level], p1[stage][level] in the stack's type, one byte op | mask << 7 per level): a lane fetches the parameters of the
// VEC levels of its vector with two conflict-free 16-byte LDS reads and their codes with one 2- / 4-byte read per stage.  A program
// with a different scale per level — what the packed surface stacks and fused pipelines produce — made every vector "mixed" on the
// kernel above: VEC x 24 bytes of operators per stage and vector out of L1 for 32 bytes of HBM traffic (137 levels of O1280, a scale
// per level: f32 1.88 ms = 0.48, f64 2.69 ms = 0.67); array-of-struct operators in LDS had been tried and lost to bank conflicts.
// NT (non-temporal loads and stores of the stack): float32 only, and not for an in-place call that skips untouched vectors —
// measured per case in profiles/r03_per_level_programs.log (f32 in place 0.68 -> 0.74, out of place 0.75 -> 0.76; f64 0.76 -> 0.73,
// in place with skipped vectors 0.71 -> 0.48).
template <typename T, int VEC, bool TRANS, bool NT>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_levels_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_pts, int n_lev, int C, int Cp,
                             const atx_level_op* __restrict__ prog, int n_stage,
                             const uint8_t* __restrict__ point_mask, int in_place) {
    // Cp >= C: vector slots per row (pitch / VEC) — slots beyond the C that hold levels are padding: neither read nor written
    using V = Pack<T, VEC>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const LevelTablesLds<T> tab = build_level_tables<T, VEC>(prog, smem, n_stage, n_lev, C, tid, kBlock);
    uint8_t* active = smem + level_tables_lds_bytes<T>(n_stage, C, VEC);  // [C]: the vector has a level that is not a plain COPY
    __syncthreads();
    for (int c = tid; c < C; c += kBlock) active[c] = level_tables_active<T, VEC>(tab, n_stage, c) ? 1 : 0;
    __syncthreads();

    const int64_t n_vec = n_pts * Cp;
    constexpr int64_t kChunk = (int64_t)kBlock * kPwUnroll;
    for (int64_t base = (int64_t)blockIdx.x * kChunk; base < n_vec; base += (int64_t)gridDim.x * kChunk) {
        const int64_t row_b = base / Cp;  // uniform: scalar unit
        const int col_b = (int)(base - row_b * Cp);
        V v[kPwUnroll];
        int64_t row[kPwUnroll];
        int col[kPwUnroll];
        bool ok[kPwUnroll], act[kPwUnroll];
#pragma unroll
        for (int u = 0; u < kPwUnroll; ++u) {
            const int off = col_b + u * kBlock + tid;  // < Cp + kChunk: 32-bit arithmetic
            const int dr = off / Cp;
            col[u] = off - dr * Cp;
            row[u] = row_b + dr;
            const int64_t vi = base + u * kBlock + tid;
            ok[u] = vi < n_vec && col[u] < C;
            if (col[u] >= C) col[u] = 0;
            act[u] = active[col[u]] != 0;
            if (in_place && !act[u]) ok[u] = false;  // untouched levels of an in-place call: nothing to move
            if (ok[u]) v[u] = NT ? pw_load_nt<T, VEC>(x + vi * VEC) : pw_load<T, VEC>(x + vi * VEC);
        }
#pragma unroll
        for (int u = 0; u < kPwUnroll; ++u) {
            if (!ok[u]) continue;
            if (act[u]) {
                const bool masked = point_mask ? (point_mask[row[u]] != 0) : false;
                apply_level_tables<T, VEC, TRANS>(tab, n_stage, col[u], v[u], masked);
            }
            if (NT) pw_store_nt<T, VEC>(y + (base + u * kBlock + tid) * VEC, v[u]);
            else pw_store<T, VEC>(y + (base + u * kBlock + tid) * VEC, v[u]);
        }
    }
}

// The same stacks with the per-vector operator table GIVEN (vec_prog, built once on the host by atx_vector_program): no
// shared memory, no barrier, no loop — one 16-byte vector per lane and one workgroup per 4 KB, the launch shape that
// streams fastest on MI355X (a plain y = a*x + b over 3.7 GB: 6.17 TB/s in this shape, 5.8 TB/s with 2-8 vectors per
// lane and / or a grid-stride loop — tools/experiments/stream_shapes.hip).  A lane fetches the operators of its column
// for all stages from the table (L1 / L2 resident: n_stage * C * 24 B) before its data arrives.
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_table_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_vec, int n_lev, int C,
                            const atx_level_op* __restrict__ prog, const atx_level_op* __restrict__ vec_prog, int n_stage,
                            const uint8_t* __restrict__ point_mask, int in_place) {
    using V = Pack<T, VEC>;
    const int64_t vi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (vi >= n_vec) return;
    int64_t row;
    int c;
    if (n_vec <= 0xffffffffll) {  // uniform: 32-bit division
        const unsigned r = (unsigned)vi / (unsigned)C;
        row = r;
        c = (int)((unsigned)vi - r * (unsigned)C);
    } else {
        row = vi / C;
        c = (int)(vi - row * C);
    }
    LevelOp<T> ops[8];
    bool act = false, need_mask = false;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < n_stage) {
            ops[s] = load_level_op<T>(vec_prog, (int64_t)s * C + c);
            act = act || ops[s].op != ATX_OP_COPY || ops[s].use_mask != 0;
            need_mask = need_mask || ops[s].use_mask != 0;
        }
    }
    if (!act && in_place) return;  // untouched levels of an in-place call: nothing to move
    V v = pw_load<T, VEC>(x + vi * VEC);
    if (act) {
        const bool masked = (need_mask && point_mask) ? (point_mask[row] != 0) : false;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s >= n_stage) break;
            if (ops[s].op != kOpMixed) {
                apply_level_op_vec<T, VEC>(ops[s], v, masked);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int l = c * VEC + e;
                    if (l < n_lev) v.v[e] = apply_level_op(load_level_op<T>(prog, (int64_t)s * n_lev + l), v.v[e], masked);
                }
            }
        }
    }
    pw_store<T, VEC>(y + vi * VEC, v);
}

// The same stacks when the program is UNIFORM over the levels (uniform_level_program: per stage one operator, or two pieces split
// on a vector boundary — rescale / convert / orog_to_z / clip / impute_nans / apply_mask over a whole stack, or "136 levels of t and
// one of orog"): the operators travel by value in the kernel arguments, so a lane's only memory traffic is its 16-byte vector (and
// one mask byte when a stage uses the point mask) — the launch shape and the instruction stream of atx_stream_copy plus a scalar
// branch per stage.  This is the shape whose speed does not move from box to box (atx_stream_copy: 2.33-2.35 ms for 137 float64
// levels of O1280 on every box of round 3, the chunked kernel 2.34-2.68 ms).  NT: non-temporal loads AND stores — nothing this
// kernel touches is touched again — measured 2.32 -> 2.22 ms f64 out of place, 2.43 -> 2.23 ms in place, f32 1.19 -> 1.13 and
// 1.24 -> 1.12 ms (0.80-0.81 of the peak, above the plain copy); with a point mask they cost 2-6 % instead (the mask bytes, shared
// by the ~35-69 lanes of a point, live in the same caches), so masked programs keep plain accesses; profiles/r03_pointwise_ab.log.
template <typename T, int VEC, bool TRANS, int U, bool NT>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_uniform_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_vec, int C, int Cp, UniformOps<T> u,
                              const uint8_t* __restrict__ point_mask, int need_rc, int in_place, unsigned act_bits) {
    // Cp >= C: vector slots per row (pitch / VEC); n_vec = n_pts * Cp; slots beyond C are padding (need_rc is set when Cp > C)
    using V = Pack<T, VEC>;
    // U vectors per lane, kBlock apart: a workgroup covers kBlock * U consecutive vectors (no loop)
    const int64_t base = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
    V v[U];
    int c[U];
    bool masked[U], go[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int64_t vi = base + (int64_t)k * kBlock;
        go[k] = vi < n_vec;
        c[k] = 0;
        masked[k] = false;
        if (go[k] && need_rc) {  // uniform: a two-piece stage or the point mask — the lane needs its (row, vector column)
            int64_t row;
            if (n_vec <= 0xffffffffll) {
                const unsigned r = (unsigned)vi / (unsigned)Cp;
                row = r;
                c[k] = (int)((unsigned)vi - r * (unsigned)Cp);
            } else {
                row = vi / Cp;
                c[k] = (int)(vi - row * Cp);
            }
            if (c[k] >= C) go[k] = false;  // padding slot of a loose pitch
            if (in_place) {  // untouched columns of an in-place call: nothing to move.  `act_bits` (host-built): bit s = the first
                // piece of stage s does something, bit 4 + s = its second piece.  (Selecting between u.stage[s].op and u.second[s].op
                // here made the compiler select between their ADDRESSES and load per lane from the kernel-argument segment, two
                // dependent loads per stage ahead of the data load: apply_mask in place 3.51 ms instead of 2.45.)
                unsigned act = 0;
#pragma unroll
                for (int s = 0; s < kMaxUniform; ++s) act |= (c[k] >= u.split[s]) ? (act_bits >> (4 + s)) : (act_bits >> s);
                go[k] = go[k] && (act & 1u) != 0;
            }
#ifndef ATX_PW_MASK_WAVE
#define ATX_PW_MASK_WAVE 0
#endif
            if (point_mask) {
                // The mask byte of a point is shared by the Cp lanes of its row: a per-lane byte load is one more vector-memory instruction
                // per wave on a kernel that has two (ATX_PW_MASK_WAVE=1: the wave fetches the 8 bytes from its first row on with ONE scalar
                // load and every lane picks its own — possible when a wave spans at most 8 rows and the 8 bytes lie inside the mask).
                bool wave_path = false;
                if (ATX_PW_MASK_WAVE && Cp >= 10 && n_vec <= 0xffffffffll) {
                    const unsigned first_vi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(vi - (threadIdx.x & (kWave - 1))));
                    const unsigned row0 = (first_vi / (unsigned)Cp) & ~7u;  // 8-byte aligned start
                    const int64_t n_rows = n_vec / Cp;
                    if ((int64_t)row0 + 8 <= n_rows && (reinterpret_cast<uintptr_t>(point_mask) & 7u) == 0) {
                        wave_path = true;
                        const unsigned long long bits = *reinterpret_cast<const unsigned long long*>(point_mask + row0);
                        const unsigned off = (unsigned)row - row0;  // < 8 + 64 / Cp
                        if (off < 8u) masked[k] = go[k] && ((bits >> (8u * off)) & 0xffull) != 0;
                        else masked[k] = go[k] && point_mask[row] != 0;
                    }
                }
                if (!wave_path && go[k]) masked[k] = point_mask[row] != 0;
            }
        }
        if (go[k]) v[k] = NT ? pw_load_nt<T, VEC>(x + vi * VEC) : pw_load<T, VEC>(x + vi * VEC);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if (!go[k]) continue;
        for (int s = 0; s < u.n_stage; ++s) {
            if (u.split[s] >= C) {  // scalar condition: one piece
                apply_level_op_vec<T, VEC, TRANS>(u.stage[s], v[k], masked[k]);
            } else {
                V other = v[k];
                apply_level_op_vec<T, VEC, TRANS>(u.stage[s], v[k], masked[k]);
                apply_level_op_vec<T, VEC, TRANS>(u.second[s], other, masked[k]);
                if (c[k] >= u.split[s]) v[k] = other;
            }
        }
        if (NT) pw_store_nt<T, VEC>(y + (base + (int64_t)k * kBlock) * VEC, v[k]);
        else pw_store<T, VEC>(y + (base + (int64_t)k * kBlock) * VEC, v[k]);
    }
}

// Programs made of RUNS of levels (RunOps: up to 4 per stage, boundaries anywhere — several variables sharing a column): the launch shape
// of the by-value kernel above — one 16-byte vector per lane, no loop, no shared memory, non-temporal accesses when no point mask is
// read — with every run's operator evaluated on the vector and the element's own kept.  Before round 4 such programs went through the
// per-level LDS kernel (0.70 / 0.75 of the peak).
template <typename T, int VEC, bool TRANS, bool NT>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_runs_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_vec, int C, int Cp, RunOps<T> runs,
                           const uint8_t* __restrict__ point_mask) {
    using V = Pack<T, VEC>;
    const int64_t vi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (vi >= n_vec) return;
    int64_t row;
    int c;
    if (n_vec <= 0xffffffffll) {
        const unsigned r = (unsigned)vi / (unsigned)Cp;
        row = r;
        c = (int)((unsigned)vi - r * (unsigned)Cp);
    } else {
        row = vi / Cp;
        c = (int)(vi - row * Cp);
    }
    if (c >= C) return;  // padding slot of a loose pitch
    const bool masked = point_mask ? point_mask[row] != 0 : false;
    V v = NT ? pw_load_nt<T, VEC>(x + vi * VEC) : pw_load<T, VEC>(x + vi * VEC);
    apply_run_ops<T, VEC, TRANS>(runs, c, v, masked);
    if (NT) pw_store_nt<T, VEC>(y + vi * VEC, v);
    else pw_store<T, VEC>(y + vi * VEC, v);
}

// ONE stage whose 16-byte vectors each hold one operator KIND (the parameters may differ from level to level: a scale per level): one
// vector per lane, no loop, the parameters from the typed per-level part of the host-built table (level_tables_layout; two 16-byte
// loads and one code word per lane, L1 / L2 resident) — the launch shape of the by-value kernel.  Measured against the per-level LDS
// kernel, same box, interleaved (profiles/r03_per_level_programs.log): f64 +4 to +8 % (0.75-0.76 -> 0.78-0.80 on a fast box, 0.67-0.70 ->
// 0.71-0.75 on a slow one), f32 in place 0.70-0.74 -> 0.74-0.78, f32 out of place 0.81 -> 0.77 and f32 with the point mask 0.73 -> 0.69
// (those stay on the LDS kernel); two stages or vectors of mixed kinds lose 10-25 % here and stay there too.
template <typename T, int VEC, bool TRANS, bool NT>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_typed_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_vec, int C, const unsigned char* __restrict__ level_tables,
                            const uint8_t* __restrict__ point_mask, int in_place) {
    using V = Pack<T, VEC>;
    using OpWord = typename OpWordOf<VEC>::type;
    const int64_t vi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (vi >= n_vec) return;
    int64_t row;
    int c;
    if (n_vec <= 0xffffffffll) {  // uniform: 32-bit division
        const unsigned r = (unsigned)vi / (unsigned)C;
        row = r;
        c = (int)((unsigned)vi - r * (unsigned)C);
    } else {
        row = vi / C;
        c = (int)(vi - row * C);
    }
    const int Lp = C * VEC;
    const T* tp0 = reinterpret_cast<const T*>(level_tables);
    const T* tp1 = tp0 + Lp;
    const uint8_t* tcd = reinterpret_cast<const uint8_t*>(tp1 + Lp);
    const unsigned code = *reinterpret_cast<const OpWord*>(tcd + c * VEC) & 0xffu;  // (the host checked: every level of the vector has this code)
    if (in_place && code == 0) return;  // untouched levels of an in-place call: nothing to move
    const V a = *reinterpret_cast<const V*>(tp0 + c * VEC);
    const V b = *reinterpret_cast<const V*>(tp1 + c * VEC);
    V v = NT ? pw_load_nt<T, VEC>(x + vi * VEC) : pw_load<T, VEC>(x + vi * VEC);
    if (code != 0) {
        const bool masked = ((code & 0x80u) && point_mask) ? (point_mask[row] != 0) : false;
        apply_level_op_params<T, VEC, TRANS>((int)(code & 0x7fu), (code & 0x80u) != 0, a, b, v, masked);
    }
    if (NT) pw_store_nt<T, VEC>(y + vi * VEC, v);
    else pw_store<T, VEC>(y + vi * VEC, v);
}

// In place with FEW active levels (1 of 137: one variable of a stack converted, a mask applied to one field): only the
// vector columns that hold an active level are visited — one (point, active column) item per lane, the column list by value in
// the kernel arguments.  The kernels above skip the loads of untouched columns too, but still walk all n_pts*C vector slots
// with 1 lane in 35 doing anything (1.53 ms for 1 of 137 levels on O1280, more than transforming the whole stack); here every
// lane has a 16-byte read-modify-write in flight.  The traffic is one 64/128-byte line per point and active column either way.
constexpr int kMaxActive = 16;
struct ActiveCols {
    int n;
    int col[kMaxActive];
};

template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
pointwise_cols_sparse_kernel(T* __restrict__ y, int64_t n_items, int n_lev, int C, int64_t pitch, ActiveCols active,
                             const atx_level_op* __restrict__ prog, const atx_level_op* __restrict__ vec_prog, int n_stage,
                             const uint8_t* __restrict__ point_mask) {
    using V = Pack<T, VEC>;
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= n_items) return;
    const int64_t p = q / active.n;
    const int a = (int)(q - p * active.n);
    int c = active.col[0];
#pragma unroll
    for (int i = 1; i < kMaxActive; ++i) c = (a == i) ? active.col[i] : c;  // the list lives in scalar registers: select, do not index
    LevelOp<T> ops[8];
    bool need_mask = false;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < n_stage) {
            ops[s] = load_level_op<T>(vec_prog, (int64_t)s * C + c);
            need_mask = need_mask || ops[s].use_mask != 0;
        }
    }
    T* at = y + p * pitch + (int64_t)c * VEC;
    V v = pw_load<T, VEC>(at);
    const bool masked = (need_mask && point_mask) ? (point_mask[p] != 0) : false;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s >= n_stage) break;
        if (ops[s].op != kOpMixed) {
            apply_level_op_vec<T, VEC>(ops[s], v, masked);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int l = c * VEC + e;
                if (l < n_lev) v.v[e] = apply_level_op(load_level_op<T>(prog, (int64_t)s * n_lev + l), v.v[e], masked);
            }
        }
    }
    pw_store<T, VEC>(at, v);
}

// ATX_FIELDS, round 3: one 16-byte vector per lane, no loop — grid.x = 1 KB-per-wave pieces of a field, grid.y = field.  The field's
// operators (uniform over the workgroup: scalar loads, scalar branches) stay in registers because the stage loop is fully unrolled
// — round 1's grid-stride kernel indexed ops[] with a run-time stage count, which put the array in scratch memory: 137 fields of O1280 ran at
// 0.40-0.44 of the HBM peak, two stages at 0.29 (tools/experiments/fields_pointwise.py).  The mask bytes of a vector's VEC points
// come as one 2- / 4-byte load (mask_vec: the mask base is aligned for it); non-temporal accesses when no mask is read.
// Measured on 137 fields of O1280 (profiles/r03_fields_pointwise.log; run-to-run noise ~3 %): requesting the data before the field's
// operators are known (out of place only — in place an untouched field must not be read) f32 0.78 -> 0.83, f64 neutral; two vectors
// per lane f64 0.75-0.77 -> 0.78-0.80 (and an in-place call that skips two fields in three 1.25 -> 0.95 ms), f32 in place 0.78 -> 0.75.
#ifndef ATX_PW_FIELDS_U32
#define ATX_PW_FIELDS_U32 1  // vectors per lane (kBlock apart), float32
#endif
#ifndef ATX_PW_FIELDS_U64
#define ATX_PW_FIELDS_U64 2  // float64
#endif
#ifndef ATX_PW_FIELDS_EARLY
#define ATX_PW_FIELDS_EARLY 1
#endif
template <typename T>
constexpr int fields_vectors_per_lane() {
    return sizeof(T) == 4 ? ATX_PW_FIELDS_U32 : ATX_PW_FIELDS_U64;
}
template <typename T, int VEC, bool TRANS, bool NT>
__global__ void __launch_bounds__(kBlock)
pointwise_fields_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n_pts, int n_lev,
                             int64_t x_pitch, int64_t y_pitch, const atx_level_op* __restrict__ prog, int n_stage,
                             const uint8_t* __restrict__ point_mask, int in_place, int mask_vec) {
    using V = Pack<T, VEC>;
    using MaskWord = typename OpWordOf<VEC>::type;
    constexpr int U = fields_vectors_per_lane<T>();
    const int l = blockIdx.y;
    const T* xs = x + (int64_t)l * x_pitch;
    T* ys = y + (int64_t)l * y_pitch;
    const int64_t n_vec = n_pts / VEC;
    const int64_t base = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
    V v[U];
    const bool early = ATX_PW_FIELDS_EARLY && !in_place;
    if (early) {
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (base + k * kBlock < n_vec) v[k] = NT ? pw_load_nt<T, VEC>(xs + (base + k * kBlock) * VEC) : pw_load<T, VEC>(xs + (base + k * kBlock) * VEC);
    }
    LevelOp<T> ops[8];
    bool act = false, need_mask = false;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        ops[s].op = ATX_OP_COPY;
        ops[s].use_mask = 0;
        ops[s].p0 = ops[s].p1 = T(0);
        if (s < n_stage) {
            ops[s] = load_level_op<T>(prog, (int64_t)s * n_lev + l);
            act = act || ops[s].op != ATX_OP_COPY || ops[s].use_mask != 0;
            need_mask = need_mask || ops[s].use_mask != 0;
        }
    }
    if (!act && in_place) return;  // untouched field: identity (R: filter.py:193-194)
    if (!early) {
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (base + k * kBlock < n_vec) v[k] = NT ? pw_load_nt<T, VEC>(xs + (base + k * kBlock) * VEC) : pw_load<T, VEC>(xs + (base + k * kBlock) * VEC);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int64_t vi = base + k * kBlock;
        if (vi >= n_vec) continue;
        if (act) {
            unsigned mbits = 0;  // byte e: the mask of point vi * VEC + e
            if (need_mask && point_mask) {
                if (mask_vec) {
                    mbits = *reinterpret_cast<const MaskWord*>(point_mask + vi * VEC);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) mbits |= (unsigned)point_mask[vi * VEC + e] << (8 * e);
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < n_stage) {
                    LevelOp<T> o = ops[s];
                    const bool use_mask = o.use_mask != 0;
                    o.use_mask = 0;
                    apply_level_op_vec<T, VEC, TRANS>(o, v[k], false);
                    if (use_mask) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if ((mbits >> (8 * e)) & 0xffu) v[k].v[e] = quiet_nan<T>();
                    }
                }
            }
        }
        if (NT) pw_store_nt<T, VEC>(ys + vi * VEC, v[k]);
        else pw_store<T, VEC>(ys + vi * VEC, v[k]);
    }
    // tail points (n_pts % VEC) by the first lanes of block 0
    if (VEC > 1 && blockIdx.x == 0) {
        const int64_t i = n_vec * VEC + threadIdx.x;
        if (threadIdx.x < VEC && i < n_pts) {
            T t = xs[i];
            if (act) {
                const bool masked = point_mask ? (point_mask[i] != 0) : false;
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < n_stage) t = apply_level_op<T, TRANS>(ops[s], t, masked);
            }
            ys[i] = t;
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

// SELF_SCAN (few thousand workgroups at most): `block_offsets` holds the raw per-workgroup COUNTS and every workgroup sums the counts before
// its own by itself (a few KB out of L2) — the single-workgroup scan launch between count and scatter goes away (three launches -> two);
// the last workgroup writes the total.
template <bool SELF_SCAN>
__global__ void __launch_bounds__(kBlock)
compact_scatter_kernel(const uint8_t* __restrict__ mask, int64_t n, const int32_t* __restrict__ block_offsets,
                       int32_t* __restrict__ index, long long* __restrict__ total) {
    __shared__ int wsum[kBlock / kWave];
    __shared__ long long before_s;
    long long before = 0;
    if (SELF_SCAN) {
        __shared__ unsigned long long psum[kBlock / kWave];
        unsigned long long mine = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) mine += (unsigned long long)block_offsets[b];
        mine = wave_sum(mine);
        if ((threadIdx.x & (kWave - 1)) == 0) psum[threadIdx.x / kWave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int i = 0; i < kBlock / kWave; ++i) t += psum[i];
            before_s = (long long)t;
            if (blockIdx.x == gridDim.x - 1) *total = (long long)t + block_offsets[blockIdx.x];
        }
        __syncthreads();
        before = before_s;
    } else {
        before = block_offsets[blockIdx.x];
    }
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
    int64_t o = (int64_t)before + wave_base + (incl - cnt);
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
    if (red == ATX_RED_MINMAX) {
        result[0] = INFINITY;
        result[1] = -INFINITY;
        return;
    }
    *result = red == ATX_RED_MIN ? INFINITY : (red == ATX_RED_MAX ? -INFINITY : 0.0);
}

// One partial per lane over a grid-stride sweep (4 independent loads in flight), 64-lane shuffle, one LDS combine per
// workgroup, ONE atomic per workgroup (a per-wave atomic on a single address serialised 100 k of them on large inputs).
constexpr int kRedUnroll = 4;
#ifndef ATX_RED_GRID
#define ATX_RED_GRID 32768  // workgroup cap = partial slots of the two-level finish.  Round 3 compared 8192 with SMALLER caps only; round 4, 137 levels of O1280, two
#endif                      // interleaved rounds: 32768 f64 min+max 0.707 -> 0.74, NaN count f32 0.70 -> 0.74, f64 0.715 -> 0.76; f32 min+max unchanged (0.68);
                            // 65536 loses (f32 min+max 0.60).  One field: unchanged (its grid is far below either cap).  profiles/r04_reduce_grid.log
constexpr int64_t kRedGrid = ATX_RED_GRID;

__device__ __forceinline__ double red_combine(double a, double b, int red) {
    if (red == ATX_RED_NANCOUNT) return a + b;
    if (a != a || b != b) return NAN;  // np.min / np.max propagate NaN
    if (red == ATX_RED_MIN) return b < a ? b : a;
    return b > a ? b : a;
}

// Two-level finish WITHOUT one atomic per workgroup on the result (a CAS loop on one address: 1 600 workgroups over one 26 MB
// field spent 10 us of a 43 us call in it, min AND max 30 us) and without an initialisation launch: with a caller-provided workspace
// every workgroup stores its partial(s) with plain stores, and a second, one-workgroup launch (reduce_final_kernel) combines them and
// writes `result` — which may then be a pinned HOST cell: no copy back either.  The kernel boundary is the only synchronisation.
// (Round 3 first tried a single launch with a ticket — the last workgroup to arrive combines — and measured it SLOWER than the
// atomics: the device-scope release every workgroup needs before taking its ticket writes the XCD's L2 back; 43 -> 60 us for one
// field, 0.66 -> 1.0 ms for a 137-level stack.  profiles/r03_small_calls.log.)
struct RedWorkspace {
    double a[kRedGrid];
    double b[kRedGrid];
};

__global__ void __launch_bounds__(kBlock)
reduce_final_kernel(const RedWorkspace* __restrict__ ws, int n, int ra, int red, double* result) {
    __shared__ double fa[kBlock / kWave], fb[kBlock / kWave];
    const double id_a = ra == ATX_RED_MIN ? INFINITY : (ra == ATX_RED_MAX ? -INFINITY : 0.0);
    double a = id_a, b = -INFINITY;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        a = red_combine(a, ws->a[i], ra);
        if (red == ATX_RED_MINMAX) b = red_combine(b, ws->b[i], ATX_RED_MAX);
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        a = red_combine(a, __shfl_down(a, off, kWave), ra);
        if (red == ATX_RED_MINMAX) b = red_combine(b, __shfl_down(b, off, kWave), ATX_RED_MAX);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        fa[threadIdx.x / kWave] = a;
        fb[threadIdx.x / kWave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double xa = fa[0], xb = fb[0];
        for (int w = 1; w < kBlock / kWave; ++w) {
            xa = red_combine(xa, fa[w], ra);
            xb = red_combine(xb, fb[w], ATX_RED_MAX);
        }
        result[0] = xa;
        if (red == ATX_RED_MINMAX) result[1] = xb;
        __threadfence_system();  // `result` may live in pinned host memory
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
reduce_kernel(const T* __restrict__ x, int64_t n_rows, int64_t row_len, int64_t pitch, int red, double* result, double* partials) {
    const double identity = red == ATX_RED_MIN ? INFINITY : (red == ATX_RED_MAX ? -INFINITY : 0.0);
    double acc = identity;
    // rows of `row_len` elements `pitch` apart; (row, col) advances by the grid stride without a division per element
    const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x, stride = (int64_t)gridDim.x * kBlock;
    const int64_t d_row = stride / row_len, d_col = stride - d_row * row_len;
    int64_t row = first / row_len, col = first - row * row_len;
    while (row < n_rows) {
        T v[kRedUnroll];
        bool ok[kRedUnroll];
#pragma unroll
        for (int u = 0; u < kRedUnroll; ++u) {
            ok[u] = row < n_rows;
            v[u] = ok[u] ? x[row * pitch + col] : T(0);
            row += d_row;
            col += d_col;
            if (col >= row_len) {
                col -= row_len;
                ++row;
            }
        }
#pragma unroll
        for (int u = 0; u < kRedUnroll; ++u) {
            if (!ok[u]) continue;
            const double d = (double)v[u];
            if (red == ATX_RED_NANCOUNT) acc += (d != d) ? 1.0 : 0.0;
            else acc = red_combine(acc, d, red);
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) acc = red_combine(acc, __shfl_down(acc, off, kWave), red);
    __shared__ double partial[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) partial[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = partial[0];
        for (int w = 1; w < kBlock / kWave; ++w) total = red_combine(total, partial[w], red);
        if (partials) {  // one slot per workgroup of the caller's workspace (a[] — or b[] for the MAX pass of a two-pass MINMAX)
            partials[blockIdx.x] = total;
        } else if (red == ATX_RED_NANCOUNT) {
            if (total != 0.0) atomicAdd(result, total);
        } else {
            atomic_minmax(result, total, red == ATX_RED_MAX);
        }
    }
}

// The same sweep with 16-byte loads — one (row, vector) item per step, the elements of a row's last vector beyond row_len
// masked — and with ATX_RED_MINMAX both extremes in ONE pass: the range check of cos_sin_from_rad (R: cos_sin_from_rad.py:73-76,
// `data.min()` then `data.max()`) read the stack twice at 4.7 TB/s (4-byte loads); this reads it once.
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
reduce_vec_kernel(const T* __restrict__ x, int64_t n_rows, int64_t row_len, int C, int64_t pitch, int red, double* result, RedWorkspace* ws) {
    using V = Pack<T, VEC>;
    const bool want_min = red == ATX_RED_MIN || red == ATX_RED_MINMAX, want_max = red == ATX_RED_MAX || red == ATX_RED_MINMAX;
    double lo = INFINITY, hi = -INFINITY, count = 0.0;
    bool seen_nan = false;
    const int64_t n_items = n_rows * C;
    const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x, stride = (int64_t)gridDim.x * kBlock;
    const int64_t d_row = stride / C;
    const int d_col = (int)(stride - d_row * C);
    int64_t row = first / C;
    int col = (int)(first - row * C);
    for (int64_t i = first; i < n_items; i += stride * kRedUnroll) {
        V v[kRedUnroll];
        int valid[kRedUnroll];
#pragma unroll
        for (int u = 0; u < kRedUnroll; ++u) {
            const bool ok = row < n_rows;
            valid[u] = ok ? (int)min((int64_t)VEC, row_len - (int64_t)col * VEC) : 0;
            if (ok) v[u] = pw_load_nt<T, VEC>(x + row * pitch + (int64_t)col * VEC);  // read once: non-temporal (0.73 -> 0.66 ms f32, 1.33 -> 1.19 ms f64)
            row += d_row;
            col += d_col;
            if (col >= C) {
                col -= C;
                ++row;
            }
        }
#pragma unroll
        for (int u = 0; u < kRedUnroll; ++u) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (e < valid[u]) {
                    const double d = (double)v[u].v[e];
                    if (d != d) {
                        seen_nan = true;
                        count += 1.0;
                    } else {
                        lo = d < lo ? d : lo;
                        hi = d > hi ? d : hi;
                    }
                }
            }
        }
    }
    if (seen_nan) lo = hi = NAN;  // np.min / np.max propagate NaN
    double a = red == ATX_RED_NANCOUNT ? count : (want_min ? lo : hi), b = hi;
    const int ra = red == ATX_RED_NANCOUNT ? ATX_RED_NANCOUNT : (want_min ? ATX_RED_MIN : ATX_RED_MAX);
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        a = red_combine(a, __shfl_down(a, off, kWave), ra);
        if (red == ATX_RED_MINMAX) b = red_combine(b, __shfl_down(b, off, kWave), ATX_RED_MAX);
    }
    __shared__ double pa[kBlock / kWave], pb[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) {
        pa[threadIdx.x / kWave] = a;
        pb[threadIdx.x / kWave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = pa[0], tb = pb[0];
        for (int w = 1; w < kBlock / kWave; ++w) {
            ta = red_combine(ta, pa[w], ra);
            tb = red_combine(tb, pb[w], ATX_RED_MAX);
        }
        if (ws) {
            ws->a[blockIdx.x] = ta;
            ws->b[blockIdx.x] = tb;
        } else if (red == ATX_RED_NANCOUNT) {
            if (ta != 0.0) atomicAdd(result, ta);
        } else {
            atomic_minmax(result, ta, !want_min);
            if (red == ATX_RED_MINMAX) atomic_minmax(result + 1, tb, true);
        }
    }
    (void)want_max;
}

// The <= 3 elements a flat array leaves after its last whole 16-byte vector, as ONE more pair of partials in the workspace
// (slot `slot`), so that the two-level finish serves this shape too: no atomics on `result`, which may be a pinned host cell.
template <typename T>
__global__ void reduce_tail_kernel(const T* __restrict__ x, int n, int red, RedWorkspace* ws, int slot) {
    double lo = INFINITY, hi = -INFINITY, count = 0.0;
    bool seen_nan = false;
    for (int i = 0; i < n; ++i) {
        const double d = (double)x[i];
        if (d != d) {
            seen_nan = true;
            count += 1.0;
        } else {
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
    }
    if (seen_nan) lo = hi = NAN;
    const bool want_min = red == ATX_RED_MIN || red == ATX_RED_MINMAX;
    ws->a[slot] = red == ATX_RED_NANCOUNT ? count : (want_min ? lo : hi);
    ws->b[slot] = hi;
}

static unsigned grid_for(int64_t items) {
    int64_t b = (items + kBlock - 1) / kBlock;
    if (b > kMaxGrid) b = kMaxGrid;
    if (b < 1) b = 1;
    return (unsigned)b;
}

template <typename T>
static int pointwise_typed(const void* x_, void* y_, int64_t n_pts, int n_lev, int64_t xp, int64_t yp, int layout,
                           const atx_level_op* prog, const atx_level_op* vec_prog, const atx_level_op* host_prog, int n_stage,
                           const uint8_t* mask, hipStream_t st) {
    const T* x = static_cast<const T*>(x_);
    T* y = static_cast<T*>(y_);
    const int in_place = (x_ == y_ && xp == yp) ? 1 : 0;
    constexpr int VEC = Vec16<T>::N;
    const bool vec_ok = aligned16(x_) && aligned16(y_) && (xp % VEC == 0) && (yp % VEC == 0);
    if (layout == ATX_COLUMNS) {
        const int64_t covered = ((int64_t)(n_lev + VEC - 1) / VEC) * VEC;
        const bool wide = vec_ok && covered <= xp && covered <= yp;
        const int C = wide ? (n_lev + VEC - 1) / VEC : n_lev;
        const size_t lds = (size_t)n_stage * C * sizeof(LevelOp<T>);  // (the older chunked kernel's per-vector table)
#ifndef ATX_PW_SPARSE
#define ATX_PW_SPARSE 1
#endif
        if (ATX_PW_SPARSE && in_place && wide && host_prog && vec_prog) {  // few active levels, in place: visit their columns only
            ActiveCols active{};
            bool fits = true;
            for (int c = 0; c < C && fits; ++c) {
                bool act = false;
                for (int s = 0; s < n_stage && !act; ++s)
                    for (int l = c * VEC; l < n_lev && l < (c + 1) * VEC && !act; ++l) {
                        const atx_level_op& o = host_prog[(int64_t)s * n_lev + l];
                        act = o.op != ATX_OP_COPY || o.use_mask != 0;
                    }
                if (!act) continue;
                if (active.n == kMaxActive) fits = false;
                else active.col[active.n++] = c;
            }
            if (fits && active.n == 0) return ATX_OK;  // nothing to do at all
            if (fits && 3 * active.n <= C) {
                for (int i = active.n; i < kMaxActive; ++i) active.col[i] = active.col[0];
                const int64_t n_items = n_pts * active.n;
                const int64_t blocks = (n_items + kBlock - 1) / kBlock;
                ATX_REQUIRE(blocks <= 0x7fffffffll, ATX_ENOTIMPL, "pointwise: %lld items exceed one launch", (long long)n_items);
                hipLaunchKernelGGL((pointwise_cols_sparse_kernel<T, VEC>), dim3((unsigned)blocks), dim3(kBlock), 0, st, y, n_items, n_lev, C, yp,
                                   active, prog, vec_prog, n_stage, mask);
                ATX_LAUNCH_CHECK("pointwise_stack_sparse");
                return ATX_OK;
            }
        }
#ifndef ATX_PW_FLAT
#define ATX_PW_FLAT 1
#endif
        // one contiguous run of vector slots (tight pitch), or — round 3 — equal loose pitches (columns aligned to 128 bytes, a spare
        // vector): the same kernels walk the pitch / VEC slots of a row and leave the padding slots alone (the row-chunk kernel below:
        // 0.51-0.57 f32, 0.69 f64 on such stacks)
        const bool tight = xp == (int64_t)C * VEC;
        if (ATX_PW_FLAT && wide && yp == xp && xp % VEC == 0 && xp / VEC <= 0x7fffffffll / 2) {
            const int Cp = (int)(xp / VEC);
            const int64_t n_vec = n_pts * Cp;
            // (1) programs uniform over the levels: operators by value, one vector per lane, no loop (pointwise_cols_uniform_kernel)
#ifndef ATX_PW_UNIFORM
#define ATX_PW_UNIFORM 1
#endif
            UniformOps<T> uni{};
            if (ATX_PW_UNIFORM && (n_vec + kBlock - 1) / kBlock <= 0x7fffffffll && uniform_level_program<T>(host_prog, n_stage, mask != nullptr, n_lev, VEC, uni)) {
                bool two_pieces = false, uses_mask = false;
                unsigned act_bits = 0;
                for (int s = 0; s < n_stage; ++s) {
                    two_pieces = two_pieces || uni.split[s] < C;
                    uses_mask = uses_mask || uni.stage[s].use_mask || uni.second[s].use_mask;
                    if (uni.stage[s].op != ATX_OP_COPY || uni.stage[s].use_mask) act_bits |= 1u << s;
                    if (uni.second[s].op != ATX_OP_COPY || uni.second[s].use_mask) act_bits |= 1u << (4 + s);
                }
                const int need_rc = (two_pieces || (uses_mask && mask) || !tight) ? 1 : 0;
                const bool trans = program_has_transcendental(host_prog, n_stage, n_lev);
#ifndef ATX_PW_UNIFORM_U_IN
#define ATX_PW_UNIFORM_U_IN 1
#endif
#ifndef ATX_PW_UNIFORM_U_OUT
#define ATX_PW_UNIFORM_U_OUT 1
#endif
#ifndef ATX_PW_UNIFORM_NT
#define ATX_PW_UNIFORM_NT 1
#endif
#ifndef ATX_PW_TRANS_U
#define ATX_PW_TRANS_U 1  // vectors per lane of programs with exp / log (A/B knob, tools/experiments/trans_ab.py)
#endif
#ifndef ATX_PW_TRANS_NT
#define ATX_PW_TRANS_NT 1  // non-temporal accesses for programs with exp / log too
#endif
#define ATX_PW_UNIFORM_LAUNCH(TR_, U_, NT_)                                                                                                \
    hipLaunchKernelGGL((pointwise_cols_uniform_kernel<T, VEC, TR_, U_, NT_>), dim3((unsigned)((n_vec + kBlock * U_ - 1) / (kBlock * U_))), \
                       dim3(kBlock), 0, st, x, y, n_vec, C, Cp, uni, uses_mask ? mask : nullptr, need_rc, in_place, act_bits)
#ifndef ATX_PW_MASK_NT
#define ATX_PW_MASK_NT 0
#endif
                const bool nt = ATX_PW_UNIFORM_NT && (ATX_PW_MASK_NT || !(uses_mask && mask));
                if (in_place) {
                    if (trans && nt && ATX_PW_TRANS_NT) ATX_PW_UNIFORM_LAUNCH(true, ATX_PW_TRANS_U, true);
                    else if (trans) ATX_PW_UNIFORM_LAUNCH(true, ATX_PW_TRANS_U, false);
                    else if (nt) ATX_PW_UNIFORM_LAUNCH(false, ATX_PW_UNIFORM_U_IN, true);
                    else ATX_PW_UNIFORM_LAUNCH(false, ATX_PW_UNIFORM_U_IN, false);
                } else {
                    if (trans && nt && ATX_PW_TRANS_NT) ATX_PW_UNIFORM_LAUNCH(true, ATX_PW_TRANS_U, true);
                    else if (trans) ATX_PW_UNIFORM_LAUNCH(true, ATX_PW_TRANS_U, false);
                    else if (nt) ATX_PW_UNIFORM_LAUNCH(false, ATX_PW_UNIFORM_U_OUT, true);
                    else ATX_PW_UNIFORM_LAUNCH(false, ATX_PW_UNIFORM_U_OUT, false);
                }
#undef ATX_PW_UNIFORM_LAUNCH
                ATX_LAUNCH_CHECK("pointwise_stack_uniform");
                return ATX_OK;
            }
            // (1b) runs of levels with boundaries anywhere (several variables in one column), out of place or in place: by value as well
#ifndef ATX_PW_RUNS
#define ATX_PW_RUNS 1
#endif
            // Measured against the routes below (tools/experiments/runs_probe.py, profiles/r04_runs_probe.log): evaluating every run's operator and
            // selecting costs a streaming kernel more than it saves — one stage of 3 runs 0.72 / 0.67 (f32 / f64) here against 0.76 / 0.72 on the
            // table and per-level kernels — except for float32 programs of two or more stages, which those kernels run at 0.52-0.58 and this one
            // at 0.65-0.66.  So only they take it.
            RunOps<T> runs{};
#ifndef ATX_PW_RUNS_ALL
#define ATX_PW_RUNS_ALL 0  // A/B: every run-structured program on the runs kernel
#endif
            if (ATX_PW_RUNS && (ATX_PW_RUNS_ALL || (sizeof(T) == 4 && n_stage >= 2)) && (n_vec + kBlock - 1) / kBlock <= 0x7fffffffll &&
                runs_level_program<T>(host_prog, n_stage, mask != nullptr, n_lev, runs)) {
                bool uses_mask = false;
                for (int s = 0; s < n_stage; ++s)
                    for (int r = 0; r < runs.n_run[s]; ++r) uses_mask = uses_mask || runs.op[s][r].use_mask != 0;
                const bool trans = program_has_transcendental(host_prog, n_stage, n_lev);
                const bool nt = !(uses_mask && mask);
                const dim3 grid((unsigned)((n_vec + kBlock - 1) / kBlock));
#define ATX_PW_RUNS_LAUNCH(TR_, NT_) \
    hipLaunchKernelGGL((pointwise_cols_runs_kernel<T, VEC, TR_, NT_>), grid, dim3(kBlock), 0, st, x, y, n_vec, C, Cp, runs, uses_mask ? mask : nullptr)
                if (trans) { if (nt) ATX_PW_RUNS_LAUNCH(true, true); else ATX_PW_RUNS_LAUNCH(true, false); }
                else { if (nt) ATX_PW_RUNS_LAUNCH(false, true); else ATX_PW_RUNS_LAUNCH(false, false); }
#undef ATX_PW_RUNS_LAUNCH
                ATX_LAUNCH_CHECK("pointwise_stack_runs");
                return ATX_OK;
            }
            // (2) operators differing from level to level.  Measured (137 levels of O1280, profiles/r03_pointwise_ab.log): the table
            // kernel (one vector per lane, no loop, operators from the host-built per-vector table) wins for
            // one-stage f32 programs without a mask, out of place (1.21 ms both) AND in place (1.22 vs 1.36 ms chunked); it loses in f64
            // (2.48 vs 2.37 ms), with two stages (f32 1.56 vs 1.28, f64 3.38 vs 2.58 ms: 24-32 B of operators per stage and vector) and
            // with a point mask in f64 (2.74 vs 2.43 ms).  ATX_PW_TABLE_RULE: 0 = round 2's rule (out of place only), 1 = this rule.
#ifndef ATX_PW_TABLE_RULE
#define ATX_PW_TABLE_RULE 1
#endif
            const bool table_narrow = !mask && sizeof(T) == 4 && n_stage == 1 && !in_place;
            const bool table_wide = ATX_PW_TABLE_RULE == 1 && !mask && sizeof(T) == 4 && n_stage == 1;
            if (tight && vec_prog && (table_narrow || table_wide) && (n_vec + kBlock - 1) / kBlock <= 0x7fffffffll &&
                !program_has_mixed_vectors<T>(host_prog, n_stage, n_lev, VEC)) {  // (mixed vectors: the chunked kernel serves them better)
                hipLaunchKernelGGL((pointwise_cols_table_kernel<T, VEC>), dim3((unsigned)((n_vec + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                                   x, y, n_vec, n_lev, C, prog, vec_prog, n_stage, mask, in_place);
                ATX_LAUNCH_CHECK("pointwise_stack");
                return ATX_OK;
            }
            // (3) one stage, one operator kind per vector (a scale per level), float64 or in place: the no-loop kernel on the typed
            // per-level part of vec_prog (pointwise_cols_typed_kernel; everything else measured faster on the per-level LDS kernel below)
#ifndef ATX_PW_TYPED
#define ATX_PW_TYPED 1
#endif
            if (ATX_PW_TYPED && tight && n_stage == 1 && (sizeof(T) == 8 || in_place) && vec_prog && aligned16(vec_prog) && host_prog &&
                (n_vec + kBlock - 1) / kBlock <= 0x7fffffffll) {
                bool one_kind = true, uses_mask = false;
                for (int l0 = 0; l0 < n_lev && one_kind; l0 += VEC)
                    for (int l = l0; l < n_lev && l < l0 + VEC; ++l) {
                        one_kind = one_kind && host_prog[l].op == host_prog[l0].op && (host_prog[l].use_mask != 0) == (host_prog[l0].use_mask != 0);
                        uses_mask = uses_mask || host_prog[l].use_mask != 0;
                    }
                if (one_kind && (sizeof(T) == 8 || !(uses_mask && mask))) {  // (float32 with the point mask: 0.69 here against 0.71-0.75)
                    const unsigned char* level_tables = reinterpret_cast<const unsigned char*>(vec_prog) +
                                                        level_tables_layout(1, n_lev, sizeof(T) == 4 ? ATX_F32 : ATX_F64).levels_offset;
                    const bool trans = program_has_transcendental(host_prog, n_stage, n_lev);
                    const bool nt = !(uses_mask && mask);
                    const dim3 grid((unsigned)((n_vec + kBlock - 1) / kBlock));
#define ATX_PW_TYPED_LAUNCH(TR_, NT_)                                                                                               \
    hipLaunchKernelGGL((pointwise_cols_typed_kernel<T, VEC, TR_, NT_>), grid, dim3(kBlock), 0, st, x, y, n_vec, C, level_tables, \
                       uses_mask ? mask : nullptr, in_place)
                    if (trans) { if (nt) ATX_PW_TYPED_LAUNCH(true, true); else ATX_PW_TYPED_LAUNCH(true, false); }
                    else { if (nt) ATX_PW_TYPED_LAUNCH(false, true); else ATX_PW_TYPED_LAUNCH(false, false); }
#undef ATX_PW_TYPED_LAUNCH
                    ATX_LAUNCH_CHECK("pointwise_stack_typed");
                    return ATX_OK;
                }
            }
            const int64_t n_chunks = (n_vec + (int64_t)kBlock * kPwUnroll - 1) / ((int64_t)kBlock * kPwUnroll);
            int64_t blocks = n_chunks > kMaxGrid ? kMaxGrid : n_chunks;
            const int64_t per = (n_chunks + blocks - 1) / blocks;
            blocks = (n_chunks + per - 1) / per;  // contiguous runs of `per` chunks: no workgroup without work
#ifndef ATX_PW_LEVELS
#define ATX_PW_LEVELS 1  // 0: round 2's chunked kernel (per-vector operators in LDS, mixed vectors from the global program)
#endif
            const size_t lds_levels = level_tables_lds_bytes<T>(n_stage, C, VEC) + (size_t)C;
            // (loose pitches: measured slower here than on the row-chunk kernel below — f64 0.57 against 0.69 — so only the by-value
            // kernel above takes them; tools/experiments/loose_pitch.py)
            if (ATX_PW_LEVELS && tight && lds_levels <= 64 * 1024) {
#ifndef ATX_PW_LEVELS_NT
#define ATX_PW_LEVELS_NT 1
#endif
                bool nt = ATX_PW_LEVELS_NT && sizeof(T) == 4 && host_prog != nullptr;
                if (nt && in_place) {  // every vector must be touched
                    for (int c = 0; c < C && nt; ++c) {
                        bool act = false;
                        for (int s = 0; s < n_stage && !act; ++s)
                            for (int l = c * VEC; l < n_lev && l < (c + 1) * VEC && !act; ++l)
                                act = host_prog[(int64_t)s * n_lev + l].op != ATX_OP_COPY || host_prog[(int64_t)s * n_lev + l].use_mask != 0;
                        nt = act;
                    }
                }
                if (program_has_transcendental(host_prog, n_stage, n_lev))
                    hipLaunchKernelGGL((pointwise_cols_levels_kernel<T, VEC, true, false>), dim3((unsigned)blocks), dim3(kBlock), lds_levels, st, x, y,
                                       n_pts, n_lev, C, Cp, prog, n_stage, mask, in_place);
                else if (nt)
                    hipLaunchKernelGGL((pointwise_cols_levels_kernel<T, VEC, false, true>), dim3((unsigned)blocks), dim3(kBlock), lds_levels, st, x, y,
                                       n_pts, n_lev, C, Cp, prog, n_stage, mask, in_place);
                else
                    hipLaunchKernelGGL((pointwise_cols_levels_kernel<T, VEC, false, false>), dim3((unsigned)blocks), dim3(kBlock), lds_levels, st, x, y,
                                       n_pts, n_lev, C, Cp, prog, n_stage, mask, in_place);
                ATX_LAUNCH_CHECK("pointwise_stack_levels");
                return ATX_OK;
            }
            if (tight) {  // very tall stacks: round 2's chunked kernel, whose per-VECTOR table is smaller (tight pitches only)
                const size_t lds_flat = lds + (size_t)C;
                if (lds_flat > 64 * 1024 && n_stage > 1) return ATX_SPLIT_PROGRAM;  // the entry point runs the stages in two halves
                ATX_REQUIRE(lds_flat <= 64 * 1024, ATX_ENOTIMPL, "pointwise: one stage over %d levels needs %zu B of LDS", n_lev, lds_flat);
                if (program_has_transcendental(host_prog, n_stage, n_lev))
                    hipLaunchKernelGGL((pointwise_cols_flat_kernel<T, VEC, true>), dim3((unsigned)blocks), dim3(kBlock), lds_flat, st, x, y, n_pts,
                                       n_lev, C, prog, n_stage, mask, in_place);
                else  // no exp / log anywhere in the program: the low-register instantiation (8 instead of 4 waves per SIMD in float64)
                    hipLaunchKernelGGL((pointwise_cols_flat_kernel<T, VEC, false>), dim3((unsigned)blocks), dim3(kBlock), lds_flat, st, x, y, n_pts,
                                       n_lev, C, prog, n_stage, mask, in_place);
                ATX_LAUNCH_CHECK("pointwise_stack");
                return ATX_OK;
            }
        }
        const size_t lds_rows = level_tables_lds_bytes<T>(n_stage, C, wide ? VEC : 1);
        if (lds_rows > 64 * 1024 && n_stage > 1) return ATX_SPLIT_PROGRAM;  // the entry point runs the stages in two halves
        ATX_REQUIRE(lds_rows <= 64 * 1024, ATX_ENOTIMPL, "pointwise: one stage over %d levels needs %zu B of LDS", n_lev, lds_rows);
        const int Cg = C < kBlock ? C : kBlock;
        const int64_t chunk = (int64_t)(kBlock / Cg) * kPwUnroll;  // rows one workgroup moves per iteration
        int64_t blocks = (n_pts + chunk - 1) / chunk;
        if (blocks > kMaxGrid) blocks = kMaxGrid;
        if (wide)
            hipLaunchKernelGGL((pointwise_cols_kernel<T, VEC>), dim3((unsigned)blocks), dim3(kBlock), lds_rows, st, x, y, n_pts, n_lev, C, xp,
                               yp, prog, n_stage, mask, in_place);
        else
            hipLaunchKernelGGL((pointwise_cols_kernel<T, 1>), dim3((unsigned)blocks), dim3(kBlock), lds_rows, st, x, y, n_pts, n_lev, C, xp,
                               yp, prog, n_stage, mask, in_place);
    } else {
        ATX_REQUIRE(n_lev <= 65535, ATX_ENOTIMPL, "pointwise: n_lev=%d exceeds grid.y", n_lev);
        {
            const int vec = vec_ok ? VEC : 1;
            const int64_t per_block = (int64_t)kBlock * fields_vectors_per_lane<T>();
            const int64_t gx_rows = (std::max<int64_t>(n_pts / vec, 1) + per_block - 1) / per_block;
            ATX_REQUIRE(gx_rows <= 0x7fffffffll, ATX_ENOTIMPL, "pointwise: %lld points per field exceed one launch", (long long)n_pts);
            {
                bool uses_mask = mask != nullptr;  // unknown program: assume it reads the mask it was given
                if (host_prog && mask) {
                    uses_mask = false;
                    for (int64_t i = 0; i < (int64_t)n_stage * n_lev; ++i) uses_mask = uses_mask || host_prog[i].use_mask != 0;
                }
                const bool trans = program_has_transcendental(host_prog, n_stage, n_lev);
                const bool nt = !uses_mask;
                const int mask_vec = (reinterpret_cast<uintptr_t>(mask) % (uintptr_t)vec) == 0 ? 1 : 0;
                const dim3 grid((unsigned)gx_rows, (unsigned)n_lev);
#define ATX_PW_ROWS_LAUNCH(V_, TR_, NT_)                                                                                                 \
    hipLaunchKernelGGL((pointwise_fields_rows_kernel<T, V_, TR_, NT_>), grid, dim3(kBlock), 0, st, x, y, n_pts, n_lev, xp, yp, prog, n_stage, \
                       mask, in_place, mask_vec)
                if (vec_ok) {
                    if (trans) { if (nt) ATX_PW_ROWS_LAUNCH(VEC, true, true); else ATX_PW_ROWS_LAUNCH(VEC, true, false); }
                    else { if (nt) ATX_PW_ROWS_LAUNCH(VEC, false, true); else ATX_PW_ROWS_LAUNCH(VEC, false, false); }
                } else {
                    if (trans) ATX_PW_ROWS_LAUNCH(1, true, false);
                    else ATX_PW_ROWS_LAUNCH(1, false, false);
                }
#undef ATX_PW_ROWS_LAUNCH
                ATX_LAUNCH_CHECK("pointwise_stack_fields");
                return ATX_OK;
            }
        }
    }
    ATX_LAUNCH_CHECK("pointwise_stack");
    return ATX_OK;
}

}  // namespace atx

using namespace atx;

extern "C" int atx_pointwise_stack(const void* x, void* y, int64_t n_pts, int64_t n_lev, int64_t x_pitch,
                                   int64_t y_pitch, int dtype, int layout, const atx_level_op* prog,
                                   const atx_level_op* vec_prog, const atx_level_op* host_prog, int32_t n_stage,
                                   const uint8_t* point_mask, void* stream) {
    ATX_REQUIRE(prog && ((x && y) || n_pts == 0), ATX_EINVAL, "atx_pointwise_stack: null pointer");  // (an empty stack may have no storage)
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
    const int rc = dtype == ATX_F32
        ? pointwise_typed<float>(x, y, n_pts, (int)n_lev, x_pitch, y_pitch, layout, prog, vec_prog, host_prog, n_stage, point_mask, s)
        : pointwise_typed<double>(x, y, n_pts, (int)n_lev, x_pitch, y_pitch, layout, prog, vec_prog, host_prog, n_stage, point_mask, s);
    if (rc != ATX_SPLIT_PROGRAM) return rc;
    // The program's per-level tables exceed the LDS budget (n_stage > 1 here): stages compose, y = s_{n-1}( ... s_0(x)), so the first
    // half goes x -> y and the second half y -> y in place — same statements in the same order, same bits.  vec_prog is laid out for the
    // whole program and does not travel with a part of it.
    const int32_t first = n_stage / 2;
    const int rc_first = atx_pointwise_stack(x, y, n_pts, n_lev, x_pitch, y_pitch, dtype, layout, prog, nullptr, host_prog, first, point_mask, stream);
    if (rc_first != ATX_OK) return rc_first;
    return atx_pointwise_stack(y, y, n_pts, n_lev, y_pitch, y_pitch, dtype, layout, prog + (int64_t)first * n_lev, nullptr,
                               host_prog ? host_prog + (int64_t)first * n_lev : nullptr, n_stage - first, point_mask, stream);
}

extern "C" int atx_mask_build(const void* m, int64_t m_stride, uint8_t* mask, int64_t n, int cmp, double threshold,
                              int dtype, void* stream) {
    ATX_REQUIRE((m && mask) || n == 0, ATX_EINVAL, "atx_mask_build: null pointer");
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
    ATX_REQUIRE(count && (mask || n == 0), ATX_EINVAL, "atx_mask_count: null pointer");
    ATX_REQUIRE(n >= 0, ATX_EINVAL, "atx_mask_count: negative n");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int st = hip_status(hipMemsetAsync(count, 0, sizeof(int64_t), s), "atx_mask_count memset");
    if (st != ATX_OK) return st;
    if (n == 0) return ATX_OK;
    const unsigned count_grid = grid_for(n) > 2048u ? 2048u : grid_for(n);  // one atomic per wave on one address: keep them few
    hipLaunchKernelGGL(mask_count_kernel, dim3(count_grid), dim3(kBlock), 0, s, mask, n, reinterpret_cast<unsigned long long*>(count));
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
    ATX_REQUIRE(count && workspace && ((mask && index) || n == 0), ATX_EINVAL, "atx_mask_to_index: null pointer");
    ATX_REQUIRE(n >= 0 && n <= INT32_MAX, ATX_EINVAL, "atx_mask_to_index: n=%lld outside int32", (long long)n);
    ATX_REQUIRE(workspace_bytes >= atx_mask_to_index_workspace(n), ATX_EWORKSPACE, "atx_mask_to_index: workspace %zu < %zu",
                workspace_bytes, atx_mask_to_index_workspace(n));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return hip_status(hipMemsetAsync(count, 0, sizeof(int64_t), s), "atx_mask_to_index memset");
    const int n_blocks = (int)((n + kChunk - 1) / kChunk);
    int32_t* block_counts = static_cast<int32_t*>(workspace);
    hipLaunchKernelGGL(compact_count_kernel, dim3(n_blocks), dim3(kBlock), 0, s, mask, n, block_counts);
    ATX_LAUNCH_CHECK("compact_count");
#ifndef ATX_COMPACT_SELF_SCAN
#define ATX_COMPACT_SELF_SCAN 4096  // workgroups up to which the scatter sums the counts before it by itself (16 M points); 0: never
#endif
    if (n_blocks <= ATX_COMPACT_SELF_SCAN) {
        hipLaunchKernelGGL(compact_scatter_kernel<true>, dim3(n_blocks), dim3(kBlock), 0, s, mask, n, block_counts, index, reinterpret_cast<long long*>(count));
        ATX_LAUNCH_CHECK("compact_scatter");
        return ATX_OK;
    }
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, n_blocks, reinterpret_cast<long long*>(count));
    ATX_LAUNCH_CHECK("compact_scan");
    hipLaunchKernelGGL(compact_scatter_kernel<false>, dim3(n_blocks), dim3(kBlock), 0, s, mask, n, block_counts, index, nullptr);
    ATX_LAUNCH_CHECK("compact_scatter");
    return ATX_OK;
}

static int reduce_rows(const void* x, int64_t n_rows, int64_t row_len, int64_t pitch, int red, double* result, int dtype,
                       void* workspace, size_t workspace_bytes, void* stream, const char* who) {
    ATX_REQUIRE(n_rows >= 0 && row_len >= 0, ATX_EINVAL, "%s: negative size", who);
    ATX_REQUIRE(result && (x || n_rows == 0 || row_len == 0), ATX_EINVAL, "%s: null pointer", who);  // (an empty array may have no storage)
    ATX_REQUIRE(pitch >= row_len, ATX_ESHAPE, "%s: pitch %lld shorter than a row of %lld", who, (long long)pitch, (long long)row_len);
    ATX_REQUIRE(red >= ATX_RED_MIN && red <= ATX_RED_MINMAX, ATX_EINVAL, "%s: bad reduction %d", who, red);
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "%s: bad dtype %d", who, dtype);
    ATX_REQUIRE(!workspace || workspace_bytes >= sizeof(RedWorkspace), ATX_EWORKSPACE, "%s: workspace %zu < %zu", who, workspace_bytes,
                sizeof(RedWorkspace));
    ATX_REQUIRE(!workspace || (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, ATX_EALIGN, "%s: workspace must be 8-byte aligned", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // 16-byte loads when every row starts on a 16-byte boundary and its last (partial) vector lies inside the pitch (a flat array is
    // one row: only the base must be aligned); MINMAX exists in this form only and falls back to two scalar passes otherwise
    const int vec = dtype == ATX_F32 ? 4 : 2;
    // a flat array (one row, no pitch to hide a partial vector in): whole vectors through the vector kernel, the <= 3 elements
    // left over through the scalar one — the library never reads past x[n)
    const int64_t tail = (n_rows == 1) ? row_len % vec : 0;
    const int64_t vec_len = row_len - tail;
    const int64_t C = (vec_len + vec - 1) / vec;
    const bool vec_ok = aligned16(x) && (n_rows == 1 || (pitch % vec == 0 && C * vec <= pitch)) && C <= 0x7fffffff;
    // With a workspace EVERY shape finishes in two levels — partials with plain stores, one combining workgroup, a plain store of the
    // result, which may therefore be a pinned host cell: the single-pass case, a flat array's tail after its last whole vector (one
    // more slot), the scalar fallback's two MINMAX passes (MIN into a[], MAX into b[]) and empty input (the identities).  Without
    // one the workgroups combine through atomics on `result`, which must then be device memory (round 3 dropped the workspace for
    // every shape but the first and ran up to 8192 CAS loops over PCIe on the caller's pinned cell — the advisor's finding).
    RedWorkspace* ws = static_cast<RedWorkspace*>(workspace);
    const int ra = red == ATX_RED_NANCOUNT ? ATX_RED_NANCOUNT : ((red == ATX_RED_MIN || red == ATX_RED_MINMAX) ? ATX_RED_MIN : ATX_RED_MAX);
    if (!ws) {
        hipLaunchKernelGGL(reduce_init_kernel, dim3(1), dim3(1), 0, s, result, red);
        ATX_LAUNCH_CHECK("reduce_init");
    }
    if (n_rows == 0 || row_len == 0) {
        if (ws) {  // nothing to combine: the identities (min +inf, max -inf, count 0), as reduce_init writes them
            hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kBlock), 0, s, ws, 0, ra, red, result);
            ATX_LAUNCH_CHECK("reduce_final");
        }
        return ATX_OK;
    }
    const int64_t grid_cap = kRedGrid - (tail > 0 ? 1 : 0);  // a tail takes one slot of the workspace
    if (vec_ok) {
        unsigned grid = 0;
        if (C > 0) {
            int64_t blocks = (n_rows * C + (int64_t)kBlock * kRedUnroll - 1) / ((int64_t)kBlock * kRedUnroll);
            grid = (unsigned)(blocks > grid_cap ? grid_cap : (blocks < 1 ? 1 : blocks));
            if (dtype == ATX_F32)
                hipLaunchKernelGGL((reduce_vec_kernel<float, 4>), dim3(grid), dim3(kBlock), 0, s, static_cast<const float*>(x), n_rows, vec_len, (int)C, pitch, red, result, ws);
            else
                hipLaunchKernelGGL((reduce_vec_kernel<double, 2>), dim3(grid), dim3(kBlock), 0, s, static_cast<const double*>(x), n_rows, vec_len, (int)C, pitch, red, result, ws);
            ATX_LAUNCH_CHECK("reduce_vec");
        }
        if (tail > 0) {
            const size_t esz = dtype == ATX_F32 ? 4 : 8;
            const void* xt = static_cast<const char*>(x) + (size_t)vec_len * esz;
            if (ws) {
                if (dtype == ATX_F32)
                    hipLaunchKernelGGL(reduce_tail_kernel<float>, dim3(1), dim3(1), 0, s, static_cast<const float*>(xt), (int)tail, red, ws, (int)grid);
                else
                    hipLaunchKernelGGL(reduce_tail_kernel<double>, dim3(1), dim3(1), 0, s, static_cast<const double*>(xt), (int)tail, red, ws, (int)grid);
                grid += 1;
            } else {  // combines into the same result cells (atomics): MINMAX as min -> result[0], max -> result[1]
                for (int pass = 0; pass < (red == ATX_RED_MINMAX ? 2 : 1); ++pass) {
                    const int r = red == ATX_RED_MINMAX ? (pass == 0 ? ATX_RED_MIN : ATX_RED_MAX) : red;
                    if (dtype == ATX_F32)
                        hipLaunchKernelGGL(reduce_kernel<float>, dim3(1), dim3(kBlock), 0, s, static_cast<const float*>(xt), (int64_t)1, tail, tail, r, result + pass, (double*)nullptr);
                    else
                        hipLaunchKernelGGL(reduce_kernel<double>, dim3(1), dim3(kBlock), 0, s, static_cast<const double*>(xt), (int64_t)1, tail, tail, r, result + pass, (double*)nullptr);
                }
            }
            ATX_LAUNCH_CHECK("reduce_tail");
        }
        if (ws) {
            hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kBlock), 0, s, ws, (int)grid, ra, red, result);
            ATX_LAUNCH_CHECK("reduce_final");
        }
        return ATX_OK;
    }
    // scalar form (unaligned base or pitch): one pass, or MIN then MAX for MINMAX
    int64_t blocks = (n_rows * row_len + (int64_t)kBlock * kRedUnroll - 1) / ((int64_t)kBlock * kRedUnroll);
    const unsigned grid = (unsigned)(blocks > kRedGrid ? kRedGrid : (blocks < 1 ? 1 : blocks));
    for (int pass = 0; pass < (red == ATX_RED_MINMAX ? 2 : 1); ++pass) {
        const int r = red == ATX_RED_MINMAX ? (pass == 0 ? ATX_RED_MIN : ATX_RED_MAX) : red;
        double* partials = ws ? (pass == 0 ? ws->a : ws->b) : nullptr;
        if (dtype == ATX_F32)
            hipLaunchKernelGGL(reduce_kernel<float>, dim3(grid), dim3(kBlock), 0, s, static_cast<const float*>(x), n_rows, row_len, pitch, r, result + pass, partials);
        else
            hipLaunchKernelGGL(reduce_kernel<double>, dim3(grid), dim3(kBlock), 0, s, static_cast<const double*>(x), n_rows, row_len, pitch, r, result + pass, partials);
    }
    ATX_LAUNCH_CHECK("reduce");
    if (ws) {
        hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kBlock), 0, s, ws, (int)grid, ra, red, result);
        ATX_LAUNCH_CHECK("reduce_final");
    }
    return ATX_OK;
}

extern "C" size_t atx_reduce_workspace(void) { return sizeof(RedWorkspace); }

extern "C" int atx_reduce(const void* x, int64_t n, int red, double* result, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return reduce_rows(x, 1, n, n, red, result, dtype, workspace, workspace_bytes, stream, "atx_reduce");
}

extern "C" int atx_reduce_stack(const void* x, int64_t n_pts, int64_t n_lev, int64_t pitch, int red, double* result,
                                int dtype, int layout, void* workspace, size_t workspace_bytes, void* stream) {
    ATX_REQUIRE(layout == ATX_COLUMNS || layout == ATX_FIELDS, ATX_EINVAL, "atx_reduce_stack: bad layout %d", layout);
    if (layout == ATX_COLUMNS) return reduce_rows(x, n_pts, n_lev, pitch, red, result, dtype, workspace, workspace_bytes, stream, "atx_reduce_stack");
    return reduce_rows(x, n_lev, n_pts, pitch, red, result, dtype, workspace, workspace_bytes, stream, "atx_reduce_stack");
}

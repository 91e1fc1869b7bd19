// Regrid kernels: precomputed index(+weight) gather over a stack of levels.
//
// Replaces, for all levels of a stack in ONE launch, the per-field statements
//   R: filters/fields/regrid.py:380  data[..., self.nearest_grid_points]
//   R: filters/fields/regrid.py:310  self.matrix @ data          (scipy csr_matvec)
//   R: filters/fields/regrid.py:420  data[..., self.mask]
// that the reference runs once per 2-D field in a Python loop (regrid.py:204-208).
//
// HBM-bound gather, no MFMA.  Design:
//  * ATX_COLUMNS stacks, fixed k <= 4 without epilogue — the headline case — run the DIRECT kernel: one
//    (target, 16-byte vector) item per lane, no shared memory, no barrier, no loop; consecutive lanes read consecutive
//    16 B of one source column and write consecutive 16 B of the output.
//  * epilogues and runtime k run the TILED kernel: a workgroup owns a tile of consecutive targets, stages their neighbour
//    indices / weights (and the per-vector operator table) in LDS once, then sweeps the flattened (target, vector) items
//    with up to 4 items per lane in flight; general CSR rows likewise from a staged slice of the CSR arrays.
//  * workgroups are dealt to XCDs in contiguous ranges (xcd_tile) so neighbouring targets that share source columns share
//    an L2; several stacks of one shape share a launch (grid.y = stack).
//  * ATX_FIELDS stacks: lane = target, neighbour indices / weights live in registers and are reused for every level of
//    the level chunk.
#include <type_traits>

#include "atx_common.hpp"

#include <cstring>

namespace atx {

// A/B knobs (build a variant with -D..., compare with tools/ab_bench.py; the defaults are the measured winners, logs under
// profiles/r01_ab_*.log): items in flight per lane of the tiled kernel (2 and 8: no gain), non-temporal output stores (+3 %),
// non-temporal index / weight loads IN THE TILED KERNEL (+2 %: every word is read once there; the direct kernel reads a
// target's words from several lanes and waves and uses plain loads), non-temporal SOURCE loads (no gain), XCD-contiguous block
// ranges off (-3 %), lanes per workgroup of the tiled kernel (64 / 128 / 512: same plateau).
#ifndef ATX_UNROLL
#define ATX_UNROLL 4
#endif
#ifndef ATX_NT_STORE
#define ATX_NT_STORE 1
#endif
#ifndef ATX_NT_IDX
#define ATX_NT_IDX 1
#endif
#ifndef ATX_NT_SRC
#define ATX_NT_SRC 0
#endif
#ifndef ATX_NO_XCD
#define ATX_NO_XCD 0
#endif
#ifndef ATX_ELL_BLOCK
#define ATX_ELL_BLOCK 256
#endif
constexpr int kEllBlock = ATX_ELL_BLOCK;  // lanes per workgroup of the columns ELL kernel
constexpr int kUnroll = ATX_UNROLL;  // items in flight per lane (columns kernels)

template <typename T, int N>
struct NativeVec {
    typedef T type __attribute__((ext_vector_type(N)));
};
template <typename T>
struct NativeVec<T, 1> {
    typedef T type;
};

template <typename T, int N>
__device__ __forceinline__ void store_out(Pack<T, N>* p, const Pack<T, N>& v) {
#if ATX_NT_STORE
    // output is written once and never re-read by this launch
    using NV = typename NativeVec<T, N>::type;
    __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&v), reinterpret_cast<NV*>(p));
#else
    *p = v;
#endif
}
template <typename T, int N>
__device__ __forceinline__ Pack<T, N> load_src(const T* p) {
#if ATX_NT_SRC
    using NV = typename NativeVec<T, N>::type;
    NV v = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
    return *reinterpret_cast<Pack<T, N>*>(&v);
#else
    return *reinterpret_cast<const Pack<T, N>*>(p);
#endif
}
#ifndef ATX_FIELDS_NT
#define ATX_FIELDS_NT 0
#endif
template <typename T>
__device__ __forceinline__ void store_scalar(T* p, T v) {
#if ATX_FIELDS_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename T>
__device__ __forceinline__ T load_once(const T* p) {
#if ATX_NT_IDX
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// ---------------------------------------------------------------------------------
// ATX_COLUMNS, fixed k (ELL).  K > 0: compile-time k; K == 0: runtime k.
// ---------------------------------------------------------------------------------
// Up to kMaxBatch stacks of identical shape share one launch (several variables or time steps on the same grid
// pair, or the N source stacks of a target-sharded multi-GPU step): blockIdx.y selects the stack, so the launch gaps
// and per-launch tails of a stack-by-stack loop disappear.  The pointers travel by value in the kernel arguments.
constexpr int kMaxBatch = 16;
struct EllBatch {
    const void* src[kMaxBatch];
    void* out[kMaxBatch];
    int n;
};

template <typename T, int VEC, int K, bool WEIGHTED, bool EPI, bool PAD>
__global__ void __launch_bounds__(kEllBlock)
regrid_cols_ell_kernel(EllBatch batch,
                       const int32_t* __restrict__ idx, const T* __restrict__ w,
                       int64_t n_tgt, int k_rt, int n_lev, int C,
                       int64_t src_pitch, int64_t out_pitch, int tile, unsigned n_tiles,
                       const atx_level_op* __restrict__ prog, int n_stage,
                       const uint8_t* __restrict__ tgt_mask, const int32_t* __restrict__ tgt_rows) {
    using V = Pack<T, VEC>;
    // items in flight per lane: the epilogue variant trades half of them for registers (its operator
    // dispatch would otherwise push the kernel from 5 to 2-4 waves per SIMD; 2 vs 4 in flight costs ~1 %)
    constexpr int kU = EPI ? (kUnroll > 2 ? 2 : kUnroll) : kUnroll;
    extern __shared__ __align__(16) unsigned char smem[];
    const int k = K > 0 ? K : k_rt;
    // LDS carve: weights (widest type first), indices, then the level program
    T* w_s = reinterpret_cast<T*>(smem);
    int32_t* idx_s = reinterpret_cast<int32_t*>(w_s + (WEIGHTED ? (size_t)tile * k : 0));
    unsigned char* prog_s = smem + (((WEIGHTED ? (size_t)tile * k * sizeof(T) : 0) + (size_t)tile * k * sizeof(int32_t) + 15) & ~size_t(15));

#if ATX_NO_XCD
    const unsigned tile_id = blockIdx.x;
#else
    const unsigned tile_id = xcd_tile(blockIdx.x, n_tiles);
#endif
    const int64_t t0 = (int64_t)tile_id * tile;
    const int nt = (int)min((int64_t)tile, n_tgt - t0);
    const int tid = threadIdx.x;

    for (int i = tid; i < nt * k; i += kEllBlock) {
        idx_s[i] = load_once(idx + t0 * k + i);
        if (WEIGHTED) w_s[i] = load_once(w + t0 * k + i);
    }
    LevelTablesLds<T> tab{};  // the operators of every level (atx_common.hpp)
    if (EPI) tab = build_level_tables<T, VEC>(prog, prog_s, n_stage, n_lev, C, tid, kEllBlock);
    __syncthreads();

    const int items = nt * C;
    const int dt = kEllBlock / C;
    const int dc = kEllBlock - dt * C;

    // stacks of a batch: grid.y (default) keeps workgroups short — measured 0.45 ms per 8-stack step on a 1/8 target shard
    // against 0.51 ms for a loop over the stacks inside the workgroup (which stages the tile once but runs 8x longer) and
    // 0.50-0.52 ms for 8 separate launches (profiles/r01_shard_balance.log)
#ifndef ATX_BATCH_LOOP
#define ATX_BATCH_LOOP 0
#endif
#if ATX_BATCH_LOOP
    for (int stack = 0; stack < batch.n; ++stack) {
#else
    {
        const int stack = blockIdx.y;
#endif
        const T* __restrict__ src = static_cast<const T*>(batch.src[stack]);
        T* __restrict__ out = static_cast<T*>(batch.out[stack]);
        int t = tid / C;
        int c = tid - t * C;

        for (int q = tid; q < items; q += kEllBlock * kU) {
            int tt[kU], cc[kU];
            bool ok[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                ok[u] = (q + u * kEllBlock) < items;
                tt[u] = ok[u] ? t : 0;
                cc[u] = ok[u] ? c : 0;
                t += dt;
                c += dc;
                if (c >= C) { c -= C; ++t; }
            }

            V acc[kU];
            if (K > 0) {
                // all K*kU loads are independent: issue them before any arithmetic
                V v[kU][K > 0 ? K : 1];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
#pragma unroll
                    for (int j = 0; j < (K > 0 ? K : 1); ++j) {
                        int64_t p = idx_s[tt[u] * K + j];
                        // absent entry of a padded row: load anything valid, skipped below (predicating the load instead was
                        // measured 20 % slower: the branch breaks up the batch of independent loads)
                        if (PAD && p < 0) p = 0;
                        v[u][j] = load_src<T, VEC>(src + p * src_pitch + (int64_t)cc[u] * VEC);
                    }
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    if (WEIGHTED) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[u].v[e] = T(0);
#pragma unroll
                        for (int j = 0; j < (K > 0 ? K : 1); ++j) {
                            const T wj = w_s[tt[u] * K + j];
                            if (!PAD || idx_s[tt[u] * K + j] >= 0) {
#pragma unroll
                                for (int e = 0; e < VEC; ++e) acc[u].v[e] = acc[u].v[e] + wj * v[u][j].v[e];
                            }
                        }
                    } else {
                        acc[u] = v[u][0];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < kU; ++u) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[u].v[e] = T(0);
                    const int base = tt[u] * k;
                    int j = 0;
                    for (; j + 2 <= k; j += 2) {
                        const int64_t pa = idx_s[base + j], pb = idx_s[base + j + 1];
                        const V va = *reinterpret_cast<const V*>(src + ((PAD && pa < 0) ? 0 : pa) * src_pitch + (int64_t)cc[u] * VEC);
                        const V vb = *reinterpret_cast<const V*>(src + ((PAD && pb < 0) ? 0 : pb) * src_pitch + (int64_t)cc[u] * VEC);
                        const T wa = WEIGHTED ? w_s[base + j] : T(1), wb = WEIGHTED ? w_s[base + j + 1] : T(1);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            if (!PAD || pa >= 0) acc[u].v[e] = acc[u].v[e] + wa * va.v[e];
                            if (!PAD || pb >= 0) acc[u].v[e] = acc[u].v[e] + wb * vb.v[e];
                        }
                    }
                    if (j < k) {
                        const int64_t pa = idx_s[base + j];
                        const V va = *reinterpret_cast<const V*>(src + ((PAD && pa < 0) ? 0 : pa) * src_pitch + (int64_t)cc[u] * VEC);
                        const T wa = WEIGHTED ? w_s[base + j] : T(1);
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (!PAD || pa >= 0) acc[u].v[e] = acc[u].v[e] + wa * va.v[e];
                    }
                }
            }

#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (!ok[u]) continue;
                const int64_t row = tgt_rows ? (int64_t)tgt_rows[t0 + tt[u]] : t0 + tt[u];  // (uniform branch)
                if (EPI) {
                    const bool masked = tgt_mask ? (tgt_mask[row] != 0) : false;
                    apply_level_tables<T, VEC>(tab, n_stage, cc[u], acc[u], masked);
                }
                store_out(reinterpret_cast<V*>(out + row * out_pitch + (int64_t)cc[u] * VEC), acc[u]);
            }
        }
    }  // stacks of the batch
}

// ---------------------------------------------------------------------------------
// ATX_COLUMNS, fixed k, "direct" form: no shared memory, no barrier, no loop.  One (target, 16-byte vector) item per
// lane; a lane reads its target's k indices / weights itself (the ~35 lanes of a target read the same words: one
// request) and then the k source vectors.  The default for k <= 4 gathers without an epilogue, see the note at its launch
// site; the tiled kernel above serves epilogues and runtime k.
// ---------------------------------------------------------------------------------
// Epilogue of the direct kernel (the fused regrid -> per-point chain, K10) without shared memory or a barrier:
//   kEpiUniform  per stage, the levels run ONE operator, or one operator up to a level and another from there on (a stack of
//                "136 levels of t, then orog" — BASELINE config 5 — or of two variables), the change falling on a 16-byte
//                vector boundary; <= kMaxUniform stages, any operators, with or without the point mask: the operators travel BY
//                VALUE in the kernel arguments (scalar registers), the dispatch on the operator is a scalar branch; with two
//                pieces both are evaluated and the lane keeps the one its vector belongs to;
//   kEpiTable    programs of the multiply-add family only (COPY / AFFINE / MUL, with or without the point mask — rescale, convert,
//                orog_to_z, apply_mask: BASELINE config 5), <= kMaxTable stages, operators differing from level to level: the
//                per-VECTOR operator table the host built once (atx_vector_program, n_stage x C entries, a few hundred bytes
//                that stay in L1) is read one entry per stage and lane, REQUESTED BEFORE THE GATHER so its latency passes under
//                it, and applied without a branch (x*p0, (x*p0)+p1 and x are all formed, the operator selects: the same two
//                roundings as the statement it replaces).  Vectors whose levels differ (marker ATX_OP_MIXED) go level by level
//                through the per-level program.
// Everything else (clip / impute / exp / log / divisions, more stages) stays on the tiled kernel: its general operator switch
// costs registers (f64: 88 VGPRs, 5 waves per SIMD instead of 8) and time the gather cannot hide (profiles/r02_ab_epilogue_routes.log).
#ifndef ATX_PAD_SELF
#define ATX_PAD_SELF 1
#endif
constexpr int kEpiNone = 0, kEpiUniform = 1, kEpiTable = 2, kEpiRuns = 3;  // kEpiRuns (round 4): up to 4 runs of levels per stage, boundaries anywhere (RunOps)
constexpr int kMaxTable = 4;

// COPY / AFFINE / MUL (+ mask) on one element, branch-free; bit-identical to apply_level_op for these operators.
template <typename T>
__device__ __forceinline__ T apply_madd_family(const LevelOp<T>& o, T x, bool masked) {
    const T m = x * o.p0;
    const T a = m + o.p1;
    T y = o.op == ATX_OP_AFFINE ? a : (o.op == ATX_OP_MUL ? m : x);
    return (o.use_mask && masked) ? quiet_nan<T>() : y;
}

template <typename T, int VEC, int K, bool WEIGHTED, bool PAD, int EPI>
__global__ void __launch_bounds__(kEllBlock)
regrid_cols_ell_direct_kernel(EllBatch batch, const int32_t* __restrict__ idx, const T* __restrict__ w, int64_t n_items,
                              int C, int64_t src_pitch, int64_t out_pitch, unsigned n_blocks, int items_per_lane,
                              UniformOps<T> uniform, const unsigned char* __restrict__ level_tables, int n_stage, int n_lev,
                              const uint8_t* __restrict__ tgt_mask,
                              const int32_t* __restrict__ tgt_rows, RunOps<T> runs) {
    using V = Pack<T, VEC>;
    const T* __restrict__ src = static_cast<const T*>(batch.src[blockIdx.y]);
    T* __restrict__ out = static_cast<T*>(batch.out[blockIdx.y]);
#ifndef ATX_DIRECT_STRIPE
#define ATX_DIRECT_STRIPE 0  // 0: one contiguous range of workgroups per XCD; > 0: stripes of that many (A/B knob); < 0: plain round-robin
#endif
    const unsigned b = ATX_DIRECT_STRIPE > 0 ? xcd_stripe(blockIdx.x, n_blocks, (unsigned)ATX_DIRECT_STRIPE)
                                             : (ATX_DIRECT_STRIPE < 0 ? blockIdx.x : xcd_tile(blockIdx.x, n_blocks));
    for (int it = 0; it < items_per_lane; ++it) {
        const int64_t q = ((int64_t)b * items_per_lane + it) * kEllBlock + threadIdx.x;
        if (q >= n_items) return;
        const unsigned t = (unsigned)((uint64_t)q / (unsigned)C);
        const int c = (int)(q - (int64_t)t * C);
        // ordered traversal: the index / weight table is stored in the order the targets are to be visited (column blocks of the
        // target grid: vertically adjacent targets meet in L2) and row t of it belongs to output row tgt_rows[t]
        const unsigned row = tgt_rows ? (unsigned)tgt_rows[t] : t;
        int32_t p[K];
        T wv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {  // plain loads: a target's words are read again by the next wave when its vectors straddle two
            p[j] = idx[(int64_t)t * K + j];
            if (WEIGHTED) wv[j] = w[(int64_t)t * K + j];
        }
        // table route: the first operators and the mask byte are requested HERE, together with the index words, so that their
        // latency passes under the gather instead of after it (loaded after the accumulation they cost 9 %: 478 vs 439 us)
        V ta[kMaxTable], tb[kMaxTable];
        unsigned tcode[kMaxTable];
        bool masked = false;
        if (EPI == kEpiUniform || EPI == kEpiRuns) masked = tgt_mask ? (tgt_mask[row] != 0) : false;
        if (EPI == kEpiTable) {  // the operators of this vector's levels, in the stack's type (level_tables_layout)
            using OpWord = typename std::conditional<VEC == 4, uint32_t, uint16_t>::type;
            const int Lp = C * VEC;
            const T* tp0 = reinterpret_cast<const T*>(level_tables);
            const T* tp1 = tp0 + (int64_t)n_stage * Lp;
            const uint8_t* tcd = reinterpret_cast<const uint8_t*>(tp1 + (int64_t)n_stage * Lp);
#pragma unroll
            for (int s = 0; s < kMaxTable; ++s) {
                tcode[s] = 0;
                if (s < n_stage) {
                    ta[s] = *reinterpret_cast<const V*>(tp0 + (int64_t)s * Lp + c * VEC);
                    tb[s] = *reinterpret_cast<const V*>(tp1 + (int64_t)s * Lp + c * VEC);
                    tcode[s] = *reinterpret_cast<const OpWord*>(tcd + (int64_t)s * Lp + c * VEC);
                }
            }
            masked = tgt_mask ? (tgt_mask[row] != 0) : false;
        }
        V v[K];
        // an absent entry of a padded row (index -1) is skipped below; its load goes to the row's FIRST column — a line this lane is
        // fetching anyway — instead of column 0, which every padded lane of the launch would share (ATX_PAD_SELF=0: column 0)
        const int32_t spare = (ATX_PAD_SELF && p[0] >= 0) ? p[0] : 0;
#pragma unroll
        for (int j = 0; j < K; ++j)
            v[j] = load_src<T, VEC>(src + (int64_t)((PAD && p[j] < 0) ? spare : p[j]) * src_pitch + (int64_t)c * VEC);
        V acc;
        if (WEIGHTED) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc.v[e] = T(0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (!PAD || p[j] >= 0) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc.v[e] = acc.v[e] + wv[j] * v[j].v[e];
                }
            }
        } else {
            acc = v[0];
        }
        if (EPI == kEpiUniform) {
            for (int s = 0; s < uniform.n_stage; ++s) {
                if (uniform.split[s] >= C) {  // scalar condition: one piece
                    apply_level_op_vec<T, VEC>(uniform.stage[s], acc, masked);
                } else {
                    V other = acc;
                    apply_level_op_vec<T, VEC>(uniform.stage[s], acc, masked);
                    apply_level_op_vec<T, VEC>(uniform.second[s], other, masked);
                    if (c >= uniform.split[s]) acc = other;
                }
            }
        } else if (EPI == kEpiRuns) {
            apply_run_ops<T, VEC>(runs, c, acc, masked);
        } else if (EPI == kEpiTable) {
#pragma unroll
            for (int s = 0; s < kMaxTable; ++s) {
                if (s < n_stage) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const unsigned code = (tcode[s] >> (8 * e)) & 0xffu;
                        LevelOp<T> o;
                        o.op = (int)(code & 0x7fu);
                        o.use_mask = (int)(code >> 7);
                        o.p0 = ta[s].v[e];
                        o.p1 = tb[s].v[e];
                        acc.v[e] = apply_madd_family(o, acc.v[e], masked);
                    }
                }
            }
        }
        store_out(reinterpret_cast<V*>(out + (int64_t)row * out_pitch + (int64_t)c * VEC), acc);
    }
}

// ---------------------------------------------------------------------------------
// ATX_COLUMNS, general CSR.  The tile's slice of (indices, data) is contiguous in
// the CSR arrays: it is copied to LDS coalesced, then every lane walks its row
// from LDS (scipy order: sum starts at 0, entries in storage order).
// ---------------------------------------------------------------------------------
template <typename T, int VEC, bool EPI>
__global__ void __launch_bounds__(kBlock)
regrid_cols_csr_kernel(const T* __restrict__ src, T* __restrict__ out,
                       const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                       const T* __restrict__ data, int64_t n_tgt, int n_lev, int C,
                       int64_t src_pitch, int64_t out_pitch, int tile, unsigned n_tiles, int cap,
                       const atx_level_op* __restrict__ prog, int n_stage,
                       const uint8_t* __restrict__ tgt_mask, const int32_t* __restrict__ tgt_rows, int stripe) {
    using V = Pack<T, VEC>;
    extern __shared__ __align__(16) unsigned char smem[];
    T* w_s = reinterpret_cast<T*>(smem);
    int32_t* idx_s = reinterpret_cast<int32_t*>(w_s + cap);
    int32_t* rp_s = idx_s + cap;
    unsigned char* prog_s = smem + (((size_t)cap * (sizeof(T) + sizeof(int32_t)) + (size_t)(tile + 1) * sizeof(int32_t) + 15) & ~size_t(15));

    const unsigned tile_id = stripe > 0 ? xcd_stripe(blockIdx.x, n_tiles, (unsigned)stripe) : xcd_tile(blockIdx.x, n_tiles);
    const int64_t t0 = (int64_t)tile_id * tile;
    const int nt = (int)min((int64_t)tile, n_tgt - t0);
    const int tid = threadIdx.x;

    for (int i = tid; i <= nt; i += kBlock) rp_s[i] = indptr[t0 + i];
    LevelTablesLds<T> tab{};  // the operators of every level (atx_common.hpp)
    if (EPI) tab = build_level_tables<T, VEC>(prog, prog_s, n_stage, n_lev, C, tid, kBlock);
    __syncthreads();
    const int64_t base = rp_s[0];
    const int nnz_tile = rp_s[nt] - rp_s[0];
    const bool staged = nnz_tile <= cap;  // block-uniform
    if (staged) {
        for (int i = tid; i < nnz_tile; i += kBlock) {
            idx_s[i] = indices[base + i];
            w_s[i] = data[base + i];
        }
    }
    __syncthreads();

    const int items = nt * C;
    for (int q = tid; q < items; q += kBlock) {
        const int t = q / C;
        const int c = q - t * C;
        const int j0 = rp_s[t] - rp_s[0], j1 = rp_s[t + 1] - rp_s[0];
        V acc;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc.v[e] = T(0);
        int jj = j0;
        for (; jj + 4 <= j1; jj += 4) {
            int64_t p[4];
            T wv[4];
            V v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                p[u] = staged ? idx_s[jj + u] : indices[base + jj + u];
                wv[u] = staged ? w_s[jj + u] : data[base + jj + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const V*>(src + p[u] * src_pitch + (int64_t)c * VEC);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc.v[e] = acc.v[e] + wv[u] * v[u].v[e];
            }
        }
        for (; jj < j1; ++jj) {
            const int64_t p = staged ? idx_s[jj] : indices[base + jj];
            const T wv = staged ? w_s[jj] : data[base + jj];
            const V v = *reinterpret_cast<const V*>(src + p * src_pitch + (int64_t)c * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc.v[e] = acc.v[e] + wv * v.v[e];
        }
        const int64_t row = tgt_rows ? (int64_t)tgt_rows[t0 + t] : t0 + t;  // ordered traversal: CSR row t is output row tgt_rows[t]
        if (EPI) {
            const bool masked = tgt_mask ? (tgt_mask[row] != 0) : false;
            apply_level_tables<T, VEC>(tab, n_stage, c, acc, masked);
        }
        store_out(reinterpret_cast<V*>(out + row * out_pitch + (int64_t)c * VEC), acc);
    }
}

// ---------------------------------------------------------------------------------
// ATX_FIELDS, fixed k.  lane = target; grid.y = level chunk.
// ---------------------------------------------------------------------------------
template <typename T, int K, bool WEIGHTED, bool EPI, bool PAD>
__global__ void __launch_bounds__(kBlock)
regrid_fields_ell_kernel(const T* __restrict__ src, T* __restrict__ out,
                         const int32_t* __restrict__ idx, const T* __restrict__ w,
                         int64_t n_tgt, int k_rt, int n_lev, int64_t src_pitch, int64_t out_pitch,
                         int lev_chunk, unsigned n_tiles,
                         const atx_level_op* __restrict__ prog, int n_stage,
                         const uint8_t* __restrict__ tgt_mask) {
    const int k = K > 0 ? K : k_rt;
    const unsigned tile_id = xcd_tile(blockIdx.x, n_tiles);
    const int64_t t = (int64_t)tile_id * kBlock + threadIdx.x;
    if (t >= n_tgt) return;
    const int l0 = blockIdx.y * lev_chunk;
    const int l1 = min(n_lev, l0 + lev_chunk);
    const bool masked = (EPI && tgt_mask) ? (tgt_mask[t] != 0) : false;

    if (K > 0) {
        int64_t p[K > 0 ? K : 1];
        T wj[K > 0 ? K : 1];
        bool present[K > 0 ? K : 1];
#pragma unroll
        for (int j = 0; j < (K > 0 ? K : 1); ++j) {
            p[j] = idx[t * K + j];
            wj[j] = WEIGHTED ? w[t * K + j] : T(1);
            present[j] = !PAD || p[j] >= 0;  // absent entry of a padded row
        }
        if (PAD) {  // an absent entry reads what the row's first entry reads (a line this lane fetches anyway), not element 0 of every field —
            // that one line, shared by every padded lane of the launch, cost 48 % (ragged 3-4 rows padded to 4: 1.12 ms against 0.76 ms)
            const int64_t spare = present[0] ? p[0] : 0;
#pragma unroll
            for (int j = 0; j < (K > 0 ? K : 1); ++j)
                if (!present[j]) p[j] = spare;
        }
#ifndef ATX_FIELDS_UNROLL
#define ATX_FIELDS_UNROLL 4
#endif
#pragma unroll ATX_FIELDS_UNROLL
        for (int l = l0; l < l1; ++l) {
            const T* s = src + (int64_t)l * src_pitch;
            T acc;
            if (WEIGHTED) {
                // all k loads first, unconditionally; the sum then skips absent entries by a select.  (Written as `if (present[j])
                // acc += w * s[p]` the padded instantiation put each load behind a divergent branch: the SAME k = 4 table ran in
                // 1.03 ms through it against 0.70 ms through the plain one.)
                T sv[K > 0 ? K : 1];
#pragma unroll
                for (int j = 0; j < (K > 0 ? K : 1); ++j) sv[j] = s[p[j]];
                acc = T(0);
#pragma unroll
                for (int j = 0; j < (K > 0 ? K : 1); ++j) {
                    const T sum = acc + wj[j] * sv[j];
                    acc = present[j] ? sum : acc;
                }
            } else {
                acc = s[p[0]];
            }
            if (EPI) {
                for (int st = 0; st < n_stage; ++st)
                    acc = apply_level_op(load_level_op<T>(prog, (int64_t)st * n_lev + l), acc, masked);
            }
            store_scalar(out + (int64_t)l * out_pitch + t, acc);
        }
    } else {  // run-time k: the row walked once per 16 fields, one accumulator per field (cf. regrid_fields_csr_kernel)
        constexpr int LC = 16;
        for (int lc = l0; lc < l1; lc += LC) {
            const int nl = min(LC, l1 - lc);  // (uniform)
            const T* s0 = src + (int64_t)lc * src_pitch;
            T acc[LC];
#pragma unroll
            for (int i = 0; i < LC; ++i) acc[i] = T(0);
            for (int j = 0; j < k; ++j) {
                const T wv = WEIGHTED ? w[t * k + j] : T(1);
                const int64_t pj = idx[t * k + j];
                if (PAD && pj < 0) continue;
#pragma unroll
                for (int i = 0; i < LC; ++i)
                    if (i < nl) acc[i] = acc[i] + wv * s0[(int64_t)i * src_pitch + pj];
            }
#pragma unroll
            for (int i = 0; i < LC; ++i) {
                if (i < nl) {
                    T v = acc[i];
                    if (EPI) {
                        for (int st = 0; st < n_stage; ++st)
                            v = apply_level_op(load_level_op<T>(prog, (int64_t)st * n_lev + lc + i), v, masked);
                    }
                    out[(int64_t)(lc + i) * out_pitch + t] = v;
                }
            }
        }
    }
}

// ATX_FIELDS, general CSR: lane = row, grid.y = chunk of kFieldsChunk fields.  The row is walked ONCE per chunk — entry by entry, the
// entry's index and weight in registers while its kFieldsChunk gathers (one per field, all independent) are in flight — with one
// accumulator per field of the chunk; every field still sums its row in storage order starting from 0 (scipy's order).  Until round 3
// the loops were nested the other way, each field re-reading every index and weight and chaining its gathers: O1280 -> 0.25 deg,
// 137 fields, ragged rows of 3-4 entries 1.97 ms (the fixed-k kernel: 0.76 ms), rows of 9-16 entries 17 ms (float32).
// Short rows (mean <= R entries): the first R entries of the row in registers for all fields of the chunk (absent ones point at the
// row's first entry and are skipped by a select), entries beyond R re-read per field.  Ragged rows of 3-4 entries: 1.97 -> 1.11 ms.
template <typename T, bool EPI, int R>
__global__ void __launch_bounds__(kBlock)
regrid_fields_csr_head_kernel(const T* __restrict__ src, T* __restrict__ out,
                              const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                              const T* __restrict__ data, int64_t n_tgt, int n_lev,
                              int64_t src_pitch, int64_t out_pitch, int lev_chunk, unsigned n_tiles,
                              const atx_level_op* __restrict__ prog, int n_stage,
                              const uint8_t* __restrict__ tgt_mask) {
    const unsigned tile_id = xcd_tile(blockIdx.x, n_tiles);
    const int64_t t = (int64_t)tile_id * kBlock + threadIdx.x;
    if (t >= n_tgt) return;
    const int l0 = blockIdx.y * lev_chunk;
    const int l1 = min(n_lev, l0 + lev_chunk);
    const int64_t j0 = indptr[t], j1 = indptr[t + 1];
    const bool masked = (EPI && tgt_mask) ? (tgt_mask[t] != 0) : false;
    int64_t p[R];
    T wj[R];
    bool present[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        present[j] = j0 + j < j1;
        p[j] = present[j] ? (int64_t)indices[j0 + j] : (j > 0 ? p[0] : 0);
        wj[j] = present[j] ? data[j0 + j] : T(0);
    }
    const int64_t rest = j0 + R;
#pragma unroll 4
    for (int l = l0; l < l1; ++l) {
        const T* s = src + (int64_t)l * src_pitch;
        T sv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) sv[j] = s[p[j]];
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const T sum = acc + wj[j] * sv[j];
            acc = present[j] ? sum : acc;
        }
        for (int64_t jj = rest; jj < j1; ++jj) acc = acc + data[jj] * s[indices[jj]];
        if (EPI) {
            for (int st = 0; st < n_stage; ++st)
                acc = apply_level_op(load_level_op<T>(prog, (int64_t)st * n_lev + l), acc, masked);
        }
        out[(int64_t)l * out_pitch + t] = acc;
    }
}

constexpr int kFieldsChunk = 16;
template <typename T, bool EPI>
__global__ void __launch_bounds__(kBlock)
regrid_fields_csr_kernel(const T* __restrict__ src, T* __restrict__ out,
                         const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                         const T* __restrict__ data, int64_t n_tgt, int n_lev,
                         int64_t src_pitch, int64_t out_pitch, unsigned n_tiles,
                         const atx_level_op* __restrict__ prog, int n_stage,
                         const uint8_t* __restrict__ tgt_mask) {
    const unsigned tile_id = xcd_tile(blockIdx.x, n_tiles);
    const int64_t t = (int64_t)tile_id * kBlock + threadIdx.x;
    if (t >= n_tgt) return;
    const int l0 = blockIdx.y * kFieldsChunk;
    const int nl = min(kFieldsChunk, n_lev - l0);  // (uniform)
    const int64_t j0 = indptr[t], j1 = indptr[t + 1];
    const T* s0 = src + (int64_t)l0 * src_pitch;
    T acc[kFieldsChunk];
#pragma unroll
    for (int i = 0; i < kFieldsChunk; ++i) acc[i] = T(0);
    for (int64_t jj = j0; jj < j1; ++jj) {
        const int64_t p = indices[jj];
        const T wv = data[jj];
#pragma unroll
        for (int i = 0; i < kFieldsChunk; ++i)
            if (i < nl) acc[i] = acc[i] + wv * s0[(int64_t)i * src_pitch + p];
    }
    const bool masked = (EPI && tgt_mask) ? (tgt_mask[t] != 0) : false;
#pragma unroll
    for (int i = 0; i < kFieldsChunk; ++i) {
        if (i < nl) {
            T v = acc[i];
            if (EPI) {
                for (int st = 0; st < n_stage; ++st)
                    v = apply_level_op(load_level_op<T>(prog, (int64_t)st * n_lev + l0 + i), v, masked);
            }
            out[(int64_t)(l0 + i) * out_pitch + t] = v;
        }
    }
}

__global__ void __launch_bounds__(kBlock)
check_indices_kernel(const int32_t* __restrict__ idx, int64_t n, int64_t n_src, unsigned long long* n_bad) {
    unsigned long long bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t v = idx[i];
        bad += (v < 0 || v >= n_src) ? 1u : 0u;
    }
    // wavefront (64-lane) shuffle reduction, then one atomic per wave
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) bad += __shfl_down(bad, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && bad) atomicAdd(n_bad, bad);
}

// ---------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------
static int pick_tile(int64_t n_tgt, int C, bool epilogue) {
    // Small tiles win: ~560 (target, vector) items per 256-lane workgroup, i.e. 16 targets of
    // 137 f32 levels, rounded up to a multiple of 4 targets (measured on O1280 -> 0.25 deg:
    // tiles of 12 / 16 beat 10, 14, 18-32; 64/128/512-lane workgroups reach the same plateau at
    // the same items-per-lane ratio — profiles/r01_ab_variants.log, r01_ab_block_sizes.log).  More, shorter
    // workgroups keep more of them in different phases (index staging / gather / store).
    // With an epilogue every workgroup first builds its operator table (a second global-load latency before the
    // barrier): tiles 2.5x larger amortise it — 137 levels f32: 0.491 ms at 16 targets, 0.469 at 32-48; f64: 0.985 ms at 8,
    // 0.878 at 24 (profiles/r01_ab_epilogue.log).
    const int items = epilogue ? 1400 : 560;
    int tile = (items / C + 2) / 4 * 4;  // nearest multiple of 4 targets: 16 (40 with epilogue) for 137 f32 levels, 8 (20) for f64
    if (tile < 8) tile = 8;
    if (tile > 256) tile = 256;
    if ((int64_t)tile > n_tgt) tile = (int)n_tgt;
    return tile;
}

static thread_local int g_tile_override = 0;  // tuning hook (atx_set_tuning): per calling thread, meant for benchmarks and tests (results never depend on it)

// The fused per-level program as the launchers see it: `prog` (device, per level) is always there when n_stage > 0; the two
// optional companions let the direct kernel take the epilogue — `vec_prog` (device: atx_vector_program of the stack's dtype)
// and `host_prog` (HOST copy of `prog`: the only way the library can SEE the program without a device round trip).
struct Epilogue {
    const atx_level_op* prog = nullptr;
    const atx_level_op* vec_prog = nullptr;
    const atx_level_op* host_prog = nullptr;
    int n_stage = 0;
    const uint8_t* mask = nullptr;
    const int32_t* tgt_rows = nullptr;  // ordered traversal (atx_regrid_ell_ordered): table row t is output row tgt_rows[t]
};

// Every operator of the program is COPY, AFFINE or MUL (masked or not) and there are <= kMaxTable stages: the direct
// kernel's table route applies.
static bool madd_family_program(const Epilogue& e, int n_lev) {
    if (!e.host_prog || !e.vec_prog || !aligned16(e.vec_prog) || e.n_stage < 1 || e.n_stage > kMaxTable) return false;
    for (int64_t i = 0; i < (int64_t)e.n_stage * n_lev; ++i) {
        const int op = e.host_prog[i].op;
        if (op != ATX_OP_COPY && op != ATX_OP_AFFINE && op != ATX_OP_MUL) return false;
    }
    return true;
}

// Per stage the levels run one operator, or one up to a level that is a multiple of `vec` and another from there on
// (<= kMaxUniform stages): the operators can travel by value (atx_common.hpp: uniform_level_program).
template <typename T>
static bool uniform_program(const Epilogue& e, int n_lev, int vec, UniformOps<T>& out) {
    return uniform_level_program<T>(e.host_prog, e.n_stage, e.mask != nullptr, n_lev, vec, out);
}

template <typename T, int VEC, int K, bool WEIGHTED, bool PAD = false>
static int launch_cols_ell(const EllBatch& batch, const int32_t* idx, const T* w, int64_t n_tgt, int k,
                           int n_lev, int64_t src_pitch, int64_t out_pitch, const Epilogue& epi, hipStream_t stream) {
    const int C = (n_lev + VEC - 1) / VEC;
    const atx_level_op* prog = epi.prog;
    const int n_stage = epi.n_stage;
    const uint8_t* tgt_mask = epi.mask;
    // Fixed-k gathers with compile-time k (1-4; padded ragged rows 3-4) take the direct kernel.  Interleaved A/B on O1280 -> 0.25 deg,
    // 137 levels (profiles/r01_ab_direct_kernel.log): k=4 f32 0.4360 vs 0.4378 ms, k=1 f32 0.1926 vs 0.1992 ms, k=4 f64 0.8334 vs
    // 0.8343 ms; 1-60 levels equal or up to 15 % faster; 2 / 4 items per lane -4 % / -9 %.  Same bits, no tile heuristic to tune.
    // With an epilogue it still does when the operators can reach it without a per-workgroup set-up: by value (uniform
    // program seen through host_prog) or, for multiply-add programs, through the host-built per-vector table (vec_prog);
    // profiles/r02_ab_epilogue_routes.log.  Round 3 tried TWO vectors per lane (columns c and c + ceil(C/2) of one target: half the
    // lanes read the index / weight words, 2 k source loads in flight per lane) — slower everywhere: k=4 f64 0.841 -> 0.912 ms,
    // k=1 f64 0.368 -> 0.420, k=4 f32 0.439 -> 0.523, k=1 f32 0.197 -> 0.236 (profiles/r03_direct_v2_experiment.log); one item per
    // lane in many short waves it stays.
#ifndef ATX_ELL_DIRECT
#define ATX_ELL_DIRECT 1
#endif
#ifndef ATX_EPI_DIRECT
#define ATX_EPI_DIRECT 1
#endif
    if constexpr (ATX_ELL_DIRECT && K > 0) {
        if (g_tile_override <= 0) {  // atx_set_tuning(tile > 0) selects the tiled kernel below (A/B, tests)
            const int64_t n_items = n_tgt * C;
            const unsigned n_blocks = (unsigned)((n_items + kEllBlock - 1) / kEllBlock);
            UniformOps<T> uniform{};
            if (!prog) {
                hipLaunchKernelGGL((regrid_cols_ell_direct_kernel<T, VEC, K, WEIGHTED, PAD, kEpiNone>), dim3(n_blocks, batch.n), dim3(kEllBlock), 0,
                                   stream, batch, idx, w, n_items, C, src_pitch, out_pitch, n_blocks, 1, uniform, nullptr, 0, n_lev, nullptr, epi.tgt_rows, RunOps<T>{});
                ATX_LAUNCH_CHECK("regrid_cols_ell_direct");
                return ATX_OK;
            }
            if (ATX_EPI_DIRECT && uniform_program<T>(epi, n_lev, VEC, uniform)) {
                hipLaunchKernelGGL((regrid_cols_ell_direct_kernel<T, VEC, K, WEIGHTED, PAD, kEpiUniform>), dim3(n_blocks, batch.n), dim3(kEllBlock),
                                   0, stream, batch, idx, w, n_items, C, src_pitch, out_pitch, n_blocks, 1, uniform, nullptr, n_stage, n_lev,
                                   tgt_mask, epi.tgt_rows, RunOps<T>{});
                ATX_LAUNCH_CHECK("regrid_cols_ell_direct_uniform");
                return ATX_OK;
            }
#ifndef ATX_EPI_RUNS
#define ATX_EPI_RUNS 1
#endif
            RunOps<T> runs{};
            if (ATX_EPI_DIRECT && ATX_EPI_RUNS && runs_level_program<T>(epi.host_prog, n_stage, tgt_mask != nullptr, n_lev, runs)) {
                // several variables in one column (runs of levels with boundaries anywhere): by value, no table read beside the gather
                hipLaunchKernelGGL((regrid_cols_ell_direct_kernel<T, VEC, K, WEIGHTED, PAD, kEpiRuns>), dim3(n_blocks, batch.n), dim3(kEllBlock),
                                   0, stream, batch, idx, w, n_items, C, src_pitch, out_pitch, n_blocks, 1, uniform, nullptr, n_stage, n_lev,
                                   tgt_mask, epi.tgt_rows, runs);
                ATX_LAUNCH_CHECK("regrid_cols_ell_direct_runs");
                return ATX_OK;
            }
            if (ATX_EPI_DIRECT && VEC == Vec16<T>::N && madd_family_program(epi, n_lev)) {  // the table is built for 16-byte vectors
                const unsigned char* level_tables = reinterpret_cast<const unsigned char*>(epi.vec_prog) +
                                                    level_tables_layout(n_stage, n_lev, sizeof(T) == 4 ? ATX_F32 : ATX_F64).levels_offset;
                hipLaunchKernelGGL((regrid_cols_ell_direct_kernel<T, VEC, K, WEIGHTED, PAD, kEpiTable>), dim3(n_blocks, batch.n), dim3(kEllBlock), 0,
                                   stream, batch, idx, w, n_items, C, src_pitch, out_pitch, n_blocks, 1, uniform, level_tables, n_stage, n_lev,
                                   tgt_mask, epi.tgt_rows, RunOps<T>{});
                ATX_LAUNCH_CHECK("regrid_cols_ell_direct_table");
                return ATX_OK;
            }
        }
    }
    int tile = g_tile_override > 0 ? g_tile_override : pick_tile(n_tgt, C, prog != nullptr);
    if ((int64_t)tile > n_tgt) tile = (int)n_tgt;
    const unsigned n_tiles = (unsigned)((n_tgt + tile - 1) / tile);
    size_t lds = (size_t)tile * k * (sizeof(int32_t) + (WEIGHTED ? sizeof(T) : 0));
    lds = (lds + 15) & ~size_t(15);
    if (prog) lds += level_tables_lds_bytes<T>(n_stage, C, VEC);
    if (prog && lds > 64 * 1024) return ATX_SPLIT_PROGRAM;  // the caller gathers without the program and applies it afterwards
    ATX_REQUIRE(lds <= 64 * 1024, ATX_ENOTIMPL, "regrid_ell: tile needs %zu B of LDS (k=%d, n_lev=%d, stages=%d)", lds, k, n_lev, n_stage);
    constexpr int KT = K > 4 ? 0 : K;  // the tiled kernel keeps k > 4 on its runtime-k loop (compile-time k would hold 4 x k vectors per lane)
    if (prog) {
        hipLaunchKernelGGL((regrid_cols_ell_kernel<T, VEC, KT, WEIGHTED, true, PAD>), dim3(n_tiles, ATX_BATCH_LOOP ? 1 : batch.n), dim3(kEllBlock), lds, stream,
                           batch, idx, w, n_tgt, k, n_lev, C, src_pitch, out_pitch, tile, n_tiles, prog, n_stage, tgt_mask, epi.tgt_rows);
    } else {
        hipLaunchKernelGGL((regrid_cols_ell_kernel<T, VEC, KT, WEIGHTED, false, PAD>), dim3(n_tiles, ATX_BATCH_LOOP ? 1 : batch.n), dim3(kEllBlock), lds, stream,
                           batch, idx, w, n_tgt, k, n_lev, C, src_pitch, out_pitch, tile, n_tiles, prog, n_stage, tgt_mask, epi.tgt_rows);
    }
    ATX_LAUNCH_CHECK("regrid_cols_ell");
    return ATX_OK;
}

template <typename T, int VEC>
static int dispatch_cols_ell(const EllBatch& batch, const int32_t* idx, const T* w, int64_t n_tgt, int k,
                             int n_lev, int64_t sp, int64_t op, bool pad, const Epilogue& e, hipStream_t st) {
    if (!w) return launch_cols_ell<T, VEC, 1, false>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
    // compile-time k up to 8 (the direct kernel: all k loads of an item in flight at once); k = 5 .. 8 moved off the runtime-k tiled
    // kernel in round 3: k = 8 0.608 -> 0.635 f32, 0.625 -> 0.657 f64, k = 6 0.636 -> 0.650 / 0.659 -> 0.683 (profiles/r03_mid_k_experiment.log)
    if (pad) {  // padded ragged rows
        switch (k) {
            case 3: return launch_cols_ell<T, VEC, 3, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            case 4: return launch_cols_ell<T, VEC, 4, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            case 5: return launch_cols_ell<T, VEC, 5, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            case 6: return launch_cols_ell<T, VEC, 6, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            case 7: return launch_cols_ell<T, VEC, 7, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            case 8: return launch_cols_ell<T, VEC, 8, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            case 12: return launch_cols_ell<T, VEC, 12, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            case 16: return launch_cols_ell<T, VEC, 16, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
            default: return launch_cols_ell<T, VEC, 0, true, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        }
    }
    switch (k) {
        case 1: return launch_cols_ell<T, VEC, 1, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 2: return launch_cols_ell<T, VEC, 2, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 3: return launch_cols_ell<T, VEC, 3, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 4: return launch_cols_ell<T, VEC, 4, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 5: return launch_cols_ell<T, VEC, 5, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 6: return launch_cols_ell<T, VEC, 6, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 7: return launch_cols_ell<T, VEC, 7, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 8: return launch_cols_ell<T, VEC, 8, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        // the two round numbers beyond 8 a k-NN regrid is configured with: all 12 / 16 source vectors in flight beat the tiled kernel's
        // runtime-k loop (k = 16: 0.46 -> 0.51 f32 in natural order, 0.58 with the targets in column blocks, f64 0.48 -> 0.52 / 0.54);
        // a runtime-k loop IN the direct kernel, 8 or 16 entries per step, measured no better than the tiled kernel and was dropped
        // (profiles/r03_long_k_direct_experiment.log)
        case 12: return launch_cols_ell<T, VEC, 12, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        case 16: return launch_cols_ell<T, VEC, 16, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
        default: return launch_cols_ell<T, VEC, 0, true>(batch, idx, w, n_tgt, k, n_lev, sp, op, e, st);
    }
}

static int pick_lev_chunk(int n_lev) {
    // enough level chunks for >= ~4 workgroups per CU even on small grids, chunks >= 8 levels
    int chunks = (n_lev + 31) / 32;
    return (n_lev + chunks - 1) / chunks;
}

template <typename T, int K, bool WEIGHTED, bool PAD = false>
static int launch_fields_ell(const T* src, T* out, const int32_t* idx, const T* w, int64_t n_tgt, int k,
                             int n_lev, int64_t sp, int64_t op, const atx_level_op* prog, int n_stage,
                             const uint8_t* m, hipStream_t st) {
    const unsigned n_tiles = (unsigned)((n_tgt + kBlock - 1) / kBlock);
    const int lev_chunk = pick_lev_chunk(n_lev);
    const unsigned n_chunks = (unsigned)((n_lev + lev_chunk - 1) / lev_chunk);
    ATX_REQUIRE(n_chunks <= 65535, ATX_ENOTIMPL, "regrid_ell: too many level chunks (%u)", n_chunks);
    if (prog) {
        hipLaunchKernelGGL((regrid_fields_ell_kernel<T, K, WEIGHTED, true, PAD>), dim3(n_tiles, n_chunks), dim3(kBlock), 0, st,
                           src, out, idx, w, n_tgt, k, n_lev, sp, op, lev_chunk, n_tiles, prog, n_stage, m);
    } else {
        hipLaunchKernelGGL((regrid_fields_ell_kernel<T, K, WEIGHTED, false, PAD>), dim3(n_tiles, n_chunks), dim3(kBlock), 0, st,
                           src, out, idx, w, n_tgt, k, n_lev, sp, op, lev_chunk, n_tiles, prog, n_stage, m);
    }
    ATX_LAUNCH_CHECK("regrid_fields_ell");
    return ATX_OK;
}

template <typename T>
static int dispatch_fields_ell(const T* src, T* out, const int32_t* idx, const T* w, int64_t n_tgt, int k,
                               int n_lev, int64_t sp, int64_t op, bool pad, const atx_level_op* prog, int n_stage,
                               const uint8_t* m, hipStream_t st) {
    if (!w) return launch_fields_ell<T, 1, false>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
    if (pad) {
        switch (k) {
            case 3: return launch_fields_ell<T, 3, true, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
            case 4: return launch_fields_ell<T, 4, true, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
            default: return launch_fields_ell<T, 0, true, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
        }
    }
    switch (k) {
        case 1: return launch_fields_ell<T, 1, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
        case 2: return launch_fields_ell<T, 2, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
        case 3: return launch_fields_ell<T, 3, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
        case 4: return launch_fields_ell<T, 4, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
        default: return launch_fields_ell<T, 0, true>(src, out, idx, w, n_tgt, k, n_lev, sp, op, prog, n_stage, m, st);
    }
}

template <typename T>
static bool cols_vector_ok(const void* src, const void* out, int64_t sp, int64_t op) {
    constexpr int VEC = Vec16<T>::N;
    return aligned16(src) && aligned16(out) && (sp % VEC == 0) && (op % VEC == 0);
}

template <typename T>
static int regrid_ell_typed(const EllBatch& batch, const int32_t* idx, const void* w_, int64_t n_tgt, int k,
                            int n_lev, int64_t sp, int64_t op, int layout, bool pad, const Epilogue& e, hipStream_t st) {
    const T* w = static_cast<const T*>(w_);
    const atx_level_op* prog = e.prog;
    const int n_stage = e.n_stage;
    const uint8_t* m = e.mask;
    if (layout == ATX_COLUMNS) {
        constexpr int VEC = Vec16<T>::N;
        // the vector path needs every row start 16-byte aligned and room for the
        // last (partial) vector inside the pitch
        const int64_t covered = ((int64_t)(n_lev + VEC - 1) / VEC) * VEC;
        bool vector_ok = covered <= sp && covered <= op;
        for (int i = 0; i < batch.n; ++i) vector_ok = vector_ok && cols_vector_ok<T>(batch.src[i], batch.out[i], sp, op);
        if (vector_ok) return dispatch_cols_ell<T, VEC>(batch, idx, w, n_tgt, k, n_lev, sp, op, pad, e, st);
        return dispatch_cols_ell<T, 1>(batch, idx, w, n_tgt, k, n_lev, sp, op, pad, e, st);
    }
    for (int i = 0; i < batch.n; ++i) {  // field-major stacks: lanes keep indices / weights in registers, one launch per stack
        const int rc = dispatch_fields_ell<T>(static_cast<const T*>(batch.src[i]), static_cast<T*>(batch.out[i]), idx, w, n_tgt, k,
                                              n_lev, sp, op, pad, prog, n_stage, m, st);
        if (rc != ATX_OK) return rc;
    }
    return ATX_OK;
}

template <typename T, int VEC>
static int launch_cols_csr(const T* src, T* out, const int32_t* indptr, const int32_t* indices, const T* data,
                           int64_t n_tgt, int64_t nnz, int n_lev, int64_t sp, int64_t op,
                           const atx_level_op* prog, int n_stage, const uint8_t* m, const int32_t* rows, hipStream_t st) {
    const int C = (n_lev + VEC - 1) / VEC;
    // (A "direct" form of this kernel — one item per lane, row walked from the CSR arrays in L1 — was measured and dropped: rows of
    // 3-4 entries 0.478 ms against 0.450 ms tiled, rows of 9-16 entries 1.06 against 1.01 ms f32, 2.03 against 2.10 ms f64; with 8
    // entries in flight and the next step's words prefetched 1.10 ms.  The extra dependent load level — row bounds, entries, source
    // columns — costs what the missing barrier saves; profiles/r02_csr_direct_experiment.log.  Nor do long rows want more loads in
    // flight: box-average coarsening O1280 -> 1 degree, ~100 entries per row, every source column read exactly once, runs at 0.86 ms
    // = 4.3 TB/s with 4 entries per step and 0.85 ms with 8 (at 84 VGPRs, 5 waves per SIMD), whatever the tile size —
    // profiles/r02_long_rows_experiment.log.)
    int tile = g_tile_override > 0 ? g_tile_override : pick_tile(n_tgt, C, prog != nullptr);
    if ((int64_t)tile > n_tgt) tile = (int)n_tgt;
    const unsigned n_tiles = (unsigned)((n_tgt + tile - 1) / tile);
    // LDS room for ~2x the mean entries of a tile (tiles above it read CSR from L2)
    const double mean = n_tgt > 0 ? (double)nnz / (double)n_tgt : 0.0;
    int cap = (int)(mean * tile * 2.0) + 64;
    if (cap > 4096) cap = 4096;
    size_t lds = (size_t)cap * (sizeof(T) + sizeof(int32_t)) + (size_t)(tile + 1) * sizeof(int32_t);
    lds = (lds + 15) & ~size_t(15);
    if (prog) lds += level_tables_lds_bytes<T>(n_stage, C, VEC);
    if (prog && lds > 64 * 1024) return ATX_SPLIT_PROGRAM;  // the caller gathers without the program and applies it afterwards
    ATX_REQUIRE(lds <= 64 * 1024, ATX_ENOTIMPL, "regrid_csr: tile needs %zu B of LDS", lds);
    // long rows in natural order: stripes of tiles per XCD (atx_common.hpp: xcd_stripe) — their cost may drift along the rows
#ifndef ATX_CSR_STRIPE
#define ATX_CSR_STRIPE 16
#endif
#ifndef ATX_CSR_STRIPE_MIN_MEAN
#define ATX_CSR_STRIPE_MIN_MEAN 8.0
#endif
    const int stripe = (!rows && mean >= ATX_CSR_STRIPE_MIN_MEAN) ? ATX_CSR_STRIPE : 0;
    if (prog) {
        hipLaunchKernelGGL((regrid_cols_csr_kernel<T, VEC, true>), dim3(n_tiles), dim3(kBlock), lds, st, src, out, indptr,
                           indices, data, n_tgt, n_lev, C, sp, op, tile, n_tiles, cap, prog, n_stage, m, rows, stripe);
    } else {
        hipLaunchKernelGGL((regrid_cols_csr_kernel<T, VEC, false>), dim3(n_tiles), dim3(kBlock), lds, st, src, out, indptr,
                           indices, data, n_tgt, n_lev, C, sp, op, tile, n_tiles, cap, prog, n_stage, m, rows, stripe);
    }
    ATX_LAUNCH_CHECK("regrid_cols_csr");
    return ATX_OK;
}

template <typename T>
static int regrid_csr_typed(const void* src_, void* out_, const int32_t* indptr, const int32_t* indices,
                            const void* data_, int64_t n_tgt, int64_t nnz, int n_lev, int64_t sp, int64_t op,
                            int layout, const atx_level_op* prog, int n_stage, const uint8_t* m, const int32_t* rows, hipStream_t st) {
    const T* src = static_cast<const T*>(src_);
    T* out = static_cast<T*>(out_);
    const T* data = static_cast<const T*>(data_);
    if (layout == ATX_COLUMNS) {
        constexpr int VEC = Vec16<T>::N;
        const int64_t covered = ((int64_t)(n_lev + VEC - 1) / VEC) * VEC;
        if (cols_vector_ok<T>(src_, out_, sp, op) && covered <= sp && covered <= op)
            return launch_cols_csr<T, VEC>(src, out, indptr, indices, data, n_tgt, nnz, n_lev, sp, op, prog, n_stage, m, rows, st);
        return launch_cols_csr<T, 1>(src, out, indptr, indices, data, n_tgt, nnz, n_lev, sp, op, prog, n_stage, m, rows, st);
    }
    ATX_REQUIRE(!rows, ATX_ENOTIMPL, "regrid_csr: an ordered traversal (tgt_rows) is available for ATX_COLUMNS stacks only");
    const unsigned n_tiles = (unsigned)((n_tgt + kBlock - 1) / kBlock);
    const double mean = n_tgt > 0 ? (double)nnz / (double)n_tgt : 0.0;
    if (mean <= 8.0) {  // short rows: their entries in registers
        const int lev_chunk = pick_lev_chunk(n_lev);
        const unsigned chunks = (unsigned)((n_lev + lev_chunk - 1) / lev_chunk);
        ATX_REQUIRE(chunks <= 65535, ATX_ENOTIMPL, "regrid_csr: too many level chunks (%u)", chunks);
#define ATX_CSR_HEAD_LAUNCH(EPI_, R_)                                                                                                   \
    hipLaunchKernelGGL((regrid_fields_csr_head_kernel<T, EPI_, R_>), dim3(n_tiles, chunks), dim3(kBlock), 0, st, src, out, indptr, indices, \
                       data, n_tgt, n_lev, sp, op, lev_chunk, n_tiles, prog, n_stage, m)
        if (prog) {
            if (mean <= 4.0) ATX_CSR_HEAD_LAUNCH(true, 4);
            else ATX_CSR_HEAD_LAUNCH(true, 8);
        } else {
            if (mean <= 4.0) ATX_CSR_HEAD_LAUNCH(false, 4);
            else ATX_CSR_HEAD_LAUNCH(false, 8);
        }
#undef ATX_CSR_HEAD_LAUNCH
        ATX_LAUNCH_CHECK("regrid_fields_csr_head");
        return ATX_OK;
    }
    const unsigned n_chunks = (unsigned)((n_lev + kFieldsChunk - 1) / kFieldsChunk);
    ATX_REQUIRE(n_chunks <= 65535, ATX_ENOTIMPL, "regrid_csr: too many level chunks (%u)", n_chunks);
    if (prog) {
        hipLaunchKernelGGL((regrid_fields_csr_kernel<T, true>), dim3(n_tiles, n_chunks), dim3(kBlock), 0, st, src, out, indptr, indices, data,
                           n_tgt, n_lev, sp, op, n_tiles, prog, n_stage, m);
    } else {
        hipLaunchKernelGGL((regrid_fields_csr_kernel<T, false>), dim3(n_tiles, n_chunks), dim3(kBlock), 0, st, src, out, indptr, indices, data,
                           n_tgt, n_lev, sp, op, n_tiles, prog, n_stage, m);
    }
    ATX_LAUNCH_CHECK("regrid_fields_csr");
    return ATX_OK;
}

static int check_stack_args(const char* fn, const void* src, const void* out, int64_t n_src, int64_t n_tgt,
                            int64_t n_lev, int64_t sp, int64_t op, int dtype, int layout) {
    ATX_REQUIRE(src && out, ATX_EINVAL, "%s: null src/out pointer", fn);
    ATX_REQUIRE(dtype == ATX_F32 || dtype == ATX_F64, ATX_EINVAL, "%s: bad dtype %d", fn, dtype);
    ATX_REQUIRE(layout == ATX_COLUMNS || layout == ATX_FIELDS, ATX_EINVAL, "%s: bad layout %d", fn, layout);
    ATX_REQUIRE(n_src > 0 && n_tgt >= 0 && n_lev > 0, ATX_EINVAL, "%s: bad sizes n_src=%lld n_tgt=%lld n_lev=%lld", fn,
                (long long)n_src, (long long)n_tgt, (long long)n_lev);
    ATX_REQUIRE(n_src <= INT32_MAX && n_tgt <= INT32_MAX && n_lev <= 65535, ATX_ENOTIMPL,
                "%s: sizes exceed int32 indexing (n_src=%lld n_tgt=%lld n_lev=%lld)", fn, (long long)n_src,
                (long long)n_tgt, (long long)n_lev);
    if (layout == ATX_COLUMNS) {
        ATX_REQUIRE(sp >= n_lev && op >= n_lev, ATX_ESHAPE, "%s: column pitch (%lld, %lld) < n_lev %lld", fn,
                    (long long)sp, (long long)op, (long long)n_lev);
    } else {
        ATX_REQUIRE(sp >= n_src && op >= n_tgt, ATX_ESHAPE, "%s: field pitch (%lld, %lld) < points (%lld, %lld)", fn,
                    (long long)sp, (long long)op, (long long)n_src, (long long)n_tgt);
    }
    return ATX_OK;
}

}  // namespace atx

using namespace atx;

extern "C" int atx_set_tuning(int tile) {
    g_tile_override = tile;
    return ATX_OK;
}

static int regrid_ell_common(const char* fn, const void* const* srcs, void* const* outs, int32_t n_stack, const int32_t* idx,
                             const void* w, int64_t n_src, int64_t n_tgt, int32_t k, int64_t n_lev, int64_t src_pitch,
                             int64_t out_pitch, int dtype, int layout, int32_t flags, const atx_level_op* prog,
                             const atx_level_op* vec_prog, const atx_level_op* host_prog, int32_t n_stage, const uint8_t* tgt_mask,
                             const int32_t* tgt_rows, void* stream) {
    ATX_REQUIRE(srcs && outs && n_stack >= 1, ATX_EINVAL, "%s: needs at least one stack", fn);
    ATX_REQUIRE(!tgt_rows || layout == ATX_COLUMNS, ATX_ENOTIMPL, "%s: an ordered traversal (tgt_rows) is available for ATX_COLUMNS stacks only", fn);
    for (int32_t i = 0; i < n_stack; ++i) {
        int st = check_stack_args(fn, srcs[i], outs[i], n_src, n_tgt, n_lev, src_pitch, out_pitch, dtype, layout);
        if (st != ATX_OK) return st;
    }
    ATX_REQUIRE(idx, ATX_EINVAL, "%s: null idx", fn);
    ATX_REQUIRE(k >= 1 && k <= 64, ATX_EINVAL, "%s: k=%d outside [1, 64]", fn, k);
    ATX_REQUIRE(w || k == 1, ATX_EINVAL, "%s: a pure gather (w == NULL) needs k == 1, got %d", fn, k);
    ATX_REQUIRE((flags & ~ATX_ELL_PADDED) == 0, ATX_EINVAL, "%s: unknown flags 0x%x", fn, flags);
    ATX_REQUIRE(!(flags & ATX_ELL_PADDED) || w, ATX_EINVAL, "%s: ATX_ELL_PADDED needs weights", fn);
    const bool pad = (flags & ATX_ELL_PADDED) != 0;
    ATX_REQUIRE((prog == nullptr) == (n_stage == 0) && n_stage >= 0 && n_stage <= 8, ATX_EINVAL,
                "%s: prog/n_stage mismatch (n_stage=%d)", fn, n_stage);
    ATX_REQUIRE(prog || (!vec_prog && !host_prog), ATX_EINVAL, "%s: vec_prog / host_prog accompany prog, which is NULL", fn);
    if (n_tgt == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Epilogue epi;
    epi.prog = prog;
    epi.vec_prog = vec_prog;
    epi.host_prog = host_prog;
    epi.n_stage = n_stage;
    epi.mask = tgt_mask;
    epi.tgt_rows = tgt_rows;
    for (int32_t first = 0; first < n_stack; first += kMaxBatch) {
        EllBatch batch;
        batch.n = n_stack - first < kMaxBatch ? n_stack - first : kMaxBatch;
        for (int i = 0; i < kMaxBatch; ++i) {
            batch.src[i] = i < batch.n ? srcs[first + i] : nullptr;
            batch.out[i] = i < batch.n ? outs[first + i] : nullptr;
        }
        int rc = dtype == ATX_F32
            ? regrid_ell_typed<float>(batch, idx, w, n_tgt, k, (int)n_lev, src_pitch, out_pitch, layout, pad, epi, s)
            : regrid_ell_typed<double>(batch, idx, w, n_tgt, k, (int)n_lev, src_pitch, out_pitch, layout, pad, epi, s);
        if (rc == ATX_SPLIT_PROGRAM) {  // the fused tables do not fit in LDS: the plain gather, then the program on its output in place
            Epilogue plain;
            plain.tgt_rows = tgt_rows;
            rc = dtype == ATX_F32
                ? regrid_ell_typed<float>(batch, idx, w, n_tgt, k, (int)n_lev, src_pitch, out_pitch, layout, pad, plain, s)
                : regrid_ell_typed<double>(batch, idx, w, n_tgt, k, (int)n_lev, src_pitch, out_pitch, layout, pad, plain, s);
            for (int i = 0; i < batch.n && rc == ATX_OK; ++i)
                rc = atx_pointwise_stack(batch.out[i], batch.out[i], n_tgt, n_lev, out_pitch, out_pitch, dtype, layout, prog, vec_prog, host_prog,
                                         n_stage, tgt_mask, stream);
        }
        if (rc != ATX_OK) return rc;
    }
    return ATX_OK;
}

extern "C" int atx_regrid_ell(const void* src, void* out, const int32_t* idx, const void* w, int64_t n_src,
                              int64_t n_tgt, int32_t k, int64_t n_lev, int64_t src_pitch, int64_t out_pitch,
                              int dtype, int layout, int32_t flags, const atx_level_op* prog, const atx_level_op* vec_prog,
                              const atx_level_op* host_prog, int32_t n_stage, const uint8_t* tgt_mask, void* stream) {
    return regrid_ell_common("atx_regrid_ell", &src, &out, 1, idx, w, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, dtype,
                             layout, flags, prog, vec_prog, host_prog, n_stage, tgt_mask, nullptr, stream);
}

extern "C" int atx_regrid_ell_batch(const void* const* srcs, void* const* outs, int32_t n_stack, const int32_t* idx,
                                    const void* w, int64_t n_src, int64_t n_tgt, int32_t k, int64_t n_lev,
                                    int64_t src_pitch, int64_t out_pitch, int dtype, int layout, int32_t flags,
                                    const atx_level_op* prog, const atx_level_op* vec_prog, const atx_level_op* host_prog,
                                    int32_t n_stage, const uint8_t* tgt_mask, void* stream) {
    return regrid_ell_common("atx_regrid_ell_batch", srcs, outs, n_stack, idx, w, n_src, n_tgt, k, n_lev, src_pitch, out_pitch,
                             dtype, layout, flags, prog, vec_prog, host_prog, n_stage, tgt_mask, nullptr, stream);
}

extern "C" int atx_regrid_ell_ordered(const void* const* srcs, void* const* outs, int32_t n_stack, const int32_t* idx,
                                      const void* w, const int32_t* tgt_rows, int64_t n_src, int64_t n_tgt, int32_t k, int64_t n_lev,
                                      int64_t src_pitch, int64_t out_pitch, int dtype, int layout, int32_t flags,
                                      const atx_level_op* prog, const atx_level_op* vec_prog, const atx_level_op* host_prog,
                                      int32_t n_stage, const uint8_t* tgt_mask, void* stream) {
    ATX_REQUIRE(tgt_rows, ATX_EINVAL, "atx_regrid_ell_ordered: null tgt_rows (use atx_regrid_ell_batch for the natural order)");
    return regrid_ell_common("atx_regrid_ell_ordered", srcs, outs, n_stack, idx, w, n_src, n_tgt, k, n_lev, src_pitch, out_pitch,
                             dtype, layout, flags, prog, vec_prog, host_prog, n_stage, tgt_mask, tgt_rows, stream);
}

static int regrid_csr_common(const char* fn, const void* src, void* out, const int32_t* indptr, const int32_t* indices,
                             const void* data, int64_t n_src, int64_t n_tgt, int64_t nnz, int64_t n_lev,
                             int64_t src_pitch, int64_t out_pitch, int dtype, int layout, const atx_level_op* prog,
                             int32_t n_stage, const uint8_t* tgt_mask, const int32_t* tgt_rows, void* stream) {
    int st = check_stack_args(fn, src, out, n_src, n_tgt, n_lev, src_pitch, out_pitch, dtype, layout);
    if (st != ATX_OK) return st;
    ATX_REQUIRE(indptr, ATX_EINVAL, "%s: null indptr", fn);
    ATX_REQUIRE(nnz >= 0 && nnz <= INT32_MAX, ATX_ENOTIMPL, "%s: nnz=%lld outside int32", fn, (long long)nnz);
    ATX_REQUIRE(nnz == 0 || (indices && data), ATX_EINVAL, "%s: null indices/data", fn);
    ATX_REQUIRE((prog == nullptr) == (n_stage == 0) && n_stage >= 0 && n_stage <= 8, ATX_EINVAL,
                "%s: prog/n_stage mismatch (n_stage=%d)", fn, n_stage);
    if (n_tgt == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto run = [&](const atx_level_op* p, int32_t stages, const uint8_t* m) {
        if (dtype == ATX_F32)
            return regrid_csr_typed<float>(src, out, indptr, indices, data, n_tgt, nnz, (int)n_lev, src_pitch, out_pitch, layout, p, stages, m, tgt_rows, s);
        return regrid_csr_typed<double>(src, out, indptr, indices, data, n_tgt, nnz, (int)n_lev, src_pitch, out_pitch, layout, p, stages, m, tgt_rows, s);
    };
    int rc = run(prog, n_stage, tgt_mask);
    if (rc == ATX_SPLIT_PROGRAM) {  // the fused tables do not fit in LDS: the plain product, then the program on its output in place
        rc = run(nullptr, 0, nullptr);
        if (rc == ATX_OK)
            rc = atx_pointwise_stack(out, out, n_tgt, n_lev, out_pitch, out_pitch, dtype, layout, prog, nullptr, nullptr, n_stage, tgt_mask, stream);
    }
    return rc;
}

extern "C" int atx_regrid_csr(const void* src, void* out, const int32_t* indptr, const int32_t* indices,
                              const void* data, int64_t n_src, int64_t n_tgt, int64_t nnz, int64_t n_lev,
                              int64_t src_pitch, int64_t out_pitch, int dtype, int layout, const atx_level_op* prog,
                              int32_t n_stage, const uint8_t* tgt_mask, void* stream) {
    return regrid_csr_common("atx_regrid_csr", src, out, indptr, indices, data, n_src, n_tgt, nnz, n_lev, src_pitch, out_pitch, dtype, layout,
                             prog, n_stage, tgt_mask, nullptr, stream);
}

extern "C" int atx_regrid_csr_ordered(const void* src, void* out, const int32_t* indptr, const int32_t* indices,
                                      const void* data, const int32_t* tgt_rows, int64_t n_src, int64_t n_tgt, int64_t nnz, int64_t n_lev,
                                      int64_t src_pitch, int64_t out_pitch, int dtype, int layout, const atx_level_op* prog,
                                      int32_t n_stage, const uint8_t* tgt_mask, void* stream) {
    ATX_REQUIRE(tgt_rows, ATX_EINVAL, "atx_regrid_csr_ordered: null tgt_rows (use atx_regrid_csr for the natural order)");
    return regrid_csr_common("atx_regrid_csr_ordered", src, out, indptr, indices, data, n_src, n_tgt, nnz, n_lev, src_pitch, out_pitch, dtype,
                             layout, prog, n_stage, tgt_mask, tgt_rows, stream);
}

extern "C" int atx_check_indices(const int32_t* idx, int64_t n, int64_t n_src, int64_t* n_bad, void* stream) {
    ATX_REQUIRE(idx && n_bad, ATX_EINVAL, "atx_check_indices: null pointer");
    ATX_REQUIRE(n >= 0 && n_src >= 0, ATX_EINVAL, "atx_check_indices: negative size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int st = hip_status(hipMemsetAsync(n_bad, 0, sizeof(int64_t), s), "atx_check_indices memset");
    if (st != ATX_OK) return st;
    if (n == 0) return ATX_OK;
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > kStreamGrid) blocks = kStreamGrid;
    hipLaunchKernelGGL(check_indices_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, idx, n, n_src,
                       reinterpret_cast<unsigned long long*>(n_bad));
    ATX_LAUNCH_CHECK("check_indices");
    return ATX_OK;
}

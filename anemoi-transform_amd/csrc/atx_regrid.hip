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
#include "atx_regrid_decl.hpp"

#include <cstdlib>
#include <cstring>

namespace atx {

thread_local int g_tile_override = 0;  // tuning hook (atx_set_tuning): per calling thread, meant for benchmarks and tests (results never depend on it)

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

// The same count with a lower bound of -1 allowed (ATX_ELL_PADDED marks absent entries with -1) and, for ordered traversals,
// over a row table that must stay inside [0, n_tgt).
__global__ void __launch_bounds__(kBlock)
check_range_kernel(const int32_t* __restrict__ idx, int64_t n, int64_t lo, int64_t hi, unsigned long long* n_bad) {
    unsigned long long bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t v = idx[i];
        bad += (v < lo || v >= hi) ? 1u : 0u;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) bad += __shfl_down(bad, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && bad) atomicAdd(n_bad, bad);
}

// ATX_VALIDATE=1 (read once per process): every gather entry point first range-checks the tables it was handed ON THE DEVICE and
// refuses the launch (ATX_EINVAL, nothing written) if an index points outside the source stack — the kernels trust their tables,
// and an out-of-range read is a GPU fault that can take the whole node with it.  A debugging aid for binders that build their own
// tables: it synchronises the stream and costs a pass over the tables, so it is off by default (the Python mirror validates on the
// host when a plan is built, GatherPlan._check).
static bool validation_on() {
    static const bool on = [] {
        const char* v = std::getenv("ATX_VALIDATE");
        return v && *v && std::strcmp(v, "0") != 0;
    }();
    return on;
}

static int validate_table(const char* fn, const char* what, const int32_t* idx, int64_t n, int64_t lo, int64_t hi, hipStream_t s) {
    if (n <= 0 || !idx) return ATX_OK;
    unsigned long long* n_bad = nullptr;
    int st = hip_status(hipMalloc(reinterpret_cast<void**>(&n_bad), sizeof(unsigned long long)), "ATX_VALIDATE: hipMalloc");
    if (st != ATX_OK) return st;
    unsigned long long host = 0;
    st = hip_status(hipMemsetAsync(n_bad, 0, sizeof(unsigned long long), s), "ATX_VALIDATE: memset");
    if (st == ATX_OK) {
        int64_t blocks = (n + kBlock - 1) / kBlock;
        if (blocks > kStreamGrid) blocks = kStreamGrid;
        hipLaunchKernelGGL(check_range_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, idx, n, lo, hi, n_bad);
        st = hip_status(hipGetLastError(), "ATX_VALIDATE: launch");
    }
    if (st == ATX_OK) st = hip_status(hipMemcpyAsync(&host, n_bad, sizeof host, hipMemcpyDeviceToHost, s), "ATX_VALIDATE: copy");
    if (st == ATX_OK) st = hip_status(hipStreamSynchronize(s), "ATX_VALIDATE: synchronise");
    (void)hipFree(n_bad);
    if (st != ATX_OK) return st;
    ATX_REQUIRE(host == 0, ATX_EINVAL, "%s: ATX_VALIDATE found %llu entries of %s outside [%lld, %lld) — launch refused", fn, host, what,
                (long long)lo, (long long)hi);
    return ATX_OK;
}

static int check_stack_args(const char* fn, const void* src, const void* out, int64_t n_src, int64_t n_tgt,
                            int64_t n_lev, int64_t sp, int64_t op, int dtype, int layout) {
    ATX_REQUIRE(src && (out || n_tgt == 0), ATX_EINVAL, "%s: null src/out pointer", fn);  // (an output of zero targets may have no storage)
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
    ATX_REQUIRE(idx || n_tgt == 0, ATX_EINVAL, "%s: null idx", fn);
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
    if (validation_on()) {
        int st = validate_table(fn, "idx", idx, n_tgt * k, pad ? -1 : 0, n_src, s);
        if (st == ATX_OK && tgt_rows) st = validate_table(fn, "tgt_rows", tgt_rows, n_tgt, 0, n_tgt, s);
        if (st != ATX_OK) return st;
    }
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
    ATX_REQUIRE(indptr || n_tgt == 0, ATX_EINVAL, "%s: null indptr", fn);
    ATX_REQUIRE(nnz >= 0 && nnz <= INT32_MAX, ATX_ENOTIMPL, "%s: nnz=%lld outside int32", fn, (long long)nnz);
    ATX_REQUIRE(nnz == 0 || (indices && data), ATX_EINVAL, "%s: null indices/data", fn);
    ATX_REQUIRE((prog == nullptr) == (n_stage == 0) && n_stage >= 0 && n_stage <= 8, ATX_EINVAL,
                "%s: prog/n_stage mismatch (n_stage=%d)", fn, n_stage);
    if (n_tgt == 0) return ATX_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (validation_on()) {
        int st = validate_table(fn, "indices", indices, nnz, 0, n_src, s);
        if (st == ATX_OK) st = validate_table(fn, "indptr", indptr, n_tgt + 1, 0, nnz + 1, s);
        if (st == ATX_OK && tgt_rows) st = validate_table(fn, "tgt_rows", tgt_rows, n_tgt, 0, n_tgt, s);
        if (st != ATX_OK) return st;
    }
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
    ATX_REQUIRE(n_bad && (idx || n == 0), ATX_EINVAL, "atx_check_indices: null pointer");
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

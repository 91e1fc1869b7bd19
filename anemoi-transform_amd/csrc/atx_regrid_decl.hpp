// Shared by the three translation units of the regrid entry points (round 4: atx_regrid.hip took 70 s to compile — the whole build's
// critical path — so its kernels are instantiated per element type in atx_regrid_f32.hip / atx_regrid_f64.hip, which compile in
// parallel, and atx_regrid.hip keeps the C ABI).
#ifndef ATX_REGRID_DECL_HPP
#define ATX_REGRID_DECL_HPP

#include "atx_common.hpp"

namespace atx {

constexpr int kMaxBatch = 16;
struct EllBatch {
    const void* src[kMaxBatch];
    void* out[kMaxBatch];
    int n;
};

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

extern thread_local int g_tile_override;  // atx_set_tuning; defined in atx_regrid.hip

template <typename T>
int regrid_ell_typed(const EllBatch& batch, const int32_t* idx, const void* w_, int64_t n_tgt, int k,
                            int n_lev, int64_t sp, int64_t op, int layout, bool pad, const Epilogue& e, hipStream_t st);
template <typename T>
int regrid_csr_typed(const void* src_, void* out_, const int32_t* indptr, const int32_t* indices,
                            const void* data_, int64_t n_tgt, int64_t nnz, int n_lev, int64_t sp, int64_t op,
                            int layout, const atx_level_op* prog, int n_stage, const uint8_t* m, const int32_t* rows, hipStream_t st);

}  // namespace atx

#endif  // ATX_REGRID_DECL_HPP

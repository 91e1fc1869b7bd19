// The regrid kernels and launchers (atx_regrid_kernels.inc) instantiated for float stacks; see atx_regrid_decl.hpp.
#include "atx_regrid_kernels.inc"

namespace atx {
template int regrid_ell_typed<float>(const EllBatch& batch, const int32_t* idx, const void* w_, int64_t n_tgt, int k,
                            int n_lev, int64_t sp, int64_t op, int layout, bool pad, const Epilogue& e, hipStream_t st);
template int regrid_csr_typed<float>(const void* src_, void* out_, const int32_t* indptr, const int32_t* indices,
                            const void* data_, int64_t n_tgt, int64_t nnz, int n_lev, int64_t sp, int64_t op,
                            int layout, const atx_level_op* prog, int n_stage, const uint8_t* m, const int32_t* rows, hipStream_t st);
}  // namespace atx

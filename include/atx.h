/*
 * atx.h — C ABI of libatx, the MI355X (gfx950) field-transform kernels.
 *
 * This is the drop-in boundary for the filter hot path of ecmwf/anemoi-transform
 * 0.4.2.  The reference is pure Python and has NO native interface on this
 * path: its filters call numpy / scipy in-process.  Every entry point below
 * therefore replaces a numpy/scipy *statement* of the reference, cited as
 * `R:` (paths relative to /root/reference/src/anemoi/transform/).  The
 * reference-side binding a maintainer would add (a ctypes stub) is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *  - Plain C types only.  Every pointer named in a signature is a DEVICE
 *    pointer (HBM) unless the comment says "host".  The caller owns every
 *    buffer; the library allocates nothing and keeps no state besides a
 *    thread-local error string.  A buffer of ZERO elements — a stack of no
 *    points, the output of a regrid to no targets — may have no storage: its
 *    pointer may be NULL, the call validates the rest, does nothing and
 *    returns ATX_OK (allocators hand out NULL for empty tensors).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *    Calls enqueue work and return without synchronising.
 *  - Return value: ATX_OK (0) or a negative ATX_E* code; atx_last_error()
 *    gives the message for the calling thread.  Nothing throws across the ABI.
 *  - Threading: every entry point is re-entrant.  The library holds no state
 *    between calls besides two PER-THREAD items — the error string and the
 *    tuning hook (atx_set_tuning) — and the RCCL binding, resolved once under
 *    std::call_once.  Several host threads may call at the same time, each on
 *    its own stream and buffers (or one thread may drive several streams); a
 *    call that fails in one thread leaves the other threads' atx_last_error()
 *    untouched.  Two calls that write the same buffer, or share a `workspace`,
 *    must be ordered by the caller (same stream, or an event).
 *    tests/test_gpu_threads.py: 4 threads x 6 rounds of atx_regrid_ell (direct
 *    and, through the per-thread tuning hook, tiled), atx_pointwise_stack,
 *    atx_reduce_stack and atx_mask_build + atx_mask_to_index at once, every
 *    result against the CPU oracle, one thread provoking ATX_EINVAL.
 *  - A *stack* is a batch of `n_lev` fields on one grid of `n_pts` points.
 *    Two HBM layouts (atx_layout):
 *      ATX_COLUMNS  element (point p, level l) at  base[p * pitch + l]
 *                   — all levels of one grid point are contiguous (a model
 *                   column).  The engine's native layout: a neighbour read is
 *                   one contiguous n_lev*B-byte run.
 *      ATX_FIELDS   element (p, l) at base[l * pitch + p] — one field after
 *                   the other, the reference's unit (`field.to_numpy()`).
 *    `pitch` is in ELEMENTS (>= n_lev for ATX_COLUMNS, >= n_pts for ATX_FIELDS).
 *  - dtype: ATX_F32 or ATX_F64; weights / scalars use the data dtype
 *    (double parameters are rounded to float for ATX_F32, as numpy does for a
 *    python-float operand).
 *  - Floating point is evaluated WITHOUT contraction (no FMA): x*a+b is two
 *    roundings, a CSR row is summed sequentially from 0 in index order, so the
 *    f64 results are bit-identical to numpy / scipy.
 */
#ifndef ATX_H
#define ATX_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the ATX_API declarations below are its ONLY dynamic symbols
 * (tests/test_host_api.py checks `nm -D --defined-only` against this header), so nothing of its C++ internals can
 * interpose, or be interposed by, another library of the process (torch's own HIP libraries, RCCL). */
#if defined(_WIN32)
#define ATX_API
#else
#define ATX_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ATX_VERSION 420 /* 0.4.2 — round 6: nothing changed but two relaxations — the declarations carry ATX_API (the library exports nothing else),
                           * and zero-element buffers may be NULL (see Conventions).
                           * round 5: no signature, enum or layout changed.  The float64 library functions are the library's own routines now
                           * (ATX_OP_EXP, ATX_OP_LOG, the tanh of ATX_COMB_SNOW_COVER, the polynomials of ATX_COMB_COS_SIN): results may differ from 0.4.1 in the
                           * last bit and stay within 1 ulp (exp, log), 2 ulp (cos, sin) and 4 ulp (tanh) of numpy's, as before.
                           * 0.4.1 — eight more multi-input operators (ATX_COMB_OPERA_CLIP .. ATX_COMB_R_TO_Q); nothing else changed.
                           * 0.4.0 — round 4: atx_vector_program takes the CAPACITY of `out` (ATX_EWORKSPACE when too small — 0.3 wrote its
                           * grown table without asking), atx_reduce* keep the no-atomics route for every shape when given a workspace.
                           * 0.3.0 — round 3: atx_reduce* take a workspace, atx_regrid_*_ordered, two more multi-input operators, and the
                           * table written by atx_vector_program grew a typed per-level part (a table built by a 0.2 library is too short for
                           * 0.3+ kernels: check atx_version() >= 300 before passing one as vec_prog) */

/* ---- status codes --------------------------------------------------------- */
enum {
    ATX_OK = 0,
    ATX_EINVAL = -1,   /* bad argument (null pointer, negative size, bad enum) -> ValueError */
    ATX_ESHAPE = -2,   /* inconsistent shapes / pitches -> AssertionError (R: filters/fields/regrid.py:377-378) */
    ATX_ENOTIMPL = -3, /* unsupported combination -> NotImplementedError (R: regrid.py:332-335) */
    ATX_EHIP = -4,     /* HIP runtime error at launch */
    ATX_EALIGN = -5,   /* pointer / pitch alignment not met for the requested layout */
    ATX_EWORKSPACE = -6, /* workspace too small */
    ATX_ECOMM = -7      /* RCCL missing or a collective failed */
};

typedef enum { ATX_F32 = 0, ATX_F64 = 1 } atx_dtype;
typedef enum { ATX_COLUMNS = 0, ATX_FIELDS = 1 } atx_layout;

/* comparison operators of apply_mask (R: filters/fields/apply_mask.py:23-36:
 * the 12 spellings map onto these 6) plus the NaN test of remove_nans
 * (R: filters/fields/remove_nans.py:101  `~np.isnan(data)`). */
typedef enum {
    ATX_CMP_GT = 0,
    ATX_CMP_LT = 1,
    ATX_CMP_EQ = 2,
    ATX_CMP_NE = 3,
    ATX_CMP_GE = 4,
    ATX_CMP_LE = 5,
    ATX_CMP_NOTNAN = 6, /* threshold ignored */
    ATX_CMP_ISNAN = 7   /* threshold ignored */
} atx_cmp;

/* per-point operators (one per reference statement). p0/p1 are the scalars. */
typedef enum {
    ATX_OP_COPY = 0,       /* y = x (bit copy)                                               */
    ATX_OP_AFFINE = 1,     /* y = x*p0 + p1        R: filters/fields/rescale.py:25           */
    ATX_OP_AFFINE_INV = 2, /* y = (x - p1) / p0    R: rescale.py:28                          */
    ATX_OP_MUL = 3,        /* y = x * p0           R: filters/fields/orog_to_z.py:59         */
    ATX_OP_DIV = 4,        /* y = x / p0           R: orog_to_z.py:77 (a true division)      */
    ATX_OP_CLIP = 5,       /* y = np.clip(x,p0,p1) R: filters/fields/clipper.py:69; a NaN
                              bound means "no bound on that side"; NaN x stays NaN          */
    ATX_OP_IMPUTE_NAN = 6, /* y = isnan(x)?p0:x    R: filters/fields/impute_nans.py:53-54    */
    ATX_OP_EXP = 7,        /* y = exp(x)           R: filters/fields/lnsp_to_sp.py:47        */
    ATX_OP_LOG = 8,        /* y = log(x)           R: lnsp_to_sp.py:65                       */
    ATX_OP_SET_NAN = 9,    /* y = NaN (canonical quiet NaN)                                  */
    ATX_OP_COUNT_ = 10
} atx_op;

/* One step of a per-level program: level l of a stack is transformed by
 * prog[s * n_lev + l] for s = 0 .. n_stage-1 in turn.  ATX_OP_COPY leaves the
 * level untouched (bit-exact pass-through of unselected fields,
 * R: filter.py:193-194).  `use_mask` != 0 additionally writes NaN where the
 * point mask is set, AFTER the arithmetic of that stage
 * (R: apply_mask.py:185 `values[self.mask] = np.nan`). */
typedef struct {
    int32_t op;       /* atx_op */
    int32_t use_mask; /* 0/1 */
    double p0;
    double p1;
} atx_level_op;

/* ---- library ------------------------------------------------------------- */
ATX_API int atx_version(void);
ATX_API const char* atx_last_error(void);          /* host string, valid until the next failing call of this thread */
ATX_API const char* atx_strerror(int code);        /* host string, static */
/* Number of HIP devices visible, or a negative ATX_EHIP. */
ATX_API int atx_device_count(void);
/* Tuning hook for benchmarks and tests: tile > 0 runs the ATX_COLUMNS regrid through the TILED kernels with that
 * many targets per workgroup; 0 = built-in choice (the direct kernel where it applies, else the tile heuristic).
 * Per CALLING THREAD (thread-local since round 3: the launches of other threads are not affected); results never depend on it,
 * only speed. */
ATX_API int atx_set_tuning(int tile);

/* ---- regrid: precomputed index(+weight) gather ---------------------------- */

/*
 * Fixed-k ("ELL") interpolation:  out[t, l] = sum_{j<k} w[t*k+j] * src[idx[t*k+j], l]
 * accumulated from 0 in j order.
 *   w == NULL : pure gather, k must be 1:  out[t, l] = src[idx[t], l]  (bit copy)
 *     R: regrid.py:380  `data = data[..., self.nearest_grid_points]`
 *     R: regrid.py:420  `data = data[..., self.mask]` (integer / compressed boolean mask)
 *     R: filters/fields/remove_nans.py:113 `data[self._mask]` (via atx_mask_to_index)
 *   w != NULL : R: regrid.py:310 `data = self.matrix @ data` for a CSR matrix whose
 *     rows all hold k entries (k-NN weights, bilinear).
 * idx: int32 [n_tgt*k], every value in [0, n_src) (check with atx_check_indices).
 * flags: 0, or ATX_ELL_PADDED (needs weights): a NEGATIVE index marks an absent entry of a padded
 *      row and is skipped (not multiplied by zero), so ragged rows of at most k entries (e.g. MIR's
 *      3-4 entries per row) run on this kernel with exactly the CSR summation order; a row of only
 *      absent entries gives 0 as scipy does.
 * w  : dtype  [n_tgt*k].
 * prog (optional, device, n_stage*n_lev entries): per-level epilogue applied to
 *   the interpolated value before it is stored (the fused regrid -> per-point
 *   chain, R: workflows/pipeline.py:46-48 without materialising intermediates);
 *   tgt_mask (optional, uint8 [n_tgt]) is the point mask used by `use_mask`.
 *   Two optional companions of prog (NULL is always valid; results never depend on them) keep the epilogue on the
 *   fastest kernel, which has no per-workgroup set-up:
 *     host_prog (HOST)   the same n_stage*n_lev entries as prog, readable by the library: when, per stage, the levels
 *               run one operator — or one up to a level and another from there on (a stack "136 levels of t, then orog"),
 *               the change on a 16-byte boundary — the operators (<= 4 stages) are passed to the kernel by value;
 *     vec_prog  (device) atx_vector_program(prog) of the stack's dtype — the per-16-byte-vector operator table, used
 *               (together with host_prog) for programs made of COPY / AFFINE / MUL with or without the mask, <= 4 stages.
 * Requirements: ATX_COLUMNS with 16-byte aligned bases and pitches that are
 *   multiples of 16 bytes take the vector path; anything else a scalar path.
 * Layouts (MI355X, O1280 -> 0.25 deg, 137 levels, fraction of the HBM peak on algorithmic bytes): ATX_COLUMNS is the gather's
 *   layout — k = 4: 0.68-0.70, k = 16: 0.52-0.59.  ATX_FIELDS (the reference's array order) runs k <= 4 and short ragged rows at
 *   0.35-0.42, but a row of more than ~8 entries re-fetches every source point once per row that uses it (k = 16: 11.5 ms against
 *   0.93 ms): convert such stacks with atx_relayout first — both conversions included that is still 4x faster (what
 *   GatherPlan.apply does by itself).  The per-point entry points run field-major stacks at full speed (0.79-0.84).
 */
#define ATX_ELL_PADDED 1
ATX_API int atx_regrid_ell(const void* src, void* out, const int32_t* idx, const void* w,
                           int64_t n_src, int64_t n_tgt, int32_t k, int64_t n_lev,
                           int64_t src_pitch, int64_t out_pitch, int dtype, int layout, int32_t flags,
                           const atx_level_op* prog, const atx_level_op* vec_prog, const atx_level_op* host_prog,
                           int32_t n_stage, const uint8_t* tgt_mask, void* stream);

/* atx_regrid_ell applied to n_stack source stacks of identical shape, dtype and pitch with ONE launch per 16 stacks
 * (ATX_COLUMNS: grid.y = stack, no launch gaps or per-launch tails; field-major stacks are launched one after the
 * other).  srcs / outs are HOST arrays of n_stack device pointers; any n_stack >= 1.
 *   R: regrid.py:204-208 — the per-field loop, when the FieldList holds several variables / time steps on one grid
 *      pair (BASELINE config 4), or the N source stacks of a target-sharded multi-GPU step. */
ATX_API int atx_regrid_ell_batch(const void* const* srcs, void* const* outs, int32_t n_stack, const int32_t* idx, const void* w,
                                 int64_t n_src, int64_t n_tgt, int32_t k, int64_t n_lev, int64_t src_pitch, int64_t out_pitch,
                                 int dtype, int layout, int32_t flags, const atx_level_op* prog, const atx_level_op* vec_prog,
                                 const atx_level_op* host_prog, int32_t n_stage, const uint8_t* tgt_mask, void* stream);

/* The same with an ORDERED traversal of the targets: row t of idx / w is the table row of OUTPUT row tgt_rows[t] (a permutation
 * of 0 .. n_tgt-1, device int32; tgt_mask stays indexed by output row).  The result is identical to atx_regrid_ell_batch on the
 * un-permuted tables, bit for bit — only the order in which the device visits the targets changes.  Visiting a lat-lon target grid
 * in column blocks (each block top to bottom) lets vertically adjacent targets, whose neighbour patches overlap, meet in an XCD's
 * L2.  Measured on O1280 -> 0.25 degree, 137 levels (profiles/r03_column_blocks_experiment.log): k = 16 +9-12 %, k = 8 +5-9 %;
 * k <= 4 is 2-7 % SLOWER (the output rows are then written band by band) — order long rows only.  ATX_COLUMNS only. */
ATX_API int atx_regrid_ell_ordered(const void* const* srcs, void* const* outs, int32_t n_stack, const int32_t* idx, const void* w,
                                   const int32_t* tgt_rows, int64_t n_src, int64_t n_tgt, int32_t k, int64_t n_lev, int64_t src_pitch,
                                   int64_t out_pitch, int dtype, int layout, int32_t flags, const atx_level_op* prog,
                                   const atx_level_op* vec_prog, const atx_level_op* host_prog, int32_t n_stage, const uint8_t* tgt_mask,
                                   void* stream);

/*
 * General CSR interpolation: out[t, l] = sum_{jj in [indptr[t], indptr[t+1])} data[jj] * src[indices[jj], l]
 * accumulated from 0 in jj order — scipy's csr_matvec.
 *   R: regrid.py:281-285 (npz keys matrix_data f64 / matrix_indices int32 / matrix_indptr int32)
 *   R: regrid.py:310
 * indptr int32 [n_tgt+1] (device), indices int32 [nnz], data dtype [nnz].
 */
ATX_API int atx_regrid_csr(const void* src, void* out, const int32_t* indptr, const int32_t* indices,
                           const void* data, int64_t n_src, int64_t n_tgt, int64_t nnz, int64_t n_lev,
                           int64_t src_pitch, int64_t out_pitch, int dtype, int layout,
                           const atx_level_op* prog, int32_t n_stage, const uint8_t* tgt_mask,
                           void* stream);
/* atx_regrid_csr with an ORDERED traversal: CSR row t (indptr[t] .. indptr[t+1]) is the row of OUTPUT point tgt_rows[t] — see
 * atx_regrid_ell_ordered; rows of 9-16 entries gain like k = 16 there.  ATX_COLUMNS only. */
ATX_API int atx_regrid_csr_ordered(const void* src, void* out, const int32_t* indptr, const int32_t* indices, const void* data,
                                   const int32_t* tgt_rows, int64_t n_src, int64_t n_tgt, int64_t nnz, int64_t n_lev, int64_t src_pitch,
                                   int64_t out_pitch, int dtype, int layout, const atx_level_op* prog, int32_t n_stage,
                                   const uint8_t* tgt_mask, void* stream);

/* Debugging aid for binders that build their own tables: with ATX_VALIDATE=1 in the environment (read once per process) every
 * atx_regrid_* call first range-checks idx / indices / indptr / tgt_rows on the device and refuses the launch with ATX_EINVAL if an
 * entry points outside the stack (the kernels trust their tables: an out-of-range read is a GPU fault).  It synchronises the stream
 * and costs a pass over the tables; off by default. */

/* Counts entries of idx[0..n) outside [0, n_src) into *n_bad (device int64,
 * zeroed by the call).  cKDTree returns n_src for "no neighbour within
 * distance_upper_bound" (R: spatial.py:630-632) — reject before gathering. */
ATX_API int atx_check_indices(const int32_t* idx, int64_t n, int64_t n_src, int64_t* n_bad, void* stream);

/* ---- per-point transforms -------------------------------------------------- */

/*
 * y[p, l] = prog_{n_stage-1}( ... prog_0(x[p, l]) )  for every level l of a stack.
 * x == y (in place) is allowed when the pitches agree.  point_mask: uint8
 * [n_pts] or NULL (required if any stage has use_mask).
 * vec_prog (device, optional): atx_vector_program(prog) of the stack's dtype; host_prog (HOST, optional): the same
 * entries as prog, readable by the library.  With both, an IN-PLACE call whose program leaves most levels alone
 * (ATX_OP_COPY) visits only the 16-byte columns holding an active level (1 of 137 levels: 5x faster).  NULL is always
 * valid for either; results never depend on them.
 *   R: filter.py:188-196 (SingleFieldFilter map over fields), rescale.py:25,28,
 *      orog_to_z.py:59,77, clipper.py:69, impute_nans.py:53-54, lnsp_to_sp.py:47,65,
 *      apply_mask.py:183-185, glacier_mask.py:33
 */
ATX_API int atx_pointwise_stack(const void* x, void* y, int64_t n_pts, int64_t n_lev,
                                int64_t x_pitch, int64_t y_pitch, int dtype, int layout,
                                const atx_level_op* prog, const atx_level_op* vec_prog, const atx_level_op* host_prog,
                                int32_t n_stage, const uint8_t* point_mask, void* stream);

/* Companion table of a per-level program, computed on the HOST (no device access), in two parts.
 * (1) Per 16-byte vector: out[s*C + c] is the operator shared by the levels c*V .. c*V+V-1 of stage s (V = 16 bytes /
 *     sizeof(dtype), C = ceil(n_lev / V); padding levels join any operator), with op = ATX_OP_MIXED (and use_mask = whether any
 *     of them uses the mask) where they differ once the parameters are rounded to dtype.
 * (2) From the next 16-byte boundary after those n_stage*C entries: the operators of EVERY level in the stack's arithmetic type,
 *     parameters and codes apart — p0[n_stage][C*V], p1[n_stage][C*V] (float or double), code[n_stage][C*V] (one byte:
 *     op | use_mask << 7; the padding of the last vector repeats the last level) — so that a lane reads the parameters of its
 *     vector's levels with two 16-byte loads: programs with a different scale per level (packed surface stacks, normalisation
 *     per level) then cost the fused regrid epilogue nothing extra (O1280 -> 0.25 deg, 137 levels: 0.44 ms with or without).
 * Returns the size of the whole table in atx_level_op entries (out == NULL: only that — always size `out` by this query);
 * out_entries is the capacity of `out` in entries: ATX_EWORKSPACE, and nothing written, when it is smaller than the table.
 * Negative on error.  Uploaded to a 16-byte aligned device buffer and passed as `vec_prog` (of the stack's dtype) it lets
 * atx_regrid_ell fuse multiply-add programs on its fastest kernel and atx_pointwise_stack run without any per-workgroup set-up;
 * vec_prog == NULL is always valid (the kernels then derive what they need themselves). */
#define ATX_OP_MIXED (-1)
ATX_API int64_t atx_vector_program(const atx_level_op* prog, int32_t n_stage, int64_t n_lev, int dtype, atx_level_op* out, int64_t out_entries);

/* ---- multi-input per-point transforms ------------------------------------------ */
/* Operators of the reference's MatchingFieldsFilter family (R: filters/fields/matching.py:90-311):
 * every input / output is a stack of the same shape, pitch and layout; level l of the
 * outputs is computed from level l of the inputs (the fields of one matching group). */
typedef enum {
    ATX_COMB_SNOW_DEPTH_M = 0, /* (sd, rsn) -> 1000.0*sd/rsn            R: filters/fields/snow_depth_m.py:42      */
    ATX_COMB_SNOW_COVER = 1,   /* (sd, rsn) -> clip(tanh(4000*(1000*sd/rsn)/clip(rsn,100,400)),0,1), >0.99 -> 1
                                                                        R: filters/fields/snow_cover.py:34-39     */
    ATX_COMB_COS_SIN = 2,      /* (x) -> (cos x, sin x); flag ATX_COMB_DEGREES: x = deg2rad(x) first
                                  R: filters/fields/cos_sin_from_rad.py:78-79, cos_sin_mean_wave_direction.py:72-76 */
    ATX_COMB_ATAN2 = 3,        /* (cos, sin) -> atan2(sin, cos); flag ATX_COMB_DEGREES: rad2deg, then wrap to [0, 360)
                                  R: cos_sin_from_rad.py:100, cos_sin_mean_wave_direction.py:97-99                 */
    ATX_COMB_W_TO_WZ = 4,      /* (w, t, q) -> (-1/(rho*g + 1e-8))*w, rho = (100*level)/(287*t*(1+0.61*q) + 1e-8)
                                  level = level_param[l]              R: filters/fields/w_to_wz.py:97-99          */
    ATX_COMB_WZ_TO_W = 5,      /* (wz, t, q) -> -1.0*rho*g*wz          R: w_to_wz.py:124-126                       */
    ATX_COMB_SUM = 6,          /* (c0, c1, ...) -> ((c0 + c1) + ...) in input order   R: filters/fields/sum.py:109-116 */
    ATX_COMB_SUB = 7,          /* (a, b) -> a - b                      R: filters/fields/accum_to_interval.py:98   */
    ATX_COMB_XY_TO_POLAR = 8,  /* (u, v) -> (hypot(u, v), mod(270 - atan2(v, u)*180/pi, 360)): speed and the direction the wind blows FROM,
                                  the "meteo" convention of earthkit.meteo.wind.array.xy_to_polar      R: filters/fields/uv_to_ddff.py:93-97   */
    ATX_COMB_POLAR_TO_XY = 9,  /* (speed, dir) -> (speed*cos(a), speed*sin(a)), a = (270 - dir)*pi/180   R: uv_to_ddff.py:121-125         */
    /* 0.4.1: the reference's remaining numpy-only per-point filters */
    ATX_COMB_OPERA_CLIP = 10,  /* (tp, qi) -> (c(tp, max)/1000, c(qi, 1)), c(v, m): v<0 -> 0, then v>=m -> m (NaN and -0.0 kept);
                                  max = level_param[l]    R: filters/fields/rodeo_opera_clipping.py:92-98, rodeo_opera_preprocessing.py:34-37 */
    ATX_COMB_OPERA_PREPROCESS = 11, /* (tp, qi, dm) -> (c(tp', max), c(qi', 1)); tp' = NaN where dm is 1 or 3, 0 where dm is 2; qi' = 0 where
                                  dm is 2; max = level_param[l]                R: rodeo_opera_preprocessing.py:83-87, :190-200        */
    ATX_COMB_ORAS6 = 12,       /* (x, siconc) -> x cleaned where siconc <= 1e-5; inputs[1] is ONE contiguous field [n_pts] shared by every
                                  level; level_param[l] = what level l is, an ATX_ORAS6_* code    R: filters/fields/oras6_clipping.py:189-215 */
    ATX_COMB_LOOKUP = 13,      /* (class) -> table[class]; level_param = double[1 + n]: n, then the value of class 0 .. n-1; a class that is
                                  not one of 0 .. n-1 (the reference's KeyError) gives NaN        R: filters/fields/land_parameters.py:71 */
    /* humidity conversions: the arithmetic is earthkit-meteo's (thermo.array; IFS saturation formulas), restated from its published form */
    ATX_COMB_R_TO_D = 14,      /* (r %, t) -> dewpoint: r == 0 -> 1e-4 first; e = r*es_water(t)/100; T(e) = (32.19 ln(e/611.21) - 17.502*273.16) /
                                  (ln(e/611.21) - 17.502)                               R: filters/fields/dewpoint.py:59-65               */
    ATX_COMB_D_TO_R = 15,      /* (td, t) -> 100*es_water(td)/es_water(t)                R: dewpoint.py:67-72                              */
    ATX_COMB_Q_TO_R = 16,      /* (q, t) or (q, t, p) -> 100*e/es_mixed(t), e = p q/(eps + (1-eps) q), eps = 287.0597/461.5250; without a
                                  pressure operand p = 100*level_param[l] (levelist, hPa)   R: filters/fields/q_to_r.py:70-75, q_height.py:117-121 */
    ATX_COMB_R_TO_Q = 17,      /* (r, t) or (r, t, p) -> eps e/(p - (1-eps) e), e = r*es_mixed(t)/100; NaN where p - e < 1e-4
                                                                                         R: q_to_r.py:77-83, q_height.py:138-142           */
    ATX_COMB_COUNT_ = 18
} atx_comb;
#define ATX_COMB_DEGREES 1
#define ATX_COMB_MAX_INPUTS 8
/* what a level of an ATX_COMB_ORAS6 stack holds (R: oras6_clipping.py:196-215) */
#define ATX_ORAS6_KEEP 0        /* passed on as it is (siconc itself)                                    */
#define ATX_ORAS6_ZERO 1        /* 0 where there is no ice (velocities, salinity, pressure, volumes, albedo) */
#define ATX_ORAS6_TEMPERATURE 2 /* 273.15 where there is no ice (sitemptop, sntemp, vasit)                   */
#define ATX_ORAS6_CELSIUS 3     /* a snow temperature archived in Celsius: + 273.15 first, then as above (:190-191) */
#define ATX_ORAS6_HEAT 4        /* 0 where there is no ice, then 0 wherever the value is >= -1e-5 (sihc, snhc) */
#define ATX_ORAS6_SURFACE 5     /* tos: raised to 271.15 - 1e-5 where it is at or below it; the ice mask is not used */
/* inputs / outputs: HOST arrays of n_in / n_out DEVICE pointers (n_in <= ATX_COMB_MAX_INPUTS, n_out <= 2).
 * level_param: device double[n_lev] or NULL (required by the W/WZ operators and by operators 10 to 13, which say above what they
 * read from it, and by operators 16 / 17 when they get no pressure operand).
 * All stacks share n_pts, n_lev, pitch and layout; the padding of the outputs (elements between a row's length and the pitch) is
 * written with zeros. */
ATX_API int atx_combine_stack(int op, const void* const* inputs, int32_t n_in, void* const* outputs, int32_t n_out,
                              int64_t n_pts, int64_t n_lev, int64_t pitch, int dtype, int layout,
                              const double* level_param, int32_t flags, void* stream);

/* ---- masks ------------------------------------------------------------------ */

/* mask[i] = (m[i*m_stride] CMP threshold) ? 1 : 0   for i < n.
 *   R: apply_mask.py:160-163  `OPERATORS[op](mask_values, threshold)` / `mask_values == mask_value`
 *   R: remove_nans.py:101     `~np.isnan(data)`  (ATX_CMP_NOTNAN)
 * m_stride in elements (1 for a flat field, the pitch for a level of an ATX_COLUMNS stack). */
ATX_API int atx_mask_build(const void* m, int64_t m_stride, uint8_t* mask, int64_t n, int cmp, double threshold,
                           int dtype, void* stream);

/* *count (device int64) = number of non-zero mask bytes.  R: numpy boolean indexing output size. */
ATX_API int atx_mask_count(const uint8_t* mask, int64_t n, int64_t* count, void* stream);

/* Stable compaction: index[0..count) = ascending positions i with mask[i] != 0;
 * *count (device int64) receives the total.  The index list turns
 * `data[bool_mask]` into atx_regrid_ell(k=1, w=NULL).
 *   R: remove_nans.py:110-116, regrid.py:420 (boolean mask)
 * workspace: device scratch of at least atx_mask_to_index_workspace(n) bytes. */
ATX_API size_t atx_mask_to_index_workspace(int64_t n);
ATX_API int atx_mask_to_index(const uint8_t* mask, int64_t n, int32_t* index, int64_t* count,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- reductions (range / validity checks) ---------------------------------- */
typedef enum { ATX_RED_MIN = 0, ATX_RED_MAX = 1, ATX_RED_NANCOUNT = 2, ATX_RED_MINMAX = 3 } atx_red;
/* result: device double[1] (double[2] for ATX_RED_MINMAX: minimum, maximum — both from ONE pass over the data);
 * min/max ignore nothing (NaN propagates like np.min); NANCOUNT returns the count as a double.
 *   R: filters/fields/cos_sin_from_rad.py:73-76 `data.min()/max()`;
 *      tests/field_filters/test_apply_mask.py:106 `np.sum(np.isnan(result))`
 * workspace (optional, NULL = none): device scratch of atx_reduce_workspace() bytes, 8-byte aligned, usable by one call at a time
 * per stream (no initialisation needed).  With it the workgroups store partials and a second one-workgroup launch combines them: no
 * per-workgroup atomics on `result`, no initialisation launch, and `result` may be a pinned HOST cell the caller reads after
 * synchronising the stream (no copy back) — for EVERY shape: lengths that are not a multiple of the 16-byte vector, unaligned
 * bases (two scalar passes for MINMAX) and empty input (the identities: +inf, -inf, 0).  Without a workspace the workgroups
 * combine through atomics on `result`, which must then be DEVICE memory. */
ATX_API size_t atx_reduce_workspace(void);
ATX_API int atx_reduce(const void* x, int64_t n, int red, double* result, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* the same over the n_pts x n_lev elements of a (pitched) stack, padding excluded */
ATX_API int atx_reduce_stack(const void* x, int64_t n_pts, int64_t n_lev, int64_t pitch, int red, double* result,
                             int dtype, int layout, void* workspace, size_t workspace_bytes, void* stream);

/* ---- k-nearest-neighbour index build ---------------------------------------- */
/* Exact k-NN on the unit sphere by chord distance — device counterpart of
 *   R: spatial.py:587-635  cKDTree(source_xyz).query(target_xyz, k [, distance_upper_bound])
 * xyz arrays are float64 [n, 3] computed on the host as R: spatial.py:132-167 does.
 * atx_knn_build sorts the sources along a Morton curve and builds an implicit box tree in
 * `workspace` (atx_knn_workspace_bytes(n_src) bytes, 256-byte aligned, owned by the caller,
 * reusable for any number of queries).  atx_knn_query writes, per target, the k nearest
 * source indices (int32, ascending distance, ties by lower index, n_src if fewer than k
 * exist) and their SQUARED distances, bit-identical to scipy's float64 arithmetic; k <= 17
 * (16 neighbours plus one to look ahead: a caller that needs cKDTree's own order among
 * EXACTLY equidistant candidates queries k+1, finds the rows with equal adjacent distances and
 * re-resolves those with cKDTree — what the host mirror's nearest_grid_points_device does). */
ATX_API size_t atx_knn_workspace_bytes(int64_t n_src);
ATX_API int atx_knn_build(const double* src_xyz, int64_t n_src, void* workspace, size_t workspace_bytes, void* stream);
ATX_API int atx_knn_query(const void* workspace, int64_t n_src, const double* tgt_xyz, int64_t n_tgt, int32_t k,
                          int32_t* idx_out, double* d2_out, void* stream);

/* ---- mask builders ------------------------------------------------------------- */
/* inside[i] = 1 if the ray from the Earth's centre through global point i hits any of the `k`
 * triangles (nb[i][j], nb[i][(j+1)%k], nb[i][(j+2)%k]) of its k nearest limited-area points
 * (Moeller-Trumbore, epsilon 1e-7) — the per-point loop of
 *   R: spatial.py:404-424 (cutout_mask) over R: spatial.py:186-233 (Triangle3D.intersect)
 * evaluated for all points at once.  global_xyz [n,3] and lam_xyz [n_lam,3] float64 unit-sphere
 * coordinates, neighbours int32 [n,k] (from atx_knn_query; k <= 17), inside uint8 [n].  A triangle with a vertex
 * index outside [0, n_lam) is skipped, never dereferenced. */
ATX_API int atx_cutout_inside(const double* global_xyz, int64_t n, const double* lam_xyz, int64_t n_lam,
                              const int32_t* neighbours, int32_t k, uint8_t* inside, void* stream);

/* ---- multi-GPU: the source exchange of a target-sharded regrid ------------------------ */
/* One process per GPU; RCCL (over xGMI on an MI355X node) is bound at first use (dlopen), so the library loads without
 * it.  The regrid path itself needs no collective: rows of the operator are independent, every rank computes a
 * contiguous slice of the target points (R: filters/fields/regrid.py:204-208 is a single-process loop — there is no
 * reference counterpart).  What has to travel is the SOURCE stack, once:
 *   atx_bcast          the whole pitched stack from its owner ("source broadcast once")
 *   atx_all_gather     every rank's stack onto every rank in one collective
 *   atx_exchange       or only the slab of source columns each peer's target slice references (ATX_COLUMNS: a
 *                      contiguous byte range), grouped send/recv — about 1/world of the bytes for lat-lon targets
 *   atx_gather_shards  optionally the target slices back onto every rank
 * All calls enqueue on `stream` of the CURRENT device and return; buffers are device memory owned by the caller.
 * A communicator belongs to the device that was current in atx_comm_init.  Errors: ATX_ECOMM + atx_last_error(). */
typedef struct atx_comm atx_comm;
#define ATX_COMM_ID_BYTES 128
/* RCCL's version code (e.g. 22205), or a negative status if RCCL cannot be loaded. */
ATX_API int atx_comm_version(void);
/* Rank 0 fills `id` (HOST, ATX_COMM_ID_BYTES) and hands it to every rank out of band (file, socket, MPI, a
 * torch.distributed store ...). */
ATX_API int atx_comm_unique_id(void* id);
/* Collective over all `world` ranks: join the job identified by `id` as `rank`, on the current HIP device. */
ATX_API int atx_comm_init(atx_comm** comm, int32_t world, int32_t rank, const void* id);
ATX_API int atx_comm_destroy(atx_comm* comm);
ATX_API int atx_comm_rank(const atx_comm* comm);
ATX_API int atx_comm_world(const atx_comm* comm);
/* buf[0..n_bytes) of rank `root` onto every rank, in place. */
ATX_API int atx_bcast(atx_comm* comm, void* buf, int64_t n_bytes, int32_t root, void* stream);
/* Every rank contributes send[0..bytes_per_rank); afterwards recv[p * bytes_per_rank ...) holds rank p's contribution on every rank
 * (recv: world * bytes_per_rank bytes; send may be the rank's own slot of recv).  The whole-stack exchange of a job in which every
 * rank owns one source stack, as ONE collective instead of `world` broadcasts. */
ATX_API int atx_all_gather(atx_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
/* send_ptrs / send_bytes / recv_ptrs / recv_bytes: HOST arrays of `world` entries, one per peer; entry p of the send
 * side goes to rank p, entry p of the receive side is filled by rank p (byte counts must match pairwise across ranks;
 * zero skips the pair; the own entry is a device-to-device copy). */
ATX_API int atx_exchange(atx_comm* comm, const void* const* send_ptrs, const int64_t* send_bytes, void* const* recv_ptrs,
                         const int64_t* recv_bytes, void* stream);
/* byte_offsets: HOST array of world + 1 non-decreasing offsets into buf; rank p owns [byte_offsets[p], byte_offsets[p+1])
 * and has filled it; afterwards every rank holds all ranges (ATX_COLUMNS: a rank's target slice is such a range). */
ATX_API int atx_gather_shards(atx_comm* comm, void* buf, const int64_t* byte_offsets, void* stream);

/* ---- measurement aid ------------------------------------------------------------ */
/* dst[0..n_bytes) = src[0..n_bytes): a plain streaming copy, one 16-byte vector per lane, one workgroup per 4 KB, never
 * tuned again.  It exists so that profiling has a FIXED kernel with a known byte count in the library's access width:
 * tools/pmc_probe.py calibrates the FETCH_SIZE / WRITE_SIZE counters on it, tools/hbm_ceiling.py reports it as the
 * device's practical 1 read : 1 write rate.  n_bytes a multiple of 16, both pointers 16-byte aligned. */
ATX_API int atx_stream_copy(const void* src, void* dst, int64_t n_bytes, void* stream);

/* ---- layout --------------------------------------------------------------- */
/* dst[p, l] = src[p, l] between layouts / pitches (LDS-tiled transpose when the
 * layouts differ, strided copy when they agree).  No reference counterpart:
 * `field.to_numpy()` (R: fields.py:178-202) is the field-major view of a stack. */
ATX_API int atx_relayout(const void* src, void* dst, int64_t n_pts, int64_t n_lev,
                         int64_t src_pitch, int64_t dst_pitch, int src_layout, int dst_layout,
                         int dtype, void* stream);

/* dst level j = src level level_map[j] for j in [0, n_map); a negative entry leaves dst level j untouched.
 * Both stacks in `layout`; level_map is a HOST array (validated before launch, passed to the kernel by value).
 *   R: filter.py:188-196 / fields.py:35-48 — the reference re-lists fields freely (a FieldList is a Python
 *      list of independent arrays); on a stack, re-listing is this level gather. */
ATX_API int atx_select_levels(const void* src, void* dst, const int32_t* level_map, int32_t n_map, int64_t n_pts,
                              int64_t n_src_lev, int64_t src_pitch, int64_t dst_pitch, int dtype, int layout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATX_H */

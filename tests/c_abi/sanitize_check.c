/* Host-side sanitizer harness (tests/test_host_sanitizers.py): a C99 program built with -fsanitize=address,undefined and linked against
 * the host-sanitized libatx (tools/build_sanitized.sh).  It drives everything the library does WITHOUT a device: argument validation of
 * every entry point (bad arguments must be refused before anything is dereferenced or launched), the host-side table builder
 * atx_vector_program over many shapes with buffers of EXACTLY the size its query reports (one byte past is an ASan report), error
 * strings, the workspace-size queries, and the dlopen'ed collective binding (ATX_RCCL_LIBRARY points at the sanitized stand-in):
 * version, unique id, communicator creation failing cleanly when there is no GPU.  Any sanitizer report aborts the program
 * (-fno-sanitize-recover=all); a wrong return code counts as a failure. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "atx.h"

static int failures = 0;
#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            printf("FAILED line %d: %s (last error: %s)\n", __LINE__, #cond, atx_last_error()); \
            ++failures;                                                   \
        }                                                                 \
    } while (0)

static void vector_programs(void) {
    /* every (stages, levels, dtype) in a grid of shapes: query the size, allocate exactly that, build, read every byte back */
    static const int levels[] = {1, 2, 3, 4, 5, 7, 8, 9, 31, 137, 1000};
    unsigned checksum = 0;
    for (int dtype = ATX_F32; dtype <= ATX_F64; ++dtype)
        for (int n_stage = 1; n_stage <= 8; ++n_stage)
            for (size_t li = 0; li < sizeof(levels) / sizeof(levels[0]); ++li) {
                const int n_lev = levels[li];
                atx_level_op* prog = (atx_level_op*)malloc(sizeof(atx_level_op) * (size_t)n_stage * (size_t)n_lev);
                for (int i = 0; i < n_stage * n_lev; ++i) {
                    prog[i].op = (i * 7 + n_stage) % 5 == 0 ? ATX_OP_CLIP : ((i % 3) ? ATX_OP_AFFINE : ATX_OP_MUL);
                    prog[i].use_mask = (i % 11) == 0;
                    prog[i].p0 = 1.0 + 0.001 * i;
                    prog[i].p1 = -0.5 * i;
                }
                const int64_t n = atx_vector_program(prog, n_stage, n_lev, dtype, NULL, 0);
                CHECK(n > 0);
                if (n > 0) {
                    atx_level_op* out = (atx_level_op*)malloc(sizeof(atx_level_op) * (size_t)n); /* exactly the reported size */
                    CHECK(atx_vector_program(prog, n_stage, n_lev, dtype, out, n - 1) == ATX_EWORKSPACE);
                    CHECK(atx_vector_program(prog, n_stage, n_lev, dtype, out, n) == n);
                    const unsigned char* bytes = (const unsigned char*)out;
                    for (size_t b = 0; b < sizeof(atx_level_op) * (size_t)n; ++b) checksum = checksum * 31u + bytes[b]; /* all of it initialised */
                    free(out);
                }
                free(prog);
            }
    CHECK(atx_vector_program(NULL, 1, 4, ATX_F32, NULL, 0) == ATX_EINVAL);
    CHECK(atx_vector_program((const atx_level_op*)(uintptr_t)16, 9, 4, ATX_F32, NULL, 0) == ATX_EINVAL);
    CHECK(atx_vector_program((const atx_level_op*)(uintptr_t)16, 1, 0, ATX_F32, NULL, 0) == ATX_EINVAL);
    CHECK(atx_vector_program((const atx_level_op*)(uintptr_t)16, 1, 4, 7, NULL, 0) == ATX_EINVAL);
    printf("vector programs: checksum %08x\n", checksum);
}

static void validation(void) {
    void* one = (void*)(uintptr_t)16; /* never dereferenced: validation refuses first */
    const void* srcs[1] = {one};
    void* outs[1] = {one};
    CHECK(atx_version() == ATX_VERSION);
    CHECK(strcmp(atx_strerror(ATX_ESHAPE), "shape mismatch") == 0);
    for (int code = 1; code >= -12; --code) CHECK(atx_strerror(code) != NULL && strlen(atx_strerror(code)) > 0); /* unknown codes too */
    /* regrid: null pointers, pitches, k, dtype / layout enums, flags, program companions */
    CHECK(atx_regrid_ell(NULL, NULL, NULL, NULL, 1, 1, 1, 1, 1, 1, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(strstr(atx_last_error(), "null") != NULL);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 8, 1, 4, 2, 4, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_ESHAPE);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 8, 3, 4, 4, 4, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell(one, one, one, one, 8, 8, 99, 4, 4, 4, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 8, 1, 4, 4, 4, 7, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 8, 1, 4, 4, 4, ATX_F32, 5, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 8, 1, 4, 4, 4, ATX_F32, ATX_COLUMNS, ATX_ELL_PADDED, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 8, 1, 4, 4, 4, ATX_F32, ATX_COLUMNS, 0, NULL, one, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 8, 1, 4, 4, 4, ATX_F32, ATX_COLUMNS, 0, one, NULL, NULL, 9, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell(one, one, one, NULL, 8, 0, 1, 4, 4, 4, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_OK); /* no targets: nothing to do */
    CHECK(atx_regrid_ell_batch(NULL, NULL, 0, one, NULL, 8, 8, 1, 4, 4, 4, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell_ordered(srcs, outs, 1, one, NULL, NULL, 8, 8, 1, 4, 4, 4, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_ell_ordered(srcs, outs, 1, one, NULL, one, 8, 8, 1, 4, 8, 8, ATX_F32, ATX_FIELDS, 0, NULL, NULL, NULL, 0, NULL, NULL) == ATX_ENOTIMPL);
    CHECK(atx_regrid_csr(one, one, NULL, one, one, 8, 8, 4, 4, 4, 4, ATX_F64, ATX_COLUMNS, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_csr(one, one, one, NULL, NULL, 8, 8, 4, 4, 4, 4, ATX_F64, ATX_COLUMNS, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_regrid_csr(one, one, one, one, one, 8, 8, -1, 4, 4, 4, ATX_F64, ATX_COLUMNS, NULL, 0, NULL, NULL) == ATX_ENOTIMPL);
    CHECK(atx_regrid_csr_ordered(one, one, one, one, one, NULL, 8, 8, 4, 4, 4, 4, ATX_F64, ATX_COLUMNS, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_check_indices(NULL, 4, 4, NULL, NULL) == ATX_EINVAL);
    /* per-point, multi-input, masks, reductions, layout, level gather, k-NN, cutout, copy */
    CHECK(atx_pointwise_stack(NULL, NULL, 1, 1, 1, 1, ATX_F64, ATX_FIELDS, NULL, NULL, NULL, 1, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_pointwise_stack(one, one, 8, 4, 4, 4, ATX_F32, ATX_COLUMNS, one, NULL, NULL, 0, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_pointwise_stack(one, one, 8, 4, 2, 4, ATX_F32, ATX_COLUMNS, one, NULL, NULL, 1, NULL, NULL) == ATX_ESHAPE);
    CHECK(atx_pointwise_stack(one, one, 0, 4, 4, 4, ATX_F32, ATX_COLUMNS, one, NULL, NULL, 1, NULL, NULL) == ATX_OK);
    CHECK(atx_mask_build(one, 1, one, 8, 99, 0.0, ATX_F32, NULL) == ATX_EINVAL);
    CHECK(atx_mask_build(NULL, 1, one, 8, ATX_CMP_GT, 0.0, ATX_F32, NULL) == ATX_EINVAL);
    CHECK(atx_mask_build(one, 1, (uint8_t*)(uintptr_t)17, 8, ATX_CMP_GT, 0.0, ATX_F32, NULL) == ATX_EALIGN);
    CHECK(atx_mask_count(NULL, 8, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_mask_to_index_workspace(1000) > 0 && atx_mask_to_index_workspace(-1) == 0);
    CHECK(atx_mask_to_index(one, 8, one, one, one, 0, NULL) == ATX_EWORKSPACE);
    CHECK(atx_reduce_workspace() >= 2 * sizeof(double));
    CHECK(atx_reduce(NULL, 8, ATX_RED_MIN, NULL, ATX_F32, NULL, 0, NULL) == ATX_EINVAL);
    CHECK(atx_reduce(one, 8, 9, one, ATX_F32, NULL, 0, NULL) == ATX_EINVAL);
    CHECK(atx_reduce(one, 8, ATX_RED_MIN, one, ATX_F32, one, 8, NULL) == ATX_EWORKSPACE);
    CHECK(atx_reduce(one, 8, ATX_RED_MIN, one, ATX_F32, (void*)(uintptr_t)17, atx_reduce_workspace(), NULL) == ATX_EALIGN);
    CHECK(atx_reduce_stack(one, 8, 4, 2, ATX_RED_MAX, one, ATX_F64, ATX_COLUMNS, NULL, 0, NULL) == ATX_ESHAPE);
    CHECK(atx_reduce_stack(one, 8, 4, 4, ATX_RED_MAX, one, ATX_F64, 3, NULL, 0, NULL) == ATX_EINVAL);
    CHECK(atx_relayout(one, one, 8, 4, 4, 8, ATX_COLUMNS, ATX_FIELDS, ATX_F32, NULL) == ATX_EINVAL); /* in place */
    CHECK(atx_relayout(NULL, one, 8, 4, 4, 8, ATX_COLUMNS, ATX_FIELDS, ATX_F32, NULL) == ATX_EINVAL);
    {
        int32_t map_ok[2] = {0, 3}, map_bad[2] = {0, 4};
        void* other = (void*)(uintptr_t)4096;
        CHECK(atx_select_levels(one, other, map_bad, 2, 8, 4, 4, 4, ATX_F32, ATX_COLUMNS, NULL) == ATX_EINVAL);
        CHECK(atx_select_levels(one, other, map_ok, 2, 8, 4, 4, 1, ATX_F32, ATX_COLUMNS, NULL) == ATX_ESHAPE);
        CHECK(atx_select_levels(NULL, other, map_ok, 2, 8, 4, 4, 4, ATX_F32, ATX_COLUMNS, NULL) == ATX_EINVAL);
    }
    CHECK(atx_knn_workspace_bytes(0) == 0 && atx_knn_workspace_bytes(1000) > 0);
    CHECK(atx_knn_build(NULL, 8, NULL, 0, NULL) == ATX_EINVAL);
    CHECK(atx_cutout_inside(NULL, 8, NULL, 8, NULL, 3, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_cutout_inside(one, 8, one, 8, one, 99, one, NULL) == ATX_EINVAL);
    CHECK(atx_stream_copy(one, one, 24, NULL) == ATX_EINVAL);
    CHECK(atx_stream_copy(one, (void*)(uintptr_t)17, 32, NULL) == ATX_EALIGN);
    CHECK(atx_stream_copy(one, one, 0, NULL) == ATX_OK);
    {
        const void* ins[1] = {one};
        void* outs1[1] = {one};
        CHECK(atx_combine_stack(99, ins, 1, outs1, 1, 8, 4, 4, ATX_F32, ATX_COLUMNS, NULL, 0, NULL) == ATX_EINVAL);
        CHECK(atx_combine_stack(ATX_COMB_SUB, NULL, 2, outs1, 1, 8, 4, 4, ATX_F32, ATX_COLUMNS, NULL, 0, NULL) == ATX_EINVAL);
    }
    CHECK(atx_set_tuning(0) == ATX_OK);
}

static void collectives(void) {
    /* the binding to the collective library (dlopen of ATX_RCCL_LIBRARY: the sanitized stand-in) and the bookkeeping around it */
    unsigned char id[ATX_COMM_ID_BYTES];
    void* one = (void*)(uintptr_t)16;
    atx_comm* comm = NULL;
    memset(id, 0, sizeof id);
    CHECK(atx_bcast(NULL, one, 16, 0, NULL) == ATX_EINVAL);
    CHECK(strstr(atx_last_error(), "communicator") != NULL);
    CHECK(atx_exchange(NULL, NULL, NULL, NULL, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_all_gather(NULL, one, one, 16, NULL) == ATX_EINVAL);
    CHECK(atx_gather_shards(NULL, one, NULL, NULL) == ATX_EINVAL);
    CHECK(atx_comm_destroy(NULL) == ATX_OK);
    CHECK(atx_comm_rank(NULL) == ATX_EINVAL && atx_comm_world(NULL) == ATX_EINVAL);
    CHECK(atx_comm_init(NULL, 1, 0, id) == ATX_EINVAL);
    CHECK(atx_comm_init(&comm, 2, 5, id) == ATX_EINVAL && strstr(atx_last_error(), "rank 5") != NULL);
    CHECK(atx_comm_unique_id(NULL) == ATX_EINVAL);
    if (getenv("ATX_RCCL_LIBRARY")) {
        CHECK(atx_comm_version() >= 20000);                 /* bound through dlopen / dlsym */
        CHECK(atx_comm_unique_id(id) == ATX_OK);             /* the stand-in's 128-byte token */
        const int rc = atx_comm_init(&comm, 1, 0, id);       /* no GPU in the build container: must fail CLEANLY (HIP or stand-in error) */
        CHECK(rc == ATX_OK || rc == ATX_EHIP || rc == ATX_ECOMM);
        if (rc == ATX_OK) {
            CHECK(atx_comm_rank(comm) == 0 && atx_comm_world(comm) == 1);
            CHECK(atx_bcast(comm, one, 16, 3, NULL) == ATX_EINVAL); /* root outside the world */
            CHECK(atx_comm_destroy(comm) == ATX_OK);
        } else {
            CHECK(comm == NULL && strlen(atx_last_error()) > 0);
        }
    } else {
        printf("ATX_RCCL_LIBRARY not set: collective binding not exercised\n");
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "--lie-about-capacity") == 0) {
        /* negative control: a caller that claims more room than it allocated.  The library writes the table it was promised room for;
         * the INSTRUMENTED library must be caught doing so (heap-buffer-overflow inside atx_vector_program) — proof that the pass is armed. */
        atx_level_op prog[2] = {{ATX_OP_AFFINE, 0, 2.0, 1.0}, {ATX_OP_COPY, 0, 0.0, 0.0}};
        const int64_t n = atx_vector_program(prog, 1, 2, ATX_F32, NULL, 0);
        atx_level_op* out = (atx_level_op*)malloc(sizeof(atx_level_op) * (size_t)(n - 1));
        const int64_t got = atx_vector_program(prog, 1, 2, ATX_F32, out, n);
        printf("not caught (%lld)\n", (long long)got);
        free(out);
        return 0;
    }
    validation();
    vector_programs();
    collectives();
    printf(failures ? "FAILED %d\n" : "ok\n", failures);
    return failures;
}

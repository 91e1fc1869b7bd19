/* include/atx.h consumed by a plain C99 translation unit: the boundary is a C ABI, not a C++ one.
 * Built and run by tests/test_host_api.py::test_header_is_plain_c_and_links (no GPU needed: only entry points that
 * validate their arguments before touching HIP are called). */
#include <stdio.h>
#include <string.h>

#include "atx.h"

int main(void) {
    int failures = 0;
    if (atx_version() != ATX_VERSION) { printf("version %d != header %d\n", atx_version(), ATX_VERSION); ++failures; }
    if (strcmp(atx_strerror(ATX_ESHAPE), "shape mismatch") != 0) ++failures;
    /* null pointers are rejected before any launch, with a message */
    if (atx_regrid_ell(NULL, NULL, NULL, NULL, 1, 1, 1, 1, 1, 1, ATX_F32, ATX_COLUMNS, 0, NULL, NULL, NULL, 0, NULL, NULL) != ATX_EINVAL) ++failures;
    if (strstr(atx_last_error(), "null") == NULL) ++failures;
    if (atx_pointwise_stack(NULL, NULL, 1, 1, 1, 1, ATX_F64, ATX_FIELDS, NULL, NULL, NULL, 1, NULL, NULL) != ATX_EINVAL) ++failures;
    if (atx_bcast(NULL, NULL, 0, 0, NULL) != ATX_EINVAL) ++failures;
    if (atx_comm_destroy(NULL) != ATX_OK) ++failures;
    {
        /* the host-side helper: per-vector form of a two-level program whose levels differ -> one MIXED float32 vector */
        atx_level_op prog[2] = {{ATX_OP_AFFINE, 0, 2.0, 1.0}, {ATX_OP_COPY, 0, 0.0, 0.0}};
        atx_level_op out[8];
        /* 1 per-vector entry (24 B) padded to 32 B, then 4 levels x (2 floats + 1 code byte) = 36 B: 68 B = 3 entries */
        const long long n = atx_vector_program(prog, 1, 2, ATX_F32, NULL, 0);
        if (n != 3) ++failures;
        if (atx_vector_program(prog, 1, 2, ATX_F32, out, n - 1) != ATX_EWORKSPACE) ++failures; /* capacity is checked */
        if (atx_vector_program(prog, 1, 2, ATX_F32, out, n) != n || out[0].op != ATX_OP_MIXED) ++failures;
        if (((const float*)((const char*)out + 32))[0] != 2.0f || ((const unsigned char*)out + 64)[0] != ATX_OP_AFFINE) ++failures;
    }
    if (atx_mask_to_index_workspace(1000) == 0) ++failures;
    if (sizeof(atx_level_op) != 24) ++failures;
    printf(failures ? "FAILED %d\n" : "ok\n", failures);
    return failures;
}

// The slice of the NCCL / RCCL public C interface that libatx binds with dlsym (atx_comm.hip), declared once so that no RCCL header
// is needed to build — and so that the test-only stand-in (tests/rccl_stub/rccl_stub.cpp) defines its functions with EXACTLY these
// types: round 4's host UBSan pass (-fsanitize=function) flagged the stand-in's `void*`-typed definitions being called through
// libatx's typed pointers.  Names and layouts are NCCL's own (nccl.h): a stable, documented ABI.
#ifndef ATX_NCCL_ABI_H
#define ATX_NCCL_ABI_H

#include <hip/hip_runtime.h>
#include <stddef.h>

#define ATX_NCCL_UNIQUE_ID_BYTES 128

typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[ATX_NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;    // ncclSuccess == 0
typedef int ncclDataType_t;  // ncclInt8 / ncclChar == 0: libatx moves bytes

typedef ncclResult_t (*atx_ncclGetVersion_t)(int*);
typedef ncclResult_t (*atx_ncclGetUniqueId_t)(ncclUniqueId*);
typedef ncclResult_t (*atx_ncclCommInitRank_t)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*atx_ncclCommDestroy_t)(ncclComm_t);
typedef const char* (*atx_ncclGetErrorString_t)(ncclResult_t);
typedef ncclResult_t (*atx_ncclBroadcast_t)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*atx_ncclAllGather_t)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*atx_ncclSend_t)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*atx_ncclRecv_t)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*atx_ncclGroup_t)(void);

#endif  // ATX_NCCL_ABI_H

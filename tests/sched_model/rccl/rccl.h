// declarations only: what csrc/rccl_loader.hpp takes the types of (the model never opens a communicator)
#pragma once
#include <hip/hip_runtime.h>
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclInternalError = 3 } ncclResult_t;
typedef enum { ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count);
}

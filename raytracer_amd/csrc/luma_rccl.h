// luma_rccl.h -- the handful of RCCL entry points the band gather uses, bound at first use.
// libluma_hip.so does not link librccl: a process that renders on one GPU (or hands its bands
// to a host frame over PCIe) never loads it; luma_hip_group_render_device binds it on demand,
// sharing the copy the process already has (a torch process has RCCL loaded under the same
// soname) or loading /opt/rocm/lib/librccl.so.1 through this library's runpath.
#pragma once

#include <rccl/rccl.h>

namespace luma {

struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char*  (*GetErrorString)(ncclResult_t);
};

// nullptr when RCCL cannot be loaded; `why` then says what failed.
const RcclApi* rccl_api(const char** why);

}  // namespace luma

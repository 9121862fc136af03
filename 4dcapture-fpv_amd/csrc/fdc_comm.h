// RCCL communicator of the sharded optimiser (SURVEY.md 8b "halo_exchange", 8e): one process per GPU, ranks own contiguous
// blocks of frames, ONE all-gather of a 1.5 KB message per iteration on the COMPUTE stream (no stream hand-over, no host
// round trip between the packing launch, the collective and the unpacking launch).
// RCCL is bound at run time (dlopen of librccl.so.1 -- the copy already in the process when PyTorch-ROCm loaded it): the
// library has no link-time dependency on it and loads on hosts without RCCL; only fdcap_comm_* need it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

namespace fdc {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;

    bool load() {
        if (lib) return true;
        // FDCAP_RCCL_LIB names the library; FDCAP_RCCL_LIB_ONLY=1 tries nothing else (hosts that must not pick up a system copy; tests)
        const char* only = getenv("FDCAP_RCCL_LIB_ONLY");
        const bool strict = only && only[0] == '1';
        const char* names[] = {getenv("FDCAP_RCCL_LIB"), strict ? nullptr : "librccl.so.1", strict ? nullptr : "librccl.so",
                               strict ? nullptr : "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !n[0]) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            const char* m = dlerror();                               // (ONE call: dlerror() clears the message it returns)
            err = std::string("librccl not found: ") + (m ? m : "");
            return false;
        }
#define FDC_SYM(field, name)                                                         \
        field = (decltype(field))dlsym(lib, name);                                   \
        if (!field) { err = std::string("librccl lacks ") + name; lib = nullptr; return false; }
        FDC_SYM(GetUniqueId, "ncclGetUniqueId")
        FDC_SYM(CommInitRank, "ncclCommInitRank")
        FDC_SYM(CommDestroy, "ncclCommDestroy")
        FDC_SYM(AllGather, "ncclAllGather")
        FDC_SYM(AllReduce, "ncclAllReduce")
        FDC_SYM(GetErrorString, "ncclGetErrorString")
#undef FDC_SYM
        return true;
    }
};

inline RcclApi& rccl() { static RcclApi api; return api; }

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

}  // namespace fdc

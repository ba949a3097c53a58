// luma_rccl.cpp -- binds RCCL at first use (see luma_rccl.h).
#include "luma_rccl.h"

#include <dlfcn.h>

#include <cstdio>
#include <mutex>

namespace luma {

const RcclApi* rccl_api(const char** why)
{
    static std::mutex lock;
    static RcclApi api{};
    static bool tried = false, ok = false;
    static char reason[384] = "";
    std::lock_guard<std::mutex> guard(lock);
    if (!tried) {
        tried = true;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) { h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL); }
        if (!h) {
            std::snprintf(reason, sizeof reason, "dlopen(librccl.so.1): %s", dlerror());
        } else {
            bool all = true;
            auto bind = [&](auto& fn, const char* name) {
                fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name));
                if (!fn && all) { all = false; std::snprintf(reason, sizeof reason, "librccl has no symbol %s", name); }
            };
            bind(api.CommInitAll, "ncclCommInitAll");
            bind(api.CommDestroy, "ncclCommDestroy");
            bind(api.GroupStart, "ncclGroupStart");
            bind(api.GroupEnd, "ncclGroupEnd");
            bind(api.Send, "ncclSend");
            bind(api.Recv, "ncclRecv");
            bind(api.GetErrorString, "ncclGetErrorString");
            ok = all;
        }
    }
    if (why) { *why = reason; }
    return ok ? &api : nullptr;
}

}  // namespace luma

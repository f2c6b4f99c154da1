// rccl_loader.hpp -- RCCL entry points resolved at first use (dlopen), so that single-GPU users of the
// library do not load the collective library at all.  Only what the one collective of this path needs:
// communicator creation over the devices of ONE process, an all-reduce, teardown.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace bbb {

struct Rccl {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;      // why loading failed (empty = loaded)
    bool ok = false;
};

inline const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { r.error = std::string("cannot load librccl: ") + dlerror(); return; }
#define BBB_RCCL_SYM(field, sym)                                                       \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(lib, #sym));               \
        if (!r.field) { r.error = "librccl lacks " #sym; return; }
        BBB_RCCL_SYM(CommInitAll, ncclCommInitAll)
        BBB_RCCL_SYM(CommDestroy, ncclCommDestroy)
        BBB_RCCL_SYM(AllReduce, ncclAllReduce)
        BBB_RCCL_SYM(GroupStart, ncclGroupStart)
        BBB_RCCL_SYM(GroupEnd, ncclGroupEnd)
        BBB_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef BBB_RCCL_SYM
        r.ok = true;
    });
    return r;
}

}  // namespace bbb

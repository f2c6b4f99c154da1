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
    decltype(&ncclCommCount) CommCount = nullptr;
    std::string error;      // why loading failed (empty = loaded)
    std::string path;       // the file the entry points came from (dladdr)
    bool reused = false;    // the process already held it (torch's copy, say): found with RTLD_NOLOAD
    bool ok = false;
};

inline const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process that already holds an RCCL (a PyTorch process maps the copy under torch/lib) must not get a second instance
        // of the library beside it -- two RCCLs with their own bootstrap state on the same devices is a configuration nobody
        // tests: look for a loaded one first (RTLD_NOLOAD never maps anything), only then load one by name.
        static const char *const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *lib = nullptr;
        for (const char *name : names) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (lib) { r.reused = true; break; }
        }
        if (!lib) {
            for (const char *name : names) {
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
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
        BBB_RCCL_SYM(CommCount, ncclCommCount)
#undef BBB_RCCL_SYM
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(r.AllReduce), &info) && info.dli_fname) r.path = info.dli_fname;
        r.ok = true;
    });
    return r;
}

}  // namespace bbb

// The model's stream LIFETIMES, held to the failure they were added for.  Round 4's crash (commit 2fe1c7f, fixed in its successor):
// a process-wide cache of PRBS region seeds remembered the hipStream_t a plan was last used on and called hipStreamSynchronize
// on it when the plan was matched or re-targeted -- after the owner of that stream (a handle's internal one) had destroyed it.
// Vector clocks cannot see that: the ORDER was fine, the handle was dead.  Here the same shape in twenty lines -- a cache entry
// that keeps a stream, the stream destroyed, the cache used again -- must be reported by the model; the fixed shape (the entry
// keeps an EVENT recorded on the stream, and waits for that) must not.  Test infrastructure only.
#include <cstdio>
#include <string>

#include "model.hpp"

struct PlanCache {
    void *buf = nullptr;
    hipStream_t last_stream = nullptr;     // the bug: a borrowed handle outliving its owner
    hipEvent_t ready = nullptr;            // the fix: an event stands for the work, whatever becomes of the stream
};

static bool reported(const char *needle) {
    for (const std::string &e : model::errors())
        if (e.find(needle) != std::string::npos) return true;
    return false;
}

int main() {
    hipStream_t owner, user;
    hipStreamCreateWithFlags(&owner, 0);
    hipStreamCreateWithFlags(&user, 0);
    PlanCache c;
    hipMalloc(&c.buf, 4096);
    hipEventCreateWithFlags(&c.ready, hipEventDisableTiming);
    // the plan is made on the owner's stream
    model::op(owner, "seed kernel", {}, {c.buf});
    c.last_stream = owner;
    hipEventRecord(c.ready, owner);
    hipStreamSynchronize(owner);
    hipStreamDestroy(owner);               // the handle that owned the stream goes away
    // fixed shape: the next user waits for the event -- no report
    hipStreamWaitEvent(user, c.ready, 0);
    model::op(user, "generator reads the seeds", {c.buf}, {});
    if (!model::errors().empty()) { std::printf("FAIL: the event-based cache was reported: %s\n", model::errors()[0].c_str()); return 1; }
    // the bug: synchronise with the remembered stream
    hipStreamSynchronize(c.last_stream);
    if (!reported("use after hipStreamDestroy")) { std::printf("FAIL: hipStreamSynchronize on a destroyed stream was not reported\n"); return 1; }
    model::clear_errors();
    // every other use of the dead handle as well: a wait, a record, a launch, a query
    hipStreamWaitEvent(c.last_stream, c.ready, 0);
    if (!reported("use after hipStreamDestroy")) { std::printf("FAIL: hipStreamWaitEvent on a destroyed stream was not reported\n"); return 1; }
    model::clear_errors();
    hipEventRecord(c.ready, c.last_stream);
    if (!reported("use after hipStreamDestroy")) { std::printf("FAIL: hipEventRecord on a destroyed stream was not reported\n"); return 1; }
    model::clear_errors();
    model::op(c.last_stream, "a kernel", {c.buf}, {});
    if (!reported("use after hipStreamDestroy")) { std::printf("FAIL: a launch on a destroyed stream was not reported\n"); return 1; }
    model::clear_errors();
    (void)hipStreamQuery(c.last_stream);
    if (!reported("use after hipStreamDestroy")) { std::printf("FAIL: hipStreamQuery on a destroyed stream was not reported\n"); return 1; }
    model::clear_errors();
    hipFree(c.buf);
    hipEventDestroy(c.ready);
    std::printf("ok lifetimes\n");
    return 0;
}

// tests/sched_model/model.hpp -- see model.cpp
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <initializer_list>
#include <string>
#include <vector>

namespace model {

typedef std::vector<uint32_t> VC;                 // one component per stream
struct Access { VC vc; int stream = 0; std::string what; uint64_t seq = 0; };
struct Buffer {
    char *base = nullptr; size_t size = 0, maplen = 0; int id = 0; std::string tag = "?";
    bool freed = false, has_write = false;
    Access last_write;
    std::vector<Access> reads;                    // since the last write, the latest per stream
};

// an operation of the caller or of a kernel stub on stream s: it reads / writes the allocations these pointers lie in
void op(hipStream_t s, const std::string &what, std::initializer_list<const void *> reads, std::initializer_list<const void *> writes);
void host_note(const std::string &what);          // a line in the trace
void reset_trace();
void tag(const void *p, const std::string &t);    // a name for an allocation, for the reports
const std::vector<std::string> &errors();
void clear_errors();
uint64_t ops_checked();

}  // namespace model

// Run-time specialisation of the conversion kernels (jit.cpp): the plan of a BufferLayoutConverter becomes a compile-time constant of
// jit_quad.hpp's kernel, compiled with hipRTC for the device's architecture, cached in memory per plan signature and on disk per source hash.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "plan.h"

namespace pstjit {

// What the generated kernel is specialised on: everything in a ConvertPlan except addresses, the point count and the parameters of the
// transformations (scale / offset / shift / mask stay run-time values, so one kernel serves every LAS header).
struct QuadSpec {
  bool src_aos = false, dst_aos = false;
  uint32_t src_stride = 0, dst_stride = 0, covered = 0;
  int blk = 256;
  uint32_t xcd = 0, nt = 1, src_words = 0, lds_per_point = 0, dst_tile_off = 0, alias = 0;
  std::vector<pstq::QEntry> entries;
  std::vector<std::string> exprs;  // empty, or one text per entry: the device expression of a PST_XF_EXPR entry ("" for the others), already validated (expr.cpp)
  uint32_t tile() const { return 4u * (uint32_t)blk; }
  uint32_t lds_bytes() const;
};

enum class Mode { Off, Async, Sync };
Mode mode();               // PST_JIT = 0 | async (default) | sync
void set_mode(int m);      // -1: the environment's setting again; 0 / 1 / 2 = Off / Async / Sync
uint64_t min_points();     // PST_JIT_MIN_POINTS: calls below it never trigger a compilation (default 2^20)

// Can this launch take a specialised kernel at all?  (LDS tile path, interleaved sides 16-byte aligned, records that fit the register
// images, at least one full tile.)  Fills `spec` when it can.
bool spec_from_plan(const ConvertPlan& plan, bool src_aos, bool dst_aos, QuadSpec* spec);

// The translation unit hipRTC compiles for `spec` (also the cache key).
std::string spec_source(const QuadSpec& spec);

struct Kernel {
  hipFunction_t fn = nullptr;
  unsigned blk = 256;
  uint32_t lds_bytes = 0, tile = 1024;
};
// Ready kernel for `spec` on the calling thread's current device, or false.  IfReady: never starts a compilation; Enqueue: a missing kernel
// is queued for the compiler thread and the call returns false at once (the caller interprets the plan meanwhile); Wait: compiles in the
// calling thread (or waits for the compiler thread if it already has this plan).
enum class Acquire { IfReady, Enqueue, Wait };
bool acquire(const QuadSpec& spec, const std::string& source, Acquire how, Kernel* out, std::string* error = nullptr);  // source = spec_source(spec)

// The same cache for any other translation unit over the embedded device headers (filter_stream.hpp's compaction kernels): `entry` is the
// extern "C" kernel of `source`; blk / lds_bytes / tile are handed back with the kernel.
bool acquire_source(const std::string& source, const char* entry, unsigned blk, uint32_t lds_bytes, uint32_t tile, Acquire how, Kernel* out,
                    std::string* error = nullptr);

// Compile without touching a device (CPU test of the generator and of the headers under hipRTC): code object bytes, or empty + error.
std::vector<char> compile_source(const std::string& source, const std::string& arch, std::string* error);

struct Stats { uint64_t compiled = 0, disk_hits = 0, memory_hits = 0, failures = 0, launches = 0; double compile_seconds = 0; };
Stats stats();

}  // namespace pstjit

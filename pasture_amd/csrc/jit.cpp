// Run-time specialisation of the layout-conversion kernels.
//
// The reference's converter is layout-generic (buffer_conversion.rs:112-234): any pair of PointLayouts, any mapping list.  The
// interpreted tile kernels (convert_kernels.hpp) serve every plan at once and pay for it -- offsets, sizes and type pairs come out of
// registers.  Here the plan becomes source text: `spec_source` writes the mapping list as a constexpr plan type for jit_quad.hpp's
// kernel, hipRTC compiles it for the device's architecture, and the code object is cached in memory (per plan signature, per device)
// and on disk (per hash of source + headers + options; PST_JIT_CACHE_DIR, default ~/.cache/pasture_amd/jit).  Compilation runs on a
// background thread (PST_JIT=async, the default): the first calls of a new plan are interpreted, later ones take the specialised
// kernel; PST_JIT=sync compiles in the calling thread, PST_JIT=0 switches the whole thing off.  hipRTC is bound with dlopen at first
// use, so the library loads (and interprets) where it is absent.
#include "jit.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <unordered_map>

#include "jit_embedded.inc"  // kJitHeaderNames / kJitHeaderTexts / kJitHeaderCount: the device headers as text (tools/embed_headers.py)

namespace pstjit {

namespace {

// ---- hipRTC, bound at first use ------------------------------------------------------------------------------------------------
struct Rtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  std::string error;
};
const Rtc& rtc() {
  static const Rtc r = [] {
    Rtc x;
    // hipRTC, comgr and the HIP runtime must come from ONE distribution: a process that imported torch runs on torch's bundled runtime
    // (pasture_amd/_capi.py), so look next to the runtime that is actually loaded first
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir = info.dli_fname;
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash);
        names.push_back(dir + "/libhiprtc.so");
        names.push_back(dir + "/libhiprtc.so.7");
      }
    }
    for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) names.push_back(n);
    for (const std::string& name : names) {
      x.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (x.lib) break;
    }
    if (!x.lib) { x.error = std::string("hipRTC not loadable: ") + dlerror(); return x; }
#define PST_RTC_SYM(field, sym)                                                                   \
  x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.lib, #sym));                              \
  if (!x.field) { x.error = "hipRTC lacks " #sym; return x; }
    PST_RTC_SYM(create, hiprtcCreateProgram)
    PST_RTC_SYM(compile, hiprtcCompileProgram)
    PST_RTC_SYM(log_size, hiprtcGetProgramLogSize)
    PST_RTC_SYM(log, hiprtcGetProgramLog)
    PST_RTC_SYM(code_size, hiprtcGetCodeSize)
    PST_RTC_SYM(code, hiprtcGetCode)
    PST_RTC_SYM(destroy, hiprtcDestroyProgram)
    PST_RTC_SYM(version, hiprtcVersion)
#undef PST_RTC_SYM
    return x;
  }();
  return r;
}

uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
std::string hash_hex(const std::string& s) {
  const uint64_t a = fnv1a(s.data(), s.size(), 0xcbf29ce484222325ull), b = fnv1a(s.data(), s.size(), 0x9ae16a3b2f90404full);
  char buf[40];
  snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
  return buf;
}

const std::vector<const char*>& compile_options(const std::string& arch, std::vector<std::string>& keep) {
  static thread_local std::vector<const char*> out;
  keep.clear();
  keep.push_back("--offload-arch=" + arch);
  keep.push_back("-O3");
  keep.push_back("-std=c++17");
  keep.push_back("-ffp-contract=off");  // (p * scale) + offset keeps its two roundings (raw_readers.rs:42-48), as in the in-tree build
  out.clear();
  for (const std::string& s : keep) out.push_back(s.c_str());
  return out;
}

// hash of everything that decides the code object besides the plan source
const std::string& toolchain_salt() {
  static const std::string s = [] {
    std::string t;
    for (int i = 0; i < kJitHeaderCount; ++i) { t += kJitHeaderNames[i]; t += '\0'; t += kJitHeaderTexts[i]; t += '\0'; }
    int major = 0, minor = 0;
    if (rtc().version) rtc().version(&major, &minor);
    t += "hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " -O3 -std=c++17 -ffp-contract=off";
    return hash_hex(t);
  }();
  return s;
}

std::string cache_dir() {
  static const std::string d = [] {
    const char* off = std::getenv("PST_JIT_CACHE");
    if (off && *off == '0') return std::string();
    std::string dir;
    if (const char* v = std::getenv("PST_JIT_CACHE_DIR")) dir = v;
    else if (const char* x = std::getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/pasture_amd/jit";
    else if (const char* h = std::getenv("HOME")) dir = std::string(h) + "/.cache/pasture_amd/jit";
    if (dir.empty()) return dir;
    std::string acc;
    for (size_t i = 0; i <= dir.size(); ++i) {  // mkdir -p
      if (i == dir.size() || dir[i] == '/') {
        if (!acc.empty()) (void)mkdir(acc.c_str(), 0755);
      }
      if (i < dir.size()) acc += dir[i];
    }
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return std::string();
    return dir;
  }();
  return d;
}

std::vector<char> read_file(const std::string& path) {
  std::vector<char> out;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return out;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n > 0) {
    out.resize((size_t)n);
    if (fread(out.data(), 1, (size_t)n, f) != (size_t)n) out.clear();
  }
  fclose(f);
  return out;
}
void write_file_atomic(const std::string& path, const std::vector<char>& data) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
  fclose(f);
  if (ok) (void)rename(tmp.c_str(), path.c_str());
  else (void)unlink(tmp.c_str());
}

// A cache file is the code object inside an envelope: magic, the 32-character key it was stored under (= hash of toolchain, architecture and
// source text: a file copied or renamed over another one is not taken), the image's length and its 128-bit FNV checksum.  hipModuleLoadData is given
// a bare pointer -- a truncated or overwritten file must be recognised HERE, before the loader walks an ELF header that promises more bytes than the
// buffer holds.  (tools/exp_jit_cache_damage.py: a truncated file made the call fail, a valid image of ANOTHER plan under this plan's name faulted the GPU.)
constexpr char kEnvelopeMagic[8] = {'P', 'S', 'T', 'J', 'I', 'T', '2', '\n'};
constexpr size_t kEnvelopeHead = 8 + 32 + 8 + 32;
std::vector<char> seal_envelope(const std::vector<char>& image, const std::string& key) {
  std::vector<char> out(kEnvelopeHead + image.size());
  const uint64_t len = image.size();
  const std::string sum = hash_hex(std::string(image.data(), image.size()));
  memcpy(out.data(), kEnvelopeMagic, 8);
  memcpy(out.data() + 8, key.data(), 32);
  memcpy(out.data() + 40, &len, 8);
  memcpy(out.data() + 48, sum.data(), 32);
  memcpy(out.data() + kEnvelopeHead, image.data(), image.size());
  return out;
}
std::vector<char> open_envelope(const std::vector<char>& file, const std::string& key) {
  std::vector<char> image;
  uint64_t len = 0;
  if (file.size() <= kEnvelopeHead || key.size() != 32 || memcmp(file.data(), kEnvelopeMagic, 8) != 0 || memcmp(file.data() + 8, key.data(), 32) != 0) return image;
  memcpy(&len, file.data() + 40, 8);
  if (len != file.size() - kEnvelopeHead) return image;
  const std::string sum = hash_hex(std::string(file.data() + kEnvelopeHead, (size_t)len));
  if (memcmp(file.data() + 48, sum.data(), 32) != 0) return image;
  image.assign(file.begin() + (long)kEnvelopeHead, file.end());
  return image;
}

// The architecture name of the calling thread's current device.  Asked on the launch path of every plan-specialised conversion and filter, so the
// answer is kept per device id (hipGetDeviceProperties fills a multi-kilobyte struct each time).
const std::string& device_arch() {
  static const std::string fallback = "gfx950";
  constexpr int kMaxDevices = 64;
  static std::string names[kMaxDevices];
  static std::atomic<bool> known[kMaxDevices];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { (void)hipGetLastError(); return fallback; }
  if (known[dev].load(std::memory_order_acquire)) return names[dev];
  std::lock_guard<std::mutex> lock(mu);
  if (!known[dev].load(std::memory_order_relaxed)) {
    hipDeviceProp_t prop;
    names[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) ? prop.gcnArchName : fallback;
    known[dev].store(true, std::memory_order_release);
  }
  return names[dev];
}

// ---- the cache --------------------------------------------------------------------------------------------------------------------
struct Entry {
  enum State { Queued, Ready, Failed } state = Queued;
  std::string source, error, entry_name = "pst_jit_convert";
  std::string arch;       // the architecture of the device the REQUESTING thread was on (the compiler thread's current device is another matter)
  std::string disk_path;  // where the code object was read from (a stale or damaged file is deleted and compiled again, once)
  bool retried = false;
  std::vector<char> code;
  std::map<int, std::pair<hipModule_t, hipFunction_t>> per_device;
  unsigned blk = 256;
  uint32_t lds_bytes = 0, tile = 1024;
};

struct Cache {
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::string, std::map<std::string, std::shared_ptr<Entry>>> by_source;  // [architecture][source text]: looked up without building a joined key per launch
  std::deque<std::shared_ptr<Entry>> queue;
  bool worker_started = false;
  int compiling = 0;       // compilations in flight (the exit handler waits for them: hipRTC's libraries must not be torn down under a compile)
  bool stopping = false;   // set at exit: the worker takes nothing more from the queue
  Stats st;
};
Cache& cache() {
  static Cache* c = new Cache;  // never destroyed: the compiler thread may outlive static destruction
  return *c;
}

void compile_entry(const std::shared_ptr<Entry>& e) {
  Cache& c = cache();
  const std::string arch = e->arch.empty() ? device_arch() : e->arch;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    c.compiling++;
  }
  std::string err;
  std::vector<char> code;
  bool from_disk = false;
  const std::string dir = cache_dir();
  std::string path;
  if (!dir.empty()) {
    const std::string key = hash_hex(toolchain_salt() + arch + e->source);
    path = dir + "/" + key + ".pstco";
    if (!e->retried) {
      code = open_envelope(read_file(path), key);
      if (code.empty()) (void)unlink(path.c_str());  // absent, or not what this library wrote under this name: compiled again below
    }
    from_disk = !code.empty();
    if (!code.empty()) code.push_back(0);  // (hipModuleLoadData takes no length: a terminator behind the image costs nothing)
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (code.empty()) {
    code = compile_source(e->source, arch, &err);
    if (!code.empty() && !path.empty()) write_file_atomic(path, seal_envelope(code, hash_hex(toolchain_salt() + arch + e->source)));
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::lock_guard<std::mutex> lock(c.mu);
  c.compiling--;
  e->disk_path = from_disk ? path : std::string();
  if (code.empty()) {
    e->state = Entry::Failed;
    e->error = err;
    c.st.failures++;
    if (std::getenv("PST_JIT_DEBUG")) fprintf(stderr, "[pst jit] compilation failed:\n%s\n", err.c_str());
  } else {
    e->code = std::move(code);
    e->state = Entry::Ready;
    if (from_disk) c.st.disk_hits++;
    else { c.st.compiled++; c.st.compile_seconds += secs; }
    if (std::getenv("PST_JIT_DEBUG")) fprintf(stderr, "[pst jit] %s in %.2f s (%zu bytes)\n", from_disk ? "loaded from disk" : "compiled", secs, e->code.size());
  }
  c.cv.notify_all();
}

void worker_main() {
  Cache& c = cache();
  for (;;) {
    std::shared_ptr<Entry> e;
    {
      std::unique_lock<std::mutex> lock(c.mu);
      c.cv.wait(lock, [&] { return !c.queue.empty() || c.stopping; });
      if (c.stopping) return;
      e = c.queue.front();
      c.queue.pop_front();
    }
    compile_entry(e);
  }
}

}  // namespace

static std::atomic<int> g_mode_override{-1};
void set_mode(int m) { g_mode_override.store(m); }
Mode mode() {
  const int o = g_mode_override.load(std::memory_order_relaxed);
  if (o >= 0) return o == 0 ? Mode::Off : o == 2 ? Mode::Sync : Mode::Async;
  static const Mode m = [] {
    const char* v = std::getenv("PST_JIT");
    if (!v || !*v) return Mode::Async;
    if (*v == '0' || !strcmp(v, "off")) return Mode::Off;
    if (!strcmp(v, "sync")) return Mode::Sync;
    return Mode::Async;
  }();
  return m;
}
uint64_t min_points() {
  static const uint64_t n = [] {
    const char* v = std::getenv("PST_JIT_MIN_POINTS");
    return v && *v ? (uint64_t)std::strtoull(v, nullptr, 10) : (uint64_t)1 << 20;
  }();
  return n;
}

uint32_t QuadSpec::lds_bytes() const {
  const uint32_t b = tile() * lds_per_point;
  return b < 256u ? 256u : b;
}

static long env_long(const char* name, long dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::strtol(v, nullptr, 10) : dflt;
}

bool spec_from_plan(const ConvertPlan& plan, bool src_aos, bool dst_aos, QuadSpec* spec) {
  const ConvertHeader& h = plan.h;
  if (!(src_aos || dst_aos) || h.in_place || h.n_entries == 0 || h.n_entries > PST_PLAN_MAX_ENTRIES) return false;
  if (src_aos && (h.src_aos & 15u)) return false;  // record tiles travel as 16-byte chunks
  if (dst_aos && (h.dst_aos & 15u)) return false;
  // the lane holds four records of either side in registers
  static const long max_words = env_long("PST_JIT_MAX_WORDS", 224);
  QuadSpec s;
  s.src_aos = src_aos; s.dst_aos = dst_aos;
  s.src_stride = src_aos ? h.src_stride : 0; s.dst_stride = dst_aos ? h.dst_stride : 0;
  s.covered = dst_aos ? (h.dst_fully_covered ? 1u : 0u) : 0u;
  if ((src_aos && h.src_stride == 0) || (dst_aos && h.dst_stride == 0)) return false;
  uint32_t src_words = 0;
  // LDS bytes per point.  Incoming regions: the source record tile, the staged wide source columns (and the target record tile when its
  // records are read-modify-written); outgoing regions: the target record tile, the staged wide target columns.  With `alias` the outgoing
  // regions overlay the incoming ones (one more barrier, half the LDS: twice the workgroups per CU).
  static const long alias_env = env_long("PST_JIT_ALIAS", 0);
  s.alias = (alias_env != 0 && !(dst_aos && !s.covered)) ? 1u : 0u;
  uint32_t in_stage = s.src_stride, out_stage = 0;
  if (s.alias) { s.dst_tile_off = 0; out_stage = s.dst_stride; }
  else { s.dst_tile_off = s.src_stride; in_stage += s.dst_stride; }
  static const long wide_min = env_long("PST_JIT_WIDE_MIN", 8);
  auto wide = [&](uint32_t size, uint64_t col) { return (long)size >= wide_min && size % 4u == 0 && (col & 15u) == 0; };
  for (uint32_t m = 0; m < h.n_entries; ++m) {
    const PlanEntry& e = plan.e[m];
    if (e.src_size == 0 || e.dst_size == 0 || e.ncomp == 0) return false;
    static const uint32_t ct_size[10] = {1, 1, 2, 2, 4, 4, 8, 8, 4, 8};
    if (e.src_ct > 9 || e.dst_ct > 9 || ct_size[e.src_ct] * e.ncomp != e.src_size || ct_size[e.dst_ct] * e.ncomp != e.dst_size) return false;
    if (!e.convert && e.src_ct != e.dst_ct) return false;
    if (src_aos && e.src_off + e.src_size > h.src_stride) return false;
    if (dst_aos && e.dst_off + e.dst_size > h.dst_stride) return false;
    pstq::QEntry q{};
    q.src_off = src_aos ? e.src_off : 0; q.dst_off = dst_aos ? e.dst_off : 0;
    q.src_size = e.src_size; q.dst_size = e.dst_size; q.ncomp = e.ncomp;
    q.src_ct = e.src_ct; q.dst_ct = e.dst_ct; q.convert = e.convert ? 1u : 0u;
    q.xf_kind = e.xf_kind; q.xf_pre = e.xf_kind ? (e.xf_on_source ? 1u : 0u) : 0u;
    if (e.xf_kind == PST_XF_EXPR) {  // a device expression: its text becomes part of the translation unit (PlanEntry::mask = its index in the plan's table)
      const std::vector<std::string>* texts = (const std::vector<std::string>*)plan.expr_texts;
      if (!texts || e.mask >= texts->size() || (*texts)[(size_t)e.mask].empty() || !(e.ncomp == 1 || e.ncomp == 3) || e.bounds) return false;
      if (s.exprs.empty()) s.exprs.resize(h.n_entries);
      s.exprs[m] = (*texts)[(size_t)e.mask];
    }
    q.bounds = (e.bounds && h.bounds_partials && e.dst_ct == 9 /*F64*/ && e.ncomp == 3) ? 1u : 0u;
    if (e.bounds && h.bounds_partials && !q.bounds) return false;
    if (!src_aos) {  // one image per distinct source column (one source -> many targets: the bit fields of raw_readers.rs:61-164)
      q.src_load = 1;
      for (uint32_t k = 0; k < m; ++k)
        if (plan.e[k].src_col == e.src_col && plan.e[k].src_size == e.src_size) { q.src_load = 0; q.src_img = s.entries[k].src_img; break; }
      if (q.src_load) {
        q.src_img = src_words;
        src_words += e.src_size;
        if (wide(e.src_size, e.src_col)) { q.src_wide = 1; q.src_stage = in_stage; in_stage += e.src_size; }
      }
    }
    if (!dst_aos && wide(e.dst_size, e.dst_col)) {
      q.dst_wide = 1;
      if (s.alias) { q.dst_stage = out_stage; out_stage += e.dst_size; }
      else { q.dst_stage = in_stage; in_stage += e.dst_size; }
    }
    s.entries.push_back(q);
  }
  s.src_words = src_aos ? 0 : src_words;
  s.lds_per_point = std::max(in_stage, out_stage);
  const uint32_t sw = src_aos ? h.src_stride : src_words, dw = dst_aos ? h.dst_stride : 0;
  if ((long)(sw + dw) > max_words) return false;
  // tile = 4 x lanes points.  ONE WAVE per workgroup (256 points) measured best on every pairing (10^8 points, eight random layouts x three
  // pairings: 64 lanes 0.74-0.80 of peak, 128 lanes 0.51-0.80, 256 lanes 0.26-0.77): the three phases of a tile -- request, shuffle, store --
  // need no barrier inside one wave, a CU holds 4-12 tiles in different phases instead of 1-3, and a tile of records is 6-25 KiB of LDS.
  static const long blk_env = env_long("PST_JIT_BLK", 64);
  s.blk = (blk_env == 64 || blk_env == 128 || blk_env == 256 || blk_env == 512) ? (int)blk_env : 64;
  if (s.lds_bytes() > 160u * 1024u - 1024u) return false;
  if (h.n < s.tile()) return false;
  // XCD-aware tile numbering (every XCD works on one contiguous eighth of the range): with one small tile per workgroup it pays on all
  // three pairings (same run as above: means 0.726 / 0.726 -> 0.770 / 0.767 for columns -> records and records -> records, 0.763 -> 0.774 for
  // records -> columns)
  static const long xcd_env = env_long("PST_JIT_XCD", 1);
  s.xcd = xcd_env != 0 ? 1u : 0u;
  // non-temporal accesses: bit 0 narrow column loads, 1 narrow column stores, 2 tile stores (LDS -> HBM), 3 tile loads (LDS-DMA).  Same-box
  // sweep (12 random layouts x 3 pairings, 10^8 points, means): all four 0.810 / 0.785 / 0.806 (records -> columns / columns -> records /
  // records -> records); without the DMA bit 0.776 / 0.775 / 0.779; none 0.770 / 0.752 / 0.772: a tile is read once, by one wave.
  static const long nt_env = env_long("PST_JIT_NT", -1);
  s.nt = nt_env >= 0 ? (uint32_t)nt_env & 15u : 15u;
  *spec = std::move(s);
  return true;
}

// expr.cpp: the component expressions of a transformation text (one, or one per component separated by top-level ';')
}  // namespace pstjit
namespace pstexpr { std::vector<std::string> split_expression_components(const std::string& expr); }
namespace pstjit {
using pstexpr::split_expression_components;
static std::string spec_source_uncached(const QuadSpec& s);
// The text is a pure function of the spec, and a converter hands the same plan to every call: a small per-thread memo keyed by the spec's
// bytes saves the ~15 us of formatting per conversion call (two of them since converter.cpp asks whether a kernel is at hand before it
// prefers the generic path) -- a multiple of the launch itself for the 1 MiB chunks pasture-io's readers convert.
std::string spec_source(const QuadSpec& s) {
  std::string key;
  key.reserve(64 + s.entries.size() * sizeof(pstq::QEntry));
  auto put = [&](const void* p, size_t n) { key.append((const char*)p, n); };
  const uint32_t head[] = {s.src_aos ? 1u : 0u, s.dst_aos ? 1u : 0u, s.src_stride, s.dst_stride, s.covered, (uint32_t)s.blk, s.xcd, s.nt, s.src_words, s.lds_per_point,
                           s.dst_tile_off, s.alias, (uint32_t)s.entries.size()};
  put(head, sizeof(head));
  for (const pstq::QEntry& e : s.entries) {  // (field by field: the struct's padding bytes are not part of the key)
    const uint32_t f[] = {e.src_off, e.dst_off, e.src_size, e.dst_size, e.ncomp, e.src_ct, e.dst_ct, e.convert, e.xf_kind, e.xf_pre, e.bounds, e.src_load, e.src_img, e.src_wide,
                          e.src_stage, e.dst_wide, e.dst_stage};
    put(f, sizeof(f));
  }
  for (const std::string& t : s.exprs) { const uint32_t len = (uint32_t)t.size(); put(&len, sizeof(len)); put(t.data(), t.size()); }
  thread_local std::unordered_map<std::string, std::string> memo;
  auto it = memo.find(key);
  if (it != memo.end()) return it->second;
  if (memo.size() >= 256) memo.clear();
  return memo.emplace(std::move(key), spec_source_uncached(s)).first->second;
}
static std::string spec_source_uncached(const QuadSpec& s) {
  std::ostringstream o;
  o << "#include \"jit_quad.hpp\"\n";
  o << "struct PstJitPlan {\n";
  o << "  static constexpr int n = " << s.entries.size() << ";\n";
  o << "  static constexpr bool src_aos = " << (s.src_aos ? "true" : "false") << ", dst_aos = " << (s.dst_aos ? "true" : "false") << ";\n";
  o << "  static constexpr uint32_t src_stride = " << s.src_stride << ", dst_stride = " << s.dst_stride << ", covered = " << s.covered << ";\n";
  o << "  static constexpr int blk = " << s.blk << ";\n";
  o << "  static constexpr uint32_t xcd = " << s.xcd << ", nt = " << s.nt << ", src_words = " << (s.src_words ? s.src_words : 1u) << ", lds_per_point = " << s.lds_per_point
    << ", dst_tile_off = " << s.dst_tile_off << ", alias = " << s.alias << ";\n";
  o << "  __host__ __device__ static constexpr pstq::QEntry entry(int m) {\n";
  o << "    constexpr pstq::QEntry t[n] = {\n";
  for (const pstq::QEntry& e : s.entries)
    o << "      {" << e.src_off << ", " << e.dst_off << ", " << e.src_size << ", " << e.dst_size << ", " << e.ncomp << ", " << e.src_ct << ", " << e.dst_ct << ", "
      << e.convert << ", " << e.xf_kind << ", " << e.xf_pre << ", " << e.bounds << ", " << e.src_img << ", " << e.src_load << ", " << e.src_wide << ", " << e.dst_wide
      << ", " << e.src_stage << ", " << e.dst_stage << "},\n";
  o << "    };\n    return t[m];\n  }\n";
  if (!s.exprs.empty()) {
    // the fused expressions (expr.cpp's names: v, x, y, z, c, i, p0 .. p3 -- the arrays travel in ConvertHeader::expr_params; a conversion captures none): one `if constexpr`
    // arm per (mapping, component); a Vec3 mapping may give one text for all components or three separated by ';' (split by expr.cpp)
    o << "  template <int M, int C, typename TI>\n  __device__ static __forceinline__ TI expr(const TI v, const TI x, const TI y, const TI z, const uint64_t i, const uint64_t (&pp)[4]) {\n";
    o << "    using namespace pstd;\n    constexpr int c = C;\n    const double* const p0 = (const double*)pp[0]; const double* const p1 = (const double*)pp[1]; const double* const p2 = (const double*)pp[2]; const double* const p3 = (const double*)pp[3];\n";
    o << "    (void)v; (void)x; (void)y; (void)z; (void)c; (void)i; (void)p0; (void)p1; (void)p2; (void)p3;\n";
    for (size_t m = 0; m < s.exprs.size(); ++m) {
      if (s.exprs[m].empty()) continue;
      const std::vector<std::string> comps = split_expression_components(s.exprs[m]);
      for (uint32_t cc = 0; cc < s.entries[m].ncomp; ++cc)
        o << "    if constexpr (M == " << m << " && C == " << cc << ") return rust_as<TI>(\n" << comps[comps.size() == 1 ? 0 : cc] << "\n    );\n";
    }
    o << "    return v;\n  }\n";
  }
  o << "};\n";
  o << "extern \"C\" __global__ __launch_bounds__(" << s.blk << ") void pst_jit_convert(const ConvertHeader h, const PlanEntry* __restrict__ entries) {\n";
  o << "  pstq::quad_convert_body<PstJitPlan>(h, entries);\n}\n";
  return o.str();
}

std::vector<char> compile_source(const std::string& source, const std::string& arch, std::string* error) {
  std::vector<char> code;
  const Rtc& r = rtc();
  if (!r.error.empty()) { if (error) *error = r.error; return code; }
  hiprtcProgram prog = nullptr;
  if (r.create(&prog, source.c_str(), "pst_jit_plan.hip", kJitHeaderCount, kJitHeaderTexts, kJitHeaderNames) != HIPRTC_SUCCESS) {
    if (error) *error = "hiprtcCreateProgram failed";
    return code;
  }
  std::vector<std::string> keep;
  const std::vector<const char*>& opts = compile_options(arch, keep);
  const hiprtcResult res = r.compile(prog, (int)opts.size(), const_cast<const char**>(opts.data()));
  if (res != HIPRTC_SUCCESS) {
    size_t n = 0;
    r.log_size(prog, &n);
    std::string log(n, '\0');
    if (n) r.log(prog, &log[0]);
    if (error) *error = "hipRTC compilation failed (" + std::to_string((int)res) + "):\n" + log;
  } else {
    size_t n = 0;
    if (r.code_size(prog, &n) == HIPRTC_SUCCESS && n) {
      code.resize(n);
      if (r.code(prog, code.data()) != HIPRTC_SUCCESS) code.clear();
    }
    if (code.empty() && error) *error = "hipRTC returned no code object";
  }
  r.destroy(&prog);
  return code;
}

bool acquire(const QuadSpec& spec, const std::string& src, Acquire how, Kernel* out, std::string* error) {
  return acquire_source(src, "pst_jit_convert", (unsigned)spec.blk, spec.lds_bytes(), spec.tile(), how, out, error);
}

// process exit: stop the compiler thread at its next wake-up and give a compilation in flight time to finish -- hipRTC's libraries (comgr,
// LLVM) run static destructors at exit, and a detached thread inside hiprtcCompileProgram at that moment races them
static void drain_at_exit() {
  Cache& c = cache();
  std::unique_lock<std::mutex> lock(c.mu);
  c.stopping = true;
  c.queue.clear();
  c.cv.notify_all();
  c.cv.wait_for(lock, std::chrono::seconds(30), [&] { return c.compiling == 0; });
}

bool acquire_source(const std::string& src, const char* entry, unsigned blk, uint32_t lds_bytes, uint32_t tile, Acquire how, Kernel* out, std::string* error) {
  const bool wait = how == Acquire::Wait;
  Cache& c = cache();
  std::shared_ptr<Entry> e;
  bool compile_here = false;
  // entries are per (architecture, source): the requesting thread's device decides the architecture, not the compiler thread's
  const std::string& arch = device_arch();
  {
    std::unique_lock<std::mutex> lock(c.mu);
    static bool at_exit = false;
    // exit handlers run last-registered-first: hipRTC (and the comgr / LLVM it links) must be LOADED before drain_at_exit is registered, or their
    // static destructors would run before the drain -- under a compilation still in flight on the compiler thread (seen in round 6: a process that
    // ended 50 ms after it queued a plan dumped core at exit)
    if (!at_exit) { (void)rtc(); at_exit = true; std::atexit(drain_at_exit); }
    auto& of_arch = c.by_source[arch];
    auto it = of_arch.find(src);
    if (it == of_arch.end()) {
      if (how == Acquire::IfReady) return false;
      e = std::make_shared<Entry>();
      e->arch = arch;
      e->source = src;
      e->entry_name = entry;
      e->blk = blk;
      e->lds_bytes = lds_bytes;
      e->tile = tile;
      of_arch.emplace(src, e);
      if (wait) {
        compile_here = true;
      } else {
        c.queue.push_back(e);
        if (!c.worker_started) {
          c.worker_started = true;
          std::thread(worker_main).detach();
        }
        c.cv.notify_all();
        return false;
      }
    } else {
      e = it->second;
      if (e->state == Entry::Queued) {
        if (!wait) return false;
        // queued for the background thread (or being compiled by another caller): wait for it
        bool still_queued = false;
        for (auto q = c.queue.begin(); q != c.queue.end(); ++q)
          if (*q == e) { c.queue.erase(q); still_queued = true; break; }
        if (still_queued) compile_here = true;
        else c.cv.wait(lock, [&] { return e->state != Entry::Queued; });
      }
    }
  }
  if (compile_here) compile_entry(e);
  std::unique_lock<std::mutex> lock(c.mu);
  if (e->state != Entry::Ready) {
    if (error) *error = e->error;
    return false;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { if (error) *error = "no current device"; return false; }
  auto pd = e->per_device.find(dev);
  if (pd == e->per_device.end()) {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t err = hipModuleLoadData(&mod, e->code.data());
    if (err == hipSuccess) err = hipModuleGetFunction(&fn, mod, e->entry_name.c_str());
    if (err != hipSuccess) {
      (void)hipGetLastError();
      if (!e->disk_path.empty() && !e->retried) {
        // a code object from the disk cache that does not load (damaged file, another toolchain's leftovers): delete it and compile again, once
        (void)unlink(e->disk_path.c_str());
        e->retried = true;
        e->state = Entry::Queued;
        e->code.clear();
        const std::shared_ptr<Entry> again = e;
        lock.unlock();
        compile_entry(again);
        lock.lock();
        if (e->state == Entry::Ready) {
          mod = nullptr; fn = nullptr;
          err = hipModuleLoadData(&mod, e->code.data());
          if (err == hipSuccess) err = hipModuleGetFunction(&fn, mod, e->entry_name.c_str());
        }
      }
    }
    if (err != hipSuccess) {
      (void)hipGetLastError();
      e->state = Entry::Failed;
      e->error = std::string("loading the compiled plan failed: ") + hipGetErrorString(err);
      c.st.failures++;
      if (error) *error = e->error;
      return false;
    }
    if (e->lds_bytes > 64u * 1024u) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
    (void)hipGetLastError();
    pd = e->per_device.emplace(dev, std::make_pair(mod, fn)).first;
  } else {
    c.st.memory_hits++;
  }
  out->fn = pd->second.second;
  out->blk = e->blk;
  out->lds_bytes = e->lds_bytes;
  out->tile = e->tile;
  c.st.launches++;
  return true;
}

Stats stats() {
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  return c.st;
}

}  // namespace pstjit

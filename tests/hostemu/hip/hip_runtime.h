// hip/hip_runtime.h - HOST stand-in used ONLY by tests/hostemu (TEST INFRASTRUCTURE, never shipped).
//
// Purpose: compile the library's own device sources (safe_adaptation_gym_amd/csrc/*.hpp, sag_api.hip)
// unchanged for x86-64 so that AddressSanitizer / UndefinedBehaviorSanitizer / pattern-initialised
// locals can look at the kernels' indexing (GPU sanitizers are not available on this pool).  It is a
// checker of the product's source, not a CPU fallback: the product loader (safe_adaptation_gym_amd/
// _native.py) only ever opens libsag.so, and hipGetDeviceCount() below reports 0 devices unless
// SAG_HOSTEMU=1 is set by the test harness.
//
// Execution model: a launch runs its workgroups one after the other; the threads of a workgroup are
// FIBERS on one OS thread (hand-rolled x86-64 context switch), scheduled round-robin and switched only
// inside the cross-lane operations:
//   __syncthreads                         rendezvous of the workgroup's live threads
//   __ballot, __shfl* (width 64)          rendezvous of the 64-thread wavefront
//   __shfl* with width 32                 rendezvous of the 32-lane half (the cooperative Doggo kernel runs
//                                         two envs per wavefront whose halves diverge)
// A thread that returns from the kernel leaves its groups.  If every live fiber waits and no group is
// complete the launch aborts ("divergent barrier").  `__shared__` is `static` (one workgroup at a time),
// "device memory" is malloc'ed (so ASan checks global-memory indexing too), streams are synchronous.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <chrono>
#include <functional>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HOSTEMU_ASAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__)
#define HOSTEMU_ASAN 1
#endif
#ifdef HOSTEMU_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif

// ---- qualifiers ---------------------------------------------------------------------------------
#define __host__
#define __device__ __attribute__((weak))   // device functions are defined in headers shared by two translation units
#define __global__
#define __constant__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define address_space(n) unused   // __attribute__((address_space(3))) -> __attribute__((unused))
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hostemu::g_blk.dyn_smem);

// ---- vector types ---------------------------------------------------------------------------------
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static inline float4 make_float4(float a, float b, float c, float d) { float4 v; v.x = a; v.y = b; v.z = c; v.w = d; return v; }
static inline int4 make_int4(int a, int b, int c, int d) { int4 v; v.x = a; v.y = b; v.z = c; v.w = d; return v; }

// ---- the fiber scheduler ------------------------------------------------------------------------
namespace hostemu {

struct Rendezvous { int arrived = 0; unsigned gen = 0; };

struct Lane {
  void* sp = nullptr;          // saved stack pointer of a suspended fiber
  char* stack = nullptr;
  size_t stack_size = 0;
  int tid = 0;
  bool done = false, started = false;
  void* fake_stack = nullptr;  // ASan
};

struct Block {
  std::vector<Lane> lanes;
  int nthreads = 0;
  uint3 block_idx{0, 0, 0};
  dim3 block_dim, grid_dim;
  Rendezvous rv_block, rv_wave[16], rv_half[32];
  int alive_block = 0, alive_wave[16] = {}, alive_half[32] = {};
  uint64_t xchg[1024];
  unsigned long progress = 0;  // bumped whenever a rendezvous completes or a lane finishes
  const std::function<void()>* body = nullptr;
  void* dyn_smem = nullptr;    // dynamic LDS of the launch (malloc'ed: ASan checks its bounds)
};

inline Block g_blk;
inline Lane* g_cur = nullptr;
inline void* g_sched_sp = nullptr;     // the scheduler's context while a fiber runs
inline void* g_sched_fake = nullptr;
inline const void* g_main_bottom = nullptr;
inline size_t g_main_size = 0;

constexpr size_t STACK_BYTES = 1u << 20;

// switch stacks: save callee-saved registers on the current stack, store sp to *from, load sp `to`
extern "C" void hostemu_switch(void** from, void* to);
asm(R"(
.text
.weak hostemu_switch
.type hostemu_switch,@function
hostemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hostemu_switch,.-hostemu_switch
)");

inline void to_scheduler() {
  Lane* me = g_cur;
#ifdef HOSTEMU_ASAN
  __sanitizer_start_switch_fiber(me->done ? nullptr : &me->fake_stack, g_main_bottom, g_main_size);
#endif
  hostemu_switch(&me->sp, g_sched_sp);
#ifdef HOSTEMU_ASAN
  __sanitizer_finish_switch_fiber(me->fake_stack, &g_main_bottom, &g_main_size);
#endif
}
inline void yield() { to_scheduler(); }

extern "C" inline void hostemu_entry() {
#ifdef HOSTEMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_main_bottom, &g_main_size);
#endif
  Lane* me = g_cur;
  (*g_blk.body)();
  me->done = true;
  Block& B = g_blk;
  B.alive_block--; B.alive_wave[me->tid >> 6]--; B.alive_half[me->tid >> 5]--;
  B.progress++;
  to_scheduler();
  abort();  // a finished fiber is never resumed
}

inline void prepare(Lane& L) {
  if (!L.stack) {
    L.stack_size = STACK_BYTES;
    L.stack = (char*)mmap(nullptr, L.stack_size + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (L.stack == MAP_FAILED) { perror("hostemu: mmap"); abort(); }
    mprotect(L.stack, 4096, PROT_NONE);  // guard page below the stack
    L.stack += 4096;
  }
  // initial frame as hostemu_switch expects it: [mxcsr/fpcw][r15 r14 r13 r12 rbx rbp][return address]
  uint64_t* top = (uint64_t*)(L.stack + L.stack_size);
  top -= 2;                       // keep 16-byte alignment at function entry (rsp % 16 == 8 after the `ret`)
  *--top = 0;                     // fake return address of hostemu_entry (never returns)
  *--top = (uint64_t)(void*)&hostemu_entry;
  for (int k = 0; k < 6; k++) *--top = 0;
  uint32_t csr[2];
  asm volatile("stmxcsr %0" : "=m"(csr[0]));
  uint16_t cw; asm volatile("fnstcw %0" : "=m"(cw)); csr[1] = cw;
  --top; memcpy(top, csr, 8);
  L.sp = top;
  L.done = false; L.started = false; L.fake_stack = nullptr;
}

// wait until `expected()` lanes have arrived at r
template <class F>
inline void rendezvous(Rendezvous& r, F alive) {
  Block& B = g_blk;
  const unsigned g = r.gen;
  r.arrived++;
  for (;;) {
    if (r.gen != g) return;
    if (r.arrived >= alive()) { r.arrived = 0; r.gen++; B.progress++; return; }
    yield();
  }
}
inline void sync_block() { rendezvous(g_blk.rv_block, [] { return g_blk.alive_block; }); }
inline void sync_group(int width) {
  const int t = g_cur->tid;
  if (width > 32) rendezvous(g_blk.rv_wave[t >> 6], [t] { return g_blk.alive_wave[t >> 6]; });
  else rendezvous(g_blk.rv_half[t >> 5], [t] { return g_blk.alive_half[t >> 5]; });
}

inline void run_block() {
  Block& B = g_blk;
  const int n = B.nthreads;
  if ((int)B.lanes.size() < n) B.lanes.resize(n);
  B.alive_block = n;
  for (int w = 0; w < 16; w++) { B.alive_wave[w] = 0; B.rv_wave[w] = Rendezvous(); }
  for (int h = 0; h < 32; h++) { B.alive_half[h] = 0; B.rv_half[h] = Rendezvous(); }
  B.rv_block = Rendezvous();
  for (int t = 0; t < n; t++) { B.lanes[t].tid = t; prepare(B.lanes[t]); B.alive_wave[t >> 6]++; B.alive_half[t >> 5]++; }
  int remaining = n;
  while (remaining > 0) {
    const unsigned long before = B.progress;
    for (int t = 0; t < n; t++) {
      Lane& L = B.lanes[t];
      if (L.done) continue;
      g_cur = &L;
#ifdef HOSTEMU_ASAN
      __sanitizer_start_switch_fiber(&g_sched_fake, L.stack, L.stack_size);
#endif
      hostemu_switch(&g_sched_sp, L.sp);
#ifdef HOSTEMU_ASAN
      __sanitizer_finish_switch_fiber(g_sched_fake, nullptr, nullptr);
#endif
      if (L.done) remaining--;
    }
    if (remaining > 0 && B.progress == before) {
      fprintf(stderr, "hostemu: divergent barrier - every live thread of block (%u) waits and no group is complete\n", B.block_idx.x);
      abort();
    }
  }
  g_cur = nullptr;
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  Block& B = g_blk;
  B.dyn_smem = shmem ? aligned_alloc(16, (shmem + 15) & ~(size_t)15) : nullptr;
  if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1 || block.x > 1024) { fprintf(stderr, "hostemu: 1-D launches of <= 1024 threads only\n"); abort(); }
  B.nthreads = (int)block.x; B.block_dim = block; B.grid_dim = grid; B.body = &body;
  for (unsigned b = 0; b < grid.x; b++) { B.block_idx = {b, 0, 0}; run_block(); }
  B.body = nullptr;
  free(B.dyn_smem); B.dyn_smem = nullptr;
}

}  // namespace hostemu

#define threadIdx (uint3{(unsigned)hostemu::g_cur->tid, 0u, 0u})
#define blockIdx (hostemu::g_blk.block_idx)
#define blockDim (hostemu::g_blk.block_dim)
#define gridDim (hostemu::g_blk.grid_dim)

// ---- cross-lane operations ----------------------------------------------------------------------
static inline void __syncthreads() { hostemu::sync_block(); }
static inline uint64_t __ballot(int pred) {
  hostemu::Block& B = hostemu::g_blk;
  const int t = hostemu::g_cur->tid, w0 = t & ~63;
  B.xchg[t] = pred ? 1 : 0;
  hostemu::sync_group(64);
  uint64_t m = 0;
  for (int l = 0; l < 64 && w0 + l < B.nthreads; l++)
    if (!B.lanes[w0 + l].done && B.xchg[w0 + l]) m |= 1ull << l;
  hostemu::sync_group(64);
  return m;
}
template <class T>
static inline T hostemu_exchange(T v, int src_lane_in_group, int width, bool valid) {
  static_assert(sizeof(T) <= 8, "shuffles move at most 8 bytes");
  hostemu::Block& B = hostemu::g_blk;
  const int t = hostemu::g_cur->tid, g0 = t & ~(width - 1);
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  B.xchg[t] = raw;
  hostemu::sync_group(width);
  T out = v;
  if (valid) { const uint64_t r = B.xchg[g0 + (src_lane_in_group & (width - 1))]; memcpy(&out, &r, sizeof(T)); }
  hostemu::sync_group(width);
  return out;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) { return hostemu_exchange(v, src, width, true); }
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  const int l = hostemu::g_cur->tid & (width - 1);
  return hostemu_exchange(v, l - (int)d, width, l - (int)d >= 0);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) {
  const int l = hostemu::g_cur->tid & (width - 1);
  return hostemu_exchange(v, l ^ m, width, (l ^ m) < width);
}

// ---- scalar intrinsics --------------------------------------------------------------------------
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline double __longlong_as_double(long long i) { double d; memcpy(&d, &i, 8); return d; }
static inline long long __double_as_longlong(double d) { long long i; memcpy(&i, &d, 8); return i; }
static inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
static inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
static inline int atomicOr(int* p, int v) { const int o = *p; *p = o | v; return o; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
// gfx950 hardware approximations -> libm (the tests' tolerances cover the ulp-level differences)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_sinf(x) sinf(6.28318530717958647692f * (x))   // argument in revolutions
#define __builtin_amdgcn_cosf(x) cosf(6.28318530717958647692f * (x))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
// v_readlane_b32 (the lane index is wave-uniform): the value of one lane, in every lane of the wavefront
#define __builtin_amdgcn_readlane(v, l) __shfl((int)(v), (int)(l), 64)
// v_readfirstlane_b32 is used on values that ARE wave-uniform (to tell the compiler so): the lane's own value
#define __builtin_amdgcn_readfirstlane(v) (v)
// v_med3_f32: the median of three = clamp(x, lo, hi) for lo <= hi
static inline float hostemu_med3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) hostemu_med3f(a, b, c)
#define DC_HALF_SELECT(a, b, half) ((half) ? (b) : (a))   // csrc/sag_doggo_coop.hpp: a v_cndmask with a constant lane mask on the GPU
#define DC_OPAQUE(x) asm volatile("" : "+r"(x))   // csrc/sag_doggo_coop.hpp: the "+v" (VGPR) constraint is the GPU's
#if !defined(__clang__)
#define __builtin_readcyclecounter() __builtin_ia32_rdtsc()
#endif

// ---- runtime API (synchronous, one "device") -------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef struct hostemu_stream* hipStream_t;
struct hostemu_event { std::chrono::steady_clock::time_point t; };
typedef hostemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { int multiProcessorCount; };
static inline const char* hipGetErrorString(hipError_t) { return "hostemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("SAG_HOSTEMU"); *n = (e && atoi(e) == 1) ? 1 : 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t bytes) { *p = (T*)malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned) { *p = (T*)malloc(bytes ? bytes : 1); return hipSuccess; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostemu_event(); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hostemu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t n) { memcpy(sym, src, n); return hipSuccess; }
static inline hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n) { memcpy(dst, sym, n); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                         \
  do {                                                                                      \
    (void)(stream);                                                                         \
    const std::function<void()> hostemu_body_ = [&]() { (kernel)(__VA_ARGS__); }; \
    hostemu::launch((grid), (block), (size_t)(shmem), hostemu_body_);                       \
  } while (0)

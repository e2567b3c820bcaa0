// CPU emulation of the small slice of the HIP device/runtime model that the sgmse_amd kernels use.
//
// TEST INFRASTRUCTURE ONLY.  This header shadows sgmse_amd/csrc/sgmse_devrt.h when the kernel sources are
// compiled with g++ (-I tests/emu) into tests/emu/libsgmse_emu.so, so that the *same* kernel code (index maths,
// LDS tiling, MFMA fragment layouts, wave shuffles, barriers) can be checked against the oracle in the build
// container, which has no GPU.  The product library (libsgmse_hip.so) is built by hipcc with the real header and
// never contains any of this.  Every workgroup runs as 64..1024 cooperative fibers on one OS thread; a wavefront
// is 64 consecutive fibers; __syncthreads / wave collectives are fiber barriers.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <algorithm>

#define SGMSE_EMULATOR 1

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct float2 { float x, y; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x2 __attribute__((vector_size(8)));
typedef uint32_t u32x4 __attribute__((vector_size(16)));

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

namespace emu {
struct Idx { unsigned x, y, z; };
extern thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_block();
float shfl_xor(float v, int mask);
float shfl_idx(float v, int src_lane);
f32x16 mfma_32x32x2(float a, float b, f32x16 c);
f32x4 mfma_16x16x4(float a, float b, f32x4 c);
f32x16 mfma_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c);
f32x16 mfma_32x32x16_f16(u32x4 a, u32x4 b, f32x16 c);
uint32_t f32_to_f16(float x);
float f16_to_f32(uint32_t h);
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

static inline void __syncthreads() { emu::sync_block(); }
static inline float __shfl_xor(float v, int mask, int width = 64) { (void)width; return emu::shfl_xor(v, mask); }
static inline float __shfl(float v, int lane, int width = 64) { (void)width; return emu::shfl_idx(v, lane); }
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) { return emu::mfma_32x32x2(a, b, c); }
static inline f32x16 mfma_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c) { return emu::mfma_32x32x16_bf16(a, b, c); }
static inline f32x16 mfma_32x32x16_f16(u32x4 a, u32x4 b, f32x16 c) { return emu::mfma_32x32x16_f16(a, b, c); }
static inline uint32_t drt_f32_to_f16(float x) { return emu::f32_to_f16(x); }
static inline float drt_f16_to_f32(uint32_t h) { return emu::f16_to_f32(h); }
struct drt_buf { char* p; };
static inline drt_buf drt_make_buf(const float* base) { return drt_buf{reinterpret_cast<char*>(const_cast<float*>(base))}; }
template <int AUX = 0>
static inline float drt_buf_load(const drt_buf& b, unsigned voff, unsigned soff) { float v; memcpy(&v, b.p + voff + soff, 4); return v; }
template <int AUX = 0>
static inline void drt_buf_store(const drt_buf& b, float v, unsigned voff, unsigned soff) { memcpy(b.p + voff + soff, &v, 4); }
template <int AUX = 0>
static inline float2 drt_buf_load2(const drt_buf& b, unsigned voff, unsigned soff) { float2 v; memcpy(&v, b.p + voff + soff, 8); return v; }
template <int AUX = 0>
static inline void drt_buf_store2(const drt_buf& b, float2 v, unsigned voff, unsigned soff) { memcpy(b.p + voff + soff, &v, 8); }
static inline unsigned long long drt_clock() { return 0; }
static inline unsigned drt_hw_id() { return 0; }
static inline unsigned drt_xcc_id() { return 0; }
static inline uint32_t drt_f32x2_to_f16x2(float x0, float x1) { return emu::f32_to_f16(x0) | (emu::f32_to_f16(x1) << 16); }
static inline float drt_sub_f16_lo(float x, uint32_t h) { volatile float r = x - emu::f16_to_f32(h & 0xffffu); return r; }
static inline float drt_sub_f16_hi(float x, uint32_t h) { volatile float r = x - emu::f16_to_f32(h >> 16); return r; }
static inline float drt_exp2(float x) { return exp2f(x); }
static inline float drt_mul_rn(float a, float b) { volatile float r = a * b; return r; }     // (volatile: no contraction whatever the flags)
static inline float drt_add_rn(float a, float b) { volatile float r = a + b; return r; }
static inline void drt_atomic_max_nonneg(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t nv; memcpy(&nv, &v, 4);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
  while (old < nv && !__atomic_compare_exchange_n(ip, &old, nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
// exchange-add steps of the GroupNorm butterfly (see sgmse_amd/csrc/sgmse_devrt.h)
template <int P>
static inline float drt_xadd(float a, float b) {
  const int lane = (int)(threadIdx.x & 63);
  const int bit = P == 16 ? 16 : P == 8 ? 8 : P == 7 ? 4 : 1;
  const float ap = emu::shfl_idx(a, lane ^ P), bp = emu::shfl_idx(b, lane ^ P);
  return (lane & bit) ? b + bp : a + ap;
}
static inline float drt_add_xor2(float a) { return a + emu::shfl_xor(a, 2); }
// whole-wave shifts by one lane (see sgmse_amd/csrc/sgmse_devrt.h): lane - 1 / lane + 1, zero at the wave's ends
static inline float drt_wave_shr1(float v) {
  const int lane = (int)(threadIdx.x & 63);
  const float t = emu::shfl_idx(v, lane > 0 ? lane - 1 : lane);
  return lane > 0 ? t : 0.f;
}
static inline float drt_wave_shl1(float v) {
  const int lane = (int)(threadIdx.x & 63);
  const float t = emu::shfl_idx(v, lane < 63 ? lane + 1 : lane);
  return lane < 63 ? t : 0.f;
}
static inline int drt_uniform(int v) { return v; }
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) { return emu::mfma_16x16x4(a, b, c); }
#define __builtin_amdgcn_iglp_opt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __expf(x) expf(x)
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  float f;
  do { memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4); } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4);
  return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline double drt_shfl_xor_f64(double v, int mask) {
  uint64_t u; memcpy(&u, &v, 8);
  uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
  float fl, fh; memcpy(&fl, &lo, 4); memcpy(&fh, &hi, 4);
  fl = emu::shfl_xor(fl, mask); fh = emu::shfl_xor(fh, mask);
  memcpy(&lo, &fl, 4); memcpy(&hi, &fh, 4);
  u = ((uint64_t)hi << 32) | lo; memcpy(&v, &u, 8);
  return v;
}

static inline void drt_wave_sync() { (void)emu::shfl_idx(0.f, 0); }
#define DRT_PIN_HERE(x) ((void)0)
#define DRT_SCHED_FENCE() ((void)0)
#define DRT_PIN8(a, b, c, d, e, f, g, h) ((void)0)
#define DRT_PIN_INT(x) ((void)0)
#define DRT_CODE_MARKER(n) ((void)0)
#define DRT_LAUNCH(kern, grid, block, stream, ...) \
  do { (void)(stream); emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); }); } while (0)

namespace drt {
typedef void* stream_t;
struct event_t { double t; };
struct graph_t { int dummy; };
inline bool is_emulator() { return true; }
inline const char* backend_name() { return "cpu-emulator(test-only)"; }
inline int set_device(int) { return 0; }
inline int malloc_dev(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 1; }
inline int free_dev(void* p) { free(p); return 0; }
inline int malloc_host(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 1; }
inline int free_host(void* p) { free(p); return 0; }
inline int memcpy_h2d(void* d, const void* s, size_t n, stream_t) { memcpy(d, s, n); return 0; }
inline int memcpy_d2h(void* d, const void* s, size_t n, stream_t) { memcpy(d, s, n); return 0; }
inline int memcpy_d2d(void* d, const void* s, size_t n, stream_t) { memmove(d, s, n); return 0; }
inline int memset_dev(void* d, int v, size_t n, stream_t) { memset(d, v, n); return 0; }
inline int stream_sync(stream_t) { return 0; }
inline int stream_create(stream_t* st) { *st = nullptr; return 0; }      // (launches are synchronous: a second stream orders nothing)
inline int stream_destroy(stream_t) { return 0; }
inline int event_create_order(struct event_t*) { return 0; }
inline int stream_wait_event(stream_t, struct event_t*) { return 0; }
inline int last_error() { return 0; }
inline const char* error_string(int) { return "emulator"; }
double wall_ms();
inline int event_create(event_t* e) { e->t = 0; return 0; }
inline int event_destroy(event_t*) { return 0; }
inline int event_record(event_t* e, stream_t) { e->t = wall_ms(); return 0; }
inline int event_sync(event_t*) { return 0; }
inline float event_elapsed_ms(const event_t& a, const event_t& b) { return float(b.t - a.t); }
// graphs: not emulated (eager execution only)
inline bool graphs_supported() { return false; }
inline int graph_begin_capture(stream_t) { return 1; }
inline int graph_end_capture(stream_t, graph_t*) { return 1; }
inline int graph_end_capture_update(stream_t, graph_t*) { return 1; }
inline void graph_abort_capture(stream_t) {}
inline int graph_launch(graph_t*, stream_t) { return 1; }
inline int graph_destroy(graph_t*) { return 0; }
inline int device_count() { return 1; }
inline size_t device_mem_total() { return size_t(8) << 30; }
}  // namespace drt

// Device/runtime shim for the gfx950 build: the HIP runtime itself plus the few names the kernel and engine
// sources use (f32x16, DRT_LAUNCH, drt::*).  tests/emu/ holds a same-named header that replaces this one when
// the sources are compiled for the CPU workgroup emulator (test infrastructure); the product is always built
// with THIS file by hipcc --offload-arch=gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x16_bf16: D[32x32] += A[32x16] * B[16x32], operands = 8 bf16 per lane packed in 4 dwords
// (lane l: row/col l & 31, k group l >> 5), fp32 accumulate; C/D layout as the 32x32x2 f32 form
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_32x32x16_f16(u32x4 a, u32x4 b, f32x16 c) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// fp32 <-> fp16 bit patterns (round to nearest even, hardware conversion)
__device__ __forceinline__ uint32_t drt_f32_to_f16(float x) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)x); }
__device__ __forceinline__ float drt_f16_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
// atomic max of non-negative floats (their bit patterns order like unsigned integers)
__device__ __forceinline__ void drt_atomic_max_nonneg(float* p, float v) {
  atomicMax(reinterpret_cast<unsigned int*>(p), __builtin_bit_cast(unsigned int, v));
}

#define DRT_LAUNCH(kern, grid, block, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), 0, (stream), __VA_ARGS__)

namespace drt {
typedef hipStream_t stream_t;
struct event_t { hipEvent_t e; };
struct graph_t { hipGraph_t g; hipGraphExec_t x; };
inline bool is_emulator() { return false; }
inline const char* backend_name() { return "hip-gfx950"; }
inline int set_device(int d) { return int(hipSetDevice(d)); }
inline int malloc_dev(void** p, size_t n) { return int(hipMalloc(p, n)); }
inline int free_dev(void* p) { return int(hipFree(p)); }
inline int memcpy_h2d(void* d, const void* s, size_t n, stream_t st) { return int(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st)); }
inline int memcpy_d2h(void* d, const void* s, size_t n, stream_t st) { return int(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st)); }
inline int memcpy_d2d(void* d, const void* s, size_t n, stream_t st) { return int(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st)); }
inline int memset_dev(void* d, int v, size_t n, stream_t st) { return int(hipMemsetAsync(d, v, n, st)); }
inline int stream_sync(stream_t st) { return int(hipStreamSynchronize(st)); }
inline int last_error() { return int(hipGetLastError()); }
inline const char* error_string(int e) { return hipGetErrorString(hipError_t(e)); }
inline int event_create(event_t* e) { return int(hipEventCreate(&e->e)); }
inline int event_destroy(event_t* e) { return int(hipEventDestroy(e->e)); }
inline int event_record(event_t* e, stream_t st) { return int(hipEventRecord(e->e, st)); }
inline int event_sync(event_t* e) { return int(hipEventSynchronize(e->e)); }
inline float event_elapsed_ms(const event_t& a, const event_t& b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, a.e, b.e); return ms; }
inline bool graphs_supported() { return true; }
inline int graph_begin_capture(stream_t st) { return int(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); }
inline int graph_end_capture(stream_t st, graph_t* g) {
  hipError_t e = hipStreamEndCapture(st, &g->g);
  if (e != hipSuccess) return int(e);
  return int(hipGraphInstantiate(&g->x, g->g, nullptr, nullptr, 0));
}
inline int graph_launch(graph_t* g, stream_t st) { return int(hipGraphLaunch(g->x, st)); }
inline int graph_destroy(graph_t* g) {
  if (g->x) (void)hipGraphExecDestroy(g->x);
  if (g->g) (void)hipGraphDestroy(g->g);
  g->x = nullptr; g->g = nullptr;
  return 0;
}
inline int device_count() { int n = 0; (void)hipGetDeviceCount(&n); return n; }
inline size_t device_mem_total() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return t; }
}  // namespace drt

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
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (no packed-fp32 helper on purpose: v_pk_fma_f32 results are not safe on a shared device -- kernels_conv_thin.h, Makefile NOPK)
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
// two fp32 -> packed fp16 pair (element 0 in the low half), round to nearest even: v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t drt_f32x2_to_f16x2(float x0, float x1) {
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2_t));
}
// x - (float)(half `hi_sel` of the packed fp16 pair h), one rounding: v_fma_mix_f32 (an fp32 fma that reads an operand as fp16) -- one
// instruction where a conversion and a subtraction would be two; bit-identical to them (the product h * -1 is exact)
__device__ __forceinline__ float drt_sub_f16_lo(float x, uint32_t h) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
__device__ __forceinline__ float drt_sub_f16_hi(float x, uint32_t h) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
// IEEE single multiply / add that the compiler must not contract into a fused multiply-add (where a reference's rounding
// sequence has to be reproduced operation by operation)
__device__ __forceinline__ float drt_mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float drt_add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float drt_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
// Raw buffer view of a global fp32 tensor slice: address = base + voff (per-lane VGPR, bytes) + soff (uniform SGPR, bytes).
// One buffer_load/store_dword per access with NO address arithmetic on the vector ALU; offsets must stay below 2^31.
struct drt_buf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ drt_buf drt_make_buf(const float* base) {
  return drt_buf{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000)};
}
// AUX: cache-policy bits of the instruction (0 default, 2 = nt: streaming data that is not read again soon)
template <int AUX = 0>
__device__ __forceinline__ float drt_buf_load(const drt_buf& b, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void drt_buf_store(const drt_buf& b, float v, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, voff, soff, AUX);
}
// the same for two consecutive floats (8-byte aligned): buffer_load/store_dwordx2
template <int AUX = 0>
__device__ __forceinline__ float2 drt_buf_load2(const drt_buf& b, unsigned voff, unsigned soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, AUX);
  return make_float2(__builtin_bit_cast(float, (unsigned)v[0]), __builtin_bit_cast(float, (unsigned)v[1]));
}
template <int AUX = 0>
__device__ __forceinline__ void drt_buf_store2(const drt_buf& b, float2 v, unsigned voff, unsigned soff) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const u2_t u = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)};
  __builtin_amdgcn_raw_buffer_store_b64(u, b.r, voff, soff, AUX);
}
// measurement: shader clock and the hardware placement of this wave (HW_REG_HW_ID, HW_REG_XCC_ID)
__device__ __forceinline__ unsigned long long drt_clock() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned drt_hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
__device__ __forceinline__ unsigned drt_xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20); }
// atomic max of non-negative floats (their bit patterns order like unsigned integers)
__device__ __forceinline__ void drt_atomic_max_nonneg(float* p, float v) {
  atomicMax(reinterpret_cast<unsigned int*>(p), __builtin_bit_cast(unsigned int, v));
}

// xor-shuffle of a double (two 32-bit shuffles)
__device__ __forceinline__ double drt_shfl_xor_f64(double v, int mask) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(u & 0xffffffffull), mask);
  const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), mask);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Exchange-add steps of the GroupNorm-statistics butterfly over the 32 lanes that hold one 32-pixel row segment (both
// 32-lane halves of a wave do the same thing).  drt_xadd<P>(a, b): a lane whose selector bit is 0 returns a + a', the others
// b + b', where x' is the value of x in the partner lane:
//   P = 16: partner lane ^ 16, selector bit 4   (v_permlane16_swap_b32: one swap + one add)
//   P = 8 : partner lane ^ 8,  selector bit 3   (DPP row_ror:8)
//   P = 7 : partner lane ^ 7 (mirror inside a group of 8), selector bit 2   (DPP row_half_mirror)
//   P = 1 : partner lane ^ 1,  selector bit 0   (DPP quad_perm [1,0,3,2])
// drt_add_xor2(a) = a + a' with partner lane ^ 2 (DPP quad_perm [2,3,0,1]).  No LDS-pipe traffic (ds_bpermute) anywhere.
template <int CTRL>
__device__ __forceinline__ float drt_dpp_add(float a) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, 0xf, 0xf, true);
  return a + __builtin_bit_cast(float, t);
}
template <int P>
__device__ __forceinline__ float drt_xadd(float a, float b) {
  if constexpr (P == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];      // (bit_cast of r[i] directly reads element 0 twice with this clang)
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
  } else {
    constexpr int ctrl = P == 8 ? 0x128 : P == 7 ? 0x141 : 0xB1;
    constexpr unsigned bit = P == 8 ? 8u : P == 7 ? 4u : 1u;
    const float t = drt_dpp_add<ctrl>(a), u = drt_dpp_add<ctrl>(b);
    return (threadIdx.x & bit) ? u : t;
  }
}
__device__ __forceinline__ float drt_add_xor2(float a) { return drt_dpp_add<0x4E>(a); }
// whole-wave shifts by one lane (DPP wave_shr:1 / wave_shl:1, one v_mov_b32_dpp each, no LDS-pipe traffic): drt_wave_shr1(v) = the
// value of lane - 1 (lane 0: 0), drt_wave_shl1(v) = the value of lane + 1 (lane 63: 0).  All lanes must be active.
__device__ __forceinline__ float drt_wave_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float drt_wave_shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
// a value every lane of the wave agrees on (the wave index), moved to a scalar register so that branches on it stay scalar
__device__ __forceinline__ int drt_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// execution barrier among the lanes of one wave around wave-private LDS traffic: the hardware runs a wave in lock step and
// executes its LDS operations in order, so this only pins the compiler's ordering (the emulator synchronises its fibers)
__device__ __forceinline__ void drt_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#define DRT_PIN_HERE(x) asm volatile("" : "+v"(x))
// the machine scheduler moves nothing across this point
#define DRT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// pins 8 register pairs at this point of the program (pure arithmetic is otherwise free to sink below a later block)
#define DRT_PIN8(a, b, c, d, e, f, g, h) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
#define DRT_PIN_INT(x) asm volatile("" : "+v"(x))
#define DRT_CODE_MARKER(n) asm volatile("; code marker %0" ::"n"(n))
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
// page-locked host memory: the source of host-to-device copies that must not block the calling thread
inline int malloc_host(void** p, size_t n) { return int(hipHostMalloc(p, n, hipHostMallocDefault)); }
inline int free_host(void* p) { return int(hipHostFree(p)); }
inline int memcpy_h2d(void* d, const void* s, size_t n, stream_t st) { return int(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st)); }
inline int memcpy_d2h(void* d, const void* s, size_t n, stream_t st) { return int(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st)); }
inline int memcpy_d2d(void* d, const void* s, size_t n, stream_t st) { return int(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st)); }
inline int memset_dev(void* d, int v, size_t n, stream_t st) { return int(hipMemsetAsync(d, v, n, st)); }
inline int stream_sync(stream_t st) { return int(hipStreamSynchronize(st)); }
// a second stream of the engine (side branches of the captured step) and the ordering events between the two (no timing)
inline int stream_create(stream_t* st) { return int(hipStreamCreateWithFlags(st, hipStreamNonBlocking)); }
inline int stream_destroy(stream_t st) { return int(hipStreamDestroy(st)); }
inline int event_create_order(event_t* e) { return int(hipEventCreateWithFlags(&e->e, hipEventDisableTiming)); }
inline int stream_wait_event(stream_t st, event_t* e) { return int(hipStreamWaitEvent(st, e->e, 0)); }
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
// end a capture whose graph has the topology of an already instantiated one (the same launch sequence with other arguments /
// grids: a new ragged batch composition, new buffers): update the executable in place instead of instantiating a new one
// (~1400 kernel nodes: instantiation is the expensive part of a re-capture).  Non-zero: the update was refused (topology
// differs) -- the caller instantiates afresh.
inline int graph_end_capture_update(stream_t st, graph_t* g) {
  hipGraph_t ng = nullptr;
  hipError_t e = hipStreamEndCapture(st, &ng);
  if (e != hipSuccess) return int(e);
  hipGraphNode_t bad = nullptr;
  hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
  e = hipGraphExecUpdate(g->x, ng, &bad, &res);
  if (e == hipSuccess && res == hipGraphExecUpdateSuccess) {
    if (g->g) (void)hipGraphDestroy(g->g);
    g->g = ng;
    return 0;
  }
  (void)hipGetLastError();
  // refused: fall back to a fresh executable of the new graph
  if (g->x) (void)hipGraphExecDestroy(g->x);
  if (g->g) (void)hipGraphDestroy(g->g);
  g->x = nullptr; g->g = ng;
  e = hipGraphInstantiate(&g->x, g->g, nullptr, nullptr, 0);
  return e == hipSuccess ? -1 : int(e);      // -1: worked, but as a new instantiation
}
// a capture that must not produce a graph (its body failed): end it and discard whatever was recorded
inline void graph_abort_capture(stream_t st) {
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(st, &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
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

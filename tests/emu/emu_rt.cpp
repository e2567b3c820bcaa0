// Fiber-based workgroup emulator (TEST INFRASTRUCTURE ONLY; see sgmse_devrt.h in this directory).
//
// A launch runs its workgroups on a small pool of OS threads; inside a workgroup every HIP thread is a fiber
// (x86-64 callee-saved context switch below).  Wave collectives (shuffle, MFMA) exchange operands through a
// per-wave, double-buffered staging area guarded by a wave-level fiber barrier, which reproduces the
// architectural lane -> element maps of gfx950:
//   v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31]
//   v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[i=4*(l>>4)+r][j=l&15]
// (cdna_hip_programming.md section 3).  Accumulation is a k-ordered fmaf chain, as on the hardware.
#include "sgmse_devrt.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdio>

#if !defined(__x86_64__)
#error "the fiber switch is written for x86-64"
#endif

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_ctx_switch
    .type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_ctx_switch, .-emu_ctx_switch
)");

namespace emu {

thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {

constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 96 * 1024;

struct WaveState {
  int arrived = 0;
  unsigned gen = 0;
  float a[2][64], b[2][64];
  float a8[2][64][8], b8[2][64][8];   // bf16 MFMA operands (widened)
};

struct Worker {
  char* stacks = nullptr;
  void* sp[kMaxThreads];
  bool done[kMaxThreads];
  Idx tid[kMaxThreads];
  int parity[kMaxThreads];
  void* main_sp = nullptr;
  int n = 0, cur = 0, ndone = 0;
  int blk_arrived = 0;
  unsigned blk_gen = 0;
  WaveState waves[kMaxThreads / 64];
  const std::function<void()>* body = nullptr;
  ~Worker() { free(stacks); }
};

thread_local Worker* tw = nullptr;

void fiber_entry();

void switch_to(Worker* w, int next) {
  int prev = w->cur;
  w->cur = next;
  t_threadIdx = w->tid[next];
  emu_ctx_switch(&w->sp[prev], w->sp[next]);
}

void yield_next(Worker* w) {
  int nx = w->cur;
  for (int i = 0; i < w->n; ++i) {
    nx = nx + 1 == w->n ? 0 : nx + 1;
    if (!w->done[nx]) break;
  }
  if (nx != w->cur) switch_to(w, nx);
}

void fiber_entry() {
  Worker* w = tw;
  (*w->body)();
  w = tw;
  w->done[w->cur] = true;
  w->ndone++;
  if (w->ndone == w->n) {
    void* dummy;
    emu_ctx_switch(&dummy, w->main_sp);
  } else {
    int nx = w->cur;
    for (;;) {
      nx = nx + 1 == w->n ? 0 : nx + 1;
      if (!w->done[nx]) break;
    }
    int prev = w->cur;
    w->cur = nx;
    t_threadIdx = w->tid[nx];
    emu_ctx_switch(&w->sp[prev], w->sp[nx]);
  }
  abort();  // a finished fiber is never resumed
}

void run_block(Worker* w, dim3 block, const std::function<void()>& body) {
  int n = int(block.x * block.y * block.z);
  if (n > kMaxThreads) { fprintf(stderr, "emu: block too large\n"); abort(); }
  if (!w->stacks) w->stacks = static_cast<char*>(aligned_alloc(64, kStack * kMaxThreads));
  w->n = n; w->ndone = 0; w->cur = 0; w->body = &body;
  w->blk_arrived = 0;
  for (int i = 0; i < (n + 63) / 64; ++i) w->waves[i].arrived = 0;
  for (int i = 0; i < n; ++i) {
    w->done[i] = false;
    w->parity[i] = 0;
    w->tid[i] = Idx{unsigned(i % block.x), unsigned((i / block.x) % block.y), unsigned(i / (block.x * block.y))};
    char* top = w->stacks + kStack * (i + 1);
    uintptr_t A = (reinterpret_cast<uintptr_t>(top) - 64) & ~uintptr_t(15);
    void** slot = reinterpret_cast<void**>(A);
    slot[0] = reinterpret_cast<void*>(&fiber_entry);
    slot[1] = nullptr;
    for (int r = 1; r <= 6; ++r) slot[-r] = nullptr;
    w->sp[i] = reinterpret_cast<void*>(A - 48);
  }
  t_threadIdx = w->tid[0];
  emu_ctx_switch(&w->main_sp, w->sp[0]);
}

inline int wave_size_of(Worker* w, int wave) {
  int rem = w->n - wave * 64;
  return rem < 64 ? rem : 64;
}

void wave_barrier(Worker* w, int wave) {
  WaveState& ws = w->waves[wave];
  unsigned g = ws.gen;
  if (++ws.arrived == wave_size_of(w, wave)) {
    ws.arrived = 0;
    ws.gen++;
  } else {
    while (ws.gen == g) yield_next(w);
  }
}

}  // namespace

void sync_block() {
  Worker* w = tw;
  unsigned g = w->blk_gen;
  if (++w->blk_arrived == w->n - w->ndone) {
    w->blk_arrived = 0;
    w->blk_gen++;
  } else {
    while (w->blk_gen == g) yield_next(w);
  }
}

float shfl_xor(float v, int mask) {
  Worker* w = tw;
  int t = w->cur, wave = t >> 6, lane = t & 63, p = w->parity[t];
  w->parity[t] ^= 1;
  WaveState& ws = w->waves[wave];
  ws.a[p][lane] = v;
  wave_barrier(w, wave);
  int src = lane ^ mask;
  return src < wave_size_of(w, wave) ? ws.a[p][src] : v;
}

float shfl_idx(float v, int src_lane) {
  Worker* w = tw;
  int t = w->cur, wave = t >> 6, lane = t & 63, p = w->parity[t];
  w->parity[t] ^= 1;
  WaveState& ws = w->waves[wave];
  ws.a[p][lane] = v;
  wave_barrier(w, wave);
  int src = src_lane & 63;
  return src < wave_size_of(w, wave) ? ws.a[p][src] : v;
}

f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
  Worker* w = tw;
  int t = w->cur, wave = t >> 6, lane = t & 63, p = w->parity[t];
  w->parity[t] ^= 1;
  WaveState& ws = w->waves[wave];
  ws.a[p][lane] = a;
  ws.b[p][lane] = b;
  wave_barrier(w, wave);
  int j = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    acc = fmaf(ws.a[p][i], ws.b[p][j], acc);            // k = 0
    acc = fmaf(ws.a[p][32 + i], ws.b[p][32 + j], acc);  // k = 1
    c[r] = acc;
  }
  return c;
}

// fp32 <-> fp16 bit patterns, round to nearest even, subnormals and infinities handled (software: g++ 11 has no _Float16)
uint32_t f32_to_f16(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t absu = u & 0x7fffffffu;
  if (absu >= 0x7f800000u) return sign | (absu > 0x7f800000u ? 0x7e00u : 0x7c00u);   // nan / inf
  if (absu >= 0x477ff000u) return sign | 0x7c00u;                                      // rounds to >= 65520 -> inf
  if (absu < 0x38800000u) {                                                            // below 2^-14: fp16 subnormal
    // value = absx / 2^-24 rounded to nearest even integer
    float ax; memcpy(&ax, &absu, 4);
    const float scaled = ax * 16777216.0f;                                             // exact
    const float r = nearbyintf(scaled);                                                // default mode: ties to even
    return sign | (uint32_t)r;                                                         // r == 1024 encodes 2^-14 correctly
  }
  uint32_t mant = absu & 0x7fffffu, exp = (absu >> 23) - 112;                          // rebias 127 -> 15
  uint32_t h = (exp << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                              // carries propagate into the exponent
  return sign | h;
}
float f16_to_f32(uint32_t h) {
  const uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, mant = h & 0x3ffu;
  float out;
  if (exp == 0) { out = (float)mant * 5.9604644775390625e-08f; uint32_t u; memcpy(&u, &out, 4); u |= sign; memcpy(&out, &u, 4); return out; }
  uint32_t u = exp == 31 ? (sign | 0x7f800000u | (mant << 13)) : (sign | ((exp + 112) << 23) | (mant << 13));
  memcpy(&out, &u, 4);
  return out;
}

// 16-bit operands: 8 per lane (lane l: row/col l & 31, k group l >> 5); products are exact in fp32, the 16-term sum is
// formed in double and rounded once (the hardware's internal order is unspecified; tests are tolerance-based)
template <class Widen>
static f32x16 mfma_32x32x16_16bit(u32x4 a, u32x4 b, f32x16 c, Widen widen) {
  Worker* w = tw;
  int t = w->cur, wave = t >> 6, lane = t & 63, p = w->parity[t];
  w->parity[t] ^= 1;
  WaveState& ws = w->waves[wave];
  for (int e = 0; e < 4; ++e) {
    ws.a8[p][lane][2 * e] = widen(a[e] & 0xffffu); ws.a8[p][lane][2 * e + 1] = widen(a[e] >> 16);
    ws.b8[p][lane][2 * e] = widen(b[e] & 0xffffu); ws.b8[p][lane][2 * e + 1] = widen(b[e] >> 16);
  }
  wave_barrier(w, wave);
  int j = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    double acc = c[r];
    for (int g = 0; g < 2; ++g)
      for (int e = 0; e < 8; ++e) acc += (double)ws.a8[p][32 * g + i][e] * (double)ws.b8[p][32 * g + j][e];
    c[r] = (float)acc;
  }
  return c;
}
f32x16 mfma_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return mfma_32x32x16_16bit(a, b, c, [](uint32_t h) { uint32_t u = h << 16; float f; memcpy(&f, &u, 4); return f; });
}
f32x16 mfma_32x32x16_f16(u32x4 a, u32x4 b, f32x16 c) { return mfma_32x32x16_16bit(a, b, c, [](uint32_t h) { return f16_to_f32(h); }); }

f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
  Worker* w = tw;
  int t = w->cur, wave = t >> 6, lane = t & 63, p = w->parity[t];
  w->parity[t] ^= 1;
  WaveState& ws = w->waves[wave];
  ws.a[p][lane] = a;
  ws.b[p][lane] = b;
  wave_barrier(w, wave);
  int j = lane & 15, q = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    int i = 4 * q + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(ws.a[p][16 * k + i], ws.b[p][16 * k + j], acc);
    c[r] = acc;
  }
  return c;
}

namespace {
struct Pool {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> threads;
  unsigned long job_id = 0;
  int pending = 0;
  bool stop = false;
  // current job
  dim3 grid, block;
  const std::function<void()>* body = nullptr;
  size_t nblocks = 0;
  std::atomic<size_t> next{0};

  void run_blocks() {
    static thread_local Worker worker;
    tw = &worker;
    t_blockDim = Idx{block.x, block.y, block.z};
    t_gridDim = Idx{grid.x, grid.y, grid.z};
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      t_blockIdx = Idx{unsigned(b % grid.x), unsigned((b / grid.x) % grid.y), unsigned(b / (size_t(grid.x) * grid.y))};
      run_block(&worker, block, *body);
    }
  }
  void thread_main() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || job_id != seen; });
        if (stop) return;
        seen = job_id;
      }
      run_blocks();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i) threads.emplace_back([this] { thread_main(); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_job.notify_all();
    for (auto& t : threads) t.join();
  }
};
std::mutex g_launch_mu;
}  // namespace

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  size_t nblocks = size_t(grid.x) * grid.y * grid.z;
  if (nblocks == 0) return;
  static int nthreads = [] {
    const char* e = getenv("SGMSE_EMU_THREADS");
    int n = e ? atoi(e) : int(std::thread::hardware_concurrency());
    return n < 1 ? 1 : (n > 64 ? 64 : n);
  }();
  static Pool* pool = new Pool(nthreads - 1);  // leaked on purpose: lives for the process
  std::lock_guard<std::mutex> launch_lock(g_launch_mu);
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    pool->grid = grid; pool->block = block; pool->body = &body; pool->nblocks = nblocks;
    pool->next.store(0);
    pool->pending = int(pool->threads.size());
    pool->job_id++;
  }
  pool->cv_job.notify_all();
  pool->run_blocks();
  {
    std::unique_lock<std::mutex> lk(pool->mu);
    pool->cv_done.wait(lk, [&] { return pool->pending == 0; });
  }
}

}  // namespace emu

namespace drt {
double wall_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace drt

// Does hipStreamWaitEvent order a second stream behind the work recorded on the first?  (round 6: the side-stream fork of engine.h)
// hipcc --offload-arch=gfx950 -O2 -o /tmp/swe tools/probes/stream_wait_event_probe.hip && /tmp/swe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void slow_fill(float* p, size_t n, float v, int spin) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = v;
  for (int k = 0; k < spin; ++k) acc = acc * 1.0000001f + 1e-9f;
  if (i < n) p[i] = (acc > 1e30f) ? 0.f : v;
}
__global__ void check(const float* p, size_t n, float v, unsigned* bad) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && p[i] != v) atomicAdd(bad, 1u);
}
int main() {
  const size_t n = 1 << 24;
  float* buf; unsigned* bad; hipMalloc(&buf, n * 4); hipMalloc(&bad, 4);
  for (int mode = 0; mode < 4; ++mode) {
    hipStream_t s1 = nullptr, s2;
    if (mode & 1) hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e; hipEventCreateWithFlags(&e, (mode & 2) ? hipEventDisableTiming : hipEventDefault);
    unsigned total = 0;
    for (int it = 0; it < 20; ++it) {
      hipMemsetAsync(buf, 0, n * 4, s1); hipMemsetAsync(bad, 0, 4, s1);
      hipStreamSynchronize(s1);
      const float v = 1.0f + it;
      for (int k = 0; k < 8; ++k) slow_fill<<<(unsigned)(n / 256), 256, 0, s1>>>(buf, n, v, 2000);
      hipEventRecord(e, s1);
      hipStreamWaitEvent(s2, e, 0);
      check<<<(unsigned)(n / 256), 256, 0, s2>>>(buf, n, v, bad);
      hipStreamSynchronize(s2);
      unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
      total += h;
      hipStreamSynchronize(s1);
    }
    printf("main stream %s, event %s: %u stale elements over 20 rounds\n", (mode & 1) ? "non-blocking" : "null", (mode & 2) ? "no-timing" : "default", total);
    hipEventDestroy(e); hipStreamDestroy(s2); if (s1) hipStreamDestroy(s1);
  }
  return 0;
}

// Minimal reproducer attempt for the round-6 finding: v_pk_fma_f32 results of conv3x3_thin_kernel changed (low element of the pair, lanes
// 48-63) whenever another process ran on the same GPU.  One kernel accumulates with __builtin_elementwise_fma on float2 (v_pk_fma_f32), one
// with two fmaf; same inputs through LDS as in the thin kernel (per-thread b128 + b64 reads of the tile, wave-uniform b128 weight reads).
//   hipcc --offload-arch=gfx950 -O3 -o pk_probe pk_fma_probe.hip;  ./pk_probe load &  ./pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PK>
__global__ __launch_bounds__(256, 3) void k(const float* in, const float* w, float* out, int nst) {
  __shared__ alignas(16) float s_in[2][4 * 18 * 68];
  __shared__ alignas(16) float s_w[2][144];
  const int tid = threadIdx.x, row = tid >> 4, cx = (tid & 15) * 4;
  f2 acc[2][4];
  double dacc[2][4];
  for (int a = 0; a < 2; ++a) for (int p = 0; p < 4; ++p) { acc[a][p] = f2{0.f, 0.f}; dacc[a][p] = 0.0; }
  const float* src = in + (size_t)blockIdx.x * 4 * 18 * 68;
  for (int st = 0; st < nst; ++st) {
    for (int i = tid; i < 4 * 18 * 68; i += 256) s_in[st & 1][i] = src[i] * (1.f + 0.001f * st);
    if (tid < 144) s_w[st & 1][tid] = w[st * 144 + tid];
    __syncthreads();
    const float* cur = s_in[st & 1]; const float* wc = s_w[st & 1];
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const float* s = cur + (q / 3) * 18 * 68 + (row + q % 3) * 68 + cx;
      const f4 v4 = *reinterpret_cast<const f4*>(s); const f2 v2 = *reinterpret_cast<const f2*>(s + 4);
      const float iv[6] = {v4[0], v4[1], v4[2], v4[3], v2[0], v2[1]};
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const f4 wq = *reinterpret_cast<const f4*>(wc + (q * 3 + dx) * 4);
        const f2 w01 = {wq[0], wq[1]}, w23 = {wq[2], wq[3]};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f2 v = {iv[p + dx], iv[p + dx]};
          if (PK == 1) { acc[0][p] = __builtin_elementwise_fma(w01, v, acc[0][p]); acc[1][p] = __builtin_elementwise_fma(w23, v, acc[1][p]); }
          else if (PK == 2) { acc[0][p] = acc[0][p] + w01 * v; acc[1][p] = acc[1][p] + w23 * v; }        // (-ffp-contract=off: v_pk_mul_f32 + v_pk_add_f32)
          else if (PK == 3) { dacc[0][p] = fma((double)w01[0], (double)v[0], dacc[0][p]); dacc[1][p] = fma((double)w23[0], (double)v[0], dacc[1][p]); }
          else { acc[0][p] = f2{fmaf(w01[0], v[0], acc[0][p][0]), fmaf(w01[1], v[1], acc[0][p][1])};
                 acc[1][p] = f2{fmaf(w23[0], v[0], acc[1][p][0]), fmaf(w23[1], v[1], acc[1][p][1])}; }
        }
      }
    }
    __syncthreads();
  }
  float* o = out + ((size_t)blockIdx.x * 256 + tid) * 16;
  for (int a = 0; a < 2; ++a) for (int p = 0; p < 4; ++p) {
    if (PK == 3) { o[(a * 4 + p) * 2] = (float)dacc[a][p]; o[(a * 4 + p) * 2 + 1] = (float)(dacc[a][p] * 0.5); }
    else { o[(a * 4 + p) * 2] = acc[a][p][0]; o[(a * 4 + p) * 2 + 1] = acc[a][p][1]; }
  }
}
// load for the second process: waves that keep the MATRIX pipe of every SIMD busy (what the network's kernels do)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_spin(float* out, int iters) {
  h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  f16v c = {0};
  for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (c[0] == 123.456f) out[0] = c[1];
}
int main(int argc, char** argv) {
  const int nb = 4096, nst = 32;
  const size_t nin = (size_t)nb * 4 * 18 * 68, nw = (size_t)nst * 144, nout = (size_t)nb * 256 * 16;
  std::vector<float> hin(nin), hw(nw);
  unsigned s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hin) v = rnd(); for (auto& v : hw) v = rnd();
  float *din, *dw, *dout; hipMalloc(&din, nin * 4); hipMalloc(&dw, nw * 4); hipMalloc(&dout, nout * 4);
  hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
  if (argc > 1 && !strcmp(argv[1], "load")) { for (int i = 0; i < 4000; ++i) k<0><<<nb, 256>>>(din, dw, dout, nst); hipDeviceSynchronize(); return 0; }
  if (argc > 1 && !strcmp(argv[1], "mfmaload")) { for (int i = 0; i < 600; ++i) mfma_spin<<<2048, 256>>>(dout, 20000); hipDeviceSynchronize(); return 0; }
  std::vector<float> ref(nout), got(nout);
  const char* names[4] = {"2 x v_fma_f32 (scalar; needs -fno-slp-vectorize)", "v_pk_fma_f32", "v_pk_mul_f32 + v_pk_add_f32 (-ffp-contract=off)", "v_fma_f64"};
  for (int pk = 0; pk < 4; ++pk) {
    auto launch = [&]() {
      if (pk == 0) k<0><<<nb, 256>>>(din, dw, dout, nst); else if (pk == 1) k<1><<<nb, 256>>>(din, dw, dout, nst);
      else if (pk == 2) k<2><<<nb, 256>>>(din, dw, dout, nst); else k<3><<<nb, 256>>>(din, dw, dout, nst);
    };
    launch();
    hipMemcpy(ref.data(), dout, nout * 4, hipMemcpyDeviceToHost);
    int badruns = 0; long lanes[4] = {0, 0, 0, 0}, elem[2] = {0, 0};
    for (int it = 0; it < 40; ++it) {
      launch();
      hipMemcpy(got.data(), dout, nout * 4, hipMemcpyDeviceToHost);
      bool bad = false;
      for (size_t i = 0; i < nout; ++i) if (memcmp(&got[i], &ref[i], 4)) { bad = true; ++lanes[((i / 16) & 63) >> 4]; ++elem[i & 1]; }
      badruns += bad;
    }
    printf("%-52s %2d of 40 launches differ from the first; by lane quarter [0-15 16-31 32-47 48-63] = [%ld %ld %ld %ld], by pair element [low high] = [%ld %ld]\n",
           names[pk], badruns, lanes[0], lanes[1], lanes[2], lanes[3], elem[0], elem[1]);
  }
  return 0;
}

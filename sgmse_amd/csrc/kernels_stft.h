// STFT / iSTFT front-end and the spectrogram amplitude transforms.
//
// Replaces torch.stft / torch.istft as configured by reference sgmse/data_module.py:190-218 (center=True with
// reflect padding, one-sided, no normalisation, periodic Hann / sqrt-Hann window) and spec_fwd / spec_back
// (data_module.py:162-188).  n_fft = 510 = 2*3*5*17 and 1534 = 2*13*59 are not powers of two and the whole
// front-end is < 0.01 % of the run time, so both transforms are direct DFTs against an fp64-generated twiddle table
// (SURVEY Appendix C gives the exact framing / envelope rules reproduced here).
#pragma once
#include <sgmse_devrt.h>

namespace sgmse {

struct StftArgs {
  const float* sig;      // [B][L]
  const float* window;   // [n_fft]
  const float2* twiddle; // [n_fft]: (cos, sin)(2*pi*m/n_fft)
  float2* spec;          // [B][F][K], F = n_fft/2+1, K = L/hop+1
  int L, n_fft, hop, F, K;
};

constexpr int kMaxNfft = 2048;

// grid = (K, B): one workgroup per frame; thread f computes bin f.
__global__ __launch_bounds__(256) void stft_kernel(StftArgs p) {
  __shared__ float s_fr[kMaxNfft];
  __shared__ float2 s_tw[kMaxNfft];
  const int k = blockIdx.x, b = blockIdx.y, N = p.n_fft, half = N / 2;
  const float* sig = p.sig + (size_t)b * p.L;
  for (int n = threadIdx.x; n < N; n += 256) {
    int idx = k * p.hop + n - half;          // position in the un-padded signal
    if (idx < 0) idx = -idx;                 // reflect (no edge repeat)
    if (idx >= p.L) idx = 2 * (p.L - 1) - idx;
    s_fr[n] = sig[idx] * p.window[n];
    s_tw[n] = p.twiddle[n];
  }
  __syncthreads();
  for (int f = threadIdx.x; f < p.F; f += 256) {
    float re = 0.f, im = 0.f;
    int m = 0;
    for (int n = 0; n < N; ++n) {
      const float2 w = s_tw[m];
      const float v = s_fr[n];
      re = fmaf(v, w.x, re);
      im = fmaf(-v, w.y, im);
      m += f;
      if (m >= N) m -= N;
    }
    p.spec[((size_t)b * p.F + f) * p.K + k] = make_float2(re, im);
  }
}

struct IstftArgs {
  const float2* spec;    // [B][F][K]
  const float* window; const float2* twiddle;
  float* out;            // [B][length]
  int n_fft, hop, F, K, length;
};

// One thread per output sample: inverse one-sided DFT of every frame that covers it, windowed, overlap-added and
// divided by the window-square envelope (torch.istft semantics incl. center crop).  grid = (ceil(length/256), B)
__global__ __launch_bounds__(256) void istft_kernel(IstftArgs p) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.length) return;
  const int b = blockIdx.y, N = p.n_fft, half = N / 2;
  const int pos = j + half;                       // position in the padded (centered) signal
  const float2* spec = p.spec + (size_t)b * p.F * p.K;
  int k_hi = pos / p.hop; if (k_hi > p.K - 1) k_hi = p.K - 1;
  int k_lo = (pos - N + p.hop) / p.hop; if (pos - N + 1 <= 0) k_lo = 0;
  if (k_lo < 0) k_lo = 0;
  float acc = 0.f, env = 0.f;
  const float invN = 1.0f / (float)N;
  for (int k = k_lo; k <= k_hi; ++k) {
    const int n = pos - k * p.hop;
    if (n < 0 || n >= N) continue;
    const float w = p.window[n];
    // irfft sample n: (X0 + (-1)^n X_{N/2} + 2 sum_{f=1}^{N/2-1} Re(X_f e^{+2 pi i f n / N})) / N   (N even)
    float s = spec[(size_t)0 * p.K + k].x;
    if ((N & 1) == 0) s += ((n & 1) ? -1.f : 1.f) * spec[(size_t)half * p.K + k].x;
    float t = 0.f;
    int m = n % N;
    const int fend = (N & 1) ? half + 1 : half;   // odd N: bins 1..half all doubled
    int mm = m;
    for (int f = 1; f < fend; ++f) {
      const float2 X = spec[(size_t)f * p.K + k];
      const float2 tw = p.twiddle[mm];
      t += X.x * tw.x - X.y * tw.y;
      mm += m;
      if (mm >= N) mm -= N;
    }
    acc += (s + 2.f * t) * invN * w;
    env += w * w;
  }
  p.out[(size_t)b * p.length + j] = acc / env;
}

// spec_fwd / spec_back, transform_type in {exponent=0, log=1, none=2}; element-wise on complex data.
struct SpecXformArgs { const float2* in; float2* out; size_t n; int type; float factor, exponent; int inverse; };

__global__ __launch_bounds__(256) void spec_xform_kernel(SpecXformArgs p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  float2 v = p.in[i];
  if (!p.inverse) {
    if (p.type == 0) {
      if (p.exponent != 1.f) {
        const float mag = sqrtf(v.x * v.x + v.y * v.y);
        const float g = mag > 0.f ? powf(mag, p.exponent) / mag : 0.f;
        v.x *= g; v.y *= g;
      }
      v.x *= p.factor; v.y *= p.factor;
    } else if (p.type == 1) {
      const float mag = sqrtf(v.x * v.x + v.y * v.y);
      const float g = mag > 0.f ? log1pf(mag) / mag : 0.f;
      v.x *= g * p.factor; v.y *= g * p.factor;
    }
  } else {
    if (p.type == 0) {
      v.x /= p.factor; v.y /= p.factor;
      if (p.exponent != 1.f) {
        const float mag = sqrtf(v.x * v.x + v.y * v.y);
        const float g = mag > 0.f ? powf(mag, 1.0f / p.exponent) / mag : 0.f;
        v.x *= g; v.y *= g;
      }
    } else if (p.type == 1) {
      v.x /= p.factor; v.y /= p.factor;
      const float mag = sqrtf(v.x * v.x + v.y * v.y);
      const float g = mag > 0.f ? (expf(mag) - 1.f) / mag : 0.f;
      v.x *= g; v.y *= g;
    }
  }
  p.out[i] = v;
}

}  // namespace sgmse

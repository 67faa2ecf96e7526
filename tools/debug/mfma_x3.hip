// What can the 3 x bf16 product structure of gemm3.hip reach with NOTHING but its LDS fragment reads and its MFMAs?
// Per wave and 16-deep k-step: 12 ds_read_b128 (3 planes x 2 row blocks x 2 operands) and 24 v_mfma_f32_32x32x16_bf16
// (six plane pairs x 2 x 2 accumulator blocks), random data, one barrier per step.  Variants: waves per SIMD (occupancy 2/3/4
// by launch bounds + LDS size), and a 64 x 128 wave tile (18 reads for 48 MFMAs).  Reports "fp32-equivalent" TFLOP/s
// (2 M N K per product block, the unit bench.py's roofline_f32x3 uses; peak on paper 416.7) and the bf16 MFMA rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/debug/mfma_x3 tools/debug/mfma_x3.hip && tools/debug/mfma_x3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <random>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PITCH = 24, PLANE = 128 * PITCH;

template <int WPE, int NB>      // NB: 32-column blocks per wave (2 -> 64 x 64 wave tile, 4 -> 64 x 128)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void k(const unsigned short* __restrict__ src,
                                                                                   float* out, int iters) {
  constexpr int BROWS = NB * 64;                                  // B rows per workgroup (2 waves across)
  constexpr int TOT = 3 * PLANE + 3 * BROWS * PITCH;
  constexpr int WANT = (WPE == 4 ? 0 : 150 * 1024 / WPE / 2);     // pad LDS so that exactly WPE workgroups fit a CU (160 KB)
  __shared__ __attribute__((aligned(16))) unsigned short T[TOT > WANT ? TOT : WANT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < TOT; i += 256) T[i] = src[(blockIdx.x * 977 + i) % (1 << 20)];
  __syncthreads();
  const int r = lane & 31, hh = lane >> 5;
  const unsigned short* TA = T + (wm * 64 + r) * PITCH + 8 * hh;
  const unsigned short* TB = T + 3 * PLANE + (wn * 32 * NB + r) * PITCH + 8 * hh;
  f16v acc[2][NB];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < NB; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
    bf16x8 a[2][3], b[NB][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i][p] = *reinterpret_cast<const bf16x8*>(TA + 32 * i * PITCH + p * PLANE);
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(TB + 32 * j * PITCH + p * BROWS * PITCH);
    }
#define X3(PA, PB)                                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j) acc[i][j] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
    X3(2, 0) X3(0, 2) X3(1, 1) X3(1, 0) X3(0, 1) X3(0, 0)
#undef X3
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < NB; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
  out[blockIdx.x * 256 + tid] = s;
}

template <int WPE, int NB>
void run(const unsigned short* src, float* out, const char* name) {
  const int blocks = 4096, iters = 1500;
  const double flop32 = 2.0 * blocks * 4 * 64 * (32.0 * NB) * 16 * (double)iters;     // fp32-equivalent
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<WPE, NB>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e0);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<WPE, NB>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double t = flop32 * 3 / (ms * 1e-3) / 1e12;
  printf("%-44s %.2f ms per launch, %.1f fp32-equivalent TFLOP/s (%.2f of 416.7), bf16 MFMA rate %.0f TFLOP/s\n", name, ms / 3, t,
         t / 416.7, 6 * t);
}

int main() {
  std::vector<unsigned short> h(1 << 20);
  std::mt19937 g(1); std::uniform_real_distribution<float> d(-1.f, 1.f);
  for (auto& v : h) { float f = d(g); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  unsigned short* src; float* out;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<2, 2>(src, out, "64x64 wave tile, 2 workgroups per CU");
    run<3, 2>(src, out, "64x64 wave tile, 3 workgroups per CU");
    run<4, 2>(src, out, "64x64 wave tile, 4 workgroups per CU");
    run<2, 4>(src, out, "64x128 wave tile, 2 workgroups per CU");
  }
  return 0;
}

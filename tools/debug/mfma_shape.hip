// Which fp32 MFMA shape delivers more FLOP/s in a GEMM-like inner loop on random data: v_mfma_f32_32x32x2_f32 or
// v_mfma_f32_16x16x4_f32?  Same FLOP per clock on paper (64 per SIMD); the chip may hold a different clock on each.
// Each wave owns a 64x64 accumulator tile (64 registers either way) and re-reads its A/B fragments from LDS by ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <random>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k(const float* __restrict__ src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float T[2][128 * 20];     // A and B tiles, [row][16 k + 4 pad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * 128 * 20; i += 256) (&T[0][0])[i] = src[(blockIdx.x * 5120 + i) % (1 << 20)];
  __syncthreads();
  if (SHAPE == 32) {
    const int r = lane & 31, hh = lane >> 5;
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float4 va = *reinterpret_cast<const float4*>(&T[0][(wm * 64 + 32 * i + r) * 20 + 8 * hh + 4 * q]);
          float4 vb = *reinterpret_cast<const float4*>(&T[1][(wn * 64 + 32 * i + r) * 20 + 8 * hh + 4 * q]);
          a[i][0] = va.x; a[i][1] = va.y; a[i][2] = va.z; a[i][3] = va.w;
          b[i][0] = vb.x; b[i][1] = vb.y; b[i][2] = vb.z; b[i][3] = vb.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    // 16x16x4: lane l holds A[l & 15][k = l >> 4], 4 k per instruction; one ds_read_b128 = 4 consecutive k-blocks of its quarter
    const int r = lane & 15, kq = lane >> 4;
    f4v acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
      float a[4][4], b[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 va = *reinterpret_cast<const float4*>(&T[0][(wm * 64 + 16 * i + r) * 20 + 4 * kq]);
        float4 vb = *reinterpret_cast<const float4*>(&T[1][(wn * 64 + 16 * i + r) * 20 + 4 * kq]);
        a[i][0] = va.x; a[i][1] = va.y; a[i][2] = va.z; a[i][3] = va.w;
        b[i][0] = vb.x; b[i][1] = vb.y; b[i][2] = vb.z; b[i][3] = vb.w;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    out[blockIdx.x * 256 + tid] = s;
  }
}

int main() {
  std::vector<float> h(1 << 20);
  std::mt19937 g(1); std::uniform_real_distribution<float> d(-1.f, 1.f);
  for (auto& v : h) v = d(g);
  float *src, *out;
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int blocks = 4096, iters = 2000;           // each iteration: 64x64 outputs x 16 k per wave
  const double flop = 2.0 * blocks * 4 * 64 * 64 * 16 * (double)iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {32, 16}) {
      for (int w = 0; w < 2; ++w) {
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      }
      hipEventRecord(e0);
      for (int w = 0; w < 3; ++w) {
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mfma %s: %.2f ms per launch, %.1f TFLOP/s\n", shape == 32 ? "32x32x2 " : "16x16x4 ", ms / 3, flop * 3 / (ms * 1e-3) / 1e12);
    }
  return 0;
}

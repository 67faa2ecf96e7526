// What in the fp32 GEMM's k-loop keeps the matrix cores at ~73 % when there are no global loads and no epilogue?
// Start from a bare LDS-fed MFMA loop (154 TFLOP/s, tools/debug/mfma_shape.hip) and add the real kernel's pieces one by one.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <random>
typedef float f16v __attribute__((ext_vector_type(16)));

// V bit 0: 40 KB of LDS per workgroup (4 workgroups per CU, as the real kernel)   bit 1: double buffer + 4 ds_write_b128 per thread and step
// bit 2: ~60 VALU instructions of "stage_fix" per step                             bit 3: short workgroups (16 steps each, many of them)
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k(const float* __restrict__ src, float* out, int iters) {
  constexpr int TILE = 128 * 20;
  __shared__ __attribute__((aligned(16))) float T[(V & 1) ? 4 * TILE : 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * TILE; i += 256) T[i] = src[(blockIdx.x * 5120 + i) % (1 << 20)];
  if (V & 1) for (int i = tid; i < 2 * TILE; i += 256) T[2 * TILE + i] = T[i];
  __syncthreads();
  const int r = lane & 31, hh = lane >> 5;
  f16v acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  float4 sa[2], sb[2];
  for (int j = 0; j < 2; ++j) { sa[j] = make_float4(tid * 1e-3f, 1.f, 2.f, 3.f); sb[j] = make_float4(tid * 2e-3f, 1.f, 2.f, 3.f); }
  for (int it = 0; it < iters; ++it) {
    const int buf = (V & 2) ? (it & 1) : 0;
    const float* TA = T + ((V & 1) ? buf * 2 * TILE : 0);
    const float* TB = TA + TILE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float a[2][4], b[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float4 va = *reinterpret_cast<const float4*>(&TA[(wm * 64 + 32 * i + r) * 20 + 8 * hh + 4 * q]);
        float4 vb = *reinterpret_cast<const float4*>(&TB[(wn * 64 + 32 * i + r) * 20 + 8 * hh + 4 * q]);
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
    if (V & 4) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* pa = reinterpret_cast<float*>(&sa[j]); float* pb = reinterpret_cast<float*>(&sb[j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (tid + it + e) < 1000000 + iters;
          pa[e] = ok ? pa[e] * 1.0001f + 0.5f : 0.f;
          pb[e] = ok ? pb[e] * 0.9999f - 0.5f : 0.f;
          pa[e] = fminf(fmaxf(pa[e], -4.f), 4.f); pb[e] = fminf(fmaxf(pb[e], -4.f), 4.f);
        }
      }
    }
    if (V & 2) {
      float* WA = T + ((V & 1) ? (buf ^ 1) * 2 * TILE : 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int f = tid + 256 * j, row = f >> 2, kk = (f & 3) * 4;
        *reinterpret_cast<float4*>(&WA[row * 20 + kk]) = sa[j];
        *reinterpret_cast<float4*>(&WA[TILE + row * 20 + kk]) = sb[j];
      }
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
  if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char* what, const float* src, float* out) {
  const bool shortwg = V & 8;
  const int blocks = shortwg ? 4096 * 25 : 4096, iters = shortwg ? 16 : 400;
  const double flop = 2.0 * blocks * 4 * 64 * 64 * 16 * (double)iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e0);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-70s %.1f TFLOP/s\n", what, flop * 3 / (ms * 1e-3) / 1e12);
}

int main() {
  std::vector<float> h(1 << 20);
  std::mt19937 g(1); std::uniform_real_distribution<float> d(-1.f, 1.f);
  for (auto& v : h) v = d(g);
  float *src, *out;
  (void)hipMalloc(&src, h.size() * 4); (void)hipMalloc(&out, 4096 * 25 * 256 * 4);
  (void)hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>("bare: fragment reads + 32 MFMA + barrier per step", src, out);
  run<1>("+ 40 KB LDS per workgroup (4 per CU)", src, out);
  run<3>("+ double buffer, 4 ds_write_b128 per thread and step", src, out);
  run<7>("+ ~60 VALU per step", src, out);
  run<15>("+ workgroups of 16 steps (prologue fill, no epilogue)", src, out);
  run<9>("40 KB LDS, workgroups of 16 steps, no writes, no VALU", src, out);
  run<8>("20 KB LDS, workgroups of 16 steps, no writes, no VALU", src, out);
  return 0;
}

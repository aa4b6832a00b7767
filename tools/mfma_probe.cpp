// Microbenchmark (GPU box): what the fp32 matrix pipe of MI355X delivers to the shapes the 1-vs-all sweep (csrc/rank.hip) can take:
// v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, one or two waves per SIMD (256-thread workgroups, one or two per CU, or one
// 512-thread workgroup), operands in registers or re-read from LDS as the sweep reads them (ds_read_b128 of a padded row image), with
// and without a workgroup barrier per 80 MFMAs.  Random operands (zeros read high: the chip holds a higher clock on them).
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_probe.cpp -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

constexpr int LS = 44;  // the sweep's LDS row stride (BK = 40 floats + 4)

// MODE 0: registers only; 1: fragments from LDS (4 x ds_read_b128 per 16 MFMAs, as rank_sweep4_kernel); 2: as 1 + two barriers per 80 MFMAs
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k32(const float* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float As[128 * LS];
  __shared__ __attribute__((aligned(16))) float Bs[128 * LS];
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, half = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 128 * LS; i += THREADS) { As[i] = src[i]; Bs[i] = src[i + 128 * LS]; }
  __syncthreads();
  const float* a_rd = As + ((wave >> 1) * 64 + l31) * LS + 4 * half;
  const float* b_rd = Bs + ((wave & 1) * 64 + l31) * LS + 4 * half;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float4 fa[2], fb[2];
  for (int x = 0; x < 2; ++x) { fa[x] = *reinterpret_cast<const float4*>(a_rd + x * 32 * LS); fb[x] = *reinterpret_cast<const float4*>(b_rd + x * 32 * LS); }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) { __syncthreads(); __syncthreads(); }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float4 na[2], nb[2];
      if (MODE >= 1) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          na[x] = *reinterpret_cast<const float4*>(a_rd + x * 32 * LS + 8 * ((j + 1) % 5));
          nb[x] = *reinterpret_cast<const float4*>(b_rd + x * 32 * LS + 8 * ((j + 1) % 5));
        }
      }
#define STEP(c)                                                                                  \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0].c, fb[0].c, acc[0][0], 0, 0, 0);         \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0].c, fb[1].c, acc[0][1], 0, 0, 0);         \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1].c, fb[0].c, acc[1][0], 0, 0, 0);         \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1].c, fb[1].c, acc[1][1], 0, 0, 0);
      STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
      if (MODE >= 1) {
#pragma unroll
        for (int x = 0; x < 2; ++x) { fa[x] = na[x]; fb[x] = nb[x]; }
      }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[(size_t)blockIdx.x * THREADS + tid] = s;
}

// the same 64x64 wave tile on v_mfma_f32_16x16x4_f32: 4 x 4 accumulators of 4 registers; one float4 fragment per 16 rows feeds 4 k-steps
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k16(const float* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float As[128 * LS];
  __shared__ __attribute__((aligned(16))) float Bs[128 * LS];
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, quad = lane >> 4, l15 = lane & 15;
  for (int i = tid; i < 128 * LS; i += THREADS) { As[i] = src[i]; Bs[i] = src[i + 128 * LS]; }
  __syncthreads();
  // lane (quad, l15): row l15 of a 16-row block, k = 4*quad .. 4*quad+3 of a 16-wide k block held as one float4 (component c = k-step c)
  const float* a_rd = As + ((wave >> 1) * 64 + l15) * LS + 4 * quad;
  const float* b_rd = Bs + ((wave & 1) * 64 + l15) * LS + 4 * quad;
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  float4 fa[4], fb[4];
  for (int x = 0; x < 4; ++x) { fa[x] = *reinterpret_cast<const float4*>(a_rd + x * 16 * LS); fb[x] = *reinterpret_cast<const float4*>(b_rd + x * 16 * LS); }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) { __syncthreads(); __syncthreads(); }
    // 40 k per iteration = 2.5 blocks of 16 k: use 5 half-blocks of 8 k (two components each) to keep the flop count of k32 (80 x 4096 flop x ...)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float4 na[4], nb[4];
      if (MODE >= 1) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          na[x] = *reinterpret_cast<const float4*>(a_rd + x * 16 * LS + 8 * ((j + 1) % 5) % 28);
          nb[x] = *reinterpret_cast<const float4*>(b_rd + x * 16 * LS + 8 * ((j + 1) % 5) % 28);
        }
      }
      // 8 k-values per j in the 32x32x2 kernel = 4 MFMA k-steps x 2 k; here one 16x16x4 MFMA = 4 k, so 2 k-steps x 16 tiles = 32 MFMAs
#define STEP16(c)                                                                                       \
  _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                        \
    _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].c, fb[b].c, acc[a][b], 0, 0, 0);
      STEP16(x) STEP16(y)
#undef STEP16
      if (MODE >= 1) {
#pragma unroll
        for (int x = 0; x < 4; ++x) { fa[x] = na[x]; fb[x] = nb[x]; }
      }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  out[(size_t)blockIdx.x * THREADS + tid] = s;
}

template <typename F>
static void run(const char* name, F launch, int blocks, int threads, int iters, double flop_per_wave_iter) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) launch(blocks, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) launch(blocks, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double waves = (double)blocks * threads / 64;
  const double tf = waves * iters * flop_per_wave_iter * reps / (ms * 1e-3) / 1e12;
  printf("%-58s blocks %4d x %3d thr  %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", name, blocks, threads, ms / reps, tf, 100.0 * tf / 157.3);
}

int main() {
  const int n = 2 * 128 * LS;
  std::vector<float> h(n);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *src, *out;
  CK(hipMalloc(&src, n * 4)); CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 1024 * 512 * 4));
  const int iters = 4000;
  const double f32 = 80.0 * 2 * 32 * 32 * 2;   // per wave and iteration: 80 MFMAs of 32x32x2
  const double f16 = 160.0 * 2 * 16 * 16 * 4;  // 160 MFMAs of 16x16x4: the same flop
#define L(K, M, T) [&](int b, int it) { hipLaunchKernelGGL((K<M, T>), dim3(b), dim3(T), 0, 0, src, out, it); }
  for (int blocks : {256, 512}) {
    run("32x32x2  registers                256 thr", L(k32, 0, 256), blocks, 256, iters, f32);
    run("32x32x2  LDS fragments            256 thr", L(k32, 1, 256), blocks, 256, iters, f32);
    run("32x32x2  LDS fragments + barriers 256 thr", L(k32, 2, 256), blocks, 256, iters, f32);
    run("16x16x4  registers                256 thr", L(k16, 0, 256), blocks, 256, iters, f16);
    run("16x16x4  LDS fragments            256 thr", L(k16, 1, 256), blocks, 256, iters, f16);
    run("16x16x4  LDS fragments + barriers 256 thr", L(k16, 2, 256), blocks, 256, iters, f16);
  }
  run("32x32x2  registers                512 thr (one WG / CU)", L(k32, 0, 512), 256, 512, iters, f32);
  run("32x32x2  LDS fragments + barriers 512 thr (one WG / CU)", L(k32, 2, 512), 256, 512, iters, f32);
  run("16x16x4  registers                512 thr (one WG / CU)", L(k16, 0, 512), 256, 512, iters, f16);
  run("16x16x4  LDS fragments + barriers 512 thr (one WG / CU)", L(k16, 2, 512), 256, 512, iters, f16);
  // how long a launch has to be to reach these rates: the same register-only kernel at 1 / 10 / 100 % of the iterations, launches
  // back to back (what a 1 ms sweep sees of the clock ramp and of launch ramp-up / drain)
  for (int it : {40, 100, 400, 1000, 4000}) {
    char name[96];
    snprintf(name, sizeof name, "32x32x2  LDS fragments + barriers 512 thr, %4d iterations", it);
    run(name, L(k32, 2, 512), 256, 512, it, f32);
  }
  return 0;
}

// Microbenchmark (GPU box): cost of kernel-to-kernel hand-offs on MI355X / ROCm 7.2 -- same stream, across two streams with an
// event (hipEventRecord + hipStreamWaitEvent), with the host running ahead.  Build: hipcc -O2 --offload-arch=gfx950 tools/xstream_probe.cpp -o gpurun_out/xstream_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(int* p, int spin) { for (int i = 0; i < spin; ++i) atomicAdd(p, 1); }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
int main() {
  int* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  const int N = 2000;
  for (int flags : {0, 1}) {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, flags ? hipStreamNonBlocking : hipStreamDefault));
    CK(hipStreamCreateWithFlags(&b, flags ? hipStreamNonBlocking : hipStreamDefault));
    for (int timing : {0, 1}) {
      std::vector<hipEvent_t> ev(2 * N);
      for (auto& e : ev) CK(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
      for (int spin : {1, 200}) {
        // same stream
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d, spin);
        double t1 = now();
        CK(hipStreamSynchronize(a));
        double t2 = now();
        printf("flags %d timing %d spin %3d  same stream : %.2f us/kernel (host enqueue %.2f)\n", flags, timing, spin, (t2 - t0) / (2 * N), (t1 - t0) / (2 * N));
        // ping-pong across two streams
        CK(hipDeviceSynchronize());
        t0 = now();
        for (int i = 0; i < N; ++i) {
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d, spin);
          CK(hipEventRecord(ev[2 * i], a));
          CK(hipStreamWaitEvent(b, ev[2 * i], 0));
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d + 64, spin);
          CK(hipEventRecord(ev[2 * i + 1], b));
          CK(hipStreamWaitEvent(a, ev[2 * i + 1], 0));
        }
        t1 = now();
        CK(hipDeviceSynchronize());
        t2 = now();
        printf("flags %d timing %d spin %3d  ping-pong   : %.2f us/kernel (host enqueue %.2f)\n", flags, timing, spin, (t2 - t0) / (2 * N), (t1 - t0) / (2 * N));
        // ping-pong reusing TWO events (what a pipeline with per-slot events does)
        CK(hipDeviceSynchronize());
        t0 = now();
        for (int i = 0; i < N; ++i) {
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d, spin);
          CK(hipEventRecord(ev[0], a));
          CK(hipStreamWaitEvent(b, ev[0], 0));
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d + 64, spin);
          CK(hipEventRecord(ev[1], b));
          CK(hipStreamWaitEvent(a, ev[1], 0));
        }
        t1 = now();
        CK(hipDeviceSynchronize());
        t2 = now();
        printf("flags %d timing %d spin %3d  ping-pong 2e: %.2f us/kernel (host enqueue %.2f)\n", flags, timing, spin, (t2 - t0) / (2 * N), (t1 - t0) / (2 * N));
      }
      for (auto& e : ev) (void)hipEventDestroy(e);
    }
    (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
  }
  return 0;
}

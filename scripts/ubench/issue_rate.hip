// Single-wavefront issue-rate / latency microbenchmark for gfx950 (what a sequential chain pays per instruction).
// build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 512
__global__ void k_dep_add(double *out, long long *cyc, double x) {
  double a = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) a = a + x;
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_indep_add(double *out, long long *cyc, double x) {
  double a = out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N / 4; i++) { a = a + x; b = b + x; c = c + x; d = d + x; }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep_fma(double *out, long long *cyc, double x) {
  double a = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) a = __builtin_fma(a, x, x);
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep_f32(float *out, long long *cyc, float x) {
  float a = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) a = a * x + x;
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_readlane(double *out, long long *cyc, double x) {
  double a = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) {
    int lo = __builtin_amdgcn_readlane(__double2loint(a), i & 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(a), i & 63);
    a = a + __hiloint2double(hi, lo);
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dpp(double *out, long long *cyc, double x) {
  double a = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xf, 0xf, true);
    a = a + __hiloint2double(hi, lo);
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_chase(double *out, long long *cyc, int stride) {
  __shared__ int nxt[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) nxt[i] = (i + stride) & 4095;
  __syncthreads();
  int p = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) p = nxt[p];
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_b128(double *out, long long *cyc, int stride) {
  __shared__ double2 rec[2048 * 5];
  for (int i = threadIdx.x; i < 2048 * 5; i += 64) rec[i] = make_double2((double)((i * 7 + 3) % 2048), 1.0);
  __syncthreads();
  int p = (threadIdx.x * 37) % 2048;
  double acc = 0;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N / 4; i++) {  // 4 x b128 of one 80-byte-stride record, dependent on the previous record
    const double2 a = rec[p * 5], b = rec[p * 5 + 1], c = rec[p * 5 + 2], d = rec[p * 5 + 3];
    acc += b.x + c.x + d.x;
    p = (int)a.x;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc + p;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_branchy(double *out, long long *cyc, int m) {
  double a = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) {
    if ((int)threadIdx.x < m + (i & 1)) a = a + 1.0;  // exec-masked block + branch
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 64 * 8 * 2); hipMalloc(&cyc, 8);
  hipMemset(out, 0, 64 * 8 * 2);
  long long h = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(name, per, ...)                                                             \
  for (int rep = 0; rep < 3; rep++) {                                                   \
    hipEventRecord(e0, 0);                                                              \
    hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, __VA_ARGS__);                     \
    hipEventRecord(e1, 0); hipEventSynchronize(e1);                                     \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                         \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                       \
    if (rep == 2) printf("%-14s %8lld ticks / %d = %7.2f ticks per step   (kernel %.1f us)\n", #name, h, per, (double)h / per, ms * 1e3); \
  }
  RUN(k_dep_add, N, out, cyc, 1.0)
  RUN(k_indep_add, N, out, cyc, 1.0)
  RUN(k_dep_fma, N, out, cyc, 1.0)
  RUN(k_dep_f32, N, (float *)out, cyc, 1.0f)
  RUN(k_readlane, N, out, cyc, 1.0)
  RUN(k_dpp, N, out, cyc, 1.0)
  RUN(k_lds_chase, N, out, cyc, 65)
  RUN(k_lds_b128, N / 4, out, cyc, 1)
  RUN(k_branchy, N, out, cyc, 32)
  // what is a tick: run a long dependent chain and compare ticks with wall time
  {
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_dep_add, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("200 launches of k_dep_add: %.1f us each (includes launch gaps)\n", ms * 1e3 / 200);
  }
  return 0;
}

// Probe: how many 64-thread workgroups are co-resident per CU on this device?
// Each workgroup spins for a fixed number of shader cycles; the launch time steps up
// each time the grid exceeds the machine's resident capacity.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64) void spin64(unsigned long long cycles, int* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
template <int T> __global__ __launch_bounds__(T) void spinT(unsigned long long cycles, int* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
template <int WORDS> __global__ __launch_bounds__(64) void spin_scratch(unsigned long long cycles, int* out, int idx) {
  volatile int arr[WORDS];
  for (int i = 0; i < WORDS; i++) arr[i] = i + idx;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = arr[(idx * 7) % WORDS];
}
template <int WORDS> void run_scratch(int* d, unsigned long long cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {4, 8, 9, 12, 16, 17, 24, 32}) {
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL(spin_scratch<WORDS>, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(spin_scratch<WORDS>, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("scratch %4d B/lane, 64-thread WGs: %2d per CU: %.3f ms\n", WORDS * 4, wgs_per_cu, ms);
  }
}
template <int BYTES> __global__ __launch_bounds__(64) void spin_lds(unsigned long long cycles, int* out, int idx) {
  __shared__ int buf[BYTES / 4];
  buf[(threadIdx.x * 17 + idx) % (BYTES / 4)] = idx;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = buf[idx % (BYTES / 4)];
}
template <int BYTES> void run_lds(int* d, unsigned long long cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {4, 6, 7, 8, 9, 12, 15, 16, 17, 20, 32}) {
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL(spin_lds<BYTES>, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(spin_lds<BYTES>, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %6d B/WG, 64-thread WGs: %2d per CU: %.3f ms\n", BYTES, wgs_per_cu, ms);
  }
}
#define SPIN_VGPR(NAME, REG) \
__global__ __launch_bounds__(64) void NAME(unsigned long long cycles, int* out) { \
  asm volatile("v_mov_b32 " REG ", 0" ::: REG); \
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); } \
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1; }
SPIN_VGPR(spin_v64, "v63")
SPIN_VGPR(spin_v96, "v95")
SPIN_VGPR(spin_v128, "v127")
SPIN_VGPR(spin_v168, "v167")
__device__ __noinline__ double probe_callee(double x, int* p) { double a[8]; for (int i = 0; i < 8; i++) a[i] = x * i + p[i & 3]; double s = 0; for (int i = 0; i < 8; i++) s += a[(i * 3) & 7] / (1.0 + a[i]); return s; }
__global__ __launch_bounds__(64, 4) void spin_call(unsigned long long cycles, int* out, int idx) {
  __shared__ int buf[10184 / 4];
  buf[(threadIdx.x * 17 + idx) % (10184 / 4)] = idx;
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  __syncthreads();
  double acc = probe_callee((double)idx, buf);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = buf[idx % 100] + (int)acc;
}
void run_call(int* d, unsigned long long cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin_call, 64, 0);
  printf("CALL occupancy query: %d\n", occ);
  for (int wgs_per_cu : {8, 9, 12, 16, 17}) {
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL(spin_call, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(spin_call, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("CALL kernel with a device function call, 64-thread WGs: %2d per CU: %.3f ms\n", wgs_per_cu, ms);
  }
}
__global__ __launch_bounds__(64, 4) void spin_combo(unsigned long long cycles, int* out, int idx) {
  __shared__ int buf[10184 / 4];
  volatile int arr[132];
  for (int i = 0; i < 132; i++) arr[i] = i + idx;
  buf[(threadIdx.x * 17 + idx) % (10184 / 4)] = idx;
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  asm volatile("s_mov_b32 s99, 0" ::: "s99");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && out) out[blockIdx.x] = buf[idx % 100] + arr[(idx * 7) % 132];
}
void run_combo(int* d, unsigned long long cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {4, 8, 9, 12, 16, 17}) {
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL(spin_combo, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(spin_combo, dim3(G), dim3(64), 0, 0, cyc, d, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("COMBO lds10184 v128 s100 scratch528, 64-thread WGs: %2d per CU: %.3f ms\n", wgs_per_cu, ms);
  }
}
template <class K> void run_v(K k, const char* name, int* d, unsigned long long cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {4, 8, 9, 12, 13, 16, 17, 20, 21, 24, 32}) {
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, cyc, d); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, cyc, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s, 64-thread WGs: %2d per CU: %.3f ms\n", name, wgs_per_cu, ms);
  }
}
int main() {
  int* d; hipMalloc(&d, 1 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned long long cyc = 200000;   // 100 MHz memtime ticks? prints reveal the unit
  for (int wgs_per_cu : {1, 2, 4, 8, 9, 10, 12, 16, 17, 24, 32, 33, 40}) {
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL(spin64, dim3(G), dim3(64), 0, 0, cyc, d); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(spin64, dim3(G), dim3(64), 0, 0, cyc, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("64-thread WGs: %2d per CU (grid %5d): %.3f ms\n", wgs_per_cu, G, ms);
  }
  for (int wgs_per_cu : {1, 2, 4, 8, 9}) {
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL(spinT<256>, dim3(G), dim3(256), 0, 0, cyc, d); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(spinT<256>, dim3(G), dim3(256), 0, 0, cyc, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("256-thread WGs: %2d per CU (grid %5d): %.3f ms\n", wgs_per_cu, G, ms);
  }
  run_combo(d, cyc);
  run_call(d, cyc);
  { int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin_combo, 64, 0); printf("COMBO occupancy query: %d\n", occ); }
  run_v(spin_v64, "VGPR 64", d, cyc); run_v(spin_v96, "VGPR 96", d, cyc); run_v(spin_v128, "VGPR 128", d, cyc); run_v(spin_v168, "VGPR 168", d, cyc);
  return 0;
  run_lds<4096>(d, cyc); run_lds<8192>(d, cyc); run_lds<10184>(d, cyc); run_lds<16384>(d, cyc); run_lds<20480>(d, cyc);
  return 0;
}

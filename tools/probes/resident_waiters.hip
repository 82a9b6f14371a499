// Probe: does a kernel whose workgroups are RESIDENT BUT WAITING (one wave of each sleeping between looks at a word, the others at a
// barrier) slow down another kernel's workgroups on OTHER CUs?  Background: the Lemke ladder's task kernel once waited like that for
// lcp_fast's verdicts, and lcp_fast's last workgroups -- whole CUs to themselves, full shader clock -- took 40-70 s instead of 3
// (profiles/r04_d_waiting_workgroups_stall.txt).  The worker below is shaped like those workgroups: 1024 threads, one per CU (128 VGPRs
// by launch bounds, 64 KB of LDS), a chain of elimination steps over a k x k matrix in global memory with a block barrier per step.
//   hipcc --offload-arch=gfx950 -O2 -o resident_waiters resident_waiters.hip && ./resident_waiters
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void worker(double* A, int k, int reps, double* out)
{
  __shared__ double pad[8192];                                        // 64 KB: one workgroup per CU, as lcp_fast's
  double* M = A + (size_t)blockIdx.x * k * k;
  const int t = threadIdx.x;
  pad[t] = 0.0;
  for (int r = 0; r < reps; r++)
    for (int j = 0; j < k - 1; j++) {
      const double piv = M[j + (size_t)k * j];
      for (int e = t; e < (k - j - 1) * (k - j - 1); e += 1024) {
        const int rr = j + 1 + e % (k - j - 1), cc = j + 1 + e / (k - j - 1);
        M[rr + (size_t)k * cc] = M[rr + (size_t)k * cc] - (M[rr + (size_t)k * j] / (piv + 3.0)) * M[j + (size_t)k * cc] * 1e-3;
      }
      __syncthreads();
    }
  if (t == 0) out[blockIdx.x] = M[k * k - 1] + pad[5];
}

// mode 1: thread 0 sleeps about a millisecond between looks at *flag, the other waves wait at a barrier; 2: the same without the other waves
// (they leave); 3: thread 0 looks without sleeping, the others at a barrier; 4: every wave sleeps and looks
template <int T> __global__ __launch_bounds__(T) void waiters(const int* flag, int mode, int max_looks, int* out)
{
  const int t = threadIdx.x;
  if (mode == 2 && t >= 64) return;
  if (t == 0 || mode == 4) {
    int looks = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && looks < max_looks) {
      if (mode != 3) for (int i = 0; i < 300; i++) __builtin_amdgcn_s_sleep(127);
      looks++;
    }
    if (t == 0) out[blockIdx.x] = looks;
  }
  if (mode != 2) __syncthreads();
}

int main()
{
  const int k = 300, NW = 20, reps = 40;
  double *A, *out; int *flag, *wout;
  CK(hipMalloc((void**)&A, (size_t)NW * k * k * 8)); CK(hipMalloc((void**)&out, NW * 8)); CK(hipMalloc((void**)&wout, 4096 * 4));
  CK(hipHostMalloc((void**)&flag, 4, hipHostMallocCoherent));
  std::vector<double> h((size_t)NW * k * k); for (size_t i = 0; i < h.size(); i++) h[i] = 1.0 + (double)(i % 17) * 0.01;
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run_worker = [&](int nw) {
    CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(worker, dim3(nw), dim3(1024), 0, s1, A, k, reps, out); CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
  };
  run_worker(NW);
  printf("worker alone: %d workgroups %.1f ms, 1 workgroup %.1f ms\n", NW, run_worker(NW), run_worker(1));
  struct Case { const char* what; int threads, grid, mode; };
  const Case cases[] = { {"800 x 128 threads, thread 0 sleeps ~1 ms between looks, the other wave at a barrier", 128, 800, 1},
                         {"800 x 256 threads, the same", 256, 440, 1},
                         {"800 x 64 threads resident (the other waves have left)", 128, 800, 2},
                         {"800 x 128 threads, thread 0 looks without sleeping, the other wave at a barrier", 128, 800, 3},
                         {"800 x 128 threads, every wave sleeps and looks", 128, 800, 4},
                         {"100 x 128 threads, thread 0 sleeps, the other wave at a barrier", 128, 100, 1} };
  for (const Case& c : cases) {
    *flag = 0;
    if (c.threads == 128) hipLaunchKernelGGL(waiters<128>, dim3(c.grid), dim3(128), 0, s2, flag, c.mode, 20000, wout);
    else hipLaunchKernelGGL(waiters<256>, dim3(c.grid), dim3(256), 0, s2, flag, c.mode, 20000, wout);
    CK(hipGetLastError());
    const float ms20 = run_worker(NW), ms1 = run_worker(1);
    *flag = 1;
    CK(hipStreamSynchronize(s2));
    printf("beside waiters (%s): %d workgroups %.1f ms, 1 workgroup %.1f ms\n", c.what, NW, ms20, ms1);
    fflush(stdout);
  }
  printf("worker alone again: %.1f ms\n", run_worker(NW));
  return 0;
}

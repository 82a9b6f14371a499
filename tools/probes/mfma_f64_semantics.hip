// What does v_mfma_f64_16x16x4_f64 compute per output element?  ("Next round" 2(i).)
// Random 16x4 * 4x16 + 16x16 tiles on the device, against CPU candidates:
//   chain_up   c = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c))))   (k ascending, every step rounded)
//   chain_down the same with k descending
//   unfused    ((((c + a0*b0) + a1*b1) + a2*b2) + a3*b3), products rounded
//   exact      c + sum a_k b_k rounded once (float128 accumulate)
// Prints one JSON line with the number of bit-exact outputs per candidate.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_f64_semantics mfma_f64_semantics.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ C, double* __restrict__ D, int tiles)
{
  const int tile = blockIdx.x, l = threadIdx.x;
  if (tile >= tiles) return;
  const double* a = A + (size_t)tile * 64; const double* b = B + (size_t)tile * 64; const double* c = C + (size_t)tile * 256;
  // A[i][k]: lane l holds i = l & 15, k = l >> 4;  B[k][j]: lane l holds j = l & 15, k = l >> 4
  const double av = a[(l & 15) * 4 + (l >> 4)];
  const double bv = b[(l >> 4) * 16 + (l & 15)];
  d4 acc;
  for (int r = 0; r < 4; r++) acc[r] = c[((l >> 4) + 4 * r) * 16 + (l & 15)];   // C/D: col = lane & 15, row = (lane >> 4) + 4 reg
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(size_t)tile * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

static bool same(double x, double y) { uint64_t a, b; memcpy(&a, &x, 8); memcpy(&b, &y, 8); return a == b || (x != x && y != y); }

int main(int argc, char** argv)
{
  const int tiles = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> A((size_t)tiles * 64), B((size_t)tiles * 64), C((size_t)tiles * 256), D((size_t)tiles * 256);
  for (int t = 0; t < tiles; t++) {
    const int mode = t % 4;          // 0: O(1) data; 1: wide exponents; 2: cancellation (c = -(a0 b0) rounded); 3: tiny c
    for (int e = 0; e < 64; e++) { A[(size_t)t * 64 + e] = U(rng) * (mode == 1 ? std::ldexp(1.0, (int)(rng() % 40) - 20) : 1.0); B[(size_t)t * 64 + e] = U(rng) * (mode == 1 ? std::ldexp(1.0, (int)(rng() % 40) - 20) : 1.0); }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
      double c = U(rng);
      if (mode == 2) c = -(A[(size_t)t * 64 + i * 4 + 0] * B[(size_t)t * 64 + 0 * 16 + j]);
      if (mode == 3) c *= 1e-17;
      C[(size_t)t * 256 + i * 16 + j] = c;
    }
  }
  double *dA, *dB, *dC, *dD;
  if (hipMalloc(&dA, A.size() * 8) != hipSuccess) { printf("{\"error\": \"no device\"}\n"); return 1; }
  hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(tiles), dim3(64), 0, 0, dA, dB, dC, dD, tiles);
  if (hipDeviceSynchronize() != hipSuccess) { printf("{\"error\": \"kernel failed\"}\n"); return 1; }
  hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
  long n = 0, up = 0, down = 0, unf = 0, ex = 0, pair = 0;
  long up_mode[4] = {0, 0, 0, 0}, n_mode[4] = {0, 0, 0, 0};
  for (int t = 0; t < tiles; t++) for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    const double* a = &A[(size_t)t * 64 + i * 4]; const double* b = &B[(size_t)t * 64 + j];
    const double c = C[(size_t)t * 256 + i * 16 + j], d = D[(size_t)t * 256 + i * 16 + j];
    double u = c; for (int k = 0; k < 4; k++) u = std::fma(a[k], b[k * 16], u);
    double w = c; for (int k = 3; k >= 0; k--) w = std::fma(a[k], b[k * 16], w);
    volatile double v = c; for (int k = 0; k < 4; k++) { volatile double p = a[k] * b[k * 16]; v = v + p; }
    __float128 q = c; for (int k = 0; k < 4; k++) q += (__float128)a[k] * (__float128)b[k * 16];
    const double p01 = std::fma(a[1], b[16], a[0] * b[0]), p23 = std::fma(a[3], b[48], a[2] * b[32]);
    const double pw = c + (p01 + p23);
    n++; n_mode[t % 4]++;
    if (same(d, u)) { up++; up_mode[t % 4]++; }
    if (same(d, w)) down++;
    if (same(d, (double)v)) unf++;
    if (same(d, (double)q)) ex++;
    if (same(d, pw)) pair++;
  }
  printf("{\"probe\": \"v_mfma_f64_16x16x4_f64 per-output arithmetic\", \"tiles\": %d, \"outputs\": %ld, \"match_fma_chain_k_ascending\": %ld, \"match_fma_chain_k_descending\": %ld, "
         "\"match_unfused_ascending\": %ld, \"match_single_rounding_exact_sum\": %ld, \"match_pairwise\": %ld, "
         "\"chain_ascending_by_mode\": {\"unit\": [%ld, %ld], \"wide_exponents\": [%ld, %ld], \"cancellation\": [%ld, %ld], \"tiny_c\": [%ld, %ld]}}\n",
         tiles, n, up, down, unf, ex, pair, up_mode[0], n_mode[0], up_mode[1], n_mode[1], up_mode[2], n_mode[2], up_mode[3], n_mode[3]);
  return 0;
}

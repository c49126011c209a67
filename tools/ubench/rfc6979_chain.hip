// What bounds the device signer at 2^16 items: the RFC 6979 retry chain.
//   A  every lane stops after its FIRST candidate (16 compressions)            = what a compacted first round costs
//   B  the library's loop (16 + 8 per rejected candidate, lockstep per wave)     = the nonce phase of the signer today
// and the distribution of rejected candidates per item: the slowest item of the batch sets the critical path of ANY
// schedule (compacted or not), because the retry chain K = HMAC(K, V 00), V = HMAC(K, V), V = HMAC(K, V) is serial.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 rfc6979_chain.hip -o rfc6979_chain
#include "../../stark-perpetual_amd/csrc/rfc6979.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace sp;

template <int MAXR>
__global__ void __launch_bounds__(128) k(const uint64_t* z, const uint64_t* d, uint64_t* out, int* rej, size_t n) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) e = n - 1;
  int r = 0;
  const u256 c = rfc6979_nonce<MAXR>(ld_u256(z + 4 * e), ld_u256(d + 4 * e), 0, &r);
  st_u256(out + 4 * e, c);
  rej[e] = r;
}

template <int MAXR>
double run(const uint64_t* z, const uint64_t* d, uint64_t* out, int* rej, size_t n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MAXR>, dim3((n + 127) / 128), dim3(128), 0, 0, z, d, out, rej, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MAXR>, dim3((n + 127) / 128), dim3(128), 0, 0, z, d, out, rej, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3;
}

int main() {
  for (int logn : {12, 16, 20}) {
    const size_t n = (size_t)1 << logn;
    std::vector<uint64_t> hz(4 * n), hd(4 * n);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < 4 * n; ++i) { hz[i] = next(); hd[i] = next(); }
    for (size_t i = 0; i < n; ++i) { hz[4 * i + 3] &= (1ull << 58) - 1; hd[4 * i + 3] &= (1ull << 58) - 1; }
    uint64_t *z, *d, *out; int* rej;
    hipMalloc(&z, 32 * n); hipMalloc(&d, 32 * n); hipMalloc(&out, 32 * n); hipMalloc(&rej, 4 * n);
    hipMemcpy(z, hz.data(), 32 * n, hipMemcpyHostToDevice);
    hipMemcpy(d, hd.data(), 32 * n, hipMemcpyHostToDevice);
    const double a = run<1>(z, d, out, rej, n);
    const double b = run<64>(z, d, out, rej, n);
    std::vector<int> hr(n);
    hipMemcpy(hr.data(), rej, 4 * n, hipMemcpyDeviceToHost);
    long long total = 0; int mx = 0; std::vector<long long> hist(40, 0);
    std::vector<int> wave_max(n / 64 ? n / 64 : 1, 0);
    for (size_t i = 0; i < n; ++i) { total += hr[i]; mx = std::max(mx, hr[i]); hist[std::min(hr[i], 39)]++; wave_max[i / 64] = std::max(wave_max[i / 64], hr[i]); }
    double wm = 0; for (int v : wave_max) wm += v;
    printf("2^%d items: first candidate only (16 compressions) %8.1f us | library loop %8.1f us | rejected candidates per item: "
           "mean %.3f, max %d (critical path 16 + 8 x %d = %d compressions), mean of the per-wave maximum %.2f (lockstep work "
           "16 + 8 x that = %.0f compressions per lane against %.0f needed)\n",
           logn, a, b, (double)total / n, mx, mx, 16 + 8 * mx, wm / wave_max.size(), 16 + 8 * wm / wave_max.size(),
           16 + 8.0 * total / n);
    printf("   per-compression time of a lone wave from A: %.2f us; B / (16 + 8 max) = %.2f us per compression\n",
           a / 16, b / (16 + 8 * mx));
    hipFree(z); hipFree(d); hipFree(out); hipFree(rej);
  }
  return 0;
}

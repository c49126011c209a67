// Probe: lane movement of the DPP quad_perm control and the ds_swizzle bit-mode xor patterns the
// quad-parallel point kernels rely on.  Prints source lane seen by each lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int PAT> __device__ int swz(int v) { return __builtin_amdgcn_ds_swizzle(v, PAT); }
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[0 * 64 + l] = dpp<(2) | (3 << 2) | (0 << 4) | (1 << 6)>(l);   // quad_perm:[2,3,0,1]
  out[1 * 64 + l] = dpp<(0) | (0 << 2) | (2 << 4) | (2 << 6)>(l);   // quad_perm:[0,0,2,2]
  out[2 * 64 + l] = swz<(4 << 10) | 0x1F>(l);                       // xor 4
  out[3 * 64 + l] = swz<(8 << 10) | 0x1F>(l);                       // xor 8
  out[4 * 64 + l] = swz<(16 << 10) | 0x1F>(l);                      // xor 16
  out[5 * 64 + l] = __shfl_xor(l, 32, 64);
  out[6 * 64 + l] = dpp<0x141>(l);                                  // row_half_mirror
}
int main() {
  int* d; hipMalloc(&d, 7 * 64 * 4);
  k<<<1, 64>>>(d);
  int h[7 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"quad_perm[2,3,0,1]", "quad_perm[0,0,2,2]", "swizzle xor 4", "swizzle xor 8", "swizzle xor 16", "shfl_xor 32", "row_half_mirror"};
  for (int r = 0; r < 7; ++r) { printf("%-20s", names[r]); for (int l = 0; l < 64; ++l) printf(" %d", h[r * 64 + l]); printf("\n"); }
  return 0;
}

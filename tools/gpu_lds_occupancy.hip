// Resident one-wave workgroups per CU as a function of the dynamic LDS size and the register budget (hipOccupancyMaxActiveBlocksPerMultiprocessor):
// where the allocation granule of the LDS puts the steps.   hipcc --offload-arch=gfx950 -O2 tools/gpu_lds_occupancy.hip -o tools/bin/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) void probe(float* out) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = out[threadIdx.x];
  out[threadIdx.x] = sm[63 - threadIdx.x];
}
template <int W> void scan() {
  int last = -1;
  for (int bytes = 8192; bytes <= 20480; bytes += 16) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, probe<W>, 64, bytes) != hipSuccess) { printf("query failed\n"); return; }
    if (per_cu != last) { printf("waves_per_eu %d: from %5d B: %d per CU\n", W, bytes, per_cu); last = per_cu; }
  }
}
int main() { scan<2>(); scan<3>(); scan<4>(); return 0; }

// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950 —
//  (1) does the builtin compile and land lane-linear at lds_base + 16 * lane?
//  (2) what does an OUT-OF-RANGE lane do: write zeros, or leave the LDS bytes untouched?
//   hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = -7.f;      // poison
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // lanes 0..31 in range (reversed order: lane l reads granule 31 - l), lanes 32..63 out of range
  const int off = lane < 32 ? (31 - lane) * 16 : 0x7ffffff0;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&lds[0], 16, off, 0, 0, 0);
  // second piece at +1 KiB through the instruction offset
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&lds[256], 16, lane * 16, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) lgkmcnt(0) expcnt(0)
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, 1024); hipMalloc(&o, 2048);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1024, o);
  std::vector<float> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  bool lin = true;
  for (int l = 0; l < 32; ++l) for (int j = 0; j < 4; ++j) lin &= r[l * 4 + j] == (float)((31 - l) * 4 + j);
  std::printf("in-range lanes lane-linear (lds[16*lane] <- src[per-lane offset]): %s\n", lin ? "yes" : "NO");
  std::printf("out-of-range lanes wrote: %g %g %g %g  (poison was -7)\n", r[32 * 4], r[32 * 4 + 1], r[63 * 4 + 2], r[63 * 4 + 3]);
  bool second = true;
  for (int i = 0; i < 256; ++i) second &= r[256 + i] == (float)i;
  std::printf("second piece at lds + 1 KiB: %s\n", second ? "ok" : "WRONG");
  return 0;
}

// Micro-benchmark: cost of ds_read_b128 / b64 at 16 / 8 / 4 / 2-byte alignment
// on gfx950 (one wave, 4096 reads per lane, conflict-light stride).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_align.hip -o /tmp/lds_align && /tmp/lds_align
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed, aligned(2))) U16x8 { uint32_t v[4]; };
struct __attribute__((packed, aligned(2))) U16x4 { uint32_t v[2]; };

template <typename T>
__global__ void k(int byte_off, int lane_stride, unsigned long long *out, uint32_t *sink) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += 64) ((uint32_t *)lds)[i] = i * 2654435761u;
  __syncthreads();
  const uint8_t *p = lds + (threadIdx.x * lane_stride) % 16384 + byte_off;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const T v = *reinterpret_cast<const T *>(p + u * 1024 + (it & 7) * 32);
      acc += v.v[0] ^ v.v[sizeof(T) / 4 - 1];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc;
}

int main() {
  unsigned long long *d_out; uint32_t *d_sink;
  hipMalloc(&d_out, 8); hipMalloc(&d_sink, 256);
  const int offs[] = {0, 8, 4, 2, 6, 14};
  for (int stride : {16, 80, 112}) {
    for (int o : offs) {
      unsigned long long t = 0;
      hipLaunchKernelGGL(k<U16x8>, dim3(1), dim3(64), 0, 0, o, stride, d_out, d_sink);
      hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
      printf("b128 lane_stride %3d byte_off %2d: %6.1f ticks per wave-read\n", stride, o, t / 4096.0);
    }
    for (int o : offs) {
      unsigned long long t = 0;
      hipLaunchKernelGGL(k<U16x4>, dim3(1), dim3(64), 0, 0, o, stride, d_out, d_sink);
      hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
      printf("b64  lane_stride %3d byte_off %2d: %6.1f ticks per wave-read\n", stride, o, t / 4096.0);
    }
  }
  return 0;
}

// Micro-benchmark: what a launch of N one-wave workgroups that do (almost) nothing costs on
// gfx950, by resource footprint - registers, LDS, scratch.  The motion search launches 8160
// such workgroups at 1080p; an instance with nothing to do measured 40 us.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wg_launch.hip -o /tmp/wg_launch && /tmp/wg_launch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// VG: highest VGPR touched (forces the allocation); SCR: a stack object (private segment)
template <int VG, bool SCR, bool READ>
__global__ void __launch_bounds__(256) k(const int *in, int *out, int n) {
  extern __shared__ int lds[];
  if (VG >= 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  else if (VG >= 95) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  else if (VG >= 63) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  int v = 0;
  if (READ) v = in[blockIdx.x & 1023];
  if (v == 12345) {  // never
    if (SCR) {
      volatile int a[16];
      a[n & 15] = n;
      v = a[(n + 1) & 15];
    }
    lds[threadIdx.x] = v;
    out[threadIdx.x] = lds[(threadIdx.x + 1) & 63] + v;
  }
}

template <int VG, bool SCR, bool READ>
static void run(const char *name, int wgs, int threads, int lds_bytes, const int *in, int *out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; i++)
    hipLaunchKernelGGL((k<VG, SCR, READ>), dim3(wgs), dim3(threads), lds_bytes, 0, in, out, 0);
  hipEventRecord(a, 0);
  const int reps = 50;
  for (int i = 0; i < reps; i++)
    hipLaunchKernelGGL((k<VG, SCR, READ>), dim3(wgs), dim3(threads), lds_bytes, 0, in, out, 0);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s wgs %5d x %3d thr lds %6d: %7.2f us per launch\n", name, wgs, threads, lds_bytes,
         ms * 1e3 / reps);
}

int main() {
  int *in, *out;
  hipMalloc(&in, 4096); hipMemset(in, 0, 4096); hipMalloc(&out, 4096);
  for (int lds : {0, 6912, 16384}) {
    run<0, false, false>("plain", 8160, 64, lds, in, out);
    run<0, false, true>("read", 8160, 64, lds, in, out);
    run<63, false, true>("read v64", 8160, 64, lds, in, out);
    run<95, false, true>("read v96", 8160, 64, lds, in, out);
    run<127, false, true>("read v128", 8160, 64, lds, in, out);
    run<127, true, true>("read v128 scratch", 8160, 64, lds, in, out);
    run<95, true, true>("read v96 scratch", 8160, 64, lds, in, out);
    run<0, true, true>("read scratch", 8160, 64, lds, in, out);
  }
  run<127, false, true>("read v128", 4080, 128, 2 * 6912, in, out);
  run<127, false, true>("read v128", 2040, 256, 4 * 6912, in, out);
  run<127, true, true>("read v128 scratch", 2040, 256, 4 * 6912, in, out);
  run<127, false, true>("read v128", 32640, 64, 6912, in, out);
  run<127, true, true>("read v128 scratch", 32640, 64, 6912, in, out);
  return 0;
}

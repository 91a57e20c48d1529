// Micro-benchmark: issue rate of the VALU instructions the motion search and the
// transforms are made of, on gfx950 - the measured number behind the "VALU floor"
// of DESIGN.md section 6 (the micro-architecture guide says 2 cycles per wave64
// VALU instruction, the first-round estimate assumed 4).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate.bin
// For each op: W waves per SIMD (workgroup of 4*W waves on one CU), every wave runs
// ITER x 32 instructions of the op on 8 independent register chains; cycles are
// measured two ways: HIP events around the launch (ns, converted to clocks with
// the device's reported engine clock) and s_memtime ticks of wave 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ITER 16384

#define OPS(X)                                                            \
  X(0, "v_add_u32", "v_add_u32 %[r], %[r], %[b]")                               \
  X(1, "v_sad_u16", "v_sad_u16 %[r], %[r], %[b], %[r]")                           \
  X(2, "v_dot2_i32_i16", "v_dot2_i32_i16 %[r], %[r], %[b], %[r]")                 \
  X(3, "v_pk_add_u16", "v_pk_add_u16 %[r], %[r], %[b]")                         \
  X(4, "v_pk_max_i16", "v_pk_max_i16 %[r], %[r], %[b]")                         \
  X(5, "v_mul_lo_u32", "v_mul_lo_u32 %[r], %[r], %[b]")                         \
  X(6, "v_mad_u64_u32", "v_mad_u64_u32 %[q], vcc, %[r], %[b], %[q]")              \
  X(7, "v_perm_b32", "v_perm_b32 %[r], %[r], %[b], %[r]")                         \
  X(8, "v_alignbit_b32", "v_alignbit_b32 %[r], %[r], %[b], 16")                 \
  X(9, "v_cndmask_b32", "v_cndmask_b32 %[r], %[r], %[b], vcc")                  \
  X(10, "v_lshlrev_b32", "v_lshlrev_b32 %[r], 1, %[r]")                       \
  X(11, "v_and_b32", "v_and_b32 %[r], %[r], %[b]")                              \
  X(12, "v_bfe_u32", "v_bfe_u32 %[r], %[r], 1, 31")                           \
  X(13, "v_add3_u32", "v_add3_u32 %[r], %[r], %[b], %[r]")                        \
  X(14, "v_max_i32", "v_max_i32 %[r], %[r], %[b]")                              \
  X(15, "v_lshl_add_u32", "v_lshl_add_u32 %[r], %[r], 1, %[b]")                 \
  X(16, "v_mad_u32_u24", "v_mad_u32_u24 %[r], %[r], %[b], %[r]")                  \
  X(17, "v_mov_b32 dpp", "v_mov_b32_dpp %[r], %[r] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(18, "v_add_u32 dpp", "v_add_u32_dpp %[r], %[r], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(19, "v_lshlrev_b64", "v_lshlrev_b64 %[q], 1, %[q]")                       \
  X(20, "v_xor_b32", "v_xor_b32 %[r], %[r], %[b]")                              \
  X(21, "v_sub_u16", "v_sub_u16 %[r], %[r], %[b]")                              \
  X(22, "v_min_u32", "v_min_u32 %[r], %[r], %[b]")                              \
  X(23, "v_pk_sub_i16", "v_pk_sub_i16 %[r], %[r], %[b]")                        \
  X(24, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %[r], 1, %[r]")                 \
  X(25, "v_dot2c_i32_i16", "v_dot2c_i32_i16 %[r], %[r], %[b]")               \
  X(26, "v_mul_i32_i24 sdwa", "v_mul_i32_i24_sdwa %[r], %[r], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1") \
  X(27, "v_lshl_add_u64", "v_lshl_add_u64 %[q], %[q], 1, %[q]")          \
  X(28, "v_ashrrev_i32", "v_ashrrev_i32 %[r], 1, %[r]")                   \
  X(29, "v_sub_u32", "v_sub_u32 %[r], %[r], %[b]")                       \
  X(30, "v_mov_b32", "v_mov_b32 %[r], %[b]")                             \
  X(31, "v_lshrrev_b32", "v_lshrrev_b32 %[r], 1, %[r]")                   \
  X(32, "v_or_b32", "v_or_b32 %[r], %[r], %[b]")                          \
  X(33, "v_cmp_lt_i32", "v_cmp_lt_i32 vcc, %[r], %[b]")                   \
  X(34, "v_cndmask_b32 e64", "v_cndmask_b32_e64 %[r], %[r], %[b], s[20:21]") \
  X(35, "v_add_co_u32", "v_add_co_u32 %[r], vcc, %[r], %[b]")             \
  X(36, "v_lshlrev_b32 (reg)", "v_lshlrev_b32 %[r], %[b], %[r]")          \
  X(37, "v_max_u32", "v_max_u32 %[r], %[r], %[b]")

template <int OP>
__global__ void k(unsigned long long *out, uint32_t *sink, uint32_t seed) {
  uint32_t r[8];
  unsigned long long q[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r[i] = seed * (threadIdx.x + 1) + i;
    q[i] = r[i];
  }
  const uint32_t b = seed | 1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
#define X(ID, NAME, ASM) \
  if (OP == ID) asm volatile(ASM : [r] "+v"(r[i]), [q] "+v"(q[i]) : [b] "v"(b) : "vcc");
        if (OP != 6) { OPS(X) }
#undef X
        if (OP == 6)
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(r[i]), "v"(b) : "vcc");
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  uint32_t a = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) a ^= r[i] ^ (uint32_t)q[i];
  sink[threadIdx.x & 63] = a;
}

int main() {
  unsigned long long *d_out;
  uint32_t *d_sink;
  hipMalloc(&d_out, 8);
  hipMalloc(&d_sink, 256);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("s_memtime ticks; device clock %d MHz (s_memtime runs at a fixed 100 MHz on this "
         "family: ticks are converted with the measured add rate below)\n", clk_khz / 1000);
  const char *names[48];
#define X(ID, NAME, ASM) names[ID] = NAME;
  OPS(X)
#undef X
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int op = 0; op < 38; op++) {
    for (int w = 1; w <= 8; w *= 2) {
      unsigned long long t = 0;
      // W waves per SIMD: one workgroup of 4 * W waves on a CU; a workgroup holds at
      // most 1024 threads, so W = 8 is two workgroups of 16 waves - launched as TWO
      // PER CU over the whole chip (256 CUs: the dispatcher fills every CU with its
      // two before anything retires), the other rows as one workgroup
      const bool chip = w == 8;
      const dim3 block(chip ? 1024 : 64 * 4 * w), grid(chip ? 512 : 1);
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {   // first run warms up clocks / code
        hipEventRecord(e0, 0);
#define L(ID) if (op == ID) hipLaunchKernelGGL(k<ID>, grid, block, 0, 0, d_out, d_sink, 12345u);
        L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
        L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31) L(32) L(33) L(34) L(35) L(36) L(37)
#undef L
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
      const double n = (double)ITER * 32 * w;
      printf("%-16s %d waves/SIMD: %7.3f ns = %6.2f clocks @%d MHz per wave-instruction per SIMD "
             "(%.4f s_memtime ticks)\n", names[op], w, 1e6 * ms / n,
             1e6 * ms / n * (clk_khz / 1e6), clk_khz / 1000, (double)t / n);
    }
  }
  return 0;
}

// Micro-benchmark: what the launch time of deblock_tail_kernel (k_tail.h) is
// made of.  Built once per TAIL_SKIP variant (tools/micro/tail_time.sh), each
// leaving one step of the kernel out; 1920x1080, 16x16 CUs with random vectors
// (every CU edge a candidate), 10 bit.
//   hipcc --offload-arch=gfx950 -O3 -DTAIL_SKIP=<mask> -I xvc_amd/csrc -I include \
//         tools/micro/tail_time.hip -o tail_time.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "k_tail.h"   // -DTAIL_TRACE: per-phase clock readings (perturbs the timing)

static PlaneView plane(uint16_t *base, int w, int h, int b) {
  PlaneView p;
  p.stride = (w + 2 * b + 127) / 128 * 128;
  p.p = base + (size_t)b * p.stride + b;
  p.w = w;
  p.h = h;
  p.border = b;
  return p;
}

static PicView picture(int w, int h, uint16_t **mem) {
  const size_t lum = (size_t)((w + 256 + 127) / 128 * 128) * (h + 256);
  const size_t chr = (size_t)((w / 2 + 128 + 127) / 128 * 128) * (h / 2 + 128);
  hipMalloc(mem, 2 * (lum + 2 * chr));
  std::vector<uint16_t> host(lum + 2 * chr);
  for (size_t i = 0; i < host.size(); i++) host[i] = 400 + (rand() % 64);
  hipMemcpy(*mem, host.data(), 2 * host.size(), hipMemcpyHostToDevice);
  PicView v;
  v.c[0] = plane(*mem, w, h, 128);
  v.c[1] = plane(*mem + lum, w / 2, h / 2, 64);
  v.c[2] = plane(*mem + lum + chr, w / 2, h / 2, 64);
  v.bd = 10;
  return v;
}

int main() {
  const int w = 1920, h = 1080, cu = 16;
  uint16_t *m0, *m1, *m2;
  PicView src = picture(w, h, &m0), dst = picture(w, h, &m1), orig = picture(w, h, &m2);
  const int ncx = (w + cu - 1) / cu, ncy = (h + cu - 1) / cu;
  std::vector<xvcgpu_cu_info> cus(ncx * ncy);
  const int ms = (w + 3) / 4, mr = (h + 3) / 4;
  std::vector<int32_t> map(ms * mr, -1);
  for (int j = 0; j < ncy; j++)
    for (int i = 0; i < ncx; i++) {
      xvcgpu_cu_info c = xvcgpu_cu_info();
      c.x = i * cu;
      c.y = j * cu;
      c.w = cu;
      c.h = (j * cu + cu <= h) ? cu : h - j * cu;
      c.cbf_luma = rand() & 1;
      c.qp_y = 32;
      c.qp_c = 31;
      c.ref_poc[0] = 0;
      c.ref_poc[1] = -1;
      const int mx = rand() % 200 - 100, my = rand() % 200 - 100;
      for (int k = 0; k < 4; k++) c.mv[0][k][0] = mx, c.mv[0][k][1] = my;
      cus[j * ncx + i] = c;
      for (int y = c.y / 4; y < (c.y + c.h) / 4; y++)
        for (int x = c.x / 4; x < (c.x + c.w) / 4; x++) map[y * ms + x] = j * ncx + i;
    }
  xvcgpu_cu_info *d_cus;
  int32_t *d_map;
  unsigned long long *part;
  hipMalloc(&d_cus, sizeof(xvcgpu_cu_info) * cus.size());
  hipMalloc(&d_map, 4 * map.size());
  hipMemcpy(d_cus, cus.data(), sizeof(xvcgpu_cu_info) * cus.size(), hipMemcpyHostToDevice);
  hipMemcpy(d_map, map.data(), 4 * map.size(), hipMemcpyHostToDevice);
  const int tiles = ((w + 63) / 64) * ((h + 63) / 64);
  hipMalloc(&part, 8 * (2 * tiles + 2));
  hipMemset(part, 0, 8 * (2 * tiles + 2));
  DbParams d;
  d.bd = 10; d.pic_w = w; d.pic_h = h; d.bipred = 0; d.beta_off = 0; d.tc_off = 0; d.sub = 4;
  d.y_begin = 0; d.y_end = h; d.cus = d_cus; d.map = d_map; d.map_stride = ms; d.map_rows = mr;
  d.comp_mask = 3;
#ifdef TAIL_TRACE
  unsigned long long *trace;
  hipMalloc(&trace, 64 * tiles);
  hipMemset(trace, 0, 64 * tiles);
  hipMemcpyToSymbol(HIP_SYMBOL(g_tail_trace), &trace, sizeof(trace));
#endif
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 200;
  for (int pass = 0; pass < 2; pass++) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++)
      hipLaunchKernelGGL(deblock_tail_kernel<true>, dim3(tiles), dim3(256), 0, 0, d, src, dst,
                         orig.c[0], 4, part);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
#ifdef TAIL_TRACE
  {  // read the per-workgroup clock readings of the last launch
    std::vector<unsigned long long> tr(8 * tiles);
    hipMemcpy(tr.data(), trace, 64 * tiles, hipMemcpyDeviceToHost);
    const char *names[6] = {"loads+cells", "V", "H", "ssd+own stores", "pad", "parts"};
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < tiles; i++) {
      if (tr[8 * i] < t0) t0 = tr[8 * i];
      if (tr[8 * i + 6] > t1) t1 = tr[8 * i + 6];
    }
    printf("  clock span first start .. last end: %llu ticks\n", t1 - t0);
    const int ntx = (w + 63) / 64, nty = (h + 63) / 64;
    for (int cls = 0; cls < 3; cls++) {   // interior, rim (not corner), corner
      double sum[7] = {0}, mx[7] = {0};
      int n = 0;
      for (int i = 0; i < tiles; i++) {
        const int tx = i % ntx, ty = i / ntx;
        const int ex = tx == 0 || tx == ntx - 1, ey = ty == 0 || ty == nty - 1;
        if (ex + ey != cls) continue;
        n++;
        for (int k = 0; k < 6; k++) {
          const double dt = (double)(tr[8 * i + k + 1] - tr[8 * i + k]);
          sum[k] += dt;
          if (dt > mx[k]) mx[k] = dt;
        }
        sum[6] += (double)(tr[8 * i] - t0);
        if ((double)(tr[8 * i + 6] - t0) > mx[6]) mx[6] = (double)(tr[8 * i + 6] - t0);
      }
      printf("  %s tiles (%d): start at avg %.0f, last end %.0f;", cls == 0 ? "interior" : cls == 1 ? "rim" : "corner", n, sum[6] / n, mx[6]);
      for (int k = 0; k < 6; k++) printf("  %s %.0f/%.0f", names[k], sum[k] / n, mx[k]);
      printf("  (avg/max ticks)\n");
    }
  }
#endif
  float ms_total = 0;
  hipEventElapsedTime(&ms_total, e0, e1);
  std::vector<unsigned long long> parts(2 * tiles);
  hipMemcpy(parts.data(), part, 16 * tiles, hipMemcpyDeviceToHost);
  unsigned long long res[2] = {0, 0};
  for (int i = 0; i < tiles; i++) res[0] += parts[2 * i], res[1] += parts[2 * i + 1];
  printf("TAIL_SKIP=%d  %.2f us per launch (back to back)  ssd=%llu n=%llu  %s\n", TAIL_SKIP,
         1e3 * ms_total / reps, res[0], res[1], hipGetErrorString(hipGetLastError()));
  return 0;
}

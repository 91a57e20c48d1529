// xvcgpu_tables.cpp -- host-side generation of the 8-bit-fraction transform
// matrices used by the kernels (reference: transform_data.cc:109-796 ships
// them as literal tables; they are the JEM definitions
//   coef = (int)(256*sqrt(N)*v + (v > 0 ? 0.5 : -0.5)),
// v = orthonormal DCT-2 / DCT-5 / DCT-8 / DST-1 / DST-7 basis).
// tests/test_abi.py::test_transform_tables_match_oracle checks every entry
// against the oracle, tests/test_oracle_vs_ref.py::test_transform_tables the
// oracle against the reference build's literals.  Pure host code: no GPU needed.
#include <math.h>
#include <string.h>

#include <vector>

#include "xvcgpu_internal.h"

namespace {

struct Tables {
  TxTableLayout layout;
  std::vector<int16_t> data;    // M[k][n], row-major
  std::vector<int16_t> data_t;  // transposed: M[n][k]
  Tables() {
    memset(&layout, 0xff, sizeof(layout));
    const double pi = 3.14159265358979323846;
    int total = 0;
    for (int t = XVC_TX_DCT2; t <= XVC_TX_DST7; t++) {
      for (int l = 1; l <= 6; l++) {
        const int N = 1 << l;
        if (N == 2 && t != XVC_TX_DCT2) continue;
        total = (total + 7) & ~7;  // 16-byte aligned rows for vector loads
        layout.off[t - 1][l] = total;
        data.resize(total + N * N);
        int16_t *m = &data[total];
        const double s = sqrt((double)N) * 256.0;
        for (int k = 0; k < N; k++) {
          for (int n = 0; n < N; n++) {
            double v;
            const double w0 = (k == 0) ? sqrt(0.5) : 1.0;
            const double w1 = (n == 0) ? sqrt(0.5) : 1.0;
            switch (t) {
              case XVC_TX_DCT2:
                v = cos(pi * (n + 0.5) * k / N) * w0 * sqrt(2.0 / N);
                break;
              case XVC_TX_DCT5:
                v = cos(pi * n * k / (N - 0.5)) * w0 * w1 *
                    sqrt(2.0 / (N - 0.5));
                break;
              case XVC_TX_DCT8:
                v = cos(pi * (k + 0.5) * (n + 0.5) / (N + 0.5)) *
                    sqrt(2.0 / (N + 0.5));
                break;
              case XVC_TX_DST1:
                v = sin(pi * (n + 1) * (k + 1) / (N + 1)) * sqrt(2.0 / (N + 1));
                break;
              default:
                v = sin(pi * (k + 0.5) * (n + 1) / (N + 0.5)) *
                    sqrt(2.0 / (N + 0.5));
                break;
            }
            m[k * N + n] = (int16_t)(int)(s * v + (v > 0 ? 0.5 : -0.5));
          }
        }
        total += N * N;
      }
    }
    // XVC_TX_DCT2_LOW: the 6-bit DCT-2 of HEVC (reference literals
    // transform_data.cc:26-107).  Its 32-point basis takes 33 magnitudes,
    // kLow[j] ~ 64 * sqrt(2) * cos(j * pi / 64) as the standard rounds them; the
    // N-point matrix is every (32 / N)-th row: M[k][n] = +-kLow at the angle
    // k * (2n + 1) * (32 / N) folded into the first quadrant.
    static const int16_t kLow[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80,
                                     78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46,
                                     43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
    for (int l = 2; l <= 5; l++) {
      const int N = 1 << l;
      total = (total + 7) & ~7;
      layout.off[XVC_TX_DCT2_LOW - 1][l] = total;
      data.resize(total + N * N);
      int16_t *m = &data[total];
      for (int k = 0; k < N; k++)
        for (int n = 0; n < N; n++) {
          const int a = (k * (2 * n + 1) * (32 / N)) & 127;   // units of pi / 64
          m[k * N + n] = (int16_t)(a <= 32 ? kLow[a] : a <= 64 ? -kLow[64 - a]
                                           : a <= 96 ? -kLow[a - 64] : kLow[128 - a]);
        }
      total += N * N;
    }
    layout.total = total;
    data_t.assign(data.size(), 0);
    for (int t = 0; t < 7; t++)
      for (int l = 1; l <= 6; l++) {
        const int off = layout.off[t][l];
        if (off < 0) continue;
        const int N = 1 << l;
        for (int k = 0; k < N; k++)
          for (int n = 0; n < N; n++) data_t[off + n * N + k] = data[off + k * N + n];
      }
  }
};

const Tables &tables() {
  static Tables t;
  return t;
}

}  // namespace

const TxTableLayout &xvcgpu_tx_layout() { return tables().layout; }
const int16_t *xvcgpu_tx_host_tables() { return tables().data.data(); }
const int16_t *xvcgpu_tx_host_tables_t() { return tables().data_t.data(); }

extern "C" xvcgpu_status xvcgpu_get_transform_matrix(int tx_type, int size,
                                                     int16_t *out) {
  if (!out) return XVCGPU_INVALID_ARGUMENT;
  if (tx_type == XVC_TX_DEFAULT) tx_type = XVC_TX_DCT2;
  if (tx_type != XVC_TX_DCT2_LOW && (tx_type < XVC_TX_DCT2 || tx_type > XVC_TX_DST7))
    return XVCGPU_INVALID_ARGUMENT;
  int l = 1;
  while ((1 << l) < size) l++;
  if ((1 << l) != size || l > 6) return XVCGPU_INVALID_ARGUMENT;
  const int off = tables().layout.off[tx_type - 1][l];
  if (off < 0) return XVCGPU_INVALID_ARGUMENT;
  memcpy(out, tables().data.data() + off, sizeof(int16_t) * size * size);
  return XVCGPU_OK;
}

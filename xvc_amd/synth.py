"""Deterministic integer-only synthetic clips.

frame n = crop of a static textured base plane at (2n mod 64, n mod 64) (a
global pan of 2 px / 1 px per frame) + one 32x32 inverted-contrast square
moving (5,3) px/frame + fresh +-2 noise from a 32-bit LCG; U,V are affine
functions of the sub-sampled luma.  8-bit content, returned at the encoder's
internal bit depth (8-bit << (bd-8), as Resampler::ConvertFrom does for 8-bit
input with internal depth 10, encoder.cc:445-480).
"""
import numpy as np

_A, _C = np.uint32(1664525), np.uint32(1013904223)


def _lcg_field(seed, shape):
    """Per-pixel 32-bit LCG stream: element i gets the i-th state from `seed`."""
    n = int(np.prod(shape))
    # closed form jump-ahead is overkill: iterate blocks of rows vectorised by
    # seeding each row from a hashed row index, then stepping along the row.
    rows, cols = shape
    s = (np.arange(rows, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)) \
        .astype(np.uint32)
    out = np.empty(shape, np.uint32)
    with np.errstate(over="ignore"):
        for x in range(cols):
            s = s * _A + _C
            out[:, x] = s
    del n
    return out


def _tri(v, period):
    """Integer triangle wave in [0, period//2]."""
    m = v % period
    return np.minimum(m, period - m)


class SyntheticClip:
    def __init__(self, width, height, bitdepth=10, seed=1234, square=True):
        self.w, self.h, self.bd = width, height, bitdepth
        self.square = square
        H, W = height + 64, width + 128
        yy, xx = np.mgrid[0:H, 0:W].astype(np.int64)
        low = (_tri(xx * 3 + yy, 211) * 120) // 105 + (_tri(yy * 5 - xx, 157) * 60) // 78
        high = (_tri(xx + 2 * yy, 14) * 30) // 7
        noise = (_lcg_field(seed, (H, W)) >> np.uint32(27)).astype(np.int64) - 16
        self.base = np.clip(30 + low + high + noise // 2, 0, 255).astype(np.int64)
        self.seed = seed

    def frame(self, n):
        """Returns [Y,U,V] uint16 planes at the internal bit depth."""
        w, h = self.w, self.h
        ox, oy = (2 * n) % 64, n % 64
        y = self.base[oy:oy + h, ox:ox + w].copy()
        sx, sy = (40 + 5 * n) % max(1, w - 32), (24 + 3 * n) % max(1, h - 32)
        if self.square:
            y[sy:sy + 32, sx:sx + 32] = 255 - y[sy:sy + 32, sx:sx + 32]
        nz = (_lcg_field(self.seed + 7919 * (n + 1), (h, w)) >> np.uint32(30)).astype(np.int64)
        y = np.clip(y + nz - 2 + (nz == 0), 0, 255)
        sub = (y[0::2, 0::2] + y[1::2, 0::2] + y[0::2, 1::2] + y[1::2, 1::2] + 2) >> 2
        u = np.clip(128 + (sub - 128) // 3, 0, 255)
        v = np.clip(128 - (sub - 128) // 4, 0, 255)
        sh = self.bd - 8
        return [(p << sh).astype(np.uint16) for p in (y, u, v)]

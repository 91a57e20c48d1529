"""Shared input generators for the parity tests (seeded, numpy only)."""
import numpy as np

import oracle_lib as ol

SIZES = [4, 8, 16, 32, 64]


def rnd_samples(rng, bd, h, w, smooth=False):
    if smooth:
        base = rng.integers(0, 1 << bd)
        a = base + rng.integers(-6, 7, size=(h, w))
        return np.clip(a, 0, (1 << bd) - 1).astype(np.uint16)
    return rng.integers(0, 1 << bd, size=(h, w), dtype=np.uint16)


def random_partition(rng, pw, ph, min_size=4):
    """Random quad/binary partition of the picture into CUs (x,y,w,h)."""
    out = []

    def split(x, y, w, h, depth):
        if x >= pw or y >= ph:
            return
        inside = x + w <= pw and y + h <= ph
        r = rng.random()
        can_h = h > min_size
        can_w = w > min_size
        if not inside or depth < 1 or (depth < 2 and r < 0.8) or (r < 0.45 and (can_h or can_w)):
            mode = rng.integers(0, 3) if inside else 0
            if mode == 0 and can_h and can_w:
                for (dx, dy) in ((0, 0), (w // 2, 0), (0, h // 2), (w // 2, h // 2)):
                    split(x + dx, y + dy, w // 2, h // 2, depth + 1)
                return
            if mode == 1 and can_w:
                split(x, y, w // 2, h, depth + 1)
                split(x + w // 2, y, w // 2, h, depth + 1)
                return
            if can_h:
                split(x, y, w, h // 2, depth + 1)
                split(x, y + h // 2, w, h // 2, depth + 1)
                return
            if can_w:
                split(x, y, w // 2, h, depth + 1)
                split(x + w // 2, y, w // 2, h, depth + 1)
                return
        out.append((x, y, w, h))

    for cy in range(0, ph, 64):
        for cx in range(0, pw, 64):
            split(cx, cy, 64, 64, 0)
    return out


def make_cus(rng, parts, bipred, l0, l1, pw, ph):
    cus = np.zeros(len(parts), ol.CU_DTYPE)
    cmap = -np.ones(((ph + 3) // 4, (pw + 3) // 4), np.int32)
    for i, (x, y, w, h) in enumerate(parts):
        c = cus[i]
        c["x"], c["y"], c["w"], c["h"] = x, y, w, h
        c["intra"] = rng.random() < 0.15
        c["cbf_luma"] = rng.random() < 0.4
        qp = int(rng.integers(20, 45))
        c["qp_y"] = qp
        c["qp_c"] = ol.chroma_qp(qp)
        if bipred:
            d = rng.integers(0, 3)
        else:
            d = 0
        i0 = int(rng.integers(0, len(l0)))
        i1 = int(rng.integers(0, len(l1)))
        c["ref_idx0"] = i0 if d != 1 else 0
        c["ref_poc"][0] = l0[i0] if d != 1 else -1
        c["ref_poc"][1] = l1[i1] if d != 0 else -1
        base = rng.integers(-40, 41, size=(2, 1, 2))
        if rng.random() < 0.2:
            mv = base + rng.integers(-20, 21, size=(2, 4, 2))
        else:
            mv = np.repeat(base, 4, axis=1)
        c["mv"] = mv
        cmap[y // 4:(y + h) // 4, x // 4:(x + w) // 4] = i
    return cus, cmap


def make_pics(rng, bd, pw, ph, border, motion=(3, -2), noise=3):
    """orig = shifted ref + noise over a textured plane -> searches do real work."""
    H, W = ph + 2 * border, pw + 2 * border
    yy, xx = np.mgrid[0:H, 0:W]
    tex = (np.sin(xx / 7.0) * np.cos(yy / 9.0) * 0.25 + np.sin((xx + yy) / 23.0) * 0.2 + 0.5)
    tex = tex * ((1 << bd) - 1) + rng.integers(-noise * 4, noise * 4 + 1, size=(H, W))
    ref = np.clip(tex, 0, (1 << bd) - 1).astype(np.uint16)
    orig = np.roll(ref, (-motion[1], -motion[0]), axis=(0, 1)).astype(np.int32)
    orig = np.clip(orig + rng.integers(-noise, noise + 1, size=(H, W)), 0,
                   (1 << bd) - 1).astype(np.uint16)
    return orig, ref

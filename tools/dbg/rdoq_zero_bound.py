#!/usr/bin/env python3
"""Experiment: how many of the blocks RdoQuant ends up zeroing completely could be
PROVEN zero from the first pass alone (a rigorous lower bound on the cost of any
coded outcome against the cost of the all-zero block)?  Settled chain state, luma
16x16 blocks.  Run on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xvc_amd import api, pipeline, synth  # noqa: E402

ENT = [0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a,
       0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9, 0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3,
       0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600,
       0x03050, 0x10f95, 0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df,
       0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00, 0x01c99, 0x166de,
       0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547,
       0x0147c, 0x1a083, 0x0138e, 0x1a8a3, 0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b,
       0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
       0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d,
       0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577, 0x007c9, 0x24ce6, 0x00763, 0x25663,
       0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5,
       0x0055e, 0x29057, 0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f,
       0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb]
ENT = np.array(ENT, np.int64)
W, H, bd = 1920, 1080, 10
CHAIN = int(os.environ.get("CHAIN", 120))
QP = int(os.environ.get("QP", 32))
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), 128)
fp = pipeline.FramePass(ctx, W, H, bd, qp=QP, rdoq=True)
for n in range(CHAIN + 1):
    O.upload(pad(clip.frame(n % 7 + 1)), 128)
    fp.run(O, R, Rec)
    ctx.sync()
    if n < CHAIN:
        R, Rec = Rec, R
d = fp.desc
cf = fp.d_coeffs.to_array(np.int16, fp.n_levels)
lv = fp.d_levels.to_array(np.int16, fp.n_levels)
off, _ = ctx.level_offsets(d.tx)
prm = fp.d_rdoq_prm.to_array(api.RDOQ_PARAMS_DTYPE, len(d.tx))
cx = fp.d_rdoq_ctx.to_array(api.RDOQ_CTX_DTYPE, 1)[0]
FWD = [26214, 23302, 20560, 18396, 16384, 14564]
INV = [40, 45, 51, 57, 64, 72]


def bits(state, b):
    return ENT[(int(state) & 127) ^ b]


def bc(b, lam):
    return (int(b) * int(lam)) >> 16


for comp, size, sigs, g1s, cbf_state in ((0, 16, cx["sig_luma"], cx["greater1_luma"], cx["root_cbf"]),
                                         (1, 8, cx["sig_chroma"], cx["greater1_chroma"], cx["cbf_chroma"])):
    sel = np.flatnonzero((d.tx["comp"] == comp) & (d.tx["w"] == size) & (d.tx["h"] == size))
    lw = int(np.log2(size))
    sig1_min = min(bits(s, 1) for s in sigs)
    g_min = min(min(bits(s, 0), bits(s, 1)) for s in g1s)
    n = zero = proven = proven_sign_only = wrong = 0
    for i in sel[::5]:
        qp = int(d.tx[i]["qp"]) + 6 * (bd - 8)
        tshift = 15 - bd - lw
        shift = 14 + qp // 6 + tshift
        cs = 15 - 2 * tshift - 2 * (bd - 8)
        iq_shift = 6 - tshift
        iq_scale = INV[qp % 6] << (qp // 6)
        lam = int(prm[i]["lambda"])
        a = np.abs(cf[off[i]:off[i] + size * size].astype(np.int64))
        q = (a * FWD[qp % 6] + (1 << (shift - 1))) >> shift
        if not q.any():
            continue
        n += 1
        is_zero = not lv[off[i]:off[i] + size * size].any()
        zero += is_zero
        zd = (a * a) << cs
        nzq = q > 0

        def dist(level):
            deq = (level * iq_scale + (1 << (iq_shift - 1))) >> iq_shift
            e = a - deq
            return (e * e) << cs
        dmin = np.minimum(dist(q), np.where(q > 1, dist(np.maximum(q - 1, 1)), dist(q)))
        zero_cost = int(zd.sum()) + bc(bits(cbf_state, 0), lam)
        base = int(zd[~nzq].sum()) + bc(bits(cbf_state, 1), lam)
        for name, r in (("sign", 32768), ("ctx", 32768 + sig1_min + g_min)):
            m = np.minimum(zd[nzq], dmin[nzq] + bc(r, lam))
            lb = base + int(m.sum()) - (bc(sig1_min, lam) + 1 if name == "ctx" else 0)
            if lb > zero_cost:
                if name == "sign":
                    proven_sign_only += 1
                else:
                    proven += 1
                    wrong += not is_zero
    # ---- third bound: the contexts a coefficient can meet, from the template of q > 0
    # neighbours (a decided level is non-zero only where q is)
    luma = comp == 0
    szc = lw                        # (lw + lh) >> 1
    proven3 = wrong3 = 0
    for i in sel[::5]:
        qp = int(d.tx[i]["qp"]) + 6 * (bd - 8)
        tshift = 15 - bd - lw
        shift = 14 + qp // 6 + tshift
        cs = 15 - 2 * tshift - 2 * (bd - 8)
        iq_shift = 6 - tshift
        iq_scale = INV[qp % 6] << (qp // 6)
        lam = int(prm[i]["lambda"])
        a = np.abs(cf[off[i]:off[i] + size * size].astype(np.int64)).reshape(size, size)
        q = (a * FWD[qp % 6] + (1 << (shift - 1))) >> shift
        if not q.any():
            continue
        is_zero = not lv[off[i]:off[i] + size * size].any()
        zd = (a * a) << cs
        qp1 = np.pad(q, ((0, 2), (0, 2)))
        lb = int(zd[q == 0].sum()) + bc(bits(cbf_state, 1), lam)
        max_sig = 0
        for y, x in zip(*np.nonzero(q)):
            nb = [qp1[y, x + 1], qp1[y, x + 2], qp1[y + 1, x + 1], qp1[y + 1, x], qp1[y + 2, x]]
            cnt = sum(1 for v in nb if v > 0)
            cnt1 = sum(1 for v in nb if v > 1)
            posxy = x + y
            start = (6 if posxy < 2 else 0) + (6 if luma and posxy < 5 else 0) + \
                    ((18 << min(szc - 3, 1)) if szc > 2 and luma else 0)
            sig1 = min(bits(sigs[start + min(nn, 5)], 1) for nn in range(cnt + 1))
            gst = (10 if posxy < 3 else (5 if posxy < 10 else 0)) if luma else 0
            ctxs = [g1s[0]] + [g1s[gst + min(nn, 4) + 1] for nn in range(cnt1 + 1)]
            lvl_min = 32768 + min(min(bits(c, 0), bits(c, 1)) for c in ctxs + [None] if c is not None)
            lvl_min = min(lvl_min, 32768 + 32768)
            best = best_last = None
            for lvl in ([int(q[y, x])] + ([int(q[y, x]) - 1] if q[y, x] > 1 else [])):
                deq = (lvl * iq_scale + (1 << (iq_shift - 1))) >> iq_shift
                e = int(a[y, x]) - deq
                c_ = ((e * e) << cs) + bc(sig1 + lvl_min, lam)
                c_last = ((e * e) << cs) + bc(lvl_min, lam)      # as the last position: no sig flag
                best = c_ if best is None else min(best, c_)
                best_last = c_last if best_last is None else min(best_last, c_last)
            m = min(int(zd[y, x]), best)
            lb += m
            # exactly one coded coefficient is the last one: the largest saving that can bring
            max_sig = max(max_sig, m - min(int(zd[y, x]), best_last))
        lb -= max_sig
        zero_cost = int(zd.sum()) + bc(bits(cbf_state, 0), lam)
        if lb > zero_cost:
            proven3 += 1
            wrong3 += not is_zero
    # ---- fourth bound: the same per-coefficient costs, and for every possible last
    # position L (a q > 0 coefficient) its last-position bits: zero is proven when
    # for all L  sum_{i < L} gain_i + (zd_L - coded_as_last_L) < cbf1 - cbf0 + lp(L)
    def grp(pos):
        if pos < 4:
            return pos
        l = int(np.floor(np.log2(pos)))
        return 2 * l + ((pos >> (l - 1)) & 1)

    def lp_axis(g, is_x):
        n = size
        gmax = grp(n - 1)
        gc = max(gmax - 1, 0)
        if luma:
            l2 = lw
            offc = 0 if l2 < 3 else (3 if l2 == 3 else (6 if l2 == 4 else (10 if l2 == 5 else 15)))
            tab = cx["last_x_luma"] if is_x else cx["last_y_luma"]
            cidx = lambda k: offc + (k >> ((l2 + 1) >> 2))      # noqa: E731
        else:
            sh = min(max(n >> 3, 0), 2)
            tab = cx["last_x_chroma"] if is_x else cx["last_y_chroma"]
            cidx = lambda k: k >> sh                             # noqa: E731
        b = sum(bits(tab[cidx(min(k, gc))], 1) for k in range(g))
        if g < gmax:
            b += bits(tab[cidx(min(g, gc))], 0)
        if g > 3:
            b += ((g - 2) >> 1) * 32768
        return b
    gsb = size // 4

    def scan_index(x, y):
        sx, sy = x >> 2, y >> 2
        s_ = sx + sy
        idx = 0
        for dd in range(s_):
            c = min(dd, gsb - 1, gsb - 1, 2 * gsb - 2 - dd)
            idx += c + 1
        sbi = idx + (min(s_, gsb - 1) - sy)
        xx, yy = x & 3, y & 3
        k = 0
        for ss in range(xx + yy):
            k += sum(1 for y2 in range(min(ss, 3), -1, -1) if ss - y2 < 4)
        k += sum(1 for y2 in range(min(xx + yy, 3), yy, -1) if xx + yy - y2 < 4)
        return sbi * 16 + k
    proven4 = wrong4 = 0
    for i in sel[::5]:
        qp = int(d.tx[i]["qp"]) + 6 * (bd - 8)
        tshift = 15 - bd - lw
        shift = 14 + qp // 6 + tshift
        cs = 15 - 2 * tshift - 2 * (bd - 8)
        iq_shift = 6 - tshift
        iq_scale = INV[qp % 6] << (qp // 6)
        lam = int(prm[i]["lambda"])
        a = np.abs(cf[off[i]:off[i] + size * size].astype(np.int64)).reshape(size, size)
        q = (a * FWD[qp % 6] + (1 << (shift - 1))) >> shift
        if not q.any():
            continue
        is_zero = not lv[off[i]:off[i] + size * size].any()
        zd = (a * a) << cs
        qp1 = np.pad(q, ((0, 2), (0, 2)))
        items = []
        for y, x in zip(*np.nonzero(q)):
            nb = [qp1[y, x + 1], qp1[y, x + 2], qp1[y + 1, x + 1], qp1[y + 1, x], qp1[y + 2, x]]
            cnt = sum(1 for v in nb if v > 0)
            cnt1 = sum(1 for v in nb if v > 1)
            posxy = x + y
            start = (6 if posxy < 2 else 0) + (6 if luma and posxy < 5 else 0) + \
                    ((18 << min(szc - 3, 1)) if szc > 2 and luma else 0)
            sig1 = min(bits(sigs[start + min(nn, 5)], 1) for nn in range(cnt + 1))
            gst = (10 if posxy < 3 else (5 if posxy < 10 else 0)) if luma else 0
            ctxs = [g1s[0]] + [g1s[gst + min(nn, 4) + 1] for nn in range(cnt1 + 1)]
            lvl_min = min(32768 + min(min(bits(c, 0), bits(c, 1)) for c in ctxs), 65536)
            best = best_last = None
            for lvl in ([int(q[y, x])] + ([int(q[y, x]) - 1] if q[y, x] > 1 else [])):
                deq = (lvl * iq_scale + (1 << (iq_shift - 1))) >> iq_shift
                e = int(a[y, x]) - deq
                c_ = ((e * e) << cs) + bc(sig1 + lvl_min, lam)
                c_last = ((e * e) << cs) + bc(lvl_min, lam)
                best = c_ if best is None else min(best, c_)
                best_last = c_last if best_last is None else min(best_last, c_last)
            z = int(zd[y, x])
            items.append((scan_index(int(x), int(y)), z - min(z, best), z - best_last,
                          bc(lp_axis(grp(int(x)), True) + lp_axis(grp(int(y)), False), lam)))
        items.sort()
        rhs0 = bc(bits(cbf_state, 1), lam) - bc(bits(cbf_state, 0), lam)
        ok = True
        run_gain = 0
        for _, g_, g_last, lpb in items:
            if run_gain + g_last >= rhs0 + lpb:
                ok = False
                break
            run_gain += g_
        if ok:
            proven4 += 1
            wrong4 += not is_zero
    print("   per-last-position bound: provably zero %d (%.0f%% of the zeroed ones), wrong %d" %
          (proven4, 100.0 * proven4 / max(1, zero), wrong4))
    print("   template-context bound: provably zero %d (%.0f%% of the zeroed ones), wrong %d" %
          (proven3, 100.0 * proven3 / max(1, zero), wrong3))
    print("comp %d %dx%d: %d walked blocks, %d end all-zero (%.0f%%); provably zero with the sign-bit "
          "bound %d, with the context-minimum bound %d (%.0f%% of the zeroed ones); proven but NOT "
          "zero (must be 0): %d" % (comp, size, size, n, zero, 100.0 * zero / max(1, n),
                                    proven_sign_only, proven, 100.0 * proven / max(1, zero), wrong))

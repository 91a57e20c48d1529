"""The all-zero proof of the RDO quantiser (rdoq_prove_zero_kernel /
rq_prove_zero_lds, xvc_amd/csrc/k_rdoq.h) is sound: a numpy restatement of the
bound the device evaluates, held against the oracle's QuantRdo (and, where it is
built, the reference's) on random blocks - whenever the bound says "QuantRdo returns
0" the quantiser does return 0, for every scan order, component, size 4..32, bit
depth and random context states; and it does prove a fair share of the blocks that
end all zero (the test is not vacuous).

The bound (see the comment above rdoq_prove_zero_kernel): candidates are the
coefficients whose plain quantised magnitude q is > 0; a candidate saves at most
zd - (min(dist(q), dist(q - 1)) + lambda * (cheapest significance "1" + sign +
cheapest continuation bin)); a last position L costs lambda * (cbf(1) - cbf(0) +
lastpos(L)) on top; QuantRdo is bound to return 0 if for every candidate L the
savings in front of it plus its own (no significance flag) stay below that."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import oracle_rdoq as oq

FWD = [26214, 23302, 20560, 18396, 16384, 14564]
INV = [40, 45, 51, 57, 64, 72]
BYPASS = 32768


@pytest.fixture(scope="module")
def xo():
    return ol.Lib("xo")


def entropy_table(xo):
    f = xo.dll.xo_entropy_bits_table
    f.restype = C.POINTER(C.c_uint32)
    return np.array([f()[i] for i in range(128)], np.int64)


def scan4(order):
    """TransformHelper::kScanCoeff4x4 (transform.cc:72-76): scan offset -> (x, y)."""
    if order == 0:
        out = []
        for s in range(7):
            y = min(s, 3)
            while y >= 0 and s - y < 4:
                out.append((s - y, y))
                y -= 1
        return out
    return [(k & 3, k >> 2) if order == 1 else (k >> 2, k & 3) for k in range(16)]


def sb_scan_index(order, gw, gh, sx, sy):
    """DeriveSubblockScan (transform.cc:1639-1683), as d_sb_scan_index."""
    if order == 1:
        return sy * gw + sx
    if order == 2:
        return sx * gh + sy
    s, idx = sx + sy, 0
    for d in range(s):
        idx += min(d, gw - 1, gh - 1, gw + gh - 2 - d) + 1
    return idx + (min(s, gh - 1) - sy)


def last_pos_group(pos):
    if pos < 4:
        return pos
    l = int(pos).bit_length() - 1
    return 2 * l + ((pos >> (l - 1)) & 1)


def last_pos_bits(ent, c, luma, w, h, scan_order, lx, ly):
    """GetLastPosBits (rdo_quant.cc:900-947) with GetCoeffLastPosCtx (cabac.cc:727-770)."""
    if scan_order == 2:
        lx, ly, w, h = ly, lx, h, w

    def ctx(pos, is_x):
        size = w if is_x else h
        if luma:
            l2 = size.bit_length() - 1
            off = 0 if l2 < 3 else {3: 3, 4: 6, 5: 10, 6: 15}.get(l2, 21)
            return (c["last_x_luma"] if is_x else c["last_y_luma"])[off + (pos >> ((l2 + 1) >> 2))]
        return (c["last_x_chroma"] if is_x else c["last_y_chroma"])[pos >> min(max(size >> 3, 0), 2)]

    bits = 0
    for is_x, p, size in ((True, lx, w), (False, ly, h)):
        g = last_pos_group(p)
        for k in range(g):
            bits += int(ent[int(ctx(k, is_x)) ^ 1])
        if g < last_pos_group(size - 1):
            bits += int(ent[int(ctx(g, is_x))])
        if g > 3:
            bits += ((g - 2) >> 1) * BYPASS
    return bits


def proves_zero(ent, bd, comp_qp, comp, scan_order, c, prm, src):
    """The device's bound; True = QuantRdo is bound to return 0."""
    h, w = src.shape
    if w < 4 or h < 4 or w > 32 or h > 32:
        return False
    luma = comp == 0
    lw, lh = w.bit_length() - 1, h.bit_length() - 1
    qpb = max(comp_qp + 6 * (bd - 8), 0)
    tshift = 15 - bd - ((lw + lh) >> 1)
    bias = (lw + lh) & 1
    scale = FWD[qpb % 6] * (181 if bias else 1)
    fq_shift = 14 + qpb // 6 + tshift + (7 if bias else 0)
    fq_offset = 1 << (fq_shift - 1)
    cost_scale = 15 - 2 * tshift - 2 * (bd - 8) + 2 * bias
    iq_shift = 6 - tshift + (8 if bias else 0)
    iq_scale = (INV[qpb % 6] << (qpb // 6)) * (181 if bias else 1)
    lam = int(prm["lambda"])
    a = np.abs(src.astype(np.int64))
    if (a == 32768).any():
        return False
    q = (a * scale + fq_offset) >> fq_shift
    ys, xs = np.nonzero(q)
    if len(ys) == 0 or len(ys) > 16:
        return False
    bc = lambda b: (int(b) * lam) >> 16
    qp = np.pad(q, ((0, 2), (0, 2)))
    sigs = c["sig_luma"] if luma else c["sig_chroma"]
    g1s = c["greater1_luma"] if luma else c["greater1_chroma"]
    cbf = c["cbf_chroma"] if not luma else (c["cbf_luma"] if int(prm["flags"]) & oq.RDOQ_INTRA_CU
                                            else c["root_cbf"])
    inv_scan = {p: k for k, p in enumerate(scan4(scan_order))}
    cands = []
    for y, x in zip(ys.tolist(), xs.tolist()):
        nb = [qp[y, x + 1], qp[y, x + 2], qp[y + 1, x + 1], qp[y + 1, x], qp[y + 2, x]]
        cnt, cnt1 = sum(v > 0 for v in nb), sum(v > 1 for v in nb)
        posxy, size = x + y, (lw + lh) >> 1
        start = (6 if posxy < 2 else 0) + (6 if luma and posxy < 5 else 0) + \
            ((18 << min(size - 3, 1)) if size > 2 and luma else 0)
        sig1 = min(int(ent[int(sigs[start + nn]) ^ 1]) for nn in range(cnt + 1))
        if (x | y) & 3 == 0 and (x | y) != 0:
            sig1 = 0
        gstart = (10 if posxy < 3 else (5 if posxy < 10 else 0)) if luma else 0
        gl = [g1s[0]] + [g1s[gstart + min(nn, 4) + 1] for nn in range(cnt1 + 1)]
        flag_min = min([BYPASS] + [int(ent[int(s) ^ b]) for s in gl for b in (0, 1)])
        lvl_min = BYPASS + flag_min

        def dist(lvl):
            if iq_shift > 0:
                deq = (lvl * iq_scale + (1 << (iq_shift - 1))) >> iq_shift
            else:
                deq = (lvl * iq_scale) << -iq_shift
            deq = min(max(deq, -32768), 32767)
            e = int(a[y, x]) - deq
            return (e * e) << cost_scale
        qv = int(q[y, x])
        d_best = min(dist(qv), dist(qv - 1)) if qv > 1 else dist(qv)
        zd = (int(a[y, x]) ** 2) << cost_scale
        coded, coded_last = d_best + bc(sig1 + lvl_min), d_best + bc(lvl_min)
        gain, gain_last = zd - min(coded, zd), zd - coded_last
        rhs = bc(int(ent[int(cbf) ^ 1])) - bc(int(ent[int(cbf)])) + \
            bc(last_pos_bits(ent, c, luma, w, h, scan_order, x, y))
        idx = (sb_scan_index(scan_order, w >> 2, h >> 2, x >> 2, y >> 2) << 4) + \
            inv_scan[(x & 3, y & 3)]
        cands.append((idx, gain, gain_last, rhs))
    for idx, _, gain_last, rhs in cands:
        before = sum(g for i, g, _, _ in cands if i < idx)
        if before + gain_last >= rhs:
            return False
    return True


def _cases(rng, n):
    for _ in range(n):
        bd = int(rng.choice([8, 10, 12]))
        comp = int(rng.integers(0, 3))
        w, h = (int(rng.choice([4, 8, 16, 32])) for _ in range(2))
        scan = int(rng.integers(0, 3)) if max(w, h) < 16 else 0
        qp = int(rng.integers(24, 46))
        # magnitudes around the smallest one that quantises to a level: the blocks for
        # which "all zero" is a close call
        lw, lh = w.bit_length() - 1, h.bit_length() - 1
        qpb = qp + 6 * (bd - 8)
        bias = (lw + lh) & 1
        fq_shift = 14 + qpb // 6 + 15 - bd - ((lw + lh) >> 1) + (7 if bias else 0)
        step = (1 << (fq_shift - 1)) / (FWD[qpb % 6] * (181 if bias else 1))
        yy, xx = np.mgrid[0:h, 0:w]
        decay = np.exp(-(xx + yy) / float(rng.choice([0.5, 1.0, 2.0])))
        src = np.rint(rng.laplace(0, 1, (h, w)) * decay * step * float(rng.choice([1.0, 2.0, 4.0])))
        src = np.clip(src, -32767, 32767).astype(np.int16)
        lam = 0.57 * 2.0 ** ((qp - 12) / 3.0) * float(rng.choice([0.5, 1.0, 2.0]))
        prm = np.zeros(1, oq.RDOQ_PARAMS_DTYPE)
        prm["lambda"] = int(lam * 65536 + 0.5)
        inv_scale = INV[(qp + 6 * (bd - 8)) % 6] << ((qp + 6 * (bd - 8)) // 6)
        prm["rd_factor"] = int(inv_scale * inv_scale / lam / 16 / (1 << (2 * (bd - 8))) + 0.5)
        prm["flags"] = oq.RDOQ_INTRA_CU if rng.integers(0, 2) else 0
        yield bd, qp, comp, scan, oq.random_contexts(rng), prm, src


def test_proof_never_zeroes_a_block_the_quantiser_codes(xo):
    ent = entropy_table(xo)
    rng = np.random.default_rng(20260929)
    n = live = zero = proved = 0
    for bd, qp, comp, scan, c, prm, src in _cases(rng, 2500):
        nnz, _ = oq.quant_rdo_oracle(xo, bd, qp, comp, scan, 1, c, prm, src)
        n += 1
        ok = proves_zero(ent, bd, qp, comp, scan, c[0], prm[0], src)
        if ok:
            assert nnz == 0, (bd, qp, comp, scan, src.shape, src)
            proved += 1
        zero += nnz == 0
    assert proved > 100 and zero > proved, (n, zero, proved)


def test_proof_against_the_reference_quantiser(xo):
    """The same on the 432 golden vectors of the reference's own QuantRdo
    (tests/golden/rdoq.npz: random and picture-initial context states)."""
    import os
    ent = entropy_table(xo)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdoq.npz"))
    proved = 0
    for i, case in enumerate(g["cases"]):
        bd, cqp, comp, scan, sign_hide, w, h, nnz = (int(v) for v in case)
        prm = g["params"][i].view(oq.RDOQ_PARAMS_DTYPE)[0]
        c = g["contexts"][i].view(oq.RDOQ_CTX_DTYPE)[0]
        if (w == 2 or h == 2):
            continue
        if proves_zero(ent, bd, cqp, comp, scan, c, prm, np.ascontiguousarray(g["src"][i][:h, :w])):
            assert nnz == 0, (i, bd, cqp, comp, scan, w, h)
            proved += 1
    print("golden vectors proved all zero:", proved)

"""Hot-path frame pass on the GPU: the composition bench.py times.

For one picture with a fixed CU partition (16x16 CUs, 16x8 / 8x16 / 8x8 at the
picture edges, as the reference's forced boundary splits produce):

  me_search        T1+T3   TZ full-pel + 9+8 point sub-pel search per CU
  mc_from_me       I1      motion compensation Y,U,V with the searched MV
  residual_batch   X1,Q,Q1,X2,R1  transform -> quant -> dequant -> inverse -> rec
  cu_info_from_me          CU metadata for the in-loop filter (device-side glue)
  deblock          D1-D4   vertical-edge pass, horizontal-edge pass
  pad_border       P1      so the result can serve as the next reference
  picture_ssd      M6      PSNR-Y parts

Everything (pictures, descriptors, decisions) stays resident in HBM; the host
only enqueues ~9 launches per picture.  What the reference derives serially
from neighbouring CUs (AMVP predictor, previous CU's MV, CABAC state for RDOQ)
is an INPUT of the batched kernels: here the predictor is the zero vector and
the quantiser is the reference's non-RDO QuantFast (scope).
"""
import math
import os

import ctypes as C

import numpy as np

from . import api

# Qp::kChromaScale_ for 4:2:0 with chroma_qp_offset_table == 1
# (quantize.cc:34-38)
_CHROMA_SCALE = list(range(30)) + [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36,
                                   36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45,
                                   46, 47, 48, 49, 50, 51]


def chroma_qp(qp):
    return _CHROMA_SCALE[max(0, min(57, qp))]


def lambda16_for_qp(qp):
    """floor(65536*sqrt(lambda)) with lambda = 0.57*2^((qp-12)/3)
    (PictureData::Init, picture_data.cc:96-97; inter_tz_search.cc:98-99)."""
    lam = 0.57 * math.pow(2.0, (qp - 12) / 3.0)
    return int(math.floor(65536.0 * math.sqrt(lam)))


_INV_QUANT_SCALES = (40, 45, 51, 57, 64, 72)       # quantize.cc:44-46
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_init_ctx_table = None


def rdoq_init_contexts(qp, pic_type=1):
    """The coefficient-coding context states a syntax writer starts a picture of
    this qp / type with (CabacContexts::ResetStates, cabac.cc:311-358), as one
    xvcgpu_rdoq_contexts record.  RDOQ reads the entropy coder's states
    (rdo_quant.cc:254): an integration snapshots its live CABAC contexts per
    batch; this repo's frame pass has no entropy coder and feeds the
    picture-initial states (data/rdoq_init_contexts.npy, captured from the
    reference by tools/gen_rdoq_contexts.py).  pic_type: 0 bi, 1 uni, 2 intra."""
    global _init_ctx_table
    if _init_ctx_table is None:
        _init_ctx_table = np.load(os.path.join(_DATA, "rdoq_init_contexts.npy"))
    raw = _init_ctx_table[max(0, min(63, qp)), pic_type]
    return np.ascontiguousarray(raw).view(api.RDOQ_CTX_DTYPE).copy()


def rdoq_host_params(qp, bitdepth, lam=None):
    """The two double-arithmetic constants of QuantRdo per component, as the
    reference computes them on the host (rdo_quant.cc:251-252, :590-594; Qp,
    quantize.cc:48-92): returns [(lambda_fix, rd_factor)] for Y, U, V.  `lam` =
    the luma lambda (default: PictureData::Init's 0.57 * 2^((qp - 12) / 3))."""
    if lam is None:
        lam = 0.57 * math.pow(2.0, (qp - 12) / 3.0)
    qpc = chroma_qp(qp)
    # Qp::GetChromaDistWeight with table 1, offset 0 (quantize.cc:82-92)
    weight_c = math.pow(2.0, -(qpc - max(0, min(57, qp))) / 3.0)
    out = []
    for c in range(3):
        lam_c = lam if c == 0 else lam / weight_c
        qpb = max(0, (qp if c == 0 else qpc) + 6 * (bitdepth - 8))
        inv_scale = float(_INV_QUANT_SCALES[qpb % 6] << (qpb // 6))
        lambda_fix = int(lam_c * 65536 + 0.5)
        rd_factor = int(inv_scale * inv_scale / lam_c / 16 / (1 << (2 * (bitdepth - 8))) + 0.5)
        out.append((lambda_fix, rd_factor))
    return out


def cu_partition(width, height, cu=16, xcd_tiles=False):
    """List of (x, y, w, h): cu x cu CUs, smaller at right/bottom edge - in raster
    order, or (xcd_tiles) region by region of a 4 x 2 tiling of the picture, raster
    inside a region.  The kernels give XCD k the k-th contiguous eighth of a job
    list (dev_common.h: xcd_job_index), and an XCD's L2 then holds what its jobs
    read: with raster order that is a band 1/8 of the picture tall plus the rows the
    TZ search's far diamonds (+-64) and the filters reach above and below it -
    (135 + 128) / 135 = 1.95 x the band at 1080p; a quarter-by-half region has the
    shorter boundary: (480 + 128)(540 + 128) / (480 * 540) = 1.57 x."""
    cols, rows = (width + cu - 1) // cu, (height + cu - 1) // cu

    def cell(cx, cy):
        x, y = cx * cu, cy * cu
        return (x, y, min(cu, width - x), min(cu, height - y))
    if not xcd_tiles:
        return [cell(cx, cy) for cy in range(rows) for cx in range(cols)]
    parts = []
    xb = [round(k * cols / 4) for k in range(5)]
    yb = [round(k * rows / 2) for k in range(3)]
    for ty in range(2):
        for tx in range(4):
            parts += [cell(cx, cy) for cy in range(yb[ty], yb[ty + 1])
                      for cx in range(xb[tx], xb[tx + 1])]
    return parts


class FrameDescriptors:
    """Host-side (numpy) descriptors of one picture's jobs; shared by the GPU
    frame pass and the CPU oracle frame pass in tests / bench.

    With `row_range=(y0, y1)` only the CUs whose top row lies in [y0, y1) get
    motion-search / residual jobs (one CTU-row shard); the CU map and the CU
    metadata array stay global (indices into the whole picture's raster list)
    because the in-loop filter looks across the shard boundary."""

    def __init__(self, width, height, qp=32, cu=16, search_range=96,
                 row_range=None, rdoq=False, bitdepth=10, xcd_tiles=False):
        self.w, self.h, self.qp = width, height, qp
        self.rdoq = rdoq
        # (region-major CU order only for a whole picture: a row shard's CUs have to be
        # a contiguous run of the raster list)
        assert not xcd_tiles or row_range is None or tuple(row_range) == (0, height)
        parts_all = cu_partition(width, height, cu, xcd_tiles)
        self.n_cus_total = len(parts_all)
        if row_range is None:
            row_range = (0, height)
        self.row_range = (max(0, row_range[0]), min(height, row_range[1]))
        own = [i for i, p in enumerate(parts_all)
               if self.row_range[0] <= p[1] < self.row_range[1]]
        self.cu_base = own[0] if own else 0
        assert own == list(range(self.cu_base, self.cu_base + len(own)))
        n = len(own)
        self.n_cus = n
        me = np.zeros(n, api.ME_DTYPE)
        tx = np.zeros(3 * n, api.TX_DTYPE)
        luma_idx = np.zeros(n, np.int32)
        cmap = -np.ones(((height + 3) // 4, (width + 3) // 4), np.int32)
        qpc = chroma_qp(qp)
        lam = lambda16_for_qp(qp)
        for gi, (x, y, w, h) in enumerate(parts_all):
            cmap[y // 4:(y + h) // 4, x // 4:(x + w) // 4] = gi
        for i, gi in enumerate(own):
            x, y, w, h = parts_all[gi]
            b = me[i]
            b["x"], b["y"], b["w"], b["h"] = x, y, w, h
            b["depth_nonzero"] = 1
            b["lambda16"] = lam
            b["search_range"] = search_range
            luma_idx[i] = 3 * i
            t = tx[3 * i]
            t["x"], t["y"], t["w"], t["h"], t["comp"], t["qp"] = x, y, w, h, 0, qp
            for c in (1, 2):
                t = tx[3 * i + c]
                t["x"], t["y"], t["w"], t["h"] = x // 2, y // 2, w // 2, h // 2
                t["comp"], t["qp"] = c, qpc
        self.me, self.tx, self.luma_idx, self.cu_map = me, tx, luma_idx, cmap
        self.qp_c = qpc
        self.rdoq_contexts = self.rdoq_params = None
        self.rdoq_lambda = 0.57 * math.pow(2.0, (qp - 12) / 3.0)
        if rdoq:
            # the quantiser the reference's encoder runs (transform_encoder.cc:230)
            tx["intra_pic"] |= api.TXF_RDOQ
            self.rdoq_contexts = rdoq_init_contexts(qp, 1)
            prm = np.zeros(3 * n, api.RDOQ_PARAMS_DTYPE)
            for c, (lf, rf) in enumerate(rdoq_host_params(qp, bitdepth, self.rdoq_lambda)):
                prm["lambda"][c::3] = lf
                prm["rd_factor"][c::3] = rf
            self.rdoq_params = prm
        self.cu_rows = (height + cu - 1) // cu
        self.cus_per_row = (width + cu - 1) // cu
        self.cu_size = cu


def run_multi(passes, origs, refs, recs, ref_pocs=None):
    """The frame passes of len(passes) independent pictures with every kernel
    launched once for all of them (xvcgpu_frame_pass_multi).  passes: FramePass
    objects of one picture size, each on its own Context - the contexts lend their
    scratch, everything is enqueued on passes[0]'s stream (share_stream() puts the
    other contexts on it)."""
    n = len(passes)
    lead = passes[0].ctx
    ctxs = (C.c_void_p * n)(*[p.ctx.h for p in passes])
    argv = (C.POINTER(api.FramePassArgs) * n)()
    for i, p in enumerate(passes):
        a = p._args()
        a.orig, a.ref, a.rec = origs[i].h_pic, refs[i].h_pic, recs[i].h_pic
        a.ref_poc = ref_pocs[i] if ref_pocs is not None else 0
        argv[i] = C.pointer(a)
    lead._check(lead.lib.xvcgpu_frame_pass_multi(
        ctxs, argv, n, api.FP_ENCODE | api.FP_DEBLOCK_V | api.FP_DEBLOCK_H | api.FP_PAD |
        api.FP_SSD))


def share_stream(passes):
    """Put the contexts of passes[1:] on passes[0]'s stream (see run_multi)."""
    stream = passes[0].ctx.stream_ptr()
    for p in passes[1:]:
        p.ctx.set_stream(stream)


class FramePass:
    """Device-resident state for running frame passes of one picture size
    (or of one CTU-row shard of it)."""

    def __init__(self, ctx, width, height, bitdepth=10, qp=32, cu=16,
                 search_range=96, row_range=None, fused=True, keep_levels=False,
                 rdoq=False, rdoq_packed=None, xcd_tiles=False):
        self.ctx = ctx
        self.rdoq = rdoq
        # RDOQ keeps only a few lanes of a wave busy per block, so its own kernel
        # packs several blocks into a wave (xvcgpu_quant_rdo_batch) between the
        # forward and the inverse half of the residual pipeline: 5 launches instead
        # of the one fused launch, an order of magnitude less time
        self.rdoq_packed = rdoq and (fused if rdoq_packed is None else rdoq_packed)
        self.tx_four_lane_only = False       # set below, from the transform blocks
        if self.rdoq_packed:
            keep_levels = True
        # keep_levels: also store the quantised coefficients of every TU (what
        # the entropy coder - or DecodePass - consumes); needs the unfused path
        self.fused = fused and width % 8 == 0 and height % 8 == 0 and not keep_levels \
            and not self.rdoq_packed
        self.w, self.h, self.bd = width, height, bitdepth
        self.desc = d = FrameDescriptors(width, height, qp, cu, search_range,
                                         row_range, rdoq, bitdepth, xcd_tiles)
        self.d_rdoq_ctx = ctx.buffer(d.rdoq_contexts) if rdoq else None
        self.d_rdoq_prm = ctx.buffer(d.rdoq_params) if rdoq else None
        self.d_me = ctx.buffer(d.me)
        # the searches are the CUs of the grid: on the 16-sample grid (almost) all 16x16
        # (the hint only where it is true: the jobs of any other shape go, 64 per wave and one
        # after the other, through the leftover kernel - a pass of 8x8 CUs would crawl)
        sq = (d.me["w"] == 16) & ((d.me["h"] == 16) | (d.me["h"] == 8)) if len(d.me) else np.zeros(0, bool)
        self.me_flags = api.ME_FULLPEL | api.ME_SUBPEL | \
            (api.ME_HINT_SQ16 if len(sq) and sq.mean() >= 0.98 else 0)
        self.me_only_sq16 = bool(len(sq)) and bool(sq.all())
        if self.me_only_sq16:        # ... all of them: no second kernel for other shapes
            self.me_flags |= api.ME_ONLY_SQ16
        self.d_tx = ctx.buffer(d.tx)
        # no block of the quantiser's general class (diagonal scan, 4x4 sub-blocks, sides up
        # to 32, at most sixteen sub-blocks): its launch can be left out (xvcgpu.h)
        t = d.tx
        self.tx_four_lane_only = bool(len(t)) and bool(
            ((t["w"] >= 4) & (t["h"] >= 4) & (t["w"] <= 32) & (t["h"] <= 32) &
             ((t["w"].astype(int) >> 2) * (t["h"].astype(int) >> 2) <= 16) &
             (((t["intra_pic"].astype(int) >> api.TXF_SCAN_SHIFT) & 3) == 0)).all())
        self.d_luma_idx = ctx.buffer(d.luma_idx)
        self.d_map = ctx.buffer(d.cu_map)
        self.d_res = ctx.alloc(api.MERES_DTYPE.itemsize * max(1, d.n_cus))
        self.d_nnz = ctx.alloc(4 * max(1, len(d.tx)))
        self.d_cus = ctx.alloc(api.CU_DTYPE.itemsize * d.n_cus_total)
        ctx._check(ctx.lib.xvcgpu_memset(ctx.h, self.d_cus.ptr, 0,
                                         api.CU_DTYPE.itemsize * d.n_cus_total))
        self.d_ssd = ctx.alloc(16)
        self.pred = ctx.picture(width, height, bitdepth)
        # whole pictures of CUs >= 8x8 end with ONE launch (xvcgpu_deblock_pad_ssd:
        # unfiltered reconstruction in `scratch` -> deblocked, padded `rec` + SSD)
        # instead of deblock V, H, pad, SSD, SSD fold
        self.scratch = None
        if cu >= 8 and width % 8 == 0 and height % 8 == 0 and d.row_range == (0, d.h) \
                and os.environ.get("XVC_TAIL_FUSED", "1") != "0":
            self.scratch = ctx.picture(width, height, bitdepth)
        self.d_levels = self.d_level_off = None
        self.n_levels = 0
        if keep_levels:
            off, total = ctx.level_offsets(d.tx)
            self.d_level_off = ctx.buffer(off)
            self.d_levels = ctx.alloc(2 * max(1, total))
            self.n_levels = total
        self.d_coeffs = ctx.alloc(2 * max(1, self.n_levels)) if self.rdoq_packed else None
        if self.rdoq_packed:    # no allocation inside the passes (they may be recorded)
            ctx._check(ctx.lib.xvcgpu_quant_rdo_reserve(ctx.h, len(d.tx), self.n_levels))

    @property
    def d_cus_own(self):
        return self.d_cus.ptr + api.CU_DTYPE.itemsize * self.desc.cu_base

    def _args(self):
        """The picture-per-call argument block (xvcgpu_frame_pass), built once;
        only the picture handles, the reference POC and row ranges change."""
        if getattr(self, "_fp_args", None) is None:
            d, a = self.desc, api.FramePassArgs()
            a.d_me, a.d_results, a.n_cus = self.d_me.ptr, self.d_res.ptr, d.n_cus
            a.max_block_size, a.qp_y, a.qp_c = d.cu_size, d.qp, d.qp_c
            a.d_nnz, a.d_cus_own, a.d_cus = self.d_nnz.ptr, self.d_cus_own, self.d_cus.ptr
            a.n_cus_total, a.d_cu_map = d.n_cus_total, self.d_map.ptr
            a.map_stride = d.cu_map.shape[1]
            a.db_y_begin, a.db_y_end, a.dbh_y_end = 0, d.h, d.h
            a.scratch_rec = self.scratch.h_pic if self.scratch is not None else None
            a.ssd_y_begin, a.ssd_y_end = 0, 1 << 30
            a.shift_bitdepth, a.d_ssd = self.bd, self.d_ssd.ptr
            if self.rdoq:
                a.d_rdoq_contexts, a.d_rdoq_params = self.d_rdoq_ctx.ptr, self.d_rdoq_prm.ptr
            if self.rdoq_packed:
                a.pred, a.d_tx, a.n_tx = self.pred.h_pic, self.d_tx.ptr, len(d.tx)
                a.d_level_off, a.d_luma_tx_index = self.d_level_off.ptr, self.d_luma_idx.ptr
                a.d_coeffs, a.d_levels = self.d_coeffs.ptr, self.d_levels.ptr
                a.n_coeffs = self.n_levels
            self._fp_args = a
        return self._fp_args

    def run_phases(self, orig, ref, rec, phases, ref_poc=0, rows=None, dbh_end=None,
                   ssd_rows=None, d_ssd=None):
        """One call for the selected phases (api.FP_*).  Needs the fused
        CompressAndEvalCbf kernel (CUs up to 16x16) for FP_ENCODE."""
        a = self._args()
        a.orig = orig.h_pic if orig is not None else None
        a.ref = ref.h_pic if ref is not None else None
        a.rec, a.ref_poc = rec.h_pic, ref_poc
        a.tx_four_lane_only = 1 if self.tx_four_lane_only else 0
        a.me_only_sq16 = 1 if self.me_only_sq16 else 0
        if rows is not None:
            a.db_y_begin, a.db_y_end = rows
            a.dbh_y_end = dbh_end if dbh_end is not None else rows[1]
        if ssd_rows is not None:
            a.ssd_y_begin, a.ssd_y_end = ssd_rows
        if d_ssd is not None:
            a.d_ssd = d_ssd
        self.ctx._check(self.ctx.lib.xvcgpu_frame_pass(self.ctx.h, C.byref(a), phases))

    def encode(self, orig, ref, rec, ref_poc=0):
        """ME -> MC -> residual -> CU metadata for the own CUs (asynchronous)."""
        ctx, d = self.ctx, self.desc
        n = d.n_cus
        if n == 0:
            return
        ctx.me_search_dev(orig, ref, self.me_flags, self.d_me.ptr, n, self.d_res.ptr, d.cu_size)
        if self.fused and d.cu_size <= 16:
            # MC + transform/quant/recon + CU metadata in one launch; the
            # prediction never leaves LDS
            if self.rdoq:
                ctx.recon_from_me_rdoq_dev(orig, ref, rec, self.d_me.ptr, self.d_res.ptr, n,
                                           d.qp, d.qp_c, ref_poc, self.d_nnz.ptr,
                                           self.d_cus_own, self.d_rdoq_ctx.ptr,
                                           self.d_rdoq_prm.ptr)
            else:
                ctx.recon_from_me_dev(orig, ref, rec, self.d_me.ptr, self.d_res.ptr, n,
                                      d.qp, d.qp_c, ref_poc, self.d_nnz.ptr, self.d_cus_own)
            return
        front_fused = self.rdoq_packed and d.cu_size <= 16    # xvcgpu_fwd_from_me
        if not front_fused:
            ctx.mc_from_me_dev(ref, self.pred, self.d_me.ptr, self.d_res.ptr, n)
        if self.rdoq_packed:
            lib, T = ctx.lib, len(d.tx)
            # fwd_from_me: the prediction goes into the reconstruction's picture, the
            # inverse half then works in place (and skips the blocks without levels)
            pred_pic = rec if front_fused else self.pred
            if front_fused:     # ... which also classifies the blocks for the quantiser
                # (with the quantiser's contexts: blocks it can prove all zero on the
                # spot never reach the walk - xvcgpu_quant_rdo_set_prove_zero)
                ctx._check(lib.xvcgpu_fwd_from_me_classify_prove(
                    ctx.h, orig.h_pic, ref.h_pic, rec.h_pic, self.d_me.ptr,
                    self.d_res.ptr, n, d.qp, d.qp_c, ref_poc, self.d_coeffs.ptr,
                    self.d_level_off.ptr, C.c_size_t(self.n_levels), self.d_levels.ptr,
                    self.d_nnz.ptr, self.d_cus_own, self.d_rdoq_ctx.ptr, self.d_rdoq_prm.ptr))
            else:
                ctx._check(lib.xvcgpu_fwd_transform_batch(
                    ctx.h, orig.h_pic, self.pred.h_pic, self.d_tx.ptr, T, self.d_coeffs.ptr,
                    self.d_level_off.ptr))
            quant = lib.xvcgpu_quant_rdo_classified_batch if front_fused else \
                lib.xvcgpu_quant_rdo_batch
            ctx._check(quant(
                ctx.h, self.bd, self.d_tx.ptr, T, self.d_coeffs.ptr, self.d_level_off.ptr,
                C.c_size_t(self.n_levels), self.d_levels.ptr, self.d_nnz.ptr,
                self.d_rdoq_ctx.ptr, self.d_rdoq_prm.ptr,
                *([self.d_cus_own] if front_fused else [])))
            if front_fused:   # blocks 3 * cu + comp, in place: U and V of a CU share a wave
                ctx._check(lib.xvcgpu_inv_transform_cu_order(
                    ctx.h, rec.h_pic, self.d_tx.ptr, n, self.d_levels.ptr, self.d_level_off.ptr,
                    self.d_nnz.ptr))
            else:
                ctx._check(lib.xvcgpu_inv_transform_batch(
                    ctx.h, pred_pic.h_pic, rec.h_pic, self.d_tx.ptr, T, self.d_levels.ptr,
                    self.d_level_off.ptr, self.d_nnz.ptr))
        elif self.rdoq:
            ctx.residual_rdoq_batch_dev(orig, self.pred, rec, self.d_tx.ptr, len(d.tx),
                                        self.d_levels.ptr if self.d_levels else None,
                                        self.d_level_off.ptr if self.d_level_off else None,
                                        self.d_nnz.ptr, self.d_rdoq_ctx.ptr, self.d_rdoq_prm.ptr)
        else:
            ctx.residual_batch_dev(orig, self.pred, rec, self.d_tx.ptr, len(d.tx),
                                   self.d_levels.ptr if self.d_levels else None,
                                   self.d_level_off.ptr if self.d_level_off else None,
                                   self.d_nnz.ptr)
        if not front_fused:      # (front_fused: the records were written on the way)
            ctx.cu_info_from_me_dev(self.d_me.ptr, self.d_res.ptr, self.d_nnz.ptr,
                                    self.d_luma_idx.ptr, n, d.qp, d.qp_c, ref_poc,
                                    self.d_cus_own)

    def kernel_steps(self, orig, ref, rec, ref_poc=0):
        """The launches of one frame pass as (name, callable) in issue order - for
        per-kernel timing (bench.py); run in order they are a frame pass."""
        ctx, d, lib = self.ctx, self.desc, self.ctx.lib
        n, T = d.n_cus, len(d.tx)
        final = rec
        if self.scratch is not None:
            rec = self.scratch
        steps = [("me_search", lambda: ctx.me_search_dev(
            orig, ref, self.me_flags, self.d_me.ptr, n, self.d_res.ptr, d.cu_size))]
        if self.fused and d.cu_size <= 16:
            if self.rdoq:
                steps.append(("recon_from_me", lambda: ctx.recon_from_me_rdoq_dev(
                    orig, ref, rec, self.d_me.ptr, self.d_res.ptr, n, d.qp, d.qp_c, ref_poc,
                    self.d_nnz.ptr, self.d_cus_own, self.d_rdoq_ctx.ptr, self.d_rdoq_prm.ptr)))
            else:
                steps.append(("recon_from_me", lambda: ctx.recon_from_me_dev(
                    orig, ref, rec, self.d_me.ptr, self.d_res.ptr, n, d.qp, d.qp_c, ref_poc,
                    self.d_nnz.ptr, self.d_cus_own)))
        else:
            front_fused = self.rdoq_packed and d.cu_size <= 16
            lv = self.d_levels.ptr if self.d_levels else None
            lo = self.d_level_off.ptr if self.d_level_off else None
            # fwd_from_me writes the prediction into the reconstruction's picture, the
            # inverse half works in place (as xvcgpu_frame_pass does)
            pred_pic = rec if front_fused else self.pred
            if front_fused:
                steps.append(("fwd_from_me", lambda: ctx._check(
                    lib.xvcgpu_fwd_from_me_classify_prove(
                        ctx.h, orig.h_pic, ref.h_pic, rec.h_pic, self.d_me.ptr,
                        self.d_res.ptr, n, d.qp, d.qp_c, ref_poc, self.d_coeffs.ptr, lo,
                        C.c_size_t(self.n_levels), lv, self.d_nnz.ptr, self.d_cus_own,
                        self.d_rdoq_ctx.ptr, self.d_rdoq_prm.ptr))))
            else:
                steps.append(("mc_from_me", lambda: ctx.mc_from_me_dev(
                    ref, self.pred, self.d_me.ptr, self.d_res.ptr, n)))
            if self.rdoq_packed:
                if not front_fused:
                    steps.append(("fwd_transform", lambda: ctx._check(
                        lib.xvcgpu_fwd_transform_batch(ctx.h, orig.h_pic, self.pred.h_pic,
                                                       self.d_tx.ptr, T, self.d_coeffs.ptr, lo))))
                steps += [
                    ("quant_rdo", lambda: ctx._check(
                        (lib.xvcgpu_quant_rdo_classified_batch if front_fused else
                         lib.xvcgpu_quant_rdo_batch)(
                            ctx.h, self.bd, self.d_tx.ptr, T, self.d_coeffs.ptr, lo,
                            C.c_size_t(self.n_levels), lv, self.d_nnz.ptr, self.d_rdoq_ctx.ptr,
                            self.d_rdoq_prm.ptr, *([self.d_cus_own] if front_fused else [])))),
                    ("inv_transform", lambda: ctx._check(
                        lib.xvcgpu_inv_transform_cu_order(ctx.h, rec.h_pic, self.d_tx.ptr, n, lv,
                                                          lo, self.d_nnz.ptr) if front_fused else
                        lib.xvcgpu_inv_transform_batch(ctx.h, pred_pic.h_pic, rec.h_pic,
                                                       self.d_tx.ptr, T, lv, lo,
                                                       self.d_nnz.ptr)))]
            elif self.rdoq:
                steps.append(("residual_rdoq", lambda: ctx.residual_rdoq_batch_dev(
                    orig, self.pred, rec, self.d_tx.ptr, T, lv, lo, self.d_nnz.ptr,
                    self.d_rdoq_ctx.ptr, self.d_rdoq_prm.ptr)))
            else:
                steps.append(("residual", lambda: ctx.residual_batch_dev(
                    orig, self.pred, rec, self.d_tx.ptr, T, lv, lo, self.d_nnz.ptr)))
            if not front_fused:
                steps.append(("cu_info", lambda: ctx.cu_info_from_me_dev(
                    self.d_me.ptr, self.d_res.ptr, self.d_nnz.ptr, self.d_luma_idx.ptr, n, d.qp,
                    d.qp_c, ref_poc, self.d_cus_own)))
        if self.scratch is not None:
            steps.append(("deblock_pad_ssd", lambda: ctx.deblock_pad_ssd_dev(
                rec, final, orig, self.d_cus.ptr, d.n_cus_total, self.d_map.ptr,
                d.cu_map.shape[1], 0, 0, 0, self.bd, self.d_ssd.ptr)))
            return steps
        steps += [
            ("deblock", lambda: ctx.deblock_dev(rec, self.d_cus.ptr, d.n_cus_total,
                                                self.d_map.ptr, d.cu_map.shape[1], 0, 0, 0, 4)),
            ("pad_border", lambda: ctx.pad_border(rec)),
            ("picture_ssd", lambda: ctx.picture_ssd_dev(orig, rec, 0, self.bd, self.d_ssd.ptr))]
        return steps

    def deblock_rows(self, rec, pass_, y0, y1):
        d = self.desc
        self.ctx.deblock_rows_dev(rec, self.d_cus.ptr, d.n_cus_total, self.d_map.ptr,
                                  d.cu_map.shape[1], pass_, y0, y1)

    def run(self, orig, ref, rec, ref_poc=0, deblock=True, pad=True, ssd=True):
        """Enqueue one whole-picture frame pass (asynchronous)."""
        ctx, d = self.ctx, self.desc
        if (self.fused or self.rdoq_packed) and d.cu_size <= 16 and d.row_range == (0, d.h):
            # the whole sequence behind one C call (xvcgpu_frame_pass)
            self.run_phases(orig, ref, rec, api.FP_ENCODE |
                            (api.FP_DEBLOCK_V | api.FP_DEBLOCK_H if deblock else 0) |
                            (api.FP_PAD if pad else 0) | (api.FP_SSD if ssd else 0), ref_poc)
            return
        if self.scratch is not None and deblock and pad and ssd:
            self.encode(orig, ref, self.scratch, ref_poc)
            ctx.deblock_pad_ssd_dev(self.scratch, rec, orig, self.d_cus.ptr, d.n_cus_total,
                                    self.d_map.ptr, d.cu_map.shape[1], 0, 0, 0, self.bd,
                                    self.d_ssd.ptr)
            return
        self.encode(orig, ref, rec, ref_poc)
        if deblock:
            ctx.deblock_dev(rec, self.d_cus.ptr, d.n_cus_total, self.d_map.ptr,
                            d.cu_map.shape[1], 0, 0, 0, 4)
        if pad:
            ctx.pad_border(rec)
        if ssd:
            ctx.picture_ssd_dev(orig, rec, 0, self.bd, self.d_ssd.ptr)

    def results(self):
        d = self.desc
        return (self.d_res.to_array(api.MERES_DTYPE, d.n_cus),
                self.d_nnz.to_array(np.int32, len(d.tx)),
                self.d_cus.to_array(api.CU_DTYPE, d.n_cus_total),
                self.d_ssd.to_array(np.uint64, 2))

    def destroy(self):
        for b in (self.d_me, self.d_tx, self.d_luma_idx, self.d_map, self.d_res,
                  self.d_nnz, self.d_cus, self.d_ssd, self.d_levels, self.d_level_off,
                  self.d_rdoq_ctx, self.d_rdoq_prm, self.d_coeffs):
            if b is not None:
                b.free()
        self.pred.destroy()
        if self.scratch is not None:
            self.scratch.destroy()
            self.scratch = None


class PipelinedFramePass:
    """The frame pass issued on two queues so that kernels of one half of the
    picture run while kernels of the other half ramp up or drain (a launch is
    only ~2 rounds of waves deep, so each kernel spends a third of its time
    partially filled - tools/trace_me.py).

        ctx_hi (high priority): top CU rows    - search, recon, deblock V
        ctx_lo (low priority):  bottom CU rows - search, recon, deblock V
        ctx_hi: waits for ctx_lo, deblock H over the picture, pad
        ctx_lo: waits for that, picture SSD (overlaps the next picture's search)

    Same kernels, same results as FramePass; only the queueing differs."""

    def __init__(self, ctx_hi, ctx_lo, width, height, bitdepth=10, qp=32, cu=16,
                 search_range=96, rdoq=False):
        self.hi, self.lo = ctx_hi, ctx_lo
        self.w, self.h, self.bd = width, height, bitdepth
        rows = (height + cu - 1) // cu
        self.y_mid = (rows // 2) * cu
        self.top = FramePass(ctx_hi, width, height, bitdepth, qp, cu, search_range,
                             row_range=(0, self.y_mid), rdoq=rdoq)
        self.bot = FramePass(ctx_lo, width, height, bitdepth, qp, cu, search_range,
                             row_range=(self.y_mid, height), rdoq=rdoq)
        # one CU metadata array for the whole picture (the H pass reads both halves)
        self.bot.d_cus.free()
        self.bot.d_cus = self.top.d_cus
        self.desc = FrameDescriptors(width, height, qp, cu, search_range)
        self.d_ssd = self.bot.d_ssd

    def run(self, orig, ref, rec, ref_poc=0):
        hi, lo, top, bot = self.hi, self.lo, self.top, self.bot
        top.encode(orig, ref, rec, ref_poc)
        top.deblock_rows(rec, 0, 0, self.y_mid)
        bot.encode(orig, ref, rec, ref_poc)
        bot.deblock_rows(rec, 0, self.y_mid, self.h)
        hi.wait_for(lo)
        top.deblock_rows(rec, 1, 0, self.h)
        hi.pad_border(rec)
        lo.wait_for(hi)
        lo.picture_ssd_dev(orig, rec, 0, self.bd, self.d_ssd.ptr)

    def sync(self):
        self.hi.sync()
        self.lo.sync()

    def results(self):
        a, b = self.top, self.bot
        res = np.concatenate([a.d_res.to_array(api.MERES_DTYPE, a.desc.n_cus),
                              b.d_res.to_array(api.MERES_DTYPE, b.desc.n_cus)])
        nnz = np.concatenate([a.d_nnz.to_array(np.int32, len(a.desc.tx)),
                              b.d_nnz.to_array(np.int32, len(b.desc.tx))])
        return (res, nnz, a.d_cus.to_array(api.CU_DTYPE, a.desc.n_cus_total),
                self.d_ssd.to_array(np.uint64, 2))

    def destroy(self):
        self.sync()
        self.bot.d_cus = None
        for fp in (self.top, self.bot):
            for b in (fp.d_me, fp.d_tx, fp.d_luma_idx, fp.d_map, fp.d_res, fp.d_nnz,
                      fp.d_cus, fp.d_ssd, fp.d_levels, fp.d_level_off,
                      fp.d_rdoq_ctx, fp.d_rdoq_prm, fp.d_coeffs):
                if b is not None:
                    b.free()
            fp.pred.destroy()
            if fp.scratch is not None:
                fp.scratch.destroy()
                fp.scratch = None


class DecodePass:
    """The decoder's reconstruction of one inter picture from parsed syntax
    (PictureDecoder::Decode, picture_decoder.cc:
    168-200, and CuDecoder::DecompressInter / DecompressComponent,
    cu_decoder.cc:102-138): per CU the MV, per TU the levels and the cbf;
    motion compensation -> Quantize::Inverse + InverseTransform + AddClip
    (or CopyFrom when cbf == 0) -> deblocking -> PadBorder.  Same kernels as
    the encoder's reconstruction, so encoder rec == decoder output by
    construction - which the tests check."""

    def __init__(self, ctx, desc, bitdepth=10):
        self.ctx, self.desc, self.bd = ctx, desc, bitdepth
        d = desc
        self.d_me = ctx.buffer(d.me)
        self.d_tx = ctx.buffer(d.tx)
        self.d_luma_idx = ctx.buffer(d.luma_idx)
        self.d_map = ctx.buffer(d.cu_map)
        self.d_cus = ctx.alloc(api.CU_DTYPE.itemsize * d.n_cus_total)
        ctx._check(ctx.lib.xvcgpu_memset(ctx.h, self.d_cus.ptr, 0,
                                         api.CU_DTYPE.itemsize * d.n_cus_total))
        self.pred = ctx.picture(d.w, d.h, bitdepth)

    def run(self, ref, rec, d_mvs, d_levels, d_level_off, d_nnz, ref_poc=0,
            deblock=True, pad=True):
        """d_mvs: xvcgpu_me_result per CU (mv_x / mv_y used); d_levels /
        d_level_off / d_nnz: per TU, as xvcgpu_residual_batch lays them out."""
        ctx, d = self.ctx, self.desc
        ctx.mc_from_me_dev(ref, self.pred, self.d_me.ptr, d_mvs, d.n_cus)
        ctx._check(ctx.lib.xvcgpu_inv_transform_batch(
            ctx.h, self.pred.h_pic, rec.h_pic, self.d_tx.ptr, len(d.tx), d_levels,
            d_level_off, d_nnz))
        ctx.cu_info_from_me_dev(self.d_me.ptr, d_mvs, d_nnz, self.d_luma_idx.ptr,
                                d.n_cus, d.qp, d.qp_c, ref_poc, self.d_cus.ptr)
        if deblock:
            ctx.deblock_dev(rec, self.d_cus.ptr, d.n_cus_total, self.d_map.ptr,
                            d.cu_map.shape[1], 0, 0, 0, 4)
        if pad:
            ctx.pad_border(rec)

    def destroy(self):
        for b in (self.d_me, self.d_tx, self.d_luma_idx, self.d_map, self.d_cus):
            b.free()
        self.pred.destroy()


def psnr_from_ssd(dist, samples):
    """SampleMetric::ComputePsnr tail (sample_metric.cc:143-154)."""
    mse = dist / samples if samples else 0.0
    return 10.0 * math.log10(255.0 * 255.0 / mse) if mse > 0 else 99.999


class IntraPictureDescriptors:
    """Host-side plan of an all-intra picture pass: a
    raster of cu x cu CUs coded in raster order.  A CU's prediction reads the
    reconstruction of its left / above-left / above / above-right neighbours,
    so CU (cx, cy) can run once wave cx + 2*cy - 1 is done: the CUs are
    grouped into those anti-diagonal waves, and every wave is one batch per
    kernel.  Jobs and transform blocks are stored wave by wave."""

    def __init__(self, width, height, qp=32, cu=16):
        assert cu in (8, 16, 32, 64)
        self.w, self.h, self.qp, self.cu = width, height, qp, cu
        self.qp_c = chroma_qp(qp)
        parts = cu_partition(width, height, cu)
        order = sorted(range(len(parts)),
                       key=lambda i: (parts[i][0] // cu + 2 * (parts[i][1] // cu), i))
        self.parts = [parts[i] for i in order]
        n = len(parts)
        self.n_cus = n
        self.luma = np.zeros(n, api.INTRA_DTYPE)
        self.chroma = np.zeros(2 * n, api.INTRA_DTYPE)      # U then V of each CU
        self.jobs3 = np.zeros(3 * n, api.INTRA_DTYPE)       # Y, U, V of each CU (= tx order)
        self.tx = np.zeros(3 * n, api.TX_DTYPE)              # Y, U, V of each CU
        wave_of = []
        for i, (x, y, w, h) in enumerate(self.parts):
            nb = (api.INTRA_HAS_LEFT if x else 0) | (api.INTRA_HAS_ABOVE if y else 0) | \
                (api.INTRA_HAS_ABOVE_LEFT if x and y else 0)
            # GetCuSizeAboveRight: the row above is complete; GetCuSizeBelowLeft: 0
            ar = max(0, min(h, width - (x + w))) if y else 0
            self.luma[i] = (x, y, w, h, 0, 0, nb, ar, 0, 0)
            for c in (1, 2):
                self.chroma[2 * i + c - 1] = (x // 2, y // 2, w // 2, h // 2, c, 0, nb,
                                              ar // 2, 0, 0)
            for c in range(3):
                t = self.tx[3 * i + c]
                s = 1 if c else 0
                t["x"], t["y"], t["w"], t["h"] = x >> s, y >> s, w >> s, h >> s
                t["comp"], t["qp"] = c, (self.qp_c if c else qp)
                t["intra_pic"] = api.TXF_INTRA_PIC
            wave_of.append(x // cu + 2 * (y // cu))
        self.jobs3[0::3] = self.luma
        self.jobs3[1::3] = self.chroma[0::2]
        self.jobs3[2::3] = self.chroma[1::2]
        wave_of = np.array(wave_of)
        self.wave_start = np.searchsorted(wave_of, np.arange(wave_of.max() + 2))
        self.level_off, self.level_total = api.Context.level_offsets(self.tx)

    def waves(self):
        for k in range(len(self.wave_start) - 1):
            a, b = int(self.wave_start[k]), int(self.wave_start[k + 1])
            if b > a:
                yield a, b

    def set_modes(self, a, b, modes):
        """Modes of CUs [a, b): luma jobs, chroma jobs (DM: the luma mode) and
        the coefficient scan of small CUs (TransformHelper::DetermineScanOrder,
        transform.cc:1614-1637)."""
        m = np.asarray(modes, np.int64).reshape(-1)
        assert len(m) == b - a
        self.luma["mode"][a:b] = m
        self.chroma["mode"][2 * a:2 * b] = np.repeat(m, 2)
        self.jobs3["mode"][3 * a:3 * b] = np.repeat(m, 3)
        small = (self.luma["w"][a:b] < 16) & (self.luma["h"][a:b] < 16)
        scan = np.where(np.abs(m - 50) < 10, 1, np.where(np.abs(m - 18) < 10, 2, 0))
        scan = np.where(small, scan, 0)
        self.tx["intra_pic"][3 * a:3 * b] = np.repeat(
            api.TXF_INTRA_PIC | (scan << api.TXF_SCAN_SHIFT), 3)


class IntraPicturePass:
    """All-intra picture on the device, wave by wave (IntraPictureDescriptors).

    encode(): per wave the SATD of all 67 luma modes (xvcgpu_intra_satd_batch),
    the mode with the smallest SATD (first on ties; chosen on the host - the
    reference adds CABAC-state dependent mode bits there, which this
    composition leaves out), prediction of Y,U,V (xvcgpu_intra_pred_batch) and
    TransformAndReconstruct with QuantFast (xvcgpu_residual_batch).
    decode(): the decoder's side - modes and levels given, prediction +
    xvcgpu_inv_transform_batch per wave, no host round trip.
    Neither is the reference's intra encoder (no RDO over modes / transforms,
    raster CU order instead of the CTU quad-tree order); the oracle's twin in
    tests/oracle_intra_picture.py is the same composition on the CPU."""

    def __init__(self, ctx, width, height, bitdepth=10, qp=32, cu=16, fused=None):
        self.ctx, self.bd = ctx, bitdepth
        self.desc = d = IntraPictureDescriptors(width, height, qp, cu)
        # CUs up to 16x16: prediction + residual pipeline in one launch per wave
        # (xvcgpu_intra_recon_batch); else prediction picture + separate launches
        self.fused = (cu <= 16) if fused is None else fused
        self.pred = ctx.picture(width, height, bitdepth)
        self.d_luma = ctx.buffer(d.luma)
        self.d_chroma = ctx.buffer(d.chroma)
        self.d_jobs3 = ctx.buffer(d.jobs3)
        self.d_tx = ctx.buffer(d.tx)
        self.d_off = ctx.buffer(d.level_off)
        self.d_levels = ctx.alloc(2 * max(1, d.level_total))
        self.d_nnz = ctx.alloc(4 * len(d.tx))
        self.d_dist = ctx.alloc(4 * api.INTRA_NUM_MODES * d.n_cus)
        self.d_modes = ctx.alloc(4 * d.n_cus)

    def _upload_range(self, buf, arr, a, b):
        sz = arr.dtype.itemsize
        self.ctx.h2d(buf.ptr + a * sz, arr[a:b])

    def encode(self, orig, rec, host_select=False):
        """host_select: fold the SATD table on the host (one round trip per
        wave, the reference's division of labour); default: on the device
        (xvcgpu_intra_select_modes), the whole picture enqueued without a sync."""
        ctx, d, lib = self.ctx, self.desc, self.ctx.lib
        J, T = api.INTRA_DTYPE.itemsize, api.TX_DTYPE.itemsize
        if not host_select and self.fused:
            M = 4 * api.INTRA_NUM_MODES
            for a, b in d.waves():
                n = b - a
                ctx._check(lib.xvcgpu_intra_satd_batch(
                    ctx.h, orig.h_pic, rec.h_pic, self.d_luma.ptr + a * J, n,
                    self.d_dist.ptr + a * M, d.cu))
                ctx._check(lib.xvcgpu_intra_select_modes(
                    ctx.h, self.d_dist.ptr + a * M, None, n, self.d_modes.ptr + 4 * a,
                    self.d_jobs3.ptr + 3 * a * J, self.d_tx.ptr + 3 * a * T, 3))
                self._recon(orig, rec, a, b)
            modes = self.d_modes.to_array(np.int32, d.n_cus)    # synchronises
            d.set_modes(0, d.n_cus, modes)
            return
        for a, b in d.waves():
            n = b - a
            ctx._check(lib.xvcgpu_intra_satd_batch(ctx.h, orig.h_pic, rec.h_pic,
                                                   self.d_luma.ptr + a * J, n,
                                                   self.d_dist.ptr, d.cu))
            dist = self.d_dist.to_array(np.uint32, n * api.INTRA_NUM_MODES) \
                .reshape(n, api.INTRA_NUM_MODES)
            d.set_modes(a, b, dist.argmin(axis=1))
            self._upload_range(self.d_tx, d.tx, 3 * a, 3 * b)
            if self.fused:
                self._upload_range(self.d_jobs3, d.jobs3, 3 * a, 3 * b)
                self._recon(orig, rec, a, b)
                continue
            self._upload_range(self.d_luma, d.luma, a, b)
            self._upload_range(self.d_chroma, d.chroma, 2 * a, 2 * b)
            self._wave(rec, a, b)
            ctx.residual_batch_dev(orig, self.pred, rec, self.d_tx.ptr + 3 * a * T, 3 * n,
                                   self.d_levels.ptr, self.d_off.ptr + 3 * a * 4,
                                   self.d_nnz.ptr + 3 * a * 4)
        ctx.sync()

    def _recon(self, orig, rec, a, b):
        ctx, J, T = self.ctx, api.INTRA_DTYPE.itemsize, api.TX_DTYPE.itemsize
        ctx._check(ctx.lib.xvcgpu_intra_recon_batch(
            ctx.h, orig.h_pic if orig is not None else None, rec.h_pic,
            self.d_jobs3.ptr + 3 * a * J, self.d_tx.ptr + 3 * a * T, 3 * (b - a),
            self.d_levels.ptr, self.d_off.ptr + 3 * a * 4, self.d_nnz.ptr + 3 * a * 4))

    def _wave(self, rec, a, b):
        ctx, lib, J = self.ctx, self.ctx.lib, api.INTRA_DTYPE.itemsize
        ctx._check(lib.xvcgpu_intra_pred_batch(ctx.h, rec.h_pic, self.pred.h_pic,
                                               self.d_luma.ptr + a * J, b - a))
        ctx._check(lib.xvcgpu_intra_pred_batch(ctx.h, rec.h_pic, self.pred.h_pic,
                                               self.d_chroma.ptr + 2 * a * J, 2 * (b - a)))

    def load(self, modes, levels, nnz):
        """Parsed syntax of a picture: mode per CU (wave order), levels / nnz as
        results() returned them."""
        d = self.desc
        d.set_modes(0, d.n_cus, modes)
        for buf, arr in ((self.d_luma, d.luma), (self.d_chroma, d.chroma), (self.d_tx, d.tx),
                         (self.d_jobs3, d.jobs3)):
            self._upload_range(buf, arr, 0, len(arr))
        self.ctx.h2d(self.d_levels.ptr, np.ascontiguousarray(levels, np.int16))
        self.ctx.h2d(self.d_nnz.ptr, np.ascontiguousarray(nnz, np.int32))

    def decode(self, rec):
        ctx, d, T = self.ctx, self.desc, api.TX_DTYPE.itemsize
        for a, b in d.waves():
            if self.fused:
                self._recon(None, rec, a, b)
                continue
            self._wave(rec, a, b)
            ctx._check(ctx.lib.xvcgpu_inv_transform_batch(
                ctx.h, self.pred.h_pic, rec.h_pic, self.d_tx.ptr + 3 * a * T, 3 * (b - a),
                self.d_levels.ptr, self.d_off.ptr + 3 * a * 4, self.d_nnz.ptr + 3 * a * 4))

    def results(self):
        d = self.desc
        return (d.luma["mode"].copy(), self.d_levels.to_array(np.int16, d.level_total),
                self.d_nnz.to_array(np.int32, len(d.tx)))

    def destroy(self):
        for b in (self.d_luma, self.d_chroma, self.d_jobs3, self.d_tx, self.d_off,
                  self.d_levels, self.d_nnz, self.d_dist, self.d_modes):
            b.free()
        self.pred.destroy()


class MixedPictureDecoder:
    """Decoder-side reconstruction of an inter picture whose CUs are a mix of
    uni-pred, bi-pred, LIC and intra CUs: what
    CuDecoder::DecompressInter / DecompressIntra (cu_decoder.cc) do CU by CU,
    scheduled as dependency waves over a raster of cu x cu CUs.  Plain inter
    CUs need no neighbour: wave 0 (prediction + inverse path for all of them).
    A LIC CU reads the reconstruction of its left / above CU, an intra CU that
    of left, above-left, above and above-right: wave = 1 + the latest of those.
    Per wave: xvcgpu_mc_lic_batch + xvcgpu_inv_transform_batch for the LIC CUs,
    one fused xvcgpu_intra_recon_batch for the intra CUs.

    syntax (numpy, CUs in raster order): kind[n] (0 uni L0, 1 bi, 2 LIC, 3 intra,
    4 affine L0), mv0[n,2], mv1[n,2] (1/16 pel), intra_mode[n], mv_affine[n,3,2]
    (corner vectors of the affine CUs), levels / nnz per (CU, comp)."""

    UNI, BI, LIC, INTRA, AFFINE = 0, 1, 2, 3, 4

    def __init__(self, ctx, width, height, bitdepth, qp, kind, mv0, mv1, intra_mode, cu=16,
                 mv_affine=None):
        self.ctx, self.bd, self.w, self.h = ctx, bitdepth, width, height
        self.fused_intra = cu <= 16     # else prediction picture + inverse path
        parts = cu_partition(width, height, cu)
        n = len(parts)
        per_row = (width + cu - 1) // cu
        qpc = chroma_qp(qp)
        wave = np.zeros(n, np.int64)
        for i, (x, y, w, h) in enumerate(parts):
            if kind[i] in (self.UNI, self.BI, self.AFFINE):
                continue
            deps = []
            if x:
                deps.append(i - 1)
            if y:
                deps.append(i - per_row)
            if kind[i] == self.INTRA:
                if x and y:
                    deps.append(i - per_row - 1)
                if y and x + w < width:
                    deps.append(i - per_row + 1)
            wave[i] = 1 + max([wave[d] for d in deps], default=0)
        # CUs ordered by (wave, kind): every (wave, kind) group is one batch
        self.order = order = sorted(range(n), key=lambda i: (wave[i], kind[i], i))
        self.parts = [parts[i] for i in order]
        self.kind = np.asarray(kind)[order]
        self.wave = wave[order]
        tx = np.zeros(3 * n, api.TX_DTYPE)
        jobs3 = np.zeros(3 * n, api.INTRA_DTYPE)
        uni, bi, lic, aff = [], [], [], []
        for k, i in enumerate(order):
            x, y, w, h = parts[i]
            for c in range(3):
                s = 1 if c else 0
                tx[3 * k + c] = (x >> s, y >> s, w >> s, h >> s, c, 0, 0, 0,
                                 qpc if c else qp, 0)
            if kind[i] == self.UNI:
                uni += [(x, y, w, h, c, 0, *mv0[i]) for c in range(3)]
            elif kind[i] == self.BI:
                bi += [(x, y, w, h, c, 0, *mv0[i], *mv1[i]) for c in range(3)]
            elif kind[i] == self.AFFINE:
                aff += [(x, y, w, h, c, 0, np.asarray(mv_affine[i], np.int32)) for c in range(3)]
            elif kind[i] == self.LIC:
                nb = (1 if y else 0) | (2 if x else 0)
                lic += [(x, y, w, h, c, nb, *mv0[i], x, max(0, y - cu), max(0, x - cu), y)
                        for c in range(3)]
            else:
                nb = (api.INTRA_HAS_LEFT if x else 0) | (api.INTRA_HAS_ABOVE if y else 0) | \
                    (api.INTRA_HAS_ABOVE_LEFT if x and y else 0)
                ar = max(0, min(h, width - (x + w))) if y else 0
                for c in range(3):
                    s = 1 if c else 0
                    jobs3[3 * k + c] = (x >> s, y >> s, w >> s, h >> s, c, intra_mode[i], nb,
                                        ar >> s, 0, 0)
        self.tx = tx
        self.level_off, self.level_total = api.Context.level_offsets(tx)
        self.d_tx, self.d_off = ctx.buffer(tx), ctx.buffer(self.level_off)
        self.d_jobs3 = ctx.buffer(jobs3)
        self.d_uni = ctx.buffer(np.array(uni, api.MC_DTYPE)) if uni else None
        self.d_bi = ctx.buffer(np.array(bi, api.MCBI_DTYPE)) if bi else None
        self.d_lic = ctx.buffer(np.array(lic, api.LIC_DTYPE)) if lic else None
        self.d_aff = ctx.buffer(np.array(aff, api.MCAFF_DTYPE)) if aff else None
        self.n_uni, self.n_bi, self.n_aff = len(uni), len(bi), len(aff)
        self.d_levels = ctx.alloc(2 * max(1, self.level_total))
        self.d_nnz = ctx.alloc(4 * 3 * n)
        self.pred = ctx.picture(width, height, bitdepth)
        # batches: (first CU, last CU) in the new order per (wave, kind)
        self.groups = []
        k = 0
        while k < n:
            e = k
            while e < n and self.wave[e] == self.wave[k] and self.kind[e] == self.kind[k]:
                e += 1
            self.groups.append((int(self.wave[k]), int(self.kind[k]), k, e))
            k = e
        self.n_inter = sum(e - a for wv, kd, a, e in self.groups
                           if kd in (self.UNI, self.BI, self.AFFINE))

    def load(self, levels_per_tx, nnz_per_tx):
        """levels_per_tx[3 * i + c]: w*h int16 of CU i (raster order), comp c."""
        lv = np.zeros(self.level_total, np.int16)
        nz = np.zeros(len(self.tx), np.int32)
        for k, i in enumerate(self.order):
            for c in range(3):
                a = np.asarray(levels_per_tx[3 * i + c], np.int16).reshape(-1)
                off = int(self.level_off[3 * k + c])
                lv[off:off + len(a)] = a
                nz[3 * k + c] = nnz_per_tx[3 * i + c]
        self.ctx.h2d(self.d_levels.ptr, lv)
        self.ctx.h2d(self.d_nnz.ptr, nz)

    def _inverse(self, rec, a, e):
        ctx, T = self.ctx, api.TX_DTYPE.itemsize
        ctx._check(ctx.lib.xvcgpu_inv_transform_batch(
            ctx.h, self.pred.h_pic, rec.h_pic, self.d_tx.ptr + 3 * a * T, 3 * (e - a),
            self.d_levels.ptr, self.d_off.ptr + 3 * a * 4, self.d_nnz.ptr + 3 * a * 4))

    def decode(self, ref0, ref1, rec):
        ctx, lib = self.ctx, self.ctx.lib
        J, T, L = api.INTRA_DTYPE.itemsize, api.TX_DTYPE.itemsize, api.LIC_DTYPE.itemsize
        if self.n_uni:
            ctx.mc_batch_dev(ref0, self.pred, self.d_uni.ptr, self.n_uni)
        if self.n_bi:
            ctx._check(lib.xvcgpu_mc_bipred_batch(ctx.h, ref0.h_pic, ref1.h_pic,
                                                  self.pred.h_pic, self.d_bi.ptr, self.n_bi))
        if self.n_aff:
            ctx._check(lib.xvcgpu_mc_affine_batch(ctx.h, ref0.h_pic, self.pred.h_pic,
                                                  self.d_aff.ptr, self.n_aff))
        if self.n_inter:
            self._inverse(rec, 0, self.n_inter)     # all plain inter CUs come first
        lic_done = 0
        for wv, kd, a, e in self.groups:
            if kd == self.LIC:
                ctx._check(lib.xvcgpu_mc_lic_batch(ctx.h, ref0.h_pic, rec.h_pic, self.pred.h_pic,
                                                   self.d_lic.ptr + 3 * lic_done * L,
                                                   3 * (e - a)))
                lic_done += e - a
                self._inverse(rec, a, e)
            elif kd == self.INTRA and not self.fused_intra:
                ctx._check(lib.xvcgpu_intra_pred_batch(ctx.h, rec.h_pic, self.pred.h_pic,
                                                       self.d_jobs3.ptr + 3 * a * J, 3 * (e - a)))
                self._inverse(rec, a, e)
            elif kd == self.INTRA:
                ctx._check(lib.xvcgpu_intra_recon_batch(
                    ctx.h, None, rec.h_pic, self.d_jobs3.ptr + 3 * a * J,
                    self.d_tx.ptr + 3 * a * T, 3 * (e - a), self.d_levels.ptr,
                    self.d_off.ptr + 3 * a * 4, self.d_nnz.ptr + 3 * a * 4))

    def destroy(self):
        for b in (self.d_tx, self.d_off, self.d_jobs3, self.d_uni, self.d_bi, self.d_lic,
                  self.d_aff, self.d_levels, self.d_nnz):
            if b is not None:
                b.free()
        self.pred.destroy()

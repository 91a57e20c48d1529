"""The capture -> the intra states' job arrays (kind 4: CompressIntra, cu_encoder.cc:518-541)
and the staging of the reconstructed neighbours a chain predicts from (LIC and intra
states): part of the harness's feed of captured inputs (tests/rd_serial.py)."""
import numpy as np

import rd_fixture as rf  # noqa: F401
from xvc_amd.cu_state import KIND_INTRA, NB_WIDTH  # noqa: F401


class Stager:
    """Sample strips of a capture (rows above, columns left, LM's luma rectangles) laid out
    in a staging picture, each copied to its place in the chain's reconstruction picture
    by one xvcgpu_copy_block.  Rows go to one-row shelves, everything taller to shelves of
    the tallest strip's height; equal content is stored once."""
    TALL = (130, 66)                    # shelf heights of the luma / chroma planes

    def __init__(self, api, width=NB_WIDTH):
        self.api, self.W = api, (width, width // 2)
        self.row = [[0, 0], [0, 0]]       # next free (x, shelf) of the one-row shelves
        self.tall = [[0, 0], [0, 0]]
        self.placed = {}                  # (cls, h, w, bytes) -> (is_row, px, shelf)
        self.data = []                    # (cls, is_row, px, shelf, array)
        self.jobs = []                    # (comp, is_row, px, shelf, dx, dy, w, h)

    def add(self, comp, dx, dy, arr):
        """arr [h, w] -> component comp at (dx, dy) of the destination; returns the job's index"""
        arr = np.ascontiguousarray(arr, np.uint16)
        h, w = arr.shape
        cls = 1 if comp else 0
        key = (cls, h, w, arr.tobytes())
        if key not in self.placed:
            is_row = h == 1
            cur = self.row[cls] if is_row else self.tall[cls]
            assert w <= self.W[cls] and h <= self.TALL[cls], (w, h)
            if cur[0] + w > self.W[cls]:
                cur[0], cur[1] = 0, cur[1] + 1
            self.placed[key] = (is_row, cur[0], cur[1])
            self.data.append((cls, is_row, cur[0], cur[1], arr))
            cur[0] += w
        is_row, px, shelf = self.placed[key]
        self.jobs.append((comp, is_row, px, shelf, dx, dy, w, h))
        return len(self.jobs) - 1

    def finish(self):
        """-> (planes of the staging picture, its height, the copy jobs)"""
        rows = [self.row[c][1] + 1 for c in range(2)]
        talls = [self.tall[c][1] + 1 for c in range(2)]
        height = max(rows[0] + self.TALL[0] * talls[0], 2 * (rows[1] + self.TALL[1] * talls[1]))
        height = (height + 63) // 64 * 64
        W = self.W[0]
        planes = [np.zeros((height, W), np.uint16), np.zeros((height // 2, W // 2), np.uint16),
                  np.zeros((height // 2, W // 2), np.uint16)]

        def sy(cls, is_row, shelf):
            return shelf if is_row else rows[cls] + self.TALL[cls] * shelf
        for cls, is_row, px, shelf, arr in self.data:
            y = sy(cls, is_row, shelf)
            for c in ((0,) if cls == 0 else (1, 2)):       # (U and V share the chroma layout)
                planes[c][y:y + arr.shape[0], px:px + arr.shape[1]] = arr
        jobs = np.zeros(len(self.jobs), self.api.COPY_BLOCK_DTYPE)
        for i, (comp, is_row, px, shelf, dx, dy, w, h) in enumerate(self.jobs):
            jobs[i] = (px, sy(1 if comp else 0, is_row, shelf), dx, dy, w, h, comp, 0)
        return planes, height, jobs


def intra_strips(sg, comp, x, y, w, h, nbits, above_right, below_left, smp, off, api):
    """the reference samples of an intra prediction ([above-left] [above: w +
    above_right] [left: h + below_left], present parts only) as strips; -> samples used"""
    o = off
    if nbits & api.INTRA_HAS_ABOVE_LEFT:
        sg.add(comp, x - 1, y - 1, smp[o:o + 1].reshape(1, 1))
        o += 1
    if nbits & api.INTRA_HAS_ABOVE:
        n = w + above_right
        sg.add(comp, x, y - 1, smp[o:o + n].reshape(1, n))
        o += n
    if nbits & api.INTRA_HAS_LEFT:
        n = h + below_left
        sg.add(comp, x - 1, y, smp[o:o + n].reshape(n, 1))
        o += n
    return o - off


def intra_jobs(self, sg):
    """Job arrays of the intra states (kind 4): one xvcgpu_intra_block per SATD
    pre-selection, and per TransformAndReconstruct call the prediction job, the
    transform block, the quantiser's parameters and the distortion candidate."""
    api, st, io = self.api, self.states, self.intra
    S = self.state_list
    intra_states = [n for n, s_ in enumerate(S) if s_["kind"] == KIND_INTRA]
    n_satd = sum(1 for n in intra_states if S[n]["satd"] >= 0)
    call_idx = [i for n in intra_states for i in S[n]["itx"]]
    nc = len(call_idx)
    self.in_satd_jobs = np.zeros(n_satd, api.INTRA_DTYPE)
    self.in_satd_call = np.zeros(n_satd, np.int64)
    self.in_pred = np.zeros(nc, api.INTRA_DTYPE)
    self.in_tx = np.zeros(nc, api.TX_DTYPE)
    self.in_prm = np.zeros(nc, api.RDOQ_PARAMS_DTYPE)
    self.in_cand = np.zeros(nc, api.CAND_DTYPE)
    self.in_off = np.zeros(nc, np.uint32)
    self.in_ctx = np.zeros(nc, np.int32)
    self.in_weight = np.zeros(nc, np.float64)
    self.in_comp = np.zeros(nc, np.int32)
    self.in_stage = np.zeros((nc, 2), np.int32)      # per call: first / count of its block copies
    self.in_wait = np.zeros(nc, np.int32)            # a read-back behind the call (end of a mode)
    self.in_want = io["itx"][np.asarray(call_idx, np.int64)] if nc else (io["itx"][:0] if io is not None else None)
    self.n_in_levels = 0
    if io is None or not intra_states:
        self.in_contexts = np.zeros(1, api.RDOQ_CTX_DTYPE)
        return
    calls, itx = io["calls"], io["itx"]
    qps = io["qps"].view(rf.QP_DTYPE).reshape(-1)
    self.in_contexts = np.ascontiguousarray(io["contexts"]).view(api.RDOQ_CTX_DTYPE).reshape(-1)
    k_satd = k_call = 0
    for n in intra_states:
        s_ = S[n]
        r = st[n]
        r["in_satd"], r["in_first"], r["in_count"] = -1, k_call, len(s_["itx"])
        staged = {}                 # component -> the samples staged last (bytes)
        if s_["satd"] >= 0:
            c = calls[s_["satd"]]
            x, y, w, h = int(c["x"]), int(c["y"]), int(c["w"]), int(c["h"])
            first = len(sg.jobs)
            used = intra_strips(sg, 0, x, y, w, h, int(c["neighbors"]), int(c["above_right"]),
                                      int(c["below_left"]), io["samples"], int(c["sample_off"]), api)
            staged[0] = io["samples"][int(c["sample_off"]):int(c["sample_off"]) + used].tobytes()
            r["nb_first"], r["nb_count"] = first, len(sg.jobs) - first
            jb = self.in_satd_jobs[k_satd]
            jb["x"], jb["y"], jb["w"], jb["h"], jb["comp"] = x, y, w, h, 0
            jb["neighbors"], jb["above_right"], jb["below_left"] = c["neighbors"], c["above_right"], c["below_left"]
            self.in_satd_jobs[k_satd] = jb
            self.in_satd_call[k_satd] = s_["satd"]
            r["in_satd"] = k_satd
            k_satd += 1
        for pos_, i in enumerate(s_["itx"]):
            t = itx[i]
            comp = int(t["comp"])
            x, y, w, h = int(t["x"]), int(t["y"]), int(t["w"]), int(t["h"])
            off = int(t["sample_off"])
            nbits = int(t["neighbors"])
            n_ref = ((1 if nbits & api.INTRA_HAS_ABOVE_LEFT else 0) +
                     (w + int(t["above_right"]) if nbits & api.INTRA_HAS_ABOVE else 0) +
                     (h + int(t["below_left"]) if nbits & api.INTRA_HAS_LEFT else 0))
            lm = int(t["mode"]) == 67
            n_lm = 0
            if lm:
                lx, ly = x << 1, y << 1
                x0, y0 = (lx - 3 if lx > 0 else lx), (ly - 2 if ly > 0 else ly)
                rw, rh = lx + 2 * w - x0, ly + 2 * h - y0
                n_lm = rw * rh
            blob = io["itx_samples"][off:off + n_ref + n_lm]
            first = len(sg.jobs)
            if staged.get(comp) != blob[:n_ref].tobytes():
                intra_strips(sg, comp, x, y, w, h, nbits, int(t["above_right"]),
                                   int(t["below_left"]), io["itx_samples"], off, api)
                staged[comp] = blob[:n_ref].tobytes()
            if lm and staged.get("lm") != blob[n_ref:].tobytes():
                sg.add(0, x0, y0, blob[n_ref:].reshape(rh, rw))
                staged["lm"] = blob[n_ref:].tobytes()
                staged.pop(0, None)        # (the rectangle overwrote the luma strips)
            self.in_stage[k_call] = (first, len(sg.jobs) - first)
            jb = self.in_pred[k_call]
            for f in ("x", "y", "w", "h", "comp", "mode", "neighbors", "above_right", "below_left"):
                jb[f] = t[f]
            self.in_pred[k_call] = jb
            b = self.in_tx[k_call]
            for f in ("x", "y", "w", "h", "comp", "tx_ver", "qp", "dst4x4"):
                b[f] = t[f]
            b["tx_hor"] = 6 if t["tx_skip"] else t["tx_hor"]
            b["intra_pic"] = api.TXF_RDOQ | (int(t["scan"]) << api.TXF_SCAN_SHIFT) | (1 if t["intra_pic"] else 0)
            self.in_tx[k_call] = b
            q = qps[int(t["qp_index"])]
            pr = self.in_prm[k_call]
            pr["lambda"], pr["rd_factor"] = q["lambda"][comp], q["rd_factor"][comp]
            pr["ctx_index"], pr["flags"] = 0, api.RDOQ_INTRA_CU
            self.in_prm[k_call] = pr
            cd = self.in_cand[k_call]
            cd["x"], cd["y"], cd["w"], cd["h"] = x, y, w, h
            cd["metric"], cd["qp"] = (7 if comp == 0 else 0), t["qp_luma"]
            self.in_cand[k_call] = cd
            self.in_off[k_call] = self.n_in_levels
            self.n_in_levels += w * h
            self.in_ctx[k_call] = t["ctx_index"]
            self.in_weight[k_call] = q["dist_weight"][comp]
            self.in_comp[k_call] = comp
            nxt = itx[s_["itx"][pos_ + 1]] if pos_ + 1 < len(s_["itx"]) else None
            self.in_wait[k_call] = int(nxt is None or int(nxt["comp"]) < comp or
                                       (comp == 0 and (int(nxt["comp"]) != 0 or int(nxt["mode"]) != int(t["mode"]))))
            k_call += 1

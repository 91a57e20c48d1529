"""Multi-GPU frame pass: one process per GPU, pictures sharded by rows of CUs.

North star: "frames shard by CTU rows across the GPUs of one node with RCCL
halo exchange over xGMI for in-loop filtering".  Shards are contiguous runs of
CU rows (16 luma lines; CTU rows when they divide evenly).  Per picture:

  A  local   motion search, motion compensation, residual pipeline and CU
             metadata for the own CUs; deblocking pass 0 (vertical edges never
             cross rows) on the own rows
  X1 halo    neighbour ranks swap the 4 luma / 2 chroma rows on each side of
             the shard boundary (pass-0 output) and the metadata of the
             boundary CU rows - packed per neighbour, one RCCL all-to-all
  B  local   deblocking pass 1 (horizontal edges) on the own rows plus the
             first edge row of the shard below, computed redundantly by both
             neighbours (SURVEY.md section 8e scheme B: no return traffic).
             Exact whenever no 4-tall CU touches a shard boundary (no deblock
             chain crosses it) - checked when the plan is built.
  X2 gather  every rank sends finished rows to the ranks whose next search can
             reach them: a shard's motion search reads the reference only
             within `reach` luma rows of its own rows (search range + MV clip
             margin + filter taps), so on a tall picture only neighbouring
             shards exchange rows - not an all-gather (packed per peer, one
             all-to-all whose segments to everybody else are empty)
  C  local   border extension; the PSNR walk over the 64-row blocks that
             START in the own rows (their sum over the ranks is the picture's;
             one all-reduce when the number is wanted, not per picture)

The orchestration is engine-agnostic: `GpuEngine` (HIP kernels through the
C-ABI, torch tensors as picture memory so RCCL can address row slabs) is the
product; tests drive the same plan with a CPU engine over gloo.
"""
import numpy as np

from . import api, pipeline


def search_reach(desc):
    """Largest vertical distance (luma rows) from a CU of `desc` to a reference
    row its motion search can touch before clipping/filter margins.  The
    window is centred on the predictor (1/16 pel, MotionVector::kPrecisionShift = 4),
    but the previous CU's
    full-pel vector is a start candidate that is only clipped to the picture
    and the diamond steps test one bound each (inter_tz_search.cc:117-123,
    :283-316), so both offsets are added to the range."""
    me = desc.me
    if len(me) == 0:
        return 0
    return int(me["search_range"].max()) + (int(np.abs(me["mvp_y"]).max()) + 15) // 16 + \
        int(np.abs(me["prev_y"]).max())


def _host():
    import ctypes as C
    from . import decoder
    L = decoder.load_host_library()
    if not getattr(L, "_shard_engine_bound", False):
        L.xvc_shard_rows.argtypes = [C.c_int] * 4 + [C.c_void_p] * 2
        L.xvc_shard_plan_create.restype = C.c_void_p
        L.xvc_shard_plan_create.argtypes = [C.c_int] * 8
        L.xvc_shard_plan_destroy.argtypes = [C.c_void_p]
        L.xvc_shard_plan_destroy.restype = None
        L.xvc_shard_plan_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.xvc_shard_plan_rows.restype = None
        L.xvc_shard_plan_valid_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.xvc_shard_plan_valid_rows.restype = None
        L.xvc_shard_plan_slabs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.xvc_shard_plan_traffic.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.xvc_shard_plan_traffic.restype = None
        L.xvc_shard_run.argtypes = [C.c_void_p, C.c_void_p]
        L.xvc_host_sharded_frame_pass.argtypes = [C.c_void_p, C.c_void_p]
        L.xvc_host_sharded_total_ssd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L._shard_engine_bound = True
    return L


def shard_rows(height, world, cu=16):
    """Split the CU rows of a picture into `world` contiguous shards.
    Returns [(y0, y1)] in luma lines (y1 clipped to height).  (xvc_shard_rows,
    xvc_amd/host/xvc_shard_engine.cc)"""
    import ctypes as C
    L = _host()
    n_rows = (height + cu - 1) // cu
    assert world <= n_rows, "more ranks than CU rows"
    out = []
    for r in range(world):
        y0, y1 = C.c_int32(), C.c_int32()
        assert L.xvc_shard_rows(height, world, cu, r, C.byref(y0), C.byref(y1)) == 0
        out.append((y0.value, y1.value))
    return out


class _PackedPlan:
    """One exchange as a single collective: every slab bound for a peer sits in
    one segment of a staging buffer (16-byte aligned pieces, in list order, so
    that both ends lay a peer's segment out alike)."""

    ALIGN = 16

    def __init__(self, sends, recvs, world, make_copier):
        import torch
        dev = (sends or recvs)[0][1].device
        self.send_buf, self.send_splits, pack = self._layout(sends, world, dev, torch)
        self.recv_buf, self.recv_splits, unpack = self._layout(recvs, world, dev, torch)
        unpack = [(stage, slab) for slab, stage in unpack]
        self.pack = make_copier(pack) if make_copier else _TorchCopier(pack)
        self.unpack = make_copier(unpack) if make_copier else _TorchCopier(unpack)
        self.peers = [p for p in range(world) if self.send_splits[p] or self.recv_splits[p]]

    @classmethod
    def _layout(cls, ops, world, dev, torch):
        splits, pieces, total = [0] * world, [], 0
        for peer in range(world):
            for p, t in ops:
                if p != peer:
                    continue
                assert t.dtype == torch.uint8 and t.dim() == 1
                n = (t.numel() + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN
                pieces.append((t, total, t.numel()))
                total += n
                splits[peer] += n
        # empty, not zeros: a fill kernel on torch's current stream could race
        # with the pack kernel on the engine's stream (padding bytes are never read)
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        return buf, splits, [(t, buf[off:off + n]) for t, off, n in pieces]

    def segment(self, buf, splits, peer):
        off = sum(splits[:peer])
        return buf[off:off + splits[peer]]


class _TorchCopier:
    """(source, destination) tensor pairs copied one by one (CPU engines)."""

    def __init__(self, pairs):
        self.pairs = pairs

    def __call__(self):
        for src, dst in self.pairs:
            dst.copy_(src)


class TorchComm:
    """Neighbour exchange over torch.distributed (RCCL on GPUs, gloo in the CPU
    tests): the slabs of one exchange are packed per peer and travel in ONE
    all_to_all_single (RCCL: a grouped send/recv per peer with data).  A batched
    point-to-point group per slab costs ~10 us of host time per operation
    through torch.distributed - 100-300 us per picture at 8 ranks, more than
    the picture's kernels (tools/p2p_host_cost.py)."""

    def __init__(self, dist, rank, world, group=None, packed=True):
        # group: a process group of all ranks dedicated to one picture chain, so
        # that the exchanges of concurrent chains do not queue behind each other
        self.dist, self.rank, self.world, self.group = dist, rank, world, group
        # RCCL orders its operations on the issuing stream.  A host transport
        # (gloo, used by tests that put several ranks on one GPU) touches the
        # device buffers from the CPU as soon as it is called: the stream has to
        # be drained first.
        self.host_transport = dist.get_backend(group) != "nccl"
        self.packed = packed
        self._p2p = {}
        self._plans = {}

    def exchange(self, sends, recvs, make_copier=None):
        """sends / recvs: lists of (peer, 1-D byte tensor); per peer the order
        of sends on one side matches the order of recvs on the other.  Every
        rank of the group calls this (a collective), with or without data.
        make_copier(pairs) -> callable: the engine's batched device copy."""
        if self.world == 1 and not sends and not recvs:
            return
        if not self.packed:
            return self._exchange_p2p(sends, recvs)
        # the callers hand in the same (cached) lists for every picture: lay the
        # staging buffers out once per list pair
        key = (id(sends), id(recvs))
        hit = self._plans.get(key)
        if hit is None or hit[0] is not sends or hit[1] is not recvs:
            plan = _PackedPlan(sends, recvs, self.world, make_copier) \
                if (sends or recvs) else None
            self._plans[key] = (sends, recvs, plan)
        else:
            plan = hit[2]
        d, g = self.dist, self.group
        # with more than one rank every shard has a neighbour
        assert plan is not None, "a rank without neighbours in a group of %d" % self.world
        plan.pack()
        if self.host_transport and plan.send_buf.is_cuda:
            # gloo with device tensors (several ranks on one GPU in tests): the
            # per-peer segments point to point, same layout as the collective
            self._drain(plan.send_buf)
            ops = []
            for p in plan.peers:
                if plan.send_splits[p]:
                    ops.append(d.P2POp(d.isend, plan.segment(plan.send_buf, plan.send_splits, p),
                                       p, group=g))
                if plan.recv_splits[p]:
                    ops.append(d.P2POp(d.irecv, plan.segment(plan.recv_buf, plan.recv_splits, p),
                                       p, group=g))
            for req in d.batch_isend_irecv(ops):
                req.wait()
        else:
            d.all_to_all_single(plan.recv_buf, plan.send_buf, plan.recv_splits,
                                plan.send_splits, group=g)
        plan.unpack()

    def _exchange_p2p(self, sends, recvs):
        """One point-to-point operation per slab, no staging copies."""
        d = self.dist
        g = self.group
        key = (id(sends), id(recvs))
        hit = self._p2p.get(key)
        if hit is None or hit[0] is not sends or hit[1] is not recvs:
            ops = [d.P2POp(d.isend, t, p, group=g) for p, t in sends] + \
                  [d.P2POp(d.irecv, t, p, group=g) for p, t in recvs]
            self._p2p[key] = (sends, recvs, ops)
        else:
            ops = hit[2]
        if not ops:
            return
        self._drain(ops[0].tensor)
        for req in d.batch_isend_irecv(ops):
            req.wait()

    def _drain(self, t):
        if self.host_transport and t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def allreduce_sum(self, t):
        self._drain(t)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class NativeComm:
    """The same neighbour exchange issued natively: ncclSend / ncclRecv of every
    slab inside one ncclGroup on libxvcgpu.so's communicator and its own stream
    (xvcgpu_comm_*), ordered with the context's kernels by two events - no
    staging copies, no torch.distributed on the data path.  ctx: the api.Context
    whose stream produces / consumes the slabs; comm: an api.Comm."""

    def __init__(self, ctx, comm):
        from . import api
        self.ctx, self.comm = ctx, comm
        self.rank, self.world = comm.rank, comm.world
        self.before, self.after = api.Event(ctx), api.Event(ctx)

    def exchange(self, sends, recvs, make_copier=None):
        if not sends and not recvs:
            return
        c = self.comm
        self.before.record(self.ctx)        # the slabs are final / free to overwrite
        c.wait_event(self.before)
        c.group_begin()
        for peer, t in sends:
            c.send_bytes(t.data_ptr(), t.numel() * t.element_size(), peer)
        for peer, t in recvs:
            c.recv_bytes(t.data_ptr(), t.numel() * t.element_size(), peer)
        c.group_end()
        c.record_event(self.after)
        self.after.wait(self.ctx)           # later kernels see the received rows

    def allreduce_sum(self, t):
        c = self.comm
        self.before.record(self.ctx)
        c.wait_event(self.before)
        c.all_reduce_sum_u64(t.data_ptr(), t.numel())
        c.sync()                            # the caller reads the sums on the host
        return t


class _Slab(__import__("ctypes").Structure):
    _fields_ = [("peer", __import__("ctypes").c_int32), ("kind", __import__("ctypes").c_int32),
                ("a", __import__("ctypes").c_int32), ("b", __import__("ctypes").c_int32)]


class ShardedFramePass:
    """Binding of the C++ shard engine (xvc_amd/host/xvc_shard_engine.{h,cc}): the plan
    - rows per rank, which rows / CU records travel in the halo and the gather exchange,
    scheme B's exactness precondition - and the five-step control (xvc_shard_run) live
    there.  This class turns the plan's slabs into the engine's memory views, hands the
    steps to the engine / transport it was given (the CPU tests: an oracle engine over
    gloo, driven by the same C++ control through callbacks) and, for a GpuEngine with a
    native communicator, calls the product path xvc_host_sharded_frame_pass (frame-pass
    phases on row ranges + ncclSend / ncclRecv groups; no Python between the steps)."""

    HALO = 4  # luma rows on each side of a shard boundary

    def __init__(self, engine, comm, rank, world, reach=None):
        import ctypes as C
        self.e, self.comm, self.rank, self.world = engine, comm, rank, world
        self.L = _host()
        self.rows = shard_rows(engine.h, world, engine.cu)
        self.y0, self.y1 = self.rows[rank]
        # rows of the reference a shard may read beyond its own: the search
        # window (+ predictor offset), the 8-sample MV clip margin, 4 filter
        # taps, one CU of slack.  Must be the same number on every rank.
        if reach is None:
            reach = engine.search_reach() + 8 + 4 + 16
        self.plan = self.L.xvc_shard_plan_create(engine.w, engine.h, engine.cu, world, rank, reach,
                                                 engine.min_cu_height_at(self.y0),
                                                 engine.min_cu_height_at(self.y1))
        # exactness precondition of the redundant-halo scheme
        assert self.plan, "4-tall CUs at a shard boundary need the ordered hand-off protocol"
        self._ops = {}
        self.up = rank - 1 if rank > 0 else None
        self.down = rank + 1 if rank < world - 1 else None
        self._cb_type = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
        self._ex_type = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int)
        self._gpu = None

    def __del__(self):
        if getattr(self, "plan", None):
            self.L.xvc_shard_plan_destroy(self.plan)
            self.plan = None

    # ---- the plan's slabs as views of the engine's memory ----
    def _slabs(self, which, direction):
        import ctypes as C
        ptr = C.POINTER(_Slab)()
        n = self.L.xvc_shard_plan_slabs(self.plan, which, direction, C.byref(ptr))
        return [(ptr[i].peer, ptr[i].kind, ptr[i].a, ptr[i].b) for i in range(n)]

    def _views(self, rec_idx, slabs):
        out = []
        for peer, kind, a, b in slabs:
            if kind == 0:
                out += [(peer, self.e.row_slab(rec_idx, 0, a, b)),
                        (peer, self.e.row_slab(rec_idx, 1, a // 2, b // 2)),
                        (peer, self.e.row_slab(rec_idx, 2, a // 2, b // 2))]
            else:
                out.append((peer, self.e.cu_slab(a, b)))
        return out

    def _exchange_ops(self, which, rec_idx):
        # the slabs are fixed views of the picture memory: built once per buffer
        key = (which, rec_idx)
        if key not in self._ops:
            self._ops[key] = (self._views(rec_idx, self._slabs(which, 0)),
                              self._views(rec_idx, self._slabs(which, 1)))
        return self._ops[key]

    def halo_ops(self, rec_idx):
        return self._exchange_ops(0, rec_idx)

    def gather_ops(self, rec_idx):
        return self._exchange_ops(1, rec_idx)

    def traffic(self):
        """{exchange: (RCCL operations, bytes) this rank sends per picture}."""
        import ctypes as C
        out = {}
        for which, name in ((0, "halo"), (1, "gather")):
            m, b = C.c_int64(), C.c_int64()
            self.L.xvc_shard_plan_traffic(self.plan, which, C.byref(m), C.byref(b))
            out[name] = (m.value, b.value)
        return out

    def valid_rows(self):
        """Rows of the local reconstruction that are up to date after run()."""
        import ctypes as C
        a, b = C.c_int32(), C.c_int32()
        self.L.xvc_shard_plan_valid_rows(self.plan, C.byref(a), C.byref(b))
        return (a.value, b.value)

    # ---- one step each (the loop-back tests interleave the ranks by hand) ----
    def phase_a(self, orig, ref_idx, rec_idx, ref_poc):
        if getattr(self.e, "one_call_phases", False):
            self.e.phase_a(orig, ref_idx, rec_idx, ref_poc, self.y0, self.y1)
            return
        self.e.encode(orig, ref_idx, rec_idx, ref_poc)
        self.e.deblock_rows(rec_idx, 0, self.y0, self.y1)

    def phase_b(self, rec_idx, y_end=None):
        if y_end is None:
            y_end = self.y1 + self.HALO if self.down is not None else self.y1
        if getattr(self.e, "one_call_phases", False):
            self.e.phase_b(rec_idx, self.y0, y_end)
            return
        self.e.deblock_rows(rec_idx, 1, self.y0, y_end)

    def phase_c(self, orig, rec_idx):
        if getattr(self.e, "one_call_phases", False):
            self.e.phase_c(orig, rec_idx, self.y0, self.y1)
            return
        self.e.pad(rec_idx)
        self.e.ssd(orig, rec_idx, self.y0, self.y1)

    def total_ssd(self):
        """(ssd, samples) of the last picture over all shards (a collective)."""
        t = self.comm.allreduce_sum(self.e.ssd_tensor())
        return int(t[0]), int(t[1])

    # ---- one picture: the C++ control ----
    def run(self, orig, ref_idx, rec_idx, ref_poc=0):
        import ctypes as C
        if isinstance(self.comm, NativeComm) and getattr(self.e, "one_call_phases", False):
            return self._run_native(orig, ref_idx, rec_idx, ref_poc)
        copier = getattr(self.e, "make_copier", None)
        err = []

        def phase(_, which, y0, y1, y_end):
            try:
                assert (y0, y1) == (self.y0, self.y1)
                if which == 0:
                    self.phase_a(orig, ref_idx, rec_idx, ref_poc)
                elif which == 1:
                    self.phase_b(rec_idx, y_end)
                else:
                    self.phase_c(orig, rec_idx)
                return 0
            except BaseException as ex:  # noqa: BLE001 (must not cross the C frame)
                err.append(ex)
                return 1

        def exchange(_, which, sends, ns, recvs, nr):
            try:
                ops = self._exchange_ops(which, rec_idx)
                assert (len(self._slabs(which, 0)), len(self._slabs(which, 1))) == (ns, nr)
                if copier is not None:
                    self.comm.exchange(*ops, copier)
                else:
                    self.comm.exchange(*ops)
                return 0
            except BaseException as ex:  # noqa: BLE001
                err.append(ex)
                return 1

        class Cb(C.Structure):
            _fields_ = [("user", C.c_void_p), ("phase", self._cb_type), ("exchange", self._ex_type)]
        cb = Cb(None, self._cb_type(phase), self._ex_type(exchange))
        rc = self.L.xvc_shard_run(self.plan, C.byref(cb))
        if err:
            raise err[0]
        assert rc == 0, rc

    def _run_native(self, orig, ref_idx, rec_idx, ref_poc):
        import ctypes as C
        e = self.e
        if self._gpu is None:
            class Gpu(C.Structure):
                _fields_ = [("ctx", C.c_void_p), ("comm", C.c_void_p), ("args", C.c_void_p),
                            ("d_cus", C.c_void_p), ("before", C.c_void_p), ("after", C.c_void_p)]
            self._gpu = Gpu(e.ctx.h, self.comm.comm.h, None, e.cu_mem.data_ptr(),
                            self.comm.before.h, self.comm.after.h)
        a = e.frame_pass_args(orig, ref_idx, rec_idx, ref_poc)
        self._gpu.args = C.addressof(a)
        rc = self.L.xvc_host_sharded_frame_pass(self.plan, C.byref(self._gpu))
        if rc:
            raise api.XvcGpuError("xvc_host_sharded_frame_pass: %d" % rc)


def chain_rows(cu_map, pic_w, pic_h, y0):
    """D of the ordered hand-off at the shard boundary y0 (host planning in the
    C++ layer: xvc_amd/host/xvc_shard_filter.cc, xvc_shard_chain_rows)."""
    import ctypes as C
    from . import decoder
    lib = decoder.load_host_library()
    m = np.ascontiguousarray(cu_map, np.int32)
    lib.xvc_shard_chain_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    return int(lib.xvc_shard_chain_rows(m.ctypes.data, m.shape[1], pic_w, pic_h, y0))


def shard_plan(cu_map, world, rows):
    """(all shards taller than their chains?, [D of every rank's upper boundary])
    - xvc_shard_filter_plan of the C++ layer."""
    import ctypes as C
    from . import decoder
    lib = decoder.load_host_library()
    m = np.ascontiguousarray(cu_map, np.int32)
    r = (C.c_int32 * (world + 1))(*[int(v) for v in rows])
    d = (C.c_int32 * world)()
    lib.xvc_shard_filter_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    st = int(lib.xvc_shard_filter_plan(m.ctypes.data, m.shape[1], world, r, d))
    if st not in (0, 50):               # XVCGPU_OK, XVCGPU_UNSUPPORTED (include/xvcgpu.h)
        raise ValueError("xvc_shard_filter_plan: status %d" % st)
    return st == 0, list(d)


class ShardedTreeFilter:
    """The in-loop filter of a picture with a REAL CU tree (binary splits down to
    4-tall CUs), sharded by CTU rows: SURVEY 8e scheme (A), the ordered hand-off -
    xvc_amd/host/xvc_shard_filter.h describes the five steps; this is the
    engine-agnostic mirror of xvc_host_shard_filter_run (the product path on
    RCCL) that the CPU tests drive with an oracle engine over gloo and the GPU
    test with the HIP engine in loop-back.

    engine: deblock_rows(pass, ya, yb) on its picture, row_slabs(ya, yb) -> the
    three planes' row slabs as byte tensors, cu_map (host), w, h.
    rows: world + 1 boundaries (multiples of 16)."""

    HALO = 4

    def __init__(self, engine, comm, rank, world, rows):
        self.e, self.comm, self.rank, self.world = engine, comm, rank, world
        self._rows = list(rows)
        self.y0, self.y1 = rows[rank], rows[rank + 1]
        self.up = rank - 1 if rank > 0 else None
        self.down = rank + 1 if rank < world - 1 else None
        # every rank plans every boundary: the same verdict everywhere, before the
        # first transfer (xvc_shard_filter_plan)
        ok, d_all = shard_plan(engine.cu_map, world, rows)
        if not ok:
            raise ValueError("a shard must be taller than the chain that enters it "
                             "(rows %r, chains %r)" % (list(rows), d_all))
        self.d_top = d_all[rank]

    def step_local(self):
        self.e.deblock_rows(0, self.y0, self.y1)
        self.e.deblock_rows(1, self.y0 + self.d_top, self.y1)

    def ops_down(self):
        H = self.HALO
        sends = [(self.down, t) for t in self.e.row_slabs(self.y1 - H, self.y1)] \
            if self.down is not None else []
        recvs = [(self.up, t) for t in self.e.row_slabs(self.y0 - H, self.y0)] \
            if self.up is not None else []
        return sends, recvs

    def step_strip(self):
        if self.up is not None:
            self.e.deblock_rows(1, self.y0, self.y0 + self.d_top)

    def ops_up(self):
        H = self.HALO
        sends = [(self.up, t) for t in self.e.row_slabs(self.y0 - H, self.y0)] \
            if self.up is not None else []
        recvs = [(self.down, t) for t in self.e.row_slabs(self.y1 - H, self.y1)] \
            if self.down is not None else []
        return sends, recvs

    def run(self):
        """The five steps under the C++ control (xvc_shard_filter_run): it plans every
        boundary and orders the steps; passes and exchanges come back here."""
        import ctypes as C
        from . import decoder
        L = decoder.load_host_library()
        pass_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
        ex_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
        err = []

        def do_pass(_, which, ya, yb):
            try:
                self.e.deblock_rows(which, ya, yb)
                return 0
            except BaseException as ex:  # noqa: BLE001 (must not cross the C frame)
                err.append(ex)
                return 1

        def do_exchange(_, send_down):
            try:
                self.comm.exchange(*(self.ops_down() if send_down else self.ops_up()))
                return 0
            except BaseException as ex:  # noqa: BLE001
                err.append(ex)
                return 1

        class Cb(C.Structure):
            _fields_ = [("user", C.c_void_p), ("do_pass", pass_t), ("exchange", ex_t)]
        cb = Cb(None, pass_t(do_pass), ex_t(do_exchange))
        m = np.ascontiguousarray(self.e.cu_map, np.int32)
        rows = (C.c_int32 * (self.world + 1))(*[int(v) for v in self._rows])
        L.xvc_shard_filter_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p]
        rc = L.xvc_shard_filter_run(m.ctypes.data, m.shape[1], self.rank, self.world, rows,
                                    C.byref(cb))
        if err:
            raise err[0]
        assert rc == 0, rc


class GpuTreeEngine:
    """ShardedTreeFilter's HIP engine: one picture in a torch tensor (so that row
    slabs can be exchanged), the CU records / map of the whole picture on the
    device; kernels through xvcgpu_deblock_rows."""

    def __init__(self, ctx, width, height, bitdepth, cus, cu_map, bipred, beta, tc, device):
        import torch
        self.torch, self.ctx = torch, ctx
        self.w, self.h, self.bd = width, height, bitdepth
        self.cu_map = np.ascontiguousarray(cu_map, np.int32)
        nbytes = ctx.lib.xvcgpu_picture_bytes(width, height)
        self.mem = torch.zeros(nbytes // 2, dtype=torch.int16, device=device)
        # the fill runs on torch's stream, everything else on the context's: it must
        # have landed before the first upload writes the same memory
        torch.cuda.current_stream(device).synchronize()
        self.picture = api.Picture(ctx, width, height, bitdepth, wrap_ptr=self.mem.data_ptr(),
                                   wrap_bytes=nbytes)
        self.geom = []
        for c in range(3):
            ptr, stride = self.picture.plane_ptr(c)
            self.geom.append(((ptr - self.mem.data_ptr()) // 2, stride,
                              api.BORDER_LUMA if c == 0 else api.BORDER_CHROMA))
        self.d_cus = ctx.buffer(np.ascontiguousarray(cus, api.CU_DTYPE))
        self.d_map = ctx.buffer(self.cu_map)
        self.n_cus, self.args = len(cus), (bipred, beta, tc)

    def deblock_rows(self, pass_, ya, yb):
        if ya < yb:
            self.ctx.deblock_rows_dev(self.picture, self.d_cus.ptr, self.n_cus, self.d_map.ptr,
                                      self.cu_map.shape[1], pass_, ya, yb, *self.args)

    def row_slabs(self, ya, yb):
        out = []
        for c in range(3):
            off, stride, border = self.geom[c]
            a, b = (ya, yb) if c == 0 else (ya // 2, yb // 2)
            start = off + a * stride - border
            out.append(self.mem[start:start + (b - a) * stride].view(self.torch.uint8))
        return out

    def destroy(self):
        self.d_cus.free()
        self.d_map.free()
        self.picture.destroy()


class GpuEngine:
    """HIP engine: pictures live in torch tensors (so RCCL can send row slabs)
    wrapped as xvcgpu pictures; kernels run on torch's current stream."""

    def __init__(self, ctx, width, height, bitdepth, qp, row_range, device,
                 n_pictures=2, cu=16, own_stream=False, rdoq=False):
        import torch
        self.torch = torch
        self.ctx, self.w, self.h, self.bd, self.cu = ctx, width, height, bitdepth, cu
        if own_stream:
            # keep the context's own stream and hand it to torch: RCCL operations
            # issued under `with torch.cuda.stream(engine.stream)` are ordered on
            # the very stream the kernels run on
            self.stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=device)
        else:
            self.stream = torch.cuda.current_stream(device)
            ctx.set_stream(self.stream.cuda_stream)
        nbytes = ctx.lib.xvcgpu_picture_bytes(width, height)
        self.mem, self.pictures, self.geom = [], [], []
        for _ in range(n_pictures):
            t = torch.zeros(nbytes // 2, dtype=torch.int16, device=device)
            pic = api.Picture(ctx, width, height, bitdepth, wrap_ptr=t.data_ptr(),
                              wrap_bytes=nbytes)
            self.mem.append(t)
            self.pictures.append(pic)
            g = []
            for c in range(3):
                ptr, stride = pic.plane_ptr(c)
                g.append(((ptr - t.data_ptr()) // 2, stride,
                          api.BORDER_LUMA if c == 0 else api.BORDER_CHROMA))
            self.geom.append(g)
        self.fp = pipeline.FramePass(ctx, width, height, bitdepth, qp=qp, cu=cu,
                                     row_range=row_range, rdoq=rdoq)
        d = self.fp.desc
        self.cus_per_row = d.cus_per_row
        # CU metadata in a torch tensor so boundary rows can be exchanged
        self.cu_mem = torch.zeros(d.n_cus_total * api.CU_DTYPE.itemsize,
                                  dtype=torch.uint8, device=device)
        self.fp.d_cus.free()
        self.fp.d_cus = _ExternalBuffer(ctx, self.cu_mem.data_ptr())
        self._parts = pipeline.cu_partition(width, height, cu)
        # PSNR parts of the own blocks, in a tensor so they can be all-reduced
        self.ssd_mem = torch.zeros(2, dtype=torch.int64, device=device)
        # the tensors' zero fills ran on torch's current stream; with own_stream the
        # kernels run on another one
        torch.cuda.current_stream(device).synchronize()

    def min_cu_height_at(self, y):
        if y <= 0 or y >= self.h:
            return 64
        hs = [p[3] for p in self._parts if p[1] == y or p[1] + p[3] == y]
        return min(hs) if hs else 64

    def encode(self, orig, ref_idx, rec_idx, ref_poc):
        self.fp.encode(orig, self.pictures[ref_idx], self.pictures[rec_idx], ref_poc)

    def deblock_rows(self, rec_idx, pass_, ya, yb):
        self.fp.deblock_rows(self.pictures[rec_idx], pass_, ya, yb)

    def pad(self, rec_idx):
        self.ctx.pad_border(self.pictures[rec_idx])

    # each phase of the sharded pass behind one C call (xvcgpu_frame_pass)
    @property
    def one_call_phases(self):
        return self.fp.fused and self.cu <= 16 and self.fp.desc.n_cus > 0

    def phase_a(self, orig, ref_idx, rec_idx, ref_poc, y0, y1):
        self.fp.run_phases(orig, self.pictures[ref_idx], self.pictures[rec_idx],
                           api.FP_ENCODE | api.FP_DEBLOCK_V, ref_poc, rows=(y0, y1))

    def phase_b(self, rec_idx, y0, y_end):
        self.fp.run_phases(None, None, self.pictures[rec_idx], api.FP_DEBLOCK_H,
                           rows=(y0, y_end), dbh_end=y_end)

    def phase_c(self, orig, rec_idx, y0, y1):
        self.fp.run_phases(orig, None, self.pictures[rec_idx], api.FP_PAD | api.FP_SSD,
                           ssd_rows=(y0, y1), d_ssd=self.ssd_mem.data_ptr())

    def search_reach(self):
        return search_reach(self.fp.desc)

    def frame_pass_args(self, orig, ref_idx, rec_idx, ref_poc):
        """The picture's xvcgpu_frame_pass_args (row ranges are the shard engine's)."""
        a = self.fp._args()
        a.orig = orig.h_pic
        a.ref = self.pictures[ref_idx].h_pic
        a.rec, a.ref_poc = self.pictures[rec_idx].h_pic, ref_poc
        a.d_ssd = self.ssd_mem.data_ptr()
        return a

    def ssd(self, orig, rec_idx, ya=0, yb=1 << 30):
        self.ctx.picture_ssd_dev(orig, self.pictures[rec_idx], 0, self.bd,
                                 self.ssd_mem.data_ptr(), ya, yb)

    def ssd_tensor(self):
        return self.ssd_mem

    def row_slab(self, rec_idx, comp, ya, yb):
        off, stride, border = self.geom[rec_idx][comp]
        # full padded rows: from the left border of row ya to the end of row yb-1
        start = off + ya * stride - border
        # as bytes: torch's NCCL / RCCL backend does not take 16-bit integer tensors
        return self.mem[rec_idx][start:start + (yb - ya) * stride].view(self.torch.uint8)

    def cu_slab(self, first_cu, n):
        s = api.CU_DTYPE.itemsize
        return self.cu_mem[first_cu * s:(first_cu + n) * s]

    def make_copier(self, pairs):
        """(source, destination) device tensor pairs -> a callable that copies
        them all in one launch on the engine's stream (xvcgpu_copy_segments)."""
        segs = np.zeros(len(pairs), api.SEG_DTYPE)
        for i, (src, dst) in enumerate(pairs):
            assert src.numel() == dst.numel() or dst.numel() >= src.numel()
            segs[i] = (src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size())
        d_segs = self.torch.empty(max(1, segs.nbytes), dtype=self.torch.uint8,
                                  device=self.mem[0].device)
        self.ctx.h2d(d_segs.data_ptr(), segs)
        ctx, n, keep = self.ctx, len(pairs), (d_segs, pairs)

        def run():
            ctx._check(ctx.lib.xvcgpu_copy_segments(ctx.h, keep[0].data_ptr(), n))
        return run


def torch_stream_of(ctx, device, own_stream):
    """Context manager: torch allocations / ops inside belong to the stream the
    engine will use."""
    import contextlib
    import torch
    if not own_stream:
        return contextlib.nullcontext()
    return torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_ptr(), device=device))


class _ExternalBuffer:
    """DeviceBuffer look-alike over memory owned by a torch tensor."""

    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr

    def to_array(self, dtype, count):
        out = np.zeros(count, dtype)
        self.ctx._check(self.ctx.lib.xvcgpu_memcpy_d2h(self.ctx.h, out.ctypes.data,
                                                       self.ptr, out.nbytes))
        return out

    def free(self):
        self.ptr = None


def make_gpu_sharded(ctx, width, height, bitdepth, qp, rank, world, device, dist,
                     group=None, own_stream=False, rdoq=False, native_comm=None):
    """The engine's kernels run on torch's current stream at the time of this
    call, or - own_stream - on the context's own stream, exposed to torch as
    `runner.e.stream`; either way call run() under `with torch.cuda.stream(
    runner.e.stream)` for a chain of its own.  `group`: the chain's own process
    group, if any."""
    rows = shard_rows(height, world)
    with torch_stream_of(ctx, device, own_stream):
        engine = GpuEngine(ctx, width, height, bitdepth, qp, rows[rank], device,
                           own_stream=own_stream, rdoq=rdoq)
    comm = NativeComm(ctx, native_comm) if native_comm is not None else \
        TorchComm(dist, rank, world, group)
    return ShardedFramePass(engine, comm, rank, world)

"""Picture-level parallel frame passes: one rank's walk over the timeline of
xvc_amd/host/xvc_picture_schedule.h (the reference's ThreadEncoder policy,
thread_encoder.cc:99-159, with GPUs and their picture slots as the workers).

The walk itself (which entries name this rank, in which order) is C++
(xvc_schedule_run); this module supplies what an entry does on the device:

  encode    the hot-path frame pass of the picture against its nearest L0
            reference, on the stream of the picture's slot, after the events of
            the pictures it reads; the reconstruction lands in the ring entry
            of the picture and an event marks it ready;
  transfer  xvcgpu_comm_send_picture / _recv_picture (RCCL over xGMI) on the
            communicator's stream, ordered by the same events.

Ring of picture buffers per rank: entry = picture index modulo ring size,
ring >= window + 2 sub-GOPs (a reference lives at most one sub-GOP either side
of its consumers).  Before an entry is overwritten the writer waits for
the picture that wrote its previous content and for everything that read it (the
events of the pictures that listed it, the sends that shipped it).

The device engine is C++ (xvc_amd/host/xvc_picture_engine.cc; GpuPictureEngine below is
its binding).  The engines used by tests/ (CPU oracle + torch.distributed gloo) implement
encode, send, recv in Python and are driven by the same C++ timeline walk
(xvc_schedule_run) through run_rank."""
import numpy as np

from . import api, pipeline


def ring_size(schedule):
    return schedule.window + 2 * schedule.sub_gop_length + 1


def run_rank(schedule, rank, engine, first_op=0, end_op=-1):
    """Walks the timeline (or entries [first_op, end_op) of it) as `rank`.
    engine.encode(p, index, ref_indices), engine.send(p, index, dst),
    engine.recv(p, index, src)."""
    if hasattr(engine, "eng"):          # the C++ engine walks the timeline itself
        return engine.run(first_op, end_op)
    P = schedule.pictures
    idx = schedule.index_of_poc

    def refs(p):
        out = []
        for l in range(2):
            for k in range(int(p["num_ref"][l])):
                j = idx[int(p["ref_poc"][l][k])]
                if j not in out:
                    out.append(j)
        return out

    schedule.run(rank,
                 lambda p, i: engine.encode(p, i, refs(p)),
                 lambda p, i, dst: engine.send(p, i, dst),
                 lambda p, i, src: engine.recv(p, i, src), first_op, end_op)


class GpuPictureEngine:
    """Binding of the C++ picture engine (xvc_amd/host/xvc_picture_engine.{h,cc}): this
    class allocates what a rank needs - `slots` contexts (streams) each with a
    FramePass's buffers, a ring of reconstructed pictures, the synthetic originals, the
    RCCL communicator when there is more than one rank - and hands it over; the walk
    (ring entries, events, frame passes, transfers) is C++.

    after_encode(picture_index): optional hook called once a picture's encode has been
    enqueued (tests download the ring entry before it is re-used)."""

    def __init__(self, ctx, schedule, rank, width, height, bitdepth, qp, origs, comm=None,
                 rdoq=True, border=128):
        import ctypes as C
        from . import schedule as sched_mod
        self.ctx, self.s, self.rank, self.comm = ctx, schedule, rank, comm
        self.w, self.h, self.bd = width, height, bitdepth
        self.ring = ring_size(schedule)
        self.slots = schedule.slots_per_rank
        self.ctxs = [ctx] + [api.Context(ctx.device) for _ in range(self.slots - 1)]
        self.fps = [pipeline.FramePass(c, width, height, bitdepth, qp=qp, rdoq=rdoq)
                    for c in self.ctxs]
        self.origs = origs                       # padded device pictures, cycled by POC
        self.recs = [ctx.picture(width, height, bitdepth) for _ in range(self.ring)]
        self.after_encode = None
        self._err = []
        L = self.L = sched_mod.lib()

        class Desc(C.Structure):
            _fields_ = [("schedule", C.c_void_p), ("rank", C.c_int32), ("n_slots", C.c_int32),
                        ("ctxs", C.c_void_p), ("slot_args", C.c_void_p), ("comm", C.c_void_p),
                        ("orig_of_picture", C.c_void_p), ("ring", C.c_int32),
                        ("recs", C.c_void_p), ("after_encode", C.c_void_p),
                        ("host_send", C.c_void_p), ("host_recv", C.c_void_p), ("user", C.c_void_p)]
        self._cb_type = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)

        def hook(_, index):
            try:
                if self.after_encode is not None:
                    self.after_encode(index)
                return 0
            except BaseException as ex:  # noqa: BLE001 (must not cross the C frame)
                self._err.append(ex)
                return 1
        self._hook = self._cb_type(hook)
        # a communicator without a native handle: a host transport (testing aid)
        native = getattr(comm, "h", None) is not None
        xfer_type = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)

        def xfer(fn):
            def call(_, entry, peer):
                try:
                    fn(self.recs[entry], peer)
                    return 0
                except BaseException as ex:  # noqa: BLE001
                    self._err.append(ex)
                    return 1
            return xfer_type(call)
        self._xfer = (xfer(comm.send_picture), xfer(comm.recv_picture)) \
            if comm is not None and not native else (None, None)
        n = len(schedule.pictures)
        self._keep = [
            (C.c_void_p * self.slots)(*[c.h for c in self.ctxs]),
            [fp._args() for fp in self.fps],
            (C.c_void_p * n)(*[self._orig(int(p["poc"])).h_pic for p in schedule.pictures]),
            (C.c_void_p * self.ring)(*[r.h_pic for r in self.recs])]
        args = (C.c_void_p * self.slots)(*[C.addressof(a) for a in self._keep[1]])
        self._keep.append(args)
        d = Desc(schedule.h, rank, self.slots, C.addressof(self._keep[0]), C.addressof(args),
                 comm.h if native else None, C.addressof(self._keep[2]), self.ring,
                 C.addressof(self._keep[3]), C.cast(self._hook, C.c_void_p),
                 C.cast(self._xfer[0], C.c_void_p) if self._xfer[0] else None,
                 C.cast(self._xfer[1], C.c_void_p) if self._xfer[1] else None, None)
        L.xvc_host_picture_engine_create.restype = C.c_void_p
        L.xvc_host_picture_engine_create.argtypes = [C.c_void_p]
        L.xvc_host_picture_engine_run.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.xvc_host_picture_engine_destroy.argtypes = [C.c_void_p]
        L.xvc_host_picture_engine_destroy.restype = None
        L.xvc_host_picture_engine_holds.argtypes = [C.c_void_p, C.c_int]
        L.xvc_host_picture_engine_encoded.argtypes = [C.c_void_p]
        self.eng = L.xvc_host_picture_engine_create(C.byref(d))
        if not self.eng:
            raise api.XvcGpuError("xvc_host_picture_engine_create failed")

    def _orig(self, poc):
        F = len(self.origs)
        if F == 1:
            return self.origs[0]
        k = poc % (2 * F - 2)
        return self.origs[k if k < F else 2 * F - 2 - k]

    @property
    def holds(self):
        """Picture index held by every ring entry (-1: none)."""
        return [self.L.xvc_host_picture_engine_holds(self.eng, e) for e in range(self.ring)]

    @property
    def encoded(self):
        return self.L.xvc_host_picture_engine_encoded(self.eng)

    def run(self, first_op=0, end_op=-1):
        """Timeline entries [first_op, end_op) as this rank (asynchronous on the slots'
        and the communicator's streams)."""
        st = self.L.xvc_host_picture_engine_run(self.eng, first_op, end_op)
        if self._err:
            raise self._err.pop(0)
        if st:
            raise api.XvcGpuError("xvc_host_picture_engine_run: %d" % st)

    def sync(self):
        for c in self.ctxs:
            c.sync()
        if self.comm:
            self.comm.sync()

    def download(self, index, border=128):
        e = index % self.ring
        assert self.L.xvc_host_picture_engine_holds(self.eng, e) == index
        self.sync()
        return self.recs[e].download(border)

    def __del__(self):
        if getattr(self, "eng", None):
            self.L.xvc_host_picture_engine_destroy(self.eng)
            self.eng = None

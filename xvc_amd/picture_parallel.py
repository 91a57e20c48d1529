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

The engines used by tests/ (CPU oracle + torch.distributed gloo) implement the
same three methods as GpuPictureEngine: encode, send, recv."""
import numpy as np

from . import api, pipeline


def ring_size(schedule):
    return schedule.window + 2 * schedule.sub_gop_length + 1


def run_rank(schedule, rank, engine, first_op=0, end_op=-1):
    """Walks the timeline (or entries [first_op, end_op) of it) as `rank`.
    engine.encode(p, index, ref_indices), engine.send(p, index, dst),
    engine.recv(p, index, src)."""
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
    """The device side of one rank: `slots` contexts (streams) each with a
    FramePass, a ring of reconstructed pictures, the synthetic originals, and -
    when there is more than one rank - the RCCL communicator."""

    def __init__(self, ctx, schedule, rank, width, height, bitdepth, qp, origs, comm=None,
                 rdoq=True, border=128):
        self.ctx, self.s, self.rank, self.comm = ctx, schedule, rank, comm
        self.w, self.h, self.bd = width, height, bitdepth
        self.ring = ring_size(schedule)
        self.slots = schedule.slots_per_rank
        self.ctxs = [ctx] + [api.Context(ctx.device) for _ in range(self.slots - 1)]
        self.fps = [pipeline.FramePass(c, width, height, bitdepth, qp=qp, rdoq=rdoq)
                    for c in self.ctxs]
        self.origs = origs                       # padded device pictures, cycled by POC
        self.recs = [ctx.picture(width, height, bitdepth) for _ in range(self.ring)]
        self.ready = [api.Event(ctx) for _ in range(self.ring)]
        self.readers = [[] for _ in range(self.ring)]   # events to wait for before overwriting
        self.holds = [-1] * self.ring
        self._pool, self._pool_i = [api.Event(ctx) for _ in range(4 * self.ring)], 0
        self.encoded = 0

    def _event(self):
        e = self._pool[self._pool_i % len(self._pool)]
        self._pool_i += 1
        return e

    def _orig(self, poc):
        F = len(self.origs)
        if F == 1:
            return self.origs[0]
        k = poc % (2 * F - 2)
        return self.origs[k if k < F else 2 * F - 2 - k]

    def _claim(self, index, wait):
        """Entry of picture `index`, safe to overwrite once `wait(ev)` has been
        applied to every reader of what it held."""
        e = index % self.ring
        if self.holds[e] >= 0:
            # ... and for whoever wrote it: a picture nobody referenced has no readers,
            # and its slot may still be at work on the entry
            wait(self.ready[e])
        for ev in self.readers[e]:
            wait(ev)
        self.readers[e] = []
        self.holds[e] = index
        return e

    def encode(self, p, index, ref_indices):
        c, fp = self.ctxs[int(p["slot"])], self.fps[int(p["slot"])]
        e = self._claim(index, lambda ev: ev.wait(c))
        if p["intra"]:
            # stands in for the intra picture of the segment: the padded original
            c._check(c.lib.xvcgpu_picture_copy(c.h, self.recs[e].h_pic,
                                               self._orig(int(p["poc"])).h_pic))
        else:
            for j in ref_indices:
                assert self.holds[j % self.ring] == j, (index, j)
                self.ready[j % self.ring].wait(c)
            r = ref_indices[0] % self.ring      # nearest L0 picture: the frame pass's reference
            fp.run(self._orig(int(p["poc"])), self.recs[r], self.recs[e],
                   ref_poc=int(self.s.pictures[ref_indices[0]]["poc"]))
        self.ready[e].record(c)
        for j in ref_indices:
            self.readers[j % self.ring].append(self.ready[e])
        self.encoded += 1

    def send(self, p, index, dst):
        e = index % self.ring
        assert self.holds[e] == index
        self.comm.wait_event(self.ready[e])
        self.comm.send_picture(self.recs[e], dst)
        ev = self._event()
        self.comm.record_event(ev)
        self.readers[e].append(ev)

    def recv(self, p, index, src):
        e = self._claim(index, self.comm.wait_event)
        self.comm.recv_picture(self.recs[e], src)
        self.comm.record_event(self.ready[e])

    def sync(self):
        for c in self.ctxs:
            c.sync()
        if self.comm:
            self.comm.sync()

    def download(self, index, border=128):
        e = index % self.ring
        assert self.holds[e] == index
        self.sync()
        return self.recs[e].download(border)

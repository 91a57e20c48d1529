"""ctypes binding of the picture-level schedule in libxvchost.so
(xvc_amd/host/xvc_picture_schedule.h): the reference's sub-GOP structure,
reference lists and ThreadEncoder policy, played on ranks x picture slots."""
import ctypes as C

import numpy as np

from . import build

PICTURE_DTYPE = np.dtype([
    ("poc", "<i4"), ("doc", "<i4"), ("tid", "<i4"), ("intra", "<i4"), ("num_ref", "<i4", (2,)),
    ("ref_poc", "<i4", (2, 5)), ("is_reference", "<i4"), ("worker", "<i4"), ("rank", "<i4"),
    ("slot", "<i4"), ("start", "<i4"), ("finish", "<i4")])
OP_DTYPE = np.dtype([("kind", "<i4"), ("picture", "<i4"), ("src_rank", "<i4"),
                     ("dst_rank", "<i4"), ("time", "<i4")])
ENCODE, TRANSFER = 0, 1


class _Picture(C.Structure):
    _fields_ = [("poc", C.c_int32), ("doc", C.c_int32), ("tid", C.c_int32), ("intra", C.c_int32),
                ("num_ref", C.c_int32 * 2), ("ref_poc", (C.c_int32 * 5) * 2),
                ("is_reference", C.c_int32), ("worker", C.c_int32), ("rank", C.c_int32),
                ("slot", C.c_int32), ("start", C.c_int32), ("finish", C.c_int32)]


_ENCODE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_Picture), C.c_int)
_XFER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_Picture), C.c_int, C.c_int)


class _Callbacks(C.Structure):
    _fields_ = [("encode", _ENCODE_CB), ("send", _XFER_CB), ("recv", _XFER_CB)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build.build()
        L = C.CDLL(build.build_host())
        L.xvc_schedule_create.restype = C.c_void_p
        L.xvc_schedule_create.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int]
        L.xvc_schedule_destroy.argtypes = [C.c_void_p]
        for f in ("num_pictures", "num_ops", "makespan", "window"):
            getattr(L, "xvc_schedule_" + f).argtypes = [C.c_void_p]
        L.xvc_schedule_pictures.restype = C.c_void_p
        L.xvc_schedule_pictures.argtypes = [C.c_void_p]
        L.xvc_schedule_ops.restype = C.c_void_p
        L.xvc_schedule_ops.argtypes = [C.c_void_p]
        L.xvc_schedule_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Callbacks), C.c_void_p]
        L.xvc_schedule_run_range.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Callbacks),
                                             C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def doc_from_poc(poc, length):
    return lib().xvc_sched_doc_from_poc(poc, length)


def poc_from_doc(doc, length):
    return lib().xvc_sched_poc_from_doc(doc, length)


def tid_from_doc(doc, length):
    return lib().xvc_sched_tid_from_doc(doc, length)


class Schedule:
    """pictures: structured array in coding order (PICTURE_DTYPE); ops: the
    timeline (OP_DTYPE).  run(rank, encode, send, recv) walks the timeline in
    C++ and calls back for the entries that name `rank`."""

    def __init__(self, num_pictures, sub_gop_length=16, num_ref_pics=2, ranks=1, slots_per_rank=1,
                 layer_cost=None):
        L = lib()
        cost = None if layer_cost is None else np.ascontiguousarray(layer_cost, np.int32)
        self.h = L.xvc_schedule_create(num_pictures, sub_gop_length, num_ref_pics, ranks,
                                       slots_per_rank, None if cost is None else cost.ctypes.data,
                                       0 if cost is None else len(cost))
        if not self.h:
            raise ValueError("invalid schedule arguments")
        n = L.xvc_schedule_num_pictures(self.h)
        buf = (C.c_char * (n * PICTURE_DTYPE.itemsize)).from_address(L.xvc_schedule_pictures(self.h))
        self.pictures = np.frombuffer(buf, PICTURE_DTYPE).copy()
        m = L.xvc_schedule_num_ops(self.h)
        buf = (C.c_char * (m * OP_DTYPE.itemsize)).from_address(L.xvc_schedule_ops(self.h))
        self.ops = np.frombuffer(buf, OP_DTYPE).copy()
        self.makespan = L.xvc_schedule_makespan(self.h)
        self.window = L.xvc_schedule_window(self.h)
        self.sub_gop_length, self.ranks, self.slots_per_rank = sub_gop_length, ranks, slots_per_rank
        self.index_of_poc = {int(p["poc"]): i for i, p in enumerate(self.pictures)}

    def run(self, rank, encode, send=None, recv=None, first_op=0, end_op=-1):
        """encode(picture_record, index), send(record, index, dst), recv(record,
        index, src); exceptions raised inside propagate after the walk stops.
        first_op / end_op: a part of the timeline only."""
        err = []

        def guard(fn, *a):
            try:
                fn(*a)
                return 0
            except BaseException as e:   # noqa: BLE001 - re-raised below
                err.append(e)
                return 1
        cb = _Callbacks(
            _ENCODE_CB(lambda u, p, i: guard(encode, self.pictures[i], i)),
            _XFER_CB(lambda u, p, i, r: guard(send, self.pictures[i], i, r) if send else 0),
            _XFER_CB(lambda u, p, i, r: guard(recv, self.pictures[i], i, r) if recv else 0))
        st = lib().xvc_schedule_run_range(self.h, rank, C.byref(cb), None, first_op, end_op)
        if err:
            raise err[0]
        if st:
            raise RuntimeError("schedule walk stopped with status %d" % st)

    def close(self):
        if self.h:
            lib().xvc_schedule_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

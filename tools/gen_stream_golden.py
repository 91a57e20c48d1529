#!/usr/bin/env python3
"""Whole-stream golden fixtures (authoring container only: needs
oracle/_ref/libxvcref.so, i.e. /root/reference).

    python tools/gen_stream_golden.py [c0] [c1] ...

For each clip: generate the integer synthetic clip (xvc_amd/synth.py, SURVEY
8d), encode it with the REFERENCE encoder through its public C API (xvcenc's
defaults, the clip's QP), decode the stream with the REFERENCE decoder and
capture, per picture in decoding order, the parsed syntax the decoder's
reconstruction stage starts from (CU tree leaves, modes, vectors, transform
types, levels) and what it produces (planes before / after the in-loop filter,
the picture MD5).  Written to tests/golden/stream_<clip>.npz.

The fixture is data (bitstream bytes, syntax records, planes); no reference
source goes into it.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
import stream_fixture as sf  # noqa: E402
from xvc_amd import synth  # noqa: E402

# name: (width, height, frames, qp, sub_gop_length (0 = default), keep planes of
# the first k pictures, explicit encoder settings)
CLIPS = {
    # BASELINE config 0: CIF, 10 frames, QP 32, xvcenc defaults
    "c0": dict(w=352, h=288, n=10, qp=32, sub_gop=0, planes=3, pre=0),
    # BASELINE config 1 (1080p QP 32), the first pictures: one intra picture +
    # one sub-GOP of 4 (hierarchical B); planes are pinned by MD5 only
    "c1": dict(w=1920, h=1080, n=5, qp=32, sub_gop=4, planes=0, pre=0),
    # BASELINE config 1 as SURVEY 8d specifies it: 33 pictures (two default
    # sub-GOPs of 16 + 1), xvcenc's defaults; MD5 only.  The decoder figure of
    # bench.py and tests/test_gpu_stream.py use it; the motion-search / RD-search
    # captures stay on the short clip above.
    "c1x": dict(w=1920, h=1080, n=33, qp=32, sub_gop=0, planes=0, pre=0),
    # CIF at the ends of the QP range (dense levels with escape codes / nearly empty
    # blocks): the RD-search captures of tools/gen_rd_golden.py at other operating
    # points than QP 32; MD5 only
    "c0q22": dict(w=352, h=288, n=5, qp=22, sub_gop=4, planes=0, pre=0),
    "c0q37": dict(w=352, h=288, n=5, qp=37, sub_gop=4, planes=0, pre=0),
    # small, for CPU-side checks of the host driver (oracle engine)
    "tiny": dict(w=136, h=72, n=5, qp=27, sub_gop=4, planes=5, pre=5),
}


def encode(lib, clip, w, h, n, qp, sub_gop, threads=-1):
    frames = np.concatenate([np.concatenate([p.reshape(-1) for p in clip.frame(i)])
                             for i in range(n)]).astype(np.uint8)
    cap = 64 << 20
    out = np.zeros(cap, np.uint8)
    lib.xr_stream_encode.restype = C.c_long
    lib.xr_stream_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_long]
    t = time.time()
    used = lib.xr_stream_encode(w, h, 8, 0, 30.0, qp, sub_gop, -1, -1, threads, None, n,
                                frames.ctypes.data, out.ctypes.data, cap)
    assert used > 0, used
    print("  encoded %d frames -> %d bytes in %.1f s" % (n, used, time.time() - t))
    return out[:used].copy()


def main(names):
    lib = C.CDLL(ol.REF_SO)
    reuse = "--reuse" in names      # keep the committed stream, only re-run the decoder side
    for name in [n for n in names if not n.startswith("--")]:
        c = CLIPS[name]
        print("clip %s: %dx%d, %d frames, QP %d" % (name, c["w"], c["h"], c["n"], c["qp"]))
        path = os.path.join(sf.GOLDEN, "stream_%s.npz" % name)
        if reuse and os.path.exists(path):
            stream = np.load(path)["stream"]
        else:
            clip = synth.SyntheticClip(c["w"], c["h"], 8)
            stream = encode(lib, clip, c["w"], c["h"], c["n"], c["qp"], c["sub_gop"])
        pics = sf.decode_with_reference(stream, keep_planes=True)
        arrays = {"stream": stream,
                  "info": np.stack([p[0] for p in pics]).view(np.uint8)}
        for i, (info, cus, lv, pre, post) in enumerate(pics):
            md5 = sf.picture_md5(post, int(info["bitdepth"]))
            assert np.array_equal(md5, info["md5"]), "host MD5 restatement differs"
            arrays["cus_%d" % i] = cus.view(np.uint8)
            arrays["levels_%d" % i] = lv
            for k in range(3):
                if i < c["planes"]:
                    arrays["post_%d_%d" % (i, k)] = post[k]
                if i < c["pre"]:
                    arrays["pre_%d_%d" % (i, k)] = pre[k]
            # invariants the reconstruction stage relies on
            inter = cus["pred_mode"] == 1
            aff = cus["affine"] == 1
            mv = cus["mv"].astype(np.int64)
            assert np.array_equal(mv[aff][:, :, 3], mv[aff][:, :, 1] + mv[aff][:, :, 2]
                                  - mv[aff][:, :, 0])
            na = inter & ~aff
            assert all(np.array_equal(mv[na][:, :, k], mv[na][:, :, 0]) for k in (1, 2, 3))
            assert not np.any(cus["lic"][aff])
            for cu in cus:
                for k in range(3):
                    if cu["cbf"][k] and not cu["tx_skip"][k]:
                        n = (int(cu["w"]) * int(cu["h"])) >> (2 if k else 0)
                        a = lv[int(cu["level_off"][k]):int(cu["level_off"][k]) + n]
                        assert bool(cu["dc_only"][k]) == (np.count_nonzero(a) == 1 and a[0] != 0)
            kinds = (int((cus["pred_mode"] == 0).sum()), int(cus["affine"].sum()),
                     int(cus["lic"].sum()), int((cus["inter_dir"] == 2).sum()))
            print("  pic %d: poc %d tid %d type %d qp %d, %d CUs (intra %d, affine %d, lic %d, "
                  "bi %d), %d levels, md5 %s" %
                  (i, info["poc"], info["tid"], info["pic_type"], info["pic_qp"], len(cus),
                   *kinds, len(lv), bytes(info["md5"]).hex()))
        np.savez_compressed(path, **arrays)
        print("  -> %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))
        update_manifest("stream_%s.npz" % name)


def update_manifest(fname):
    """tests/golden/MANIFEST.md5: one line per fixture (test_golden_manifest)."""
    import hashlib
    mpath = os.path.join(sf.GOLDEN, "MANIFEST.md5")
    lines = [l for l in open(mpath).read().split("\n") if l.strip() and l.split()[1] != fname]
    digest = hashlib.md5(open(os.path.join(sf.GOLDEN, fname), "rb").read()).hexdigest()
    lines.append("%s  %s" % (digest, fname))
    open(mpath, "w").write("\n".join(sorted(lines, key=lambda l: l.split()[1])) + "\n")


if __name__ == "__main__":
    main(sys.argv[1:] or ["tiny", "c0"])

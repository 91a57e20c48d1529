"""Reconstruction of REAL reference streams on the device: the
parsed syntax of every picture of streams the reference encoder produced
(tests/golden/stream_*.npz, captured from the reference decoder) goes through
the C++ host decoder (xvc_gpu::PictureDecoder, libxvchost.so) and the C-ABI;
every picture must equal what the reference decoder output - the committed
planes where the fixture holds them, the picture MD5 (the value the stream
itself carries for its reference pictures) everywhere, and the oracle decoder's
planes (which localise a mismatch)."""
import time

import numpy as np
import pytest

import stream_fixture as sf
from xvc_amd import decoder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def decode_stream_on_gpu(api, ctx, fx, check, one_launch_intra=True):
    from xvc_amd import decoder
    w, h, bd = (int(fx.info[0][k]) for k in ("width", "height", "bitdepth"))
    dec = decoder.PictureDecoder(ctx, w, h, bd)
    dec.one_launch_intra(one_launch_intra)
    done = {}
    try:
        for i in range(fx.n):
            info = fx.info[i]
            ps, cs = sf.to_syntax(info, fx.cus(i))
            refs = [[done[int(info["ref_poc"][l][k])] for k in range(int(info["num_ref"][l]))]
                    for l in range(2)]
            rec = ctx.picture(w, h, bd)
            dec.decode(ps, cs, fx.levels(i), refs, rec)
            ctx.sync()
            done[int(info["poc"])] = rec
            check(i, rec, dec)
    finally:
        dec.destroy()
        for p in done.values():
            p.destroy()


@pytest.mark.parametrize("name,one_launch", [("tiny", True), ("c0", True), ("c1", True),
                                             ("c1x", True), ("tiny", False), ("c1", False),
                                             ("c0q22", True), ("c0q37", True)])
def test_gpu_reconstructs_reference_stream(gpu, name, one_launch):
    """one_launch: the intra picture's dependency waves as ONE cooperative launch
    (xvcgpu_intra_recon_waves, the default) or as a launch set per wave."""
    api, ctx = gpu
    fx = sf.StreamFixture(name)
    oracle = sf.oracle_decode_stream([(fx.info[i], fx.cus(i), fx.levels(i))
                                      for i in range(fx.n)])

    def check(i, rec, dec):
        info = fx.info[i]
        got = rec.download(0)
        opic = oracle[i][0]
        for c in range(3):
            if not np.array_equal(got[c], opic.planes[c]):
                bad = np.argwhere(got[c] != opic.planes[c])
                raise AssertionError("%s pic %d (poc %d) comp %d: %d samples differ from the "
                                     "oracle, first at (y, x) = %s" %
                                     (name, i, info["poc"], c, len(bad), bad[0]))
        if fx.has_planes(i):
            for c, e in enumerate(fx.planes(i)):
                assert np.array_equal(got[c], e), (name, i, c)
        md5 = sf.picture_md5(got, int(info["bitdepth"]))
        assert np.array_equal(md5, info["md5"]), "%s pic %d MD5" % (name, i)
        if info["padded"]:      # the border a later picture's motion vectors reach into
            gp = rec.download(80)
            for c in range(3):
                b = 80 >> (1 if c else 0)
                full = opic.full[c]
                ob = opic.border >> (1 if c else 0)
                assert np.array_equal(gp[c], full[ob - b:full.shape[0] - ob + b,
                                                  ob - b:full.shape[1] - ob + b]), (name, i, c)
        if int(info["pic_type"]) == 2:      # an intra picture: its waves took one launch
            assert (dec.launches < 10) == one_launch, (dec.waves, dec.launches)

    decode_stream_on_gpu(api, ctx, fx, check, one_launch)


def test_gpu_stream_decode_rate(gpu):
    """Not a parity test: prints pictures/s of the 1080p stream (device + host
    planning, syntax already parsed) for the record."""
    api, ctx = gpu
    fx = sf.StreamFixture("c1")
    t0 = time.time()
    info = []

    def check(i, rec, dec):
        info.append((dec.waves, dec.launches))

    decode_stream_on_gpu(api, ctx, fx, check)
    dt = time.time() - t0
    print("\n1080p stream: %d pictures in %.3f s (%.1f pictures/s incl. python marshalling); "
          "waves / launches per picture: %s" % (fx.n, dt, fx.n / dt, info))


@pytest.mark.parametrize("name,lanes", [("tiny", 1), ("c1x", 1), ("tiny", 3), ("c0", 2), ("c1x", 3),
                                        ("c1x", 4)])
def test_decode_sequence_equals_stream_md5(gpu, name, lanes):
    """The whole stream through ONE call (PictureDecoder::DecodeSequence: the plan of
    picture i + 1 made on a worker thread while picture i is uploaded and launched; in
    inter pictures the trailing waves of intra CUs in one cooperative launch): every
    picture's MD5 equals the stream's.  lanes > 1: the pictures dealt over that many
    contexts (PictureDecoder::AddLane) - a picture's kernels wait for the pictures it
    references by events, the B pictures of a temporal layer run side by side."""
    api, ctx = gpu
    fx = sf.StreamFixture(name)
    w, h, bd = (int(fx.info[0][k]) for k in ("width", "height", "bitdepth"))
    dec = decoder.PictureDecoder(ctx, w, h, bd)
    lane_ctxs = [api.Context(0) for _ in range(lanes - 1)]
    for c in lane_ctxs:
        dec.add_lane(c)
    if lanes > 1:
        with pytest.raises(api.XvcGpuError):
            dec.add_lane(ctx)                 # the decoder's own context is lane 0
    infos = [fx.info[i] for i in range(fx.n)]
    pos = {int(infos[i]["poc"]): i for i in range(fx.n)}
    ref_index = np.full((fx.n, 2, 5), -1, np.int32)
    pictures = []
    for i, info in enumerate(infos):
        ps, cs = sf.to_syntax(info, fx.cus(i))
        pictures.append((ps, cs, np.ascontiguousarray(fx.levels(i))))
        for l in range(2):
            for k in range(int(info["num_ref"][l])):
                ref_index[i, l, k] = pos[int(info["ref_poc"][l][k])]
    recs = [ctx.picture(w, h, bd) for _ in range(fx.n)]
    for _ in range(2 if lanes == 1 else 4):       # again: the later runs re-use every buffer
        dec.decode_sequence(pictures, ref_index, recs)
        ctx.sync()
        for i, info in enumerate(infos):
            got = recs[i].download(0)
            assert np.array_equal(sf.picture_md5(got, bd), info["md5"]), (name, lanes, i)
    # a reference that is not decoded yet is refused
    bad = ref_index.copy()
    bad[1, 0, 0] = fx.n - 1
    with pytest.raises(api.XvcGpuError):
        dec.decode_sequence(pictures, bad, recs)
    ctx.sync()
    dec.destroy()
    for c in lane_ctxs:
        c.close()
    for p in recs:
        p.destroy()

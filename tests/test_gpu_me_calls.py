"""xvcgpu_me_search on the motion searches of a real encoder run
(tests/golden/me_calls_*.npz: every MotionEstNormal call the reference encoder
made while coding the stream fixtures, with its real AMVP predictor,
previous-CU vector, block shape 4x4..64x64, fullpel-MV flag and search range).
The reference pictures are the stream's reconstructions, decoded on the device
first; the originals are the synthetic frames at the internal bit depth."""
import numpy as np
import pytest

import stream_fixture as sf
import test_me_calls as tmc
from xvc_amd import decoder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def decode_stream(ctx, fx):
    """-> {poc: device picture} of the whole stream (GPU PictureDecoder)."""
    w, h = int(fx.info[0]["width"]), int(fx.info[0]["height"])
    dec = decoder.PictureDecoder(ctx, w, h, 10)
    pics = {}
    for i in range(fx.n):
        ps, cus = sf.to_syntax(fx.info[i], fx.cus(i))
        rec = ctx.picture(w, h, 10)
        info = fx.info[i]
        refs = [[pics[int(info["ref_poc"][l][k])] for k in range(int(info["num_ref"][l]))]
                for l in range(2)]
        dec.decode(ps, cus, fx.levels(i), refs, rec)
        pics[int(fx.info[i]["poc"])] = rec
    ctx.sync()
    return pics, w, h


def me_blocks(api, calls):
    b = np.zeros(len(calls), api.ME_DTYPE)
    for k in ("x", "y", "w", "h", "depth_nonzero", "fullpel_mv", "mvp_x", "mvp_y", "prev_x",
              "prev_y", "lambda16", "search_range"):
        b[k] = calls[k]
    b["fullpel_mv"] |= np.where(calls["use_lic"] != 0, 2, 0).astype(np.uint8)   # XVC_ME_USE_LIC
    return b


@pytest.mark.parametrize("name", ["tiny", "c0", "c1", "c0q22", "c0q37"])
def test_me_search_reproduces_encoder_motion_searches(gpu, name):
    api, ctx = gpu
    fx = sf.StreamFixture(name)
    pics, w, h = decode_stream(ctx, fx)
    calls = tmc.load_calls(name)
    done = 0
    for poc in sorted(set(calls["poc"].tolist())):
        O = ctx.picture(w, h, 10)
        O.upload([tmc.original_luma(w, h, poc), None, None], tmc.BL)
        for ref_poc in sorted(set(calls["ref_poc"][calls["poc"] == poc].tolist())):
            sel = calls[(calls["poc"] == poc) & (calls["ref_poc"] == ref_poc)]
            res = ctx.me_search(O, pics[ref_poc], me_blocks(api, sel),
                                flags=api.ME_FULLPEL | api.ME_SUBPEL | api.ME_LIC_JOBS)
            if (sel["use_lic"] != 0).any():     # not announced: reported, not computed
                r0 = ctx.me_search(O, pics[ref_poc], me_blocks(api, sel[:64]))
                lic0 = sel["use_lic"][:64] != 0
                assert (r0["subpel_dist"][lic0] == 0xffffffff).all()
                assert (r0["subpel_dist"][~lic0] != 0xffffffff).all()
            for k_res, k_call in (("fullpel_x", "fullpel_x"), ("fullpel_y", "fullpel_y"),
                                  ("mv_x", "mv_x"), ("mv_y", "mv_y"), ("subpel_dist", "dist")):
                bad = np.nonzero(res[k_res].astype(np.int64) != sel[k_call].astype(np.int64))[0]
                assert len(bad) == 0, (name, poc, ref_poc, k_res, len(bad), tuple(sel[bad[0]]),
                                       tuple(res[bad[0]]))
            done += len(sel)
        O.destroy()
    assert done == len(calls) and done > 15000
    for p in pics.values():
        p.destroy()

"""The rest of a real encoder run's RD search on the device
(tests/golden/rd_calls_*.npz, tools/gen_rd_golden.py): every bi-prediction
refinement step, affine motion search, merge-candidate ranking and inter-CU
TransformAndReconstruct the reference encoder made while it coded the stream
fixtures - with the CABAC context states its quantiser really read and, for CUs
with local illumination compensation, the neighbouring reconstruction of that
moment - replayed as batches through the C-ABI (tests/rd_replay.py) and
compared call by call: vectors, distortions, sorted merge costs, RdoQuant's
levels (CRC) and counts, reconstruction blocks (CRC), returned distortions."""
import numpy as np
import pytest

import rd_fixture as rf
import rd_replay
import stream_fixture as sf
from test_gpu_me_calls import decode_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


@pytest.fixture(scope="module", params=["tiny", "c0", "c1", "c0q22", "c0q37"])
def replay(gpu, request):
    api, ctx = gpu
    name = request.param
    if not __import__("os").path.exists(rf.path(name)):
        pytest.skip("no rd_calls fixture for " + name)
    fx = sf.StreamFixture(name)
    pics, w, h = decode_stream(ctx, fx)
    r = rd_replay.Replay(api, ctx, name, pics, w, h)
    yield r
    r.destroy()
    for p in pics.values():
        p.destroy()


def test_bipred_refinement_steps(replay):
    done, bad, skipped = replay.bi_steps()
    assert done > 1000 and bad == 0, (done, bad, getattr(replay, "first_bad", None))


def test_affine_motion_searches(replay):
    done, bad = replay.affine_steps()
    assert bad == 0, (done, bad, getattr(replay, "first_bad", None))


def test_merge_candidate_rankings(replay):
    done, bad = replay.merges()
    assert done > 100 and bad == 0, (done, bad, getattr(replay, "first_bad", None))


def test_transform_calls_with_scratch_destinations(replay):
    replay.dz_done = replay.dz_bad = 0
    done, bad = replay.transform_calls_scratch()
    assert done > 1000 and bad == 0, (done, bad, getattr(replay, "first_bad", None))
    assert replay.dz_done > 100 and replay.dz_bad == 0, (replay.dz_done, replay.dz_bad)


def test_transform_and_reconstruct_calls(replay):
    done, bad, layers, dz_done, dz_bad = replay.transform_calls()
    assert done > 10000 and bad == 0, (done, bad, layers, getattr(replay, "first_bad", None))
    assert dz_done > 1000 and dz_bad == 0, (dz_done, dz_bad)


def test_all_zero_proof_on_captured_calls(replay):
    """xvcgpu_quant_rdo_batch with the all-zero proof forced on / off over every
    captured inter TransformAndReconstruct call (live context snapshots): the same
    levels as the fused path - which test_transform_calls_with_scratch_destinations
    holds against the reference - and the proof does take blocks off the walk."""
    replay.check_prove_zero = True
    replay.pz_done = replay.pz_bad = replay.pz_zero = replay.pz_walked = replay.pz_proved = 0
    try:
        done, bad = replay.transform_calls_scratch()
    finally:
        replay.check_prove_zero = False
    assert bad == 0 and replay.pz_bad == 0, (done, bad, replay.pz_bad,
                                             getattr(replay, "first_bad", None))
    assert replay.pz_done > 1000 and replay.pz_proved > 0, (replay.pz_done, replay.pz_proved)
    print("all-zero proof: %d calls, %d end all zero, %d walked without the proof, %d proved"
          % (replay.pz_done, replay.pz_zero, replay.pz_walked, replay.pz_proved))

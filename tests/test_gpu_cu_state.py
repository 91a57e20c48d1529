"""A real picture's RD search walked one CU STATE at a time, in the order the reference
encoder issued it (tests/rd_serial.py, xvc_amd/host/xvc_cu_state.cc): every step a batch
of one with a read-back wherever the reference reads a result - the regime a bit-exact
encoder can actually present.  Every result equals what the reference encoder got.  All
states of the picture are walked: merge rankings, merge-candidate evaluations, CompressInter
with and without local illumination compensation, CompressIntra."""
import numpy as np
import pytest

import rd_serial
import stream_fixture as sf
from test_gpu_me_calls import decode_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def _run(api, ctx, name, poc, n_states):
    fx = sf.StreamFixture(name)
    pics, w, h = decode_stream(ctx, fx)
    sp = rd_serial.SerialPicture(api, name, poc)
    run = rd_serial.SerialRun(api, ctx, sp, pics, w, h)
    n = min(n_states, len(sp.states))
    stats = run.run_serial(0, n)
    res = run.check(0, n)
    run.destroy()
    for p in pics.values():
        p.destroy()
    return sp, stats, res


@pytest.mark.parametrize("name,poc,n_states", [("tiny", 2, 1 << 30), ("c1", 2, 6000)])
def test_serial_walk_equals_reference(gpu, name, poc, n_states):
    api, ctx = gpu
    sp, stats, res = _run(api, ctx, name, poc, n_states)
    print(name, sp.summary(), res, "%.1f us / state, %.1f API calls, %.2f round trips per state" % (
        1e6 * stats.seconds / max(stats.states, 1), stats.api_calls / max(stats.states, 1),
        stats.round_trips / max(stats.states, 1)))
    assert stats.states > 1000
    # every state of the picture is walked: the LIC states too (tiny: 3648 of 8151; the
    # neighbouring reconstruction of that moment staged in front of each)
    assert sp.summary()["unsupported"] == 0 and stats.skipped == 0 and stats.states == min(n_states, len(sp.states))
    lic = (sp.states["flags"] & rd_serial.STATE_LIC) != 0
    assert lic.sum() == (3648 if name == "tiny" else 0)
    for k in ("me", "bi", "merge", "calls", "dist_zero", "intra_satd", "intra_calls"):
        assert res[k][0] > 100 and res[k][1] == 0, (k, res)
    # the intra states (CompressIntra: the SATD pre-selection, every PredictAndTransform
    # alternative of the luma and chroma modes, LM chroma included)
    assert sp.summary()["intra"] > (200 if name == "tiny" else 20), sp.summary()
    assert (sp.in_want["mode"] == 67).sum() > 50
    assert res["affine"][1] == 0, res
    if name == "c1":
        assert res["affine"][0] > 100


def _run_chained(api, ctx, name, poc, n_states, by_position, refs_form=True, live=False):
    fx = sf.StreamFixture(name)
    pics, w, h = decode_stream(ctx, fx)
    sp = rd_serial.SerialPicture(api, name, poc)
    run = rd_serial.ChainedRun(api, ctx, sp, pics, w, h, rd_serial.ref_lists_of(name, poc))
    run.refs_form = refs_form
    run.no_copies = refs_form      # (the older form keeps the block copies too)
    n = min(n_states, len(sp.states))
    stats = run.run_chained(0, n, by_position, live=live)
    res = run.check(0, n, searches=False)
    res.update(run.check_chained(0, n))
    bad = repr(getattr(run, "first_bad", None))     # (views of pinned memory: before destroy)
    run.destroy()
    for p in pics.values():
        p.destroy()
    return sp, stats, res, bad


@pytest.mark.parametrize("name,poc,n_states,by_position,refs_form", [
    ("tiny", 2, 1 << 30, False, True), ("tiny", 2, 1 << 30, True, True),
    ("tiny", 2, 1 << 30, True, False), ("c0", 4, 1 << 30, True, True),
    ("c1", 2, 1 << 30, True, True), ("c1", 2, 4000, True, False)])      # (c1: the whole 1080p picture)
def test_chained_states_equal_reference(gpu, name, poc, n_states, by_position, refs_form):
    """The same states as ONE enqueue each (or per visit of a CU position), the folds
    between the searches on the device (xvcgpu_cs_*_fold): no read-back inside a chain.
    The searches' jobs are composed by the folds - the predictor EvalStartMvp picks, the
    refinement's jobs, the affine bootstrap, the evaluation's motion - so every result
    below depends on them: all equal to the reference's, and so are the folds' own
    intermediates (final predictor, bits with the default prices, costs, choices).
    refs_form: every step of SearchMotion into all the CU's reference pictures as one
    launch (xvcgpu_*_refs, the jobs side by side) instead of one launch per picture."""
    api, ctx = gpu
    sp, stats, res, bad = _run_chained(api, ctx, name, poc, n_states, by_position, refs_form)
    print(name, by_position, res, "%.1f us / state, %.1f API calls, %.2f round trips per state" % (
        1e6 * stats.seconds / max(stats.states, 1), stats.api_calls / max(stats.states, 1),
        stats.round_trips / max(stats.states, 1)))
    for k, (done, wrong) in res.items():
        assert wrong == 0, (k, res, bad)
    # (an intra state waits twice: behind the SATD pre-selection - the host sorts with the
    # mode bits - and at its end)
    assert stats.round_trips <= stats.states + sp.summary()["intra"]
    assert sp.summary()["unsupported"] == 0 and stats.states == min(n_states, len(sp.states))
    for k in ("cands", "finals", "eval_motion", "calls", "merge", "merge_fold", "merge_slot_motion",
              "intra_satd", "intra_calls"):
        assert res[k][0] > 100, (k, res)
    if name == "tiny":       # the LIC states' SearchMotion runs through the folds too (XVC_CS_LIC)
        assert res["cands"][0] == 24412 and res["finals"][0] == 5253 and res["me"][0] == 0, res
    # every merge ranking went through the device's fold (xvcgpu_cs_merge_fold), and nearly
    # every merge candidate's evaluation predicted from the slot the fold filled (the rest:
    # affine merges and candidates the harness could not tell apart)
    assert res["merge_fold"][0] == res["merge"][0]
    merge_evals = int(((sp.states["kind"] == rd_serial.KIND_EVAL) & (sp.states["supported"] != 0))[
        :min(n_states, len(sp.states))].sum())
    assert res["merge_slot_motion"][0] > 0.8 * merge_evals, (res, merge_evals)


@pytest.mark.parametrize("name,poc,n_states", [("tiny", 2, 1 << 30), ("c1", 2, 6000)])
def test_live_chains_equal_reference(gpu, name, poc, n_states):
    """The chains a live encoder could issue (rd_serial.program(live=True)): a wait after
    every CompressInter and in front of an evaluation's gated second transform pass, a merge
    ranking and its candidates' evaluations as one chain with the ranking folded on the
    device (xvcgpu_cs_merge_fold fills the evaluation slots).  Same results as the
    reference's, between one and two waits per state."""
    api, ctx = gpu
    sp, stats, res, bad = _run_chained(api, ctx, name, poc, n_states, False, True, live=True)
    print(name, "live", res, "%.1f us / state, %.2f round trips per state" % (
        1e6 * stats.seconds / max(stats.states, 1), stats.round_trips / max(stats.states, 1)))
    for k, (done, wrong) in res.items():
        assert wrong == 0, (k, res, bad)
    for k in ("cands", "finals", "eval_motion", "calls", "merge_fold", "merge_slot_motion", "intra_satd",
              "intra_calls"):
        assert res[k][0] > 100, (k, res)
    assert 0.5 * stats.states < stats.round_trips < 2 * stats.states


@pytest.mark.parametrize("name,poc,k,n,threads,live", [("tiny", 2, 5, 1500, 1, False),
                                                       ("c1", 2, 8, 1500, 1, False),
                                                       ("c1", 2, 9, 1200, 3, False),
                                                       ("c1", 2, 8, 1000, 2, True),
                                                       # the k an encoder can reach (the sub-GOP's
                                                       # top layers, two sub-GOPs in flight)
                                                       ("c1", 2, 16, 600, 4, False)])
def test_engine_chains_equal_reference(gpu, name, poc, k, n, threads, live):
    """k chains on ONE context through the engine (xvc_host_cs_run_programs_engine): every
    round the chains' next steps grouped by kind, one launch per kind with the chains' jobs
    side by side in the grid (xvcgpu_cs_segs_launch) - each chain walks its own stretch of
    the picture, so their steps do not line up.  threads = 1: a round's groups dealt over
    three streams; threads = 3: three engines on three host threads, a stream and every
    third chain each.  live: the chains a live encoder could issue (a wait wherever the host's
    entropy coder decides).  Every chain's results equal the reference's, and the launches
    are fewer than the steps."""
    api, ctx = gpu
    fx = sf.StreamFixture(name)
    pics, w, h = decode_stream(ctx, fx)
    sp = rd_serial.SerialPicture(api, name, poc)
    ectx = api.Context(0)
    ectx.use_own_stream()
    lists = rd_serial.ref_lists_of(name, poc)
    runs = [rd_serial.ChainedRun(api, ectx, sp, pics, w, h, lists) for _ in range(k)]
    n = min(n, len(sp.states) // k)
    firsts = [sp.position_start(c * (len(sp.states) - n - 64) // max(k - 1, 1)) for c in range(k)]
    more = [api.Context(0) for _ in range(max(2, threads - 1))]   # further streams for the rounds' groups / engines
    for c in more:
        c.use_own_stream()
    stats = rd_serial.ChainedRun.run_engine(runs, firsts, n, by_position=True, verify=True, streams=more,
                                            threads=threads, live=live)
    steps = sum(int((r.program(f, n, True, True, live=live)["opcode"] != rd_serial.OP_SYNC).sum())
                for r, f in zip(runs, firsts))
    print(name, "engine k=%d: %d states, %d launches for %d steps, %.1f us per state" % (
        k, stats.states, stats.api_calls, steps, 1e6 * stats.seconds / max(stats.states, 1)))
    # (the LIC and intra states' steps have no batched form: tiny's chains are mostly those)
    assert stats.states > 0.3 * k * n and stats.api_calls < (1.0 if name == "tiny" else 0.8) * steps
    for r, f in zip(runs, firsts):
        res = r.check(f, n, searches=False)
        res.update(r.check_chained(f, n))
        bad = repr(getattr(r, "first_bad", None))
        for key, (done, wrong) in res.items():
            assert wrong == 0, (key, res, bad)
    for r in runs:
        r.destroy()
    for c in more:
        c.close()
    ectx.close()
    for p in pics.values():
        p.destroy()


def test_interleaved_chains_equal_reference(gpu):
    """Three replays of the picture on their own contexts driven by one host thread
    (xvc_host_cs_run_programs_interleaved: a chain issued while the others execute):
    each run's results equal the reference's."""
    api, ctx = gpu
    fx = sf.StreamFixture("tiny")
    pics, w, h = decode_stream(ctx, fx)
    sp = rd_serial.SerialPicture(api, "tiny", 2)
    ctxs = [api.Context(0) for _ in range(3)]
    for c in ctxs:
        c.use_own_stream()
    lists = rd_serial.ref_lists_of("tiny", 2)
    runs = [rd_serial.ChainedRun(api, c, sp, pics, w, h, lists) for c in ctxs]
    stats = rd_serial.ChainedRun.run_interleaved(runs)
    assert stats.states > 1000 and stats.states % 3 == 0 and stats.round_trips < stats.states
    for r in runs:
        res = r.check(0, len(sp.states), searches=False)
        for k, (done, wrong) in res.items():
            assert wrong == 0, (k, res)
        assert res["calls"][0] > 100
    for r in runs:
        r.destroy()
    for c in ctxs:
        c.close()
    for p in pics.values():
        p.destroy()

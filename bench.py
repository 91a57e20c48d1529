#!/usr/bin/env python3
"""bench.py -- hot-path frame passes per second on MI355X.

One "step" = one hot-path frame pass (xvc_amd/pipeline.py) over one 1920x1080
4:2:0 picture of a synthetic clip at QP 32, internal bit depth 10: TZ + sub-pel
motion search, motion compensation, transform/quant/dequant/inverse/recon,
deblocking, border extension and PSNR-Y parts - every kernel of SURVEY.md
section 8a's M/T/I/X/Q/D/P rows, chained picture to picture (the deblocked,
padded reconstruction of step i is the reference of step i+1).  Pictures,
descriptors and decisions are resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: every picture is sharded by rows of CUs across the ranks (CTU-row
shards); the in-loop filter exchanges a 4-row halo with
the neighbouring shards and the reconstructed rows are all-gathered over
RCCL/xGMI so every rank holds the next reference picture.  Same pictures as
N = 1 => "scaling": "strong".

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is ~0.16 ms: the defaults run long enough (~0.2 s) for the GPU's
    # clocks to settle and for every frame of the cycle to be visited many times
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames")
    ap.add_argument("--settle", type=int, default=-1,
                    help="untimed frame passes before the warmup so that the picture chains "
                         "are in their steady state when the clock starts (a chain codes every "
                         "picture against the previous RECONSTRUCTION: quantisation noise builds "
                         "up over the first few hundred pictures - 2092 coded luma blocks in a "
                         "chain's first picture, 8146 after 600 - and with it the quantiser's "
                         "load; without this a 20-step and a 1000-step run time different "
                         "workloads).  -1 = automatic: 1500 per chain on one GPU, 0 otherwise")
    ap.add_argument("--cpu-frames", type=int, default=12,
                    help="cap on the single-thread CPU oracle frame passes (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cu-order", choices=["tiles", "raster"], default="tiles",
                    help="order of the CU list of the (unsharded) frame pass: region-major over "
                         "a 4 x 2 tiling of the picture, so that each XCD's eighth of every "
                         "job list is one compact region (pipeline.cu_partition), or raster")
    ap.add_argument("--tail-priority", choices=["auto", "on", "off"], default="auto",
                    help="the short kernels that end a pass (inverse transform, fused tail) "
                         "on a high-priority stream of their chain "
                         "(xvcgpu_set_short_kernel_priority); auto = off: measured in round 4 it "
                         "LOSES - 7124 -> 3726 passes/s at 1080p, 1542 -> 1305 at 2160p, 583 -> "
                         "522 at 4320p (three chains): the two event hand-overs per pass cost "
                         "more than the earlier start of the short kernels gains")
    ap.add_argument("--quant", choices=["rdoq", "fast"], default="rdoq",
                    help="quantiser of the transform stage: rdoq = RdoQuant::QuantRdo with "
                         "CoeffSignHideRdo, what the reference's encoder always runs "
                         "(encoder_settings.h:59); fast = its non-RDO QuantFast")
    ap.add_argument("--no-decode", action="store_true",
                    help="skip the stream-decode figure (reconstruction of the committed "
                         "1080p reference stream, tests/golden/stream_c1.npz)")
    ap.add_argument("--two-queue", action="store_true",
                    help="N=1: issue the halves of every picture on a high- and a "
                         "low-priority stream (pipeline.PipelinedFramePass; measured "
                         "7 %% slower than one queue: 17 launches + events per step)")
    ap.add_argument("--graph", action="store_true",
                    help="N=1: replay one recorded HIP graph per step instead of "
                         "launching the kernels separately (the same rate on the settled "
                         "state: the steps are GPU-bound, not launch-bound)")
    ap.add_argument("--schedule", choices=["auto", "chains", "subgop", "rows"], default="auto",
                    help="chains: independent picture chains, each picture referencing the one "
                         "before (N=1 default); subgop: hierarchical sub-GOP 16 coded with the "
                         "reference's ThreadEncoder policy on ranks x picture slots, references "
                         "shipped by RCCL (N>1 default up to 1080p); rows: CTU-row shards of "
                         "every picture with RCCL halo exchange (N>1 default above 1080p)")
    ap.add_argument("--slots", type=int, default=3, help="subgop: pictures in flight per GPU")
    ap.add_argument("--chains", type=int, default=0,
                    help="(0 = automatic: 3, or 1 when one rank's share of a picture "
                         "exceeds 3840x2160, where a picture fills the chip by itself) "
                         "independent picture chains in flight, issued round-robin on "
                         "their own HIP streams (like the reference's picture-level "
                         "threads): a picture of one chain fills the drain of another's "
                         "kernels (1080p: 6870 / 8030 / 8740 frame passes/s with 1 / 2 / 3 "
                         "chains, tools/two_chains.py); with N > 1 also hides the RCCL "
                         "exchanges of one chain behind the kernels of the others")
    ap.add_argument("--force-sharded", action="store_true",
                    help="testing aid: take the multi-GPU code path (row-sharded engine, "
                         "process groups, RCCL exchanges) even with one rank; launch "
                         "through torch.distributed.run --nproc-per-node 1")
    ap.add_argument("--kernel-times", action="store_true", default=True)
    return ap.parse_args()


def kernel_source_md5():
    """MD5 over the HIP sources (xvc_amd/csrc): ties a PMC traffic profile to
    the kernels it was measured on."""
    import hashlib
    m = hashlib.md5()
    d = os.path.join(ROOT, "xvc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        m.update(open(os.path.join(d, f), "rb").read())
    return m.hexdigest()


def stream_decode_figure(ctx, api):
    """Reconstruction of the committed 1080p reference stream (the first five
    pictures of BASELINE config 1 as the reference encoder coded them: CU trees,
    modes, vectors, levels parsed by the reference decoder) through the C++ host
    decoder + the C-ABI, and - where the reference build travels with the repo -
    the reference decoder on the same stream on one host core.  Every picture's
    MD5 is checked against the stream's."""
    import stream_fixture as sf
    from xvc_amd import decoder
    # the decode figure: BASELINE config 1 as SURVEY 8d specifies it (33 pictures,
    # two default sub-GOPs of 16 + 1); the search replays below use the short
    # stream their captures were made on
    dfx = sf.StreamFixture("c1x")
    fx = sf.StreamFixture("c1")
    w, h, bd = (int(fx.info[0][k]) for k in ("width", "height", "bitdepth"))
    dec = decoder.PictureDecoder(ctx, w, h, bd)

    def decoder_run(f, syn, pics, per_picture=None):
        done = {}
        if not hasattr(f, "_levels"):   # out of the .npz once (every access inflates it again)
            f._levels = [np.ascontiguousarray(f.levels(i)) for i in range(f.n)]
            f._infos = [f.info[i] for i in range(f.n)]
            pos = {int(f._infos[i]["poc"]): i for i in range(f.n)}
            f._ref_index = np.full((f.n, 2, 5), -1, np.int32)
            for i in range(f.n):
                info = f._infos[i]
                for l in range(2):
                    for k in range(int(info["num_ref"][l])):
                        f._ref_index[i, l, k] = pos[int(info["ref_poc"][l][k])]
        if per_picture is None:
            # the whole sequence in one call: the C++ layer plans picture i + 1 on a
            # worker thread while picture i is uploaded and launched
            dec.decode_sequence([(syn[i][0], syn[i][1], f._levels[i]) for i in range(f.n)],
                                f._ref_index, pics)
            ctx.sync()
            return
        for i in range(f.n):
            info = f._infos[i]
            refs = [[done[int(info["ref_poc"][l][k])] for k in range(int(info["num_ref"][l]))]
                    for l in range(2)]
            t0 = time.perf_counter()
            dec.decode(syn[i][0], syn[i][1], f._levels[i], refs, pics[i])
            ctx.sync()
            per_picture[i] += time.perf_counter() - t0
            done[int(info["poc"])] = pics[i]
        ctx.sync()

    dsyn = [sf.to_syntax(dfx.info[i], dfx.cus(i)) for i in range(dfx.n)]
    dpics = [ctx.picture(w, h, bd) for _ in range(dfx.n)]
    decoder_run(dfx, dsyn, dpics)
    ok = all(np.array_equal(sf.picture_md5(dpics[i].download(0), bd), dfx.info[i]["md5"])
             for i in range(dfx.n))
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        decoder_run(dfx, dsyn, dpics)
    dt = (time.perf_counter() - t0) / reps
    # by picture type (a sync after every picture: the sum exceeds the pipelined run)
    per = [0.0] * dfx.n
    for _ in range(reps):
        decoder_run(dfx, dsyn, dpics, per)
    kinds = {}
    for i in range(dfx.n):
        k = {0: "B", 1: "P", 2: "I"}[int(dsyn[i][0]["pic_type"][0])]
        kinds.setdefault(k, []).append(1e3 * per[i] / reps)
    out = {"stream": "1920x1080 QP 32, %d pictures (1 intra + two hierarchical-B sub-GOPs of 16) "
                     "coded by the reference encoder with xvcenc's defaults "
                     "(tests/golden/stream_c1x.npz)" % dfx.n,
           "pictures_per_s": dfx.n / dt, "ms_per_picture": 1e3 * dt / dfx.n,
           "ms_by_picture_type": {k: {"pictures": len(v), "mean_ms": sum(v) / len(v),
                                      "pictures_per_s": 1e3 * len(v) / sum(v)}
                                  for k, v in kinds.items()},
           "md5_match": bool(ok), "includes": "host planning + upload of the parsed syntax; the "
                                              "planning of picture i + 1 on a worker thread while "
                                              "picture i is issued (PictureDecoder::DecodeSequence)"}
    # an inter picture inside the pipelined run: what the sequence takes beyond its
    # intra picture (ms_by_picture_type waits after every picture)
    if "I" in kinds and any(k != "I" for k in kinds):
        n_inter = sum(len(v) for k, v in kinds.items() if k != "I")
        out["ms_per_inter_picture_pipelined"] = max(1e3 * dt - sum(kinds["I"]), 0.0) / n_inter
    for p in dpics:
        p.destroy()
    # the short stream, decoded for the search replays
    syn = [sf.to_syntax(fx.info[i], fx.cus(i)) for i in range(fx.n)]
    pics = [ctx.picture(w, h, bd) for _ in range(fx.n)]
    decoder_run(fx, syn, pics)
    out["encoder_me_batches"] = encoder_me_figure(ctx, api, fx, pics, w, h)
    out["encoder_rd_batches"] = encoder_rd_figure(ctx, api, fx, pics, w, h)
    out["encoder_rd_serial"] = encoder_rd_serial_figure(ctx, api, fx, pics, w, h)
    if out["encoder_rd_serial"]:
        out["encoder_rd_serial"]["side_by_side"] = rd_side_by_side(out)
    dec.destroy()
    for p in pics:
        p.destroy()
    import oracle_lib as ol
    if ol.have_ref():
        import ctypes as C
        lib = C.CDLL(ol.REF_SO)
        lib.xr_stream_decode.argtypes = [C.c_void_p, C.c_long, C.c_int]
        buf = np.ascontiguousarray(dfx.stream, np.uint8)
        t0 = time.perf_counter()
        n = 0
        while n < 2 or time.perf_counter() - t0 < 2.0:
            assert lib.xr_stream_decode(buf.ctypes.data, len(buf), 0) == dfx.n
            n += 1
        out["cpu_reference_pictures_per_s"] = n * dfx.n / (time.perf_counter() - t0)
        out["cpu_reference"] = "the reference decoder (parse + reconstruct), 1 thread"
        lib.xr_stream_release()
    return out


def encoder_me_figure(ctx, api, fx, pics, w, h):
    """Second workload: the motion searches the reference encoder actually made
    for one 1080p picture of that stream (POC 2, both reference pictures: every
    CU shape its RD search tried, 4x4 to 64x64, with the real AMVP predictors,
    previous-CU vectors, fullpel-MV CUs; tests/golden/me_calls_c1.npz, captured
    by tools/gen_me_golden.py) as two xvcgpu_me_search batches against the
    decoded reference pictures.  Results are checked against the reference's."""
    import test_me_calls as tmc
    calls = tmc.load_calls("c1")
    calls = calls[calls["use_lic"] == 0]
    poc = int(calls["poc"][0])
    by_poc = {int(fx.info[i]["poc"]): pics[i] for i in range(fx.n)}
    O = ctx.picture(w, h, 10)
    O.upload([tmc.original_luma(w, h, poc), None, None], tmc.BL)
    batches, ok = [], True
    for ref_poc in sorted(set(calls["ref_poc"].tolist())):
        sel = calls[calls["ref_poc"] == ref_poc]
        b = np.zeros(len(sel), api.ME_DTYPE)
        for k in ("x", "y", "w", "h", "depth_nonzero", "fullpel_mv", "mvp_x", "mvp_y", "prev_x",
                  "prev_y", "lambda16", "search_range"):
            b[k] = sel[k]
        db, dr = ctx.buffer(b), ctx.alloc(api.MERES_DTYPE.itemsize * len(sel))
        batches.append((by_poc[ref_poc], db, dr, sel))
    flags = api.ME_FULLPEL | api.ME_SUBPEL

    def run():
        for ref, db, dr, sel in batches:
            ctx.me_search_dev(O, ref, flags, db.ptr, len(sel), dr.ptr, 64)
    run()
    ctx.sync()
    for ref, db, dr, sel in batches:
        res = dr.to_array(api.MERES_DTYPE, len(sel))
        ok &= bool(np.array_equal(res["mv_x"], sel["mv_x"]) and
                   np.array_equal(res["mv_y"], sel["mv_y"]) and
                   np.array_equal(res["subpel_dist"], sel["dist"]))
    reps = 10
    ctx.timer_begin()
    for _ in range(reps):
        run()
    ms = ctx.timer_end() / reps
    n = int(len(calls))
    samples = int((calls["w"].astype(np.int64) * calls["h"]).sum())
    out = {"workload": "all %d uni-directional motion searches of the reference encoder's RD "
                       "search for one 1080p B picture (POC %d vs POC %s), block shapes 4x4..64x64"
                       % (n, poc, "/".join(str(int(r)) for r in sorted(set(calls["ref_poc"].tolist())))),
           "ms": ms, "searches_per_s": n / (ms * 1e-3), "block_samples": samples,
           "ns_per_block_sample": 1e6 * ms / samples, "matches_reference": ok}
    for _, db, dr, _ in batches:
        db.free()
        dr.free()
    O.destroy()
    import oracle_lib as ol
    if ol.have_ref():       # the reference's own classes on a bounded sample, one thread
        xr = ol.Lib("xr")
        rec = {p: np.ascontiguousarray(np.pad(pic.download(0)[0], tmc.BL, mode="edge"))
               for p, pic in by_poc.items() if p in set(calls["ref_poc"].tolist())}
        orig = tmc.original_luma(w, h, poc)
        t0, k = time.perf_counter(), 0
        for c in calls[::37]:
            st = tmc.me_struct(c)
            fp, _ = xr.tz_search(10, st, w, h, orig, rec[int(c["ref_poc"])], tmc.BL)
            if not c["fullpel_mv"]:
                xr.subpel_search(10, st, w, h, orig, rec[int(c["ref_poc"])], tmc.BL, fp)
            k += 1
            if time.perf_counter() - t0 > 3.0:
                break
        out["cpu_reference_searches_per_s"] = k / (time.perf_counter() - t0)
        out["cpu_reference"] = "TzSearch::Search + SubpelSearch of oracle/_ref, 1 thread, every 37th call"
    return out


def encoder_rd_serial_figure(ctx, api, fx, pics, w, h, n_states=4000):
    """Fourth workload: the SAME picture's RD search in the order the reference can
    actually issue it (tests/golden/rd_order_c1.npz, tests/rd_serial.py,
    xvc_amd/host/xvc_cu_state.cc).  CuEncoder::CompressCu evaluates one CU state at a
    time - merge ranking, each merge candidate, CompressInter (SearchMotion [+ affine]
    -> CompressAndEvalCbf) per mode - and every state starts from what the previous one
    decided; the whole-picture batches above are a shape no bit-exact encoder can
    present.  Here the first `n_states` states of the picture are walked in that order:
      serial   each step a batch of one CU through the entry points as they are, with a
               read-back wherever the reference reads a result ("latency /
               launch bounds the per-CU batches; report it as such");
      chained  all states of one visit of a CU position as ONE enqueue, the folds between
               the searches on the device (xvcgpu_cs_*_fold, default bit prices), one
               read-back at the end;
    on k contexts at once (k independent pictures in flight: the sub-GOP's top layer).
    Every search result, every priced candidate, every SearchMotion choice, every
    TransformAndReconstruct (count, levels, distortion) is compared with the reference
    encoder's."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cu_state_walk
    import rd_fixture as rf
    import rd_serial
    if not (os.path.exists(rf.path("c1")) and os.path.exists(os.path.join(rf.GOLDEN, "rd_order_c1.npz"))):
        return None
    by_poc = {int(fx.info[i]["poc"]): pics[i] for i in range(fx.n)}
    poc = 2
    spent = {}
    t_ = time.perf_counter()
    sp = rd_serial.SerialPicture(api, "c1", poc)
    spent["state_table"] = time.perf_counter() - t_

    def timed_walk(label, *a, **kw):
        t0_ = time.perf_counter()
        r = cu_state_walk.walk(*a, **kw)
        spent[label] = time.perf_counter() - t0_
        return r
    serial = timed_walk("serial", api, "c1", poc, n_states, [1, 4], "serial", decoded=(by_poc, w, h), sp=sp)
    chained = timed_walk("chained", api, "c1", poc, n_states, [1, 4], "chained", decoded=(by_poc, w, h), sp=sp)
    live = timed_walk("live", api, "c1", poc, n_states, [1, 4], "live", decoded=(by_poc, w, h), sp=sp)
    # many pictures in flight: k chains through the execution engine, four engines on four
    # host threads and streams (each chain its own stretch of the picture)
    # (k = 128 / 256: profiles/r05_cu_state_walk_engine.json - building 256 chains' device
    # arrays takes longer than the rest of this figure)
    engine = timed_walk("engine", api, "c1", poc, min(n_states, 1200), [16, 64], "engine",
                        decoded=(by_poc, w, h), sp=sp, engine_threads=4, reps=1)
    s1, c1, l1 = serial["chains"]["1"], chained["chains"]["1"], live["chains"]["1"]
    ok = all(e.get("matches_reference") for r in (serial, chained, live, engine)
             for e in r["chains"].values())
    # the clip whose encode allows local illumination compensation (136x72, POC 2): nearly
    # half of its states are LIC states - walked with the neighbouring reconstruction of that
    # moment staged in front of each (xvcgpu_bipred_search_lic, XVC_INTER_LIC predictions)
    lic_fig = None
    if os.path.exists(rf.path("tiny")) and os.path.exists(os.path.join(rf.GOLDEN, "rd_order_tiny.npz")):
        spl = rd_serial.SerialPicture(api, "tiny", 2)
        forms = {m: timed_walk("lic_" + m, api, "tiny", 2, 1 << 30, [1], m, sp=spl)
                 for m in ("serial", "chained", "live")}
        ok = ok and all(f["chains"]["1"].get("matches_reference") for f in forms.values())
        lic_fig = {
            "workload": "136x72 B picture POC 2 of the reference-coded LIC clip: all %d CU states" %
                        forms["serial"]["states_in_picture"],
            "lic_states": int(((spl.states["flags"] & rd_serial.STATE_LIC) != 0).sum()),
            "intra_states": forms["serial"]["summary"]["intra"],
            "unsupported_states": forms["serial"]["summary"]["unsupported"],
            "us_per_cu_state": {m: f["chains"]["1"]["us_per_cu_state"] for m, f in forms.items()},
            "round_trips_per_state": {m: f["chains"]["1"]["round_trips_per_state"] for m, f in forms.items()},
            "form": "a LIC state's SearchMotion runs through the device folds (XVC_CS_LIC): "
                    "EvalStartMvp on compensated predictions (XVC_INTER_LIC jobs + SAD), the searches "
                    "per picture with the AC-only metrics, xvcgpu_bipred_search_lic per refinement "
                    "slot, use_lic in the folds' syntax; its merge ranking is folded on the device; "
                    "the neighbouring reconstruction of that moment is staged from the capture",
            "matches_reference": all(f["chains"]["1"].get("matches_reference") for f in forms.values())}
    return {
        "workload": "1080p B picture POC %d of the reference-coded stream: %d CU states in the "
                    "reference's issue order (%d merge rankings, %d merge-candidate evaluations, "
                    "%d CompressInter with evaluation, %d without, %d CompressIntra with %d "
                    "TransformAndReconstruct calls); walked: the %d states from state %d on - the "
                    "stretch whose mix of state kinds is closest to the picture's (%d CompressIntra "
                    "with %d calls: the picture's first states hold most of its intra states)" % (
                        poc, serial["states_in_picture"], serial["summary"]["merge_rank"],
                        serial["summary"]["eval"], serial["summary"]["inter"],
                        serial["summary"]["motion_only"], serial["summary"]["intra"],
                        serial["summary"]["intra_calls"], serial["states_walked"],
                        serial["first_state_walked"], serial["stretch"]["intra"],
                        serial["stretch"]["intra_calls"]),
        "us_per_cu_state": s1["us_per_cu_state"],
        "launches_per_state": {"entry_point_calls": s1["api_calls_per_state"]},
        # not of this run: read from the committed profiles, labelled with their file
        "quoted": quoted_walk_profiles(),
        "round_trips_per_state": s1["round_trips_per_state"],
        "pictures_per_s": {k: v["pictures_per_s"] for k, v in serial["chains"].items()},
        "us_by_state_kind": s1["us_by_kind"],
        "intra_picture_dag": intra_picture_dag(sp, c1.get("us_by_kind") or s1["us_by_kind"]),
        "chained": {"us_per_cu_state": c1["us_per_cu_state"],
                    "entry_point_calls_per_state": c1["api_calls_per_state"],
                    "round_trips_per_state": c1["round_trips_per_state"],
                    "states_per_chain": c1["states_per_chain"],
                    "fewer_round_trips": s1["round_trips_per_state"] / c1["round_trips_per_state"],
                    "pictures_per_s": {k: v["pictures_per_s"] for k, v in chained["chains"].items()},
                    "pictures_per_s_one_host_thread": {
                        k: v["one_thread"]["pictures_per_s"] for k, v in chained["chains"].items()
                        if "one_thread" in v},
                    "form": "every step of SearchMotion into all the CU's reference pictures as one "
                            "launch (xvcgpu_*_refs), a state's transform blocks as one launch, "
                            "one read-back per result array and chain",
                    "compared": c1.get("compared")},
        "live": {"us_per_cu_state": l1["us_per_cu_state"],
                 "entry_point_calls_per_state": l1["api_calls_per_state"],
                 "round_trips_per_state": l1["round_trips_per_state"],
                 "pictures_per_s": {k: v["pictures_per_s"] for k, v in live["chains"].items()},
                 "pictures_per_s_one_host_thread": {
                     k: v["one_thread"]["pictures_per_s"] for k, v in live["chains"].items()
                     if "one_thread" in v},
                 "form": "the chains a live encoder could issue: a wait wherever the reference's "
                         "control reads a cost that needs the host's entropy coder - after every "
                         "CompressInter / merge position, and inside an evaluation in front of the "
                         "gated second transform pass; nothing is taken from the capture's knowledge "
                         "of how a state ended except where listed below",
                 "decided_on_the_device": [
                     "EvalStartMvp / EvalFinalMvpIdx, GetInterPredBits (default prices), SearchRefIdx's "
                     "folds, the list SearchBiIterative searches (xvcgpu_cs_start / uni_fold)",
                     "the refinement's costs, the three-way choice, affine against plain, HasZeroMvd "
                     "(xvcgpu_cs_bi_fold) -> the evaluation's prediction jobs",
                     "the merge ranking: cost in double, stable sort, the 1.25 x cut -> the ranked "
                     "candidates' evaluation slots (xvcgpu_cs_merge_fold)"],
                 "decided_on_the_host": [
                     "GetCuCostWithoutSplit of every evaluation (the RDO writer's bits: entropy coder, "
                     "out of scope) and with it: best_cu_cost handed to the next candidate, "
                     "skip_evaluated, the break on a cbf-free winner (cu_encoder.cc:598-628)",
                     "cost_full > best_cu_cost * 1.1 in front of the second transform pass "
                     "(inter_search.cc:347-361): a wait inside the state",
                     "the CU recursion itself (split decisions, mode order)"],
                 "still_from_the_capture": [
                     "whether a CompressInter went on to its evaluation (HasZeroMvd is on the device, "
                     "but the replay has no evaluation jobs for the states that returned early)",
                     "which ranked merge candidates were evaluated (the fold fills all slots below "
                     "the count; the replay runs the ones the reference ran)",
                     "an intra state's kept luma modes and chroma mode list (the SATD sort adds the "
                     "host's mode bits; the walk runs the PredictAndTransform calls the reference ran)",
                     "a LIC state's neighbouring reconstruction (lic_picture: a live encoder's own picture)"],
                 "compared": l1.get("compared")},
        "engine": {"pictures_per_s": {k: v["pictures_per_s"] for k, v in engine["chains"].items()},
                   "launches_per_state": {k: v["launches_per_state"] for k, v in engine["chains"].items()},
                   "round_trips_per_state": {k: v["round_trips_per_state"]
                                             for k, v in engine["chains"].items()},
                   "host_threads_and_streams": 4,
                   "states_per_chain_walked": engine["states_walked"],
                   "form": "k pictures' chains in flight (the chained form's programs, every chain on "
                           "its own stretch of the picture): a round takes the next step of every "
                           "chain and issues one launch per step kind, grid y = chain "
                           "(xvcgpu_cs_segs_launch; xvc_host_cs_run_programs_engine); a chain at its "
                           "read-back sits out until the round's event has passed"},
        "lic_picture": lic_fig,
        "seconds_spent_measuring": {k: round(v, 1) for k, v in spent.items()},
        "compared": s1.get("compared"),
        "matches_reference": bool(ok),
        "reading": "every form is bound by the chain of dependent kernels per state (each search is "
                   "one CU's worth of work: 4-30 us on a few CUs); what the forms differ in is how "
                   "often the host waits (round_trips_per_state) and how many launches a state takes "
                   "(entry_point_calls_per_state); more than four chains on streams of their own do "
                   "not add up (about four dependent-kernel streams of a process run side by side), "
                   "more work per launch does: the engine's figures"}


def encoder_rd_figure(ctx, api, fx, pics, w, h):
    """Third workload: the REST of the reference encoder's RD search for that
    1080p B picture (tests/golden/rd_calls_c1.npz, tools/gen_rd_golden.py): every
    bi-prediction refinement step (FullSearch +-4 + sub-pel on the 2 * orig -
    other-list target), every affine motion search, every merge-candidate
    ranking and every inter-CU TransformAndReconstruct - the latter quantised
    with the CABAC context states RdoQuant::QuantRdo really read at that moment
    - as device batches (tests/rd_replay.py), every call compared with the
    reference's result.  The searches are one batch per picture pair; the merge
    candidates and the TransformAndReconstruct calls are ONE batch each - the RD
    recursion revisits every position at every size, so each candidate / CU state
    predicts into its own slot of a scratch picture (xvcgpu_inter_pred_batch_to)
    with its original copied beside (xvcgpu_copy_blocks), as the reference's temp
    buffers are."""
    import rd_fixture as rf
    import rd_replay
    if not os.path.exists(rf.path("c1")):
        return None
    by_poc = {int(fx.info[i]["poc"]): pics[i] for i in range(fx.n)}
    r = rd_replay.Replay(api, ctx, "c1", by_poc, w, h)
    r.bi_steps()        # warm-up (originals uploaded, kernels loaded)
    r.timing = {}
    bi_n, bi_bad, bi_skip = r.bi_steps()
    af_n, af_bad = r.affine_steps()
    mg_n, mg_bad = r.merges()
    r.dz_done = r.dz_bad = 0
    tx_n, tx_bad = r.transform_calls_scratch()
    dz_bad = r.dz_bad
    t = {k: 1e3 * v for k, v in r.timing.items()}
    poc = int(r.rd["evals"]["poc"][0])
    out = {"workload": "the rest of the reference encoder's RD search for the 1080p B picture "
                       "POC %d: %d bi-prediction refinement steps, %d affine motion searches, "
                       "%d merge rankings (x 5 candidates), %d TransformAndReconstruct calls of "
                       "%d CU states (%d CABAC context snapshots)" %
                       (poc, bi_n, af_n, mg_n, tx_n, len(r.rd["evals"]), len(r.rd["contexts"])),
           "ms": {"bi_steps": t.get("bi_steps"), "affine_searches": t.get("affine_steps"),
                  "merge_rankings": t.get("merges"),
                  "transform_and_reconstruct": t.get("scratch_transform_calls"),
                  "transform_distortions": t.get("scratch_transform_dist"),
                  "cbf_zero_distortions": t.get("scratch_dist_zero")},
           "ms_includes": "host job preparation excluded; descriptor upload, launches, result "
                          "download included (wall clock around each batch)",
           "transform_scratch": {"cu_states_x_rounds": r.timing.get("scratch_units"),
                                 "picture": "4096 x %d" % r.timing.get("scratch_rows", 0)},
           "matches_reference": bool(bi_bad == 0 and af_bad == 0 and mg_bad == 0 and tx_bad == 0
                                     and dz_bad == 0),
           "mismatches": {"bi": bi_bad, "affine": af_bad, "merge": mg_bad, "transform": tx_bad,
                          "dist_zero": dz_bad},
           "bi_steps_of_lic_cus_not_replayed": bi_skip,
           "transform_calls_per_s": (tx_n / (1e-3 * t["scratch_transform_calls"])
                                     if t.get("scratch_transform_calls") else None)}
    import oracle_lib as ol
    if ol.have_ref():       # the reference's own MotionEstNormal(kFullSearch), one thread
        import ctypes as C
        st = r.rd["steps"]
        st = st[(st["kind"] == rf.KIND_BI) & ((st["flags"] & rf.FLAG_LIC) == 0)]
        key = (int(st["other_ref_poc"][0]), int(st["ref_poc"][0]))
        sel = st[(st["other_ref_poc"] == key[0]) & (st["ref_poc"] == key[1])][:4000]
        jobs = np.zeros(len(sel), api.BI_DTYPE)
        b = jobs["blk"]
        for k in ("x", "y", "w", "h", "lambda16"):
            b[k] = sel[k]
        b["fullpel_mv"] = (sel["flags"] & rf.FLAG_FULLPEL) != 0
        i = np.arange(len(sel))
        k0 = sel["start_mvp_idx"].astype(np.int64)
        b["mvp_x"], b["mvp_y"] = sel["mvp"][i, k0, 0, 0], sel["mvp"][i, k0, 0, 1]
        b["search_range"] = 4
        jobs["blk"] = b
        jobs["other_mv_x"], jobs["other_mv_y"] = sel["other_mv"][:, 0, 0], sel["other_mv"][:, 0, 1]
        jobs["boot_mv_x"], jobs["boot_mv_y"] = sel["boot"][:, 0, 0], sel["boot"][:, 0, 1]
        res = np.zeros(len(sel), api.MERES_DTYPE)
        lib = C.CDLL(ol.REF_SO)
        planes = [rd_replay.original_planes(w, h, poc)[0]] + [
            np.ascontiguousarray(np.pad(by_poc[p].download(0)[0], rd_replay.BL, mode="edge"))
            for p in key]
        args = []
        for a in planes:
            args += [C.c_void_p(a.ctypes.data + 2 * rd_replay.BL * (a.shape[1] + 1)),
                     C.c_ssize_t(a.shape[1])]
        t0 = time.perf_counter()
        lib.xr_bipred_search_many(10, C.c_void_p(jobs.ctypes.data), len(sel), w, h, *args,
                                  C.c_void_p(res.ctypes.data))
        dt = time.perf_counter() - t0
        assert np.array_equal(res["mv_x"], sel["mv"][:, 0, 0])
        out["cpu_reference_bi_steps_per_s"] = len(sel) / dt
        out["cpu_reference"] = ("InterSearch::MotionEstNormal(kFullSearch, bipred) of oracle/_ref on "
                                "the first %d steps, 1 thread (incl. one picture set-up)" % len(sel))
        out["bi_steps_per_s"] = bi_n / (1e-3 * t["bi_steps"]) if t.get("bi_steps") else None
    r.destroy()
    return out


def pad_planes(planes, border):
    return [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge"))
            for c, p in enumerate(planes)]


def usable_cpus():
    """Hardware threads this process may actually use: the affinity mask capped
    by the cgroup CPU quota (a 256-thread box with cpu.max = 16 CPUs runs 256
    OpenMP threads slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(args, clip, bd, border):
    """The same frame passes on the host, chained like the GPU steps (each
    reconstruction is the next reference).  Kind "reference" when the
    reference build travels with the repo (oracle/_ref/libxvcref.so, compiled
    by oracle/Makefile from the reference's own sources): its classes - TZ
    search, sub-pel search, interpolation, transforms, QuantRdo (or QuantFast), deblocking,
    PadBorder, ComparePicture, with its SSE2/AVX2 kernels - run the composition
    (ref_harness.cc xr_frame_pass; tests pin it equal to the oracle's).
    Otherwise kind "port": the plain-C oracle.  The per-CU loops are spread
    over the CPUs this process may use (OpenMP, identical results); the
    reported value is the all-threads one, `cores` = threads used.  About 15 s
    of CPU work in all."""
    import oracle_frame
    import oracle_lib as ol
    from xvc_amd import pipeline
    desc = pipeline.FrameDescriptors(args.width, args.height, args.qp,
                                     rdoq=args.quant == "rdoq", bitdepth=bd)
    frames = [pad_planes(clip.frame(i), border) for i in range(args.frames + 1)]
    F = len(frames) - 1

    def run(lib, reference, n_max, seconds, threads):
        ref, n = frames[0], 0
        t0 = time.perf_counter()
        while n < n_max and (n == 0 or time.perf_counter() - t0 < seconds):
            k = n % (2 * F - 2) if F > 1 else 0
            orig = frames[1 + (k if k < F else 2 * F - 2 - k)]
            ref = oracle_frame.frame_pass(desc, bd, orig, ref, border, lib=lib,
                                          threads=threads, reference=reference)[0]
            n += 1
        return n, time.perf_counter() - t0

    cores = usable_cpus()
    xo = ol.Lib("xo")
    np1, dtp1 = run(xo, False, max(1, int(args.cpu_frames)), 4.0, 1)
    if not ol.have_ref():
        nc, dtc = run(xo, False, 2000, 7.0, cores)
        return {
            "value": nc / dtc, "unit": "frame passes/s", "cores": cores, "kind": "port",
            "single_thread_value": np1 / dtp1,
            "sample": "%dx%d workload, chained frame passes, C oracle (gcc -O2): %d passes "
                      "in %.1f s on %d threads (OpenMP over the CUs; deblock/pad/SSD "
                      "serial); %d passes in %.1f s on 1 thread" %
                      (args.width, args.height, nc, dtc, cores, np1, dtp1),
        }
    xr = ol.Lib("xr")
    n1, dt1 = run(xr, True, 2000, 4.0, 1)
    nc, dtc = run(xr, True, 2000, 7.0, cores)
    return {
        "value": nc / dtc, "unit": "frame passes/s", "cores": cores, "kind": "reference",
        "single_thread_value": n1 / dt1, "port_single_thread_value": np1 / dtp1,
        "sample": "%dx%d workload, chained frame passes run by the reference's own classes "
                  "with its SIMD kernels (g++ -O2, oracle/_ref): %d passes in %.1f s on %d "
                  "threads (OpenMP over the CUs; deblock/pad/PSNR serial as in the "
                  "reference), %d passes in %.1f s on 1 thread; plain-C oracle: %d passes "
                  "in %.1f s on 1 thread" %
                  (args.width, args.height, nc, dtc, cores, n1, dt1, np1, dtp1),
    }


RDOQ_PACKED = None if os.environ.get("RDOQ_FUSED") is None else False


def algorithmic_bytes(d, W, H):
    """Algorithmic bytes per launch of every kernel of the pass (DESIGN.md
    section 4); d: the pass's FrameDescriptors."""
    S = 2
    n_luma = sum(int(b["w"]) * int(b["h"]) for b in d.me)
    n_all = int(1.5 * n_luma)
    # samples PadBorder writes: 128 around luma, 64 around each chroma plane
    border = ((W + 256) * (H + 256) - W * H) + 2 * ((W // 2 + 128) * (H // 2 + 128) - W * H // 4)
    return {
        # each plane read once: original + reference luma of the CUs
        "me_search": 2 * n_luma * S,
        # orig + reference read, reconstruction written, all three planes
        "recon_from_me": 3 * n_all * S,
        "mc_from_me": 2 * n_all * S,                 # reference read, prediction written
        "fwd_from_me": 4 * n_all * S,                # orig + ref read, pred + coefficients written
        "fwd_transform": 3 * n_all * S,              # orig + pred read, coefficients written
        "quant_rdo": 2 * n_all * S,                  # coefficients read, levels written
        "inv_transform": 3 * n_all * S,              # levels + pred read, rec written
        "residual_rdoq": 3 * n_all * S, "residual": 3 * n_all * S,
        "cu_info": 84 * d.n_cus,
        "deblock": 2 * n_all * S + 16 * (n_luma // 16),
        "pad_border": border * S,
        "picture_ssd": 2 * n_luma * S,
        # unfiltered reconstruction read, filtered picture + its border written,
        # original luma read, CU map / records
        "deblock_pad_ssd": 2 * n_all * S + n_luma * S + 16 * (n_luma // 16) + border * S,
    }


class HostStagedComm:
    """TESTING AID (XVC_BENCH_BACKEND=gloo): the interface of api.Comm with the
    picture going through host memory and torch.distributed gloo, so that several
    ranks can share the one GPU of a test box (RCCL refuses that).  Never used
    for a measurement."""

    def __init__(self, ctx, dist, world, rank):
        self.ctx, self.dist, self.world, self.rank, self.h = ctx, dist, world, rank, None

    def wait_event(self, ev):
        ev.synchronize()

    def record_event(self, ev):
        ev.record(self.ctx)

    def send_picture(self, pic, dst):
        import torch
        for a in pic.download(128):
            self.dist.send(torch.from_numpy(np.ascontiguousarray(a).view(np.int16)), dst)

    def recv_picture(self, pic, src):
        import torch
        planes = []
        for c in range(3):
            b = 128 >> (c > 0)
            a = np.zeros(((pic.h >> (c > 0)) + 2 * b, (pic.w >> (c > 0)) + 2 * b), np.uint16)
            self.dist.recv(torch.from_numpy(a.view(np.int16)), src)
            planes.append(a)
        pic.upload(planes, 128)

    def sync(self):
        self.ctx.sync()

    def destroy(self):
        pass




def intra_picture_dag(sp, us_by_kind):
    """States / critical path of the picture's state-dependency DAG (tools/state_dag.py: what
    a state reads of its spatial neighbours' decisions and of its CU's merge ranking; the
    CABAC context chain is NOT modelled - with it the walk is serial): how many states of ONE
    picture a walk that is handed the context snapshots could have in flight."""
    import state_dag
    names = ["merge_rank", "eval", "inter", "motion_only", "intra"]
    out = {"by_count": state_dag.critical_path(sp.states)}
    try:
        w = [float(us_by_kind[n]) for n in names]
        out["by_chained_us"] = state_dag.critical_path(sp.states, w)
    except (KeyError, TypeError):
        pass
    out["note"] = ("upper bound: spatial-neighbour and merge-ranking dependencies only; the entropy "
                   "coder's context state (restored per CU from everything coded before it, "
                   "cu_encoder.cc:166-170) makes the reference's own order serial")
    return out



def quoted_ref_encoder_cpu():
    """The reference encoder's own time on the clip the walk's picture comes from, measured by
    tools/ref_encoder_time.py on a GPU box's host (a 1080p inter picture costs the reference
    tens of seconds: too long for this run) - the newest committed
    profiles/rNN_ref_encoder_cpu.json, quoted with its file name."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_ref_encoder_cpu.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except ValueError:
        return None
    d["quoted_from"] = os.path.relpath(files[-1], ROOT) + " (not re-measured by this run)"
    return d


def rd_side_by_side(out):
    """Three figures for ONE 1080p B picture's RD search, side by side: what the reference
    encoder needs for a picture on the host's cores (quoted: quoted_ref_encoder_cpu), what
    the device walk reaches in the reference's issue order with k = 16 pictures in flight
    (this run: engine), and the ceiling the walk is chasing - the same searches and
    TransformAndReconstruct calls as whole-picture batches (this run: encoder_me_batches +
    encoder_rd_batches; a shape no bit-exact encoder can present)."""
    rs = out["encoder_rd_serial"]
    res = {}
    cpu = quoted_ref_encoder_cpu()
    if cpu:
        res["reference_cpu"] = {
            "pictures_per_s": {k: v["pictures_per_s"] for k, v in cpu["runs"].items()},
            "threads": {k: v["threads"] for k, v in cpu["runs"].items()},
            "host_cores": cpu.get("host_cores"), "quoted_from": cpu["quoted_from"],
            "what": "the whole encode of the clip's pictures (RD search + entropy coding + "
                    "filters): an upper bound on the time of the RD search alone"}
    eng = (rs.get("engine") or {}).get("pictures_per_s") or {}
    res["device_walk_pictures_per_s"] = {"engine_k16": eng.get("16"), "engine": eng,
                                         "chained_k1": (rs.get("chained") or {}).get("pictures_per_s")}
    try:
        ms = float(out["encoder_me_batches"]["ms"]) + sum(
            float(v) for v in out["encoder_rd_batches"]["ms"].values() if v)
        res["batch_ceiling"] = {"ms_per_picture": ms, "pictures_per_s": 1e3 / ms,
                                "what": "every search and TransformAndReconstruct call of the "
                                        "picture as whole-picture batches (intra states not "
                                        "included)"}
    except (KeyError, TypeError, ValueError):
        pass
    return res


def quoted_walk_profiles():
    """Kernel launches per CU state of the serial / chained walks, from the NEWEST committed
    rocprofv3 --kernel-trace --stats summaries (profiles/rNN_cu_state_<form>_kernel_stats.csv,
    tools/runs/rNN: `cu_state_walk.py --mode <form> --states 3200 --k 1 --no-check`).  Quoted,
    not measured by this run: each entry names its file."""
    import csv
    import glob
    out = {"note": "rocprofv3 summaries of tools/cu_state_walk.py --states 3200 --k 1 --no-check, "
                   "committed under profiles/ (not re-measured by this run; launches include the "
                   "result copies)"}
    for form in ("serial", "chained", "live"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_cu_state_%s_kernel_stats.csv" % form)))
        if not files:
            continue
        calls = sum(int(r["Calls"]) for r in csv.DictReader(open(files[-1])))
        out[form] = {"kernel_launches_per_state": round(calls / 3200.0, 2),
                     "file": os.path.relpath(files[-1], ROOT)}
    return out


def quoted_cpu_baseline():
    """The N > 1 lines carry the CPU baseline of the N = 1 run (it is measured on rank 0 at
    N = 1 only, as the contract asks): quoted from the newest committed bench line under
    profiles/, with its source named."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")), reverse=True):
        try:
            with open(path) as f:
                cpu = json.load(f).get("cpu_baseline")
        except (OSError, ValueError):
            continue
        if cpu:
            cpu = dict(cpu)
            cpu["quoted_from"] = os.path.relpath(path, ROOT) + " (the N = 1 run; not re-measured here)"
            return cpu
    return None


def main_pictures(args, torch, api, pipeline, synth, world, rank, local_rank):
    """N ranks x `slots` picture slots code the pictures of hierarchical sub-GOPs
    (length 16, two references per list as the reference configures itself) in
    the order and on the workers the reference's ThreadEncoder policy gives
    (xvc_amd/host/xvc_picture_schedule.h); each finished reference picture goes
    by RCCL send / recv to the ranks that list it (libxvcgpu.so's communicator,
    its own stream).  A step = one picture = one hot-path frame pass against its
    nearest L0 reference.  Warm-up and timed part are two consecutive parts of
    one timeline; the timed part codes exactly --steps pictures."""
    from xvc_amd import picture_parallel, schedule
    bd, border = 10, api.BORDER_LUMA
    W, H = args.width, args.height
    rdoq = args.quant == "rdoq"
    ctx = api.Context(local_rank)
    comm, dist = None, None
    if world > 1:
        import torch.distributed as dist      # control plane only: id hand-over, barrier, max
        dist.init_process_group("gloo")
        if os.environ.get("XVC_BENCH_BACKEND", "nccl") == "gloo":    # testing aid, see class
            comm, rccl_world = HostStagedComm(ctx, dist, world, rank), 0
        else:
            ids = [api.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = api.Comm(ctx, ids[0], world, rank)
            rccl_world = ctx.lib.xvcgpu_comm_world(comm.h)
    else:
        rccl_world = 1
    n_pictures = 1 + args.warmup + args.steps
    sched = schedule.Schedule(n_pictures, 16, 2, world, args.slots)
    clip = synth.SyntheticClip(W, H, bd)
    origs = []
    for n in range(args.frames):
        p = ctx.picture(W, H, bd)
        p.upload(pad_planes(clip.frame(n), border), border)
        origs.append(p)
    eng = picture_parallel.GpuPictureEngine(ctx, sched, rank, W, H, bd, args.qp, origs, comm,
                                            rdoq=rdoq)
    # the timeline position where the timed part starts: just before the
    # (1 + warmup + 1)-th encode
    enc = [i for i, o in enumerate(sched.ops) if o["kind"] == schedule.ENCODE]
    split = enc[1 + args.warmup]

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    picture_parallel.run_rank(sched, rank, eng, 0, split)
    barrier()
    t0 = time.perf_counter()
    picture_parallel.run_rank(sched, rank, eng, split, -1)
    barrier()
    dt = time.perf_counter() - t0
    coded = eng.encoded
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([coded], dtype=torch.int64)
        dist.all_reduce(c)
        coded = int(c.item())
    assert coded == n_pictures, (coded, n_pictures)
    # checksums of the last pictures (POC -> CRC of the three planes), from
    # whichever rank holds them: equal for every N
    tail = {}
    for i, p in enumerate(sched.pictures):
        if int(p["poc"]) >= n_pictures - 8 and int(p["rank"]) == rank:
            tail[int(p["poc"])] = ctx.picture_crc(eng.recs[i % eng.ring], 1).hex()
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, tail)
        tail = {k: v for part in parts for k, v in part.items()}

    # the same workload on ONE GPU (rank 0 alone, the others wait): what the
    # multi-GPU figure should be compared with - the N = 1 default of this
    # script is the chain workload, whose references are one picture away
    one_gpu = None
    if world > 1 and rank == 0:
        s1 = schedule.Schedule(n_pictures, 16, 2, 1, args.slots)
        e1 = picture_parallel.GpuPictureEngine(ctx, s1, 0, W, H, bd, args.qp, origs, None,
                                               rdoq=rdoq)
        enc1 = [i for i, o in enumerate(s1.ops) if o["kind"] == schedule.ENCODE]
        picture_parallel.run_rank(s1, 0, e1, 0, enc1[1 + args.warmup])
        e1.sync()
        t1 = time.perf_counter()
        picture_parallel.run_rank(s1, 0, e1, enc1[1 + args.warmup], -1)
        e1.sync()
        one_gpu = args.steps / (time.perf_counter() - t1)
    roof = cpu = None
    fp = eng.fps[0]
    if rank == 0:
        # per-kernel launch durations of the pass alone on rank 0 (the other ranks idle)
        recs = [ctx.picture(W, H, bd), ctx.picture(W, H, bd)]
        recs[0].upload(pad_planes(clip.frame(0), border), border)
        times, reps, F = {}, 5, len(origs)
        for i in range(1, F):
            ref, rec = recs[(i + 1) % 2], recs[i % 2]
            for name, fn in fp.kernel_steps(origs[i], ref, rec, ref_poc=i - 1):
                fn()
                ctx.sync()
                ctx.timer_begin()
                for _ in range(reps):
                    fn()
                times[name] = times.get(name, 0.0) + ctx.timer_end() / reps
            fp.run(origs[i], ref, rec, ref_poc=i - 1)
        times = {k: v / (F - 1) for k, v in times.items()}
        dom = max(times, key=times.get)
        alg = algorithmic_bytes(fp.desc, W, H)
        achieved = alg[dom] / (times[dom] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0,
                "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                "algorithmic_bytes": alg[dom], "ms_per_launch": times[dom], "measured": "alone",
                "all_kernels_ms": {k: round(v, 4) for k, v in times.items()}}
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline(args, clip, bd, border)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        n_xfer = int((sched.ops["kind"] == schedule.TRANSFER).sum())
        print(json.dumps({
            "metric": "hot-path frame passes/s (TZ + sub-pel ME, MC, transform + %s + "
                      "dequant + inverse, deblock, pad, PSNR parts), bit-exact vs the "
                      "reference's classes" % ("RDOQ" if rdoq else "QuantFast"),
            "value": args.steps / dt, "unit": "frame passes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "pictures_in_flight": world * args.slots, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "rccl_world_size": rccl_world,
            "tail_crc": {str(k): tail[k] for k in sorted(tail)},
            "same_workload_on_one_gpu": one_gpu,
            # the N = 1 point of THIS curve is `bench.py --gpus 1 --schedule subgop` (the
            # default N = 1 run times the three-chain workload); efficiency against it:
            "efficiency_vs_same_workload_on_one_gpu":
                (args.steps / dt) / (world * one_gpu) if one_gpu else None,
            "config": {"workload": "%dx%d yuv420p 30fps synthetic, QP %d, internal bitdepth 10, "
                                   "16x16 CUs, TZ range 96, %s" %
                                   (W, H, args.qp, "RDOQ" if rdoq else "QuantFast"),
                       "cus_per_picture": fp.desc.n_cus_total,
                       "regime": "sub-GOPs of 16 in coding order, every sub-GOP's key picture "
                                 "coded against the previous key picture's reconstruction",
                       "parallelism": "picture-level: sub-GOP 16 (layers of 1, 1, 2, 4, 8 "
                                      "pictures), ThreadEncoder policy on %d rank(s) x %d picture "
                                      "slots, each picture against its nearest L0 reference; "
                                      "%d reference transfers of %.1f MB by %s for "
                                      "%d pictures" % (world, args.slots, n_xfer,
                                                       origs[0].nbytes() / 1e6,
                                                       "RCCL send/recv" if rccl_world else
                                                       "HOST STAGING (testing aid)", n_pictures),
                       "schedule_makespan_pictures": int(sched.makespan)},
            "roofline": roof,
            "cpu_baseline": cpu if cpu is not None or world == 1 else quoted_cpu_baseline()}))
    if comm is not None:
        comm.sync()
        comm.destroy()
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    # HIP maps streams onto 4 hardware queues by default; torch / RCCL take
    # some, and two chains whose streams land on one queue run one after the
    # other (3 chains beside RCCL: 7530 -> 8650 frame passes/s with 8 queues).
    # Must be set before the runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    from xvc_amd import api, pipeline, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # testing aids: all ranks on one GPU (XVC_BENCH_DEVICE=0) need a transport that
    # accepts that (XVC_BENCH_BACKEND=gloo; RCCL refuses duplicate devices)
    backend = os.environ.get("XVC_BENCH_BACKEND", "nccl")
    if "XVC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["XVC_BENCH_DEVICE"])
    mode = args.schedule
    if mode == "auto":
        mode = "chains" if world == 1 and not args.force_sharded else \
            ("subgop" if args.width * args.height <= 1920 * 1088 and not args.force_sharded
             else "rows")
    if mode == "subgop":
        if args.gpus != world and world > 1:
            raise SystemExit("--gpus must equal WORLD_SIZE")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
        if "XVC_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["XVC_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        return main_pictures(args, torch, api, pipeline, synth, world, rank, local_rank)
    multi = world > 1 or args.force_sharded     # row-sharded engine + process groups
    if multi:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")

    bd, border = 10, api.BORDER_LUMA
    W, H = args.width, args.height
    rdoq = args.quant == "rdoq"
    ctx = api.Context(local_rank)
    clip = synth.SyntheticClip(W, H, bd)

    ts0 = None
    if multi:
        from xvc_amd import sharded
        # every chain on its context's own stream (handed to torch so that the
        # RCCL operations are ordered on it) - not torch's default stream
        # the exchange natively on RCCL (libxvcgpu.so's communicator, one per chain);
        # XVC_BENCH_COMM=torch keeps torch.distributed's all_to_all_single on the data path
        native = backend == "nccl" and os.environ.get("XVC_BENCH_COMM", "native") == "native"

        def new_comm(c):
            if not native:
                return None
            ids = [api.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            return api.Comm(c, ids[0], world, rank)
        comm0 = new_comm(ctx)
        rccl_world = ctx.lib.xvcgpu_comm_world(comm0.h) if comm0 is not None else \
            (dist.get_world_size() if backend == "nccl" else 0)
        runner = sharded.make_gpu_sharded(ctx, W, H, bd, args.qp, rank, world,
                                          torch.device("cuda", local_rank), dist,
                                          own_stream=True, rdoq=rdoq, native_comm=comm0)
        ts0 = runner.e.stream
    else:
        runner = None

    # resident inputs: `frames` original pictures + ping-pong reconstructions
    origs = []
    for n in range(1, args.frames + 1):
        p = ctx.picture(W, H, bd)
        p.upload(pad_planes(clip.frame(n), border), border)
        origs.append(p)
    pipelined = runner is None and args.two_queue and not args.graph
    ctx_lo = None
    if runner is None:
        recs = [ctx.picture(W, H, bd), ctx.picture(W, H, bd)]
        fp = pipeline.FramePass(ctx, W, H, bd, qp=args.qp, rdoq=rdoq, rdoq_packed=RDOQ_PACKED,
                                xcd_tiles=args.cu_order == "tiles")
        if pipelined:
            # two queues on the device: top half of every picture on a
            # high-priority stream, bottom half on a low-priority one
            ctx_lo = api.Context(local_rank)
            ctx.use_priority_stream(True)
            ctx_lo.use_priority_stream(False)
            pfp = pipeline.PipelinedFramePass(ctx, ctx_lo, W, H, bd, qp=args.qp, rdoq=rdoq)
    else:
        recs = runner.e.pictures
        fp = runner.e.fp
    recs[0].upload(pad_planes(clip.frame(0), border), border)
    ctx.sync()

    # ---- further independent chains (own context = own stream, own
    # reconstructions and job buffers; the originals are shared, each chain
    # starts at another phase of the frame cycle) ----
    # (one GPU: three chains at every size - 7680x4320: 438 / 509 / 515 passes/s with
    # 1 / 2 / 3; row shards of larger pictures keep one chain per rank)
    auto_chains = 3 if (not multi or W * H // world <= 3840 * 2160) else 1
    n_chains = 1 if pipelined else (args.chains or auto_chains)
    extra = []          # (ctx, runner-or-None, frame pass, recs, phase, torch stream)
    F = len(origs)
    cycle_len = 2 * F - 2 if F > 1 else 1
    for c in range(1, n_chains):
        cctx = api.Context(local_rank)
        phase = (c * cycle_len) // n_chains
        first = pad_planes(clip.frame(phase if phase < F else 2 * F - 2 - phase), border)
        if multi:
            ccomm = new_comm(cctx)
            crun = sharded.make_gpu_sharded(cctx, W, H, bd, args.qp, rank, world,
                                            torch.device("cuda", local_rank), dist,
                                            group=None if native else dist.new_group(),
                                            own_stream=True, rdoq=rdoq, native_comm=ccomm)
            ts = crun.e.stream
            crecs, cfp = crun.e.pictures, crun.e.fp
        else:
            ts, crun = None, None
            crecs = [cctx.picture(W, H, bd), cctx.picture(W, H, bd)]
            cfp = pipeline.FramePass(cctx, W, H, bd, qp=args.qp, rdoq=rdoq, rdoq_packed=RDOQ_PACKED,
                                     xcd_tiles=args.cu_order == "tiles")
        crecs[0].upload(first, border)
        cctx.sync()
        extra.append((cctx, crun, cfp, crecs, phase, ts))

    tail_priority = args.tail_priority == "on"
    if tail_priority:
        for c in [ctx] + [e[0] for e in extra]:
            c.set_short_kernel_priority(True)

    recordings = {}

    def orig_at(j):
        # frames 1..F then back down: consecutive pictures are always one
        # frame apart (no artificial scene cut when the clip wraps around)
        k = j % cycle_len
        return origs[k if k < F else 2 * F - 2 - k]

    def step(i, record_only=False):
        if n_chains > 1:
            c, j = i % n_chains, i // n_chains      # chain, its own step counter
            if c > 0:
                cctx, crun, cfp, crecs, phase, ts = extra[c - 1]
                o = orig_at(j + phase)
                if crun is not None:
                    with torch.cuda.stream(ts):
                        crun.run(o, j % 2, (j + 1) % 2, ref_poc=j)
                elif args.graph:
                    kk = (j + phase) % cycle_len
                    key = (c, kk, j % 2)
                    if key not in recordings:
                        ref_c, rec_c = crecs[j % 2], crecs[(j + 1) % 2]
                        recordings[key] = cctx.record(
                            lambda: cfp.run(o, ref_c, rec_c, ref_poc=kk))
                    if not record_only:
                        cctx.replay(recordings[key])
                else:
                    cfp.run(o, crecs[j % 2], crecs[(j + 1) % 2], ref_poc=j)
                return
            i = j
        F = len(origs)
        k = i % (2 * F - 2) if F > 1 else 0
        o = origs[k if k < F else 2 * F - 2 - k]
        ref, rec = recs[i % 2], recs[(i + 1) % 2]
        if runner is not None:
            with torch.cuda.stream(ts0):
                runner.run(o, i % 2, (i + 1) % 2, ref_poc=i)
        elif pipelined:
            pfp.run(o, ref, rec, ref_poc=i)
        elif not args.graph:
            fp.run(o, ref, rec, ref_poc=i)
        else:
            # one frame pass = one HIP graph launch: the sequence of launches
            # for (this original, this ping-pong parity) is recorded once,
            # before the warmup, and replayed (all kernels run every step)
            key = (0, k, i % 2)
            if key not in recordings:
                recordings[key] = ctx.record(lambda: fp.run(o, ref, rec, ref_poc=k))
            if not record_only:
                ctx.replay(recordings[key])

    def barrier():
        ctx.sync()
        for e in extra:
            e[0].sync()
        if ctx_lo is not None:
            ctx_lo.sync()
            ctx.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    if runner is None and args.graph:
        for i in range(2 * cycle_len * n_chains):
            step(i, record_only=True)
    settle = args.settle
    if settle < 0:
        settle = 1500 * n_chains if (runner is None and not pipelined) else 0
    settle -= settle % (2 * n_chains)      # keep the ping-pong parity and the chains' turn
    for i in range(settle):
        step(i)
    for i in range(settle, settle + args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_begin()
    for i in range(settle + args.warmup, settle + args.warmup + args.steps):
        step(i)
    if ctx_lo is not None:
        ctx.wait_for(ctx_lo)  # the event timer sits on the high-priority queue
    t_issued = time.perf_counter() - t0    # host time to issue every launch of the region
    gpu_ms = ctx.timer_end()
    barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # PSNR-Y of the last reconstructed picture (sanity, not timed)
    if runner is not None:
        ssd = runner.total_ssd()   # per-shard parts, all-reduced once here
    else:
        ssd = (pfp if pipelined else fp).d_ssd.to_array(np.uint64, 2)
    psnr_y = pipeline.psnr_from_ssd(int(ssd[0]), int(ssd[1])) if ssd[1] else None

    # ---- roofline of the dominant kernel: HIP events around it on the
    # context's stream, per launch, same data as the timed region ----
    roof = None
    if rank == 0:
        d = fp.desc
        # Per-kernel launch durations, averaged over one whole cycle of the
        # frame sequence (the searches are data dependent: 0.08-0.13 ms per
        # picture): for every step of the cycle each kernel is replayed
        # `reps` times on that step's inputs between two HIP events, then the
        # step itself runs to advance the reconstruction chain.
        F = len(origs)
        cycle = 2 * F - 2 if F > 1 else 1
        reps = 5
        times = {}

        def timed(fn):
            fn()
            ctx.sync()
            ctx.timer_begin()
            for _ in range(reps):
                fn()
            return ctx.timer_end() / reps

        # continue chain 0 where the timed region left it
        base = -(-(settle + args.warmup + args.steps) // n_chains)
        for i in range(base, base + cycle):
            k = i % cycle
            o = origs[k if k < F else 2 * F - 2 - k]
            ref, rec = recs[i % 2], recs[(i + 1) % 2]
            # every launch of the pass in issue order (each leaves what the next
            # one reads), then the pass itself to advance the chain
            for name, fn in fp.kernel_steps(o, ref, rec, ref_poc=i):
                times[name] = times.get(name, 0.0) + timed(fn)
            fp.run(o, ref, rec, ref_poc=i)
        times = {k: v / cycle for k, v in times.items()}
        dom = max(times, key=times.get)

        # The same kernel with the other chains in flight (what a kernel trace
        # of this command shows): one more cycle of the round-robin, chain 0's
        # launch of the search bracketed by two events per step, read afterwards.
        in_flight = None
        if n_chains > 1 and not multi and cycle <= 31:
            start = (base + cycle) * n_chains
            for i in range(start, start + cycle * n_chains):
                if i % n_chains:
                    step(i)
                    continue
                j = i // n_chains
                o = orig_at(j)
                ref, rec = recs[j % 2], recs[(j + 1) % 2]
                slot = 2 * (j - base - cycle)
                ctx.timer_mark(slot)
                ctx.me_search_dev(o, ref, fp.me_flags, fp.d_me.ptr,
                                  d.n_cus, fp.d_res.ptr, d.cu_size)
                ctx.timer_mark(slot + 1)
                fp.run(o, ref, rec, ref_poc=j)     # the whole pass (search repeated)
            barrier()
            ms = [ctx.timer_between(2 * q, 2 * q + 1) for q in range(cycle)]
            in_flight = {"kernel": "me_search", "chains": n_chains,
                         "ms_per_launch": sum(ms) / len(ms)}
        alg = algorithmic_bytes(d, W, H)
        # The profile carries the MD5 of the kernel sources it was taken from:
        # a figure from other kernels than the ones running now is not reported.
        traffic = None
        kname = dom
        try:
            prof = json.load(open(os.path.join(
                ROOT, "profiles",
                "traffic_current.json" if args.quant == "rdoq" else "traffic_current_fast.json")))
            kname = {"me_search": "me_search_sq16_kernel",
                     "recon_from_me": "recon_from_me_kernel", "quant_rdo": "quant_rdo_packed4_kernel",
                     "fwd_transform": "residual_wave_kernel<1", "inv_transform": "residual_wave_kernel<2",
                     "mc_from_me": "mc_from_me_kernel",
                     "picture_ssd": "picture_ssd_kernel", "pad_border": "pad_border_kernel",
                     "deblock": "deblock_pass_kernel<true>",
                     "deblock_pad_ssd": "deblock_tail_kernel"}.get(dom, dom)
            if (W == 1920 and H == 1080 and not multi and
                    prof.get("kernel_source_md5") == kernel_source_md5() and
                    prof.get("quant") == args.quant):
                hit = [v for k, v in prof["kernels"].items() if kname in k]
                traffic = hit[0]["hbm_bytes_per_launch"] if hit else None
        except (OSError, KeyError, ValueError):
            traffic = None
        # What the kernel is really bound by: the vector instructions it issues.
        # wave64 instructions per launch (SQ_INSTS_VALU, a separate --pmc pass of this
        # command, profiles/issue_current.json, same MD5 rule) x the measured issue
        # cost of the kernels' instruction mix (3.6 clocks per instruction and SIMD,
        # profiles/r03_valu_rate.txt + tools/isa_mix.py) over 1024 SIMDs at 2.4 GHz.
        issue = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "issue_current.json")))
            if (W == 1920 and H == 1080 and not multi and
                    prof.get("kernel_source_md5") == kernel_source_md5() and
                    prof.get("quant") == args.quant):
                hit = [v for k, v in prof["kernels"].items() if kname in k]
                if hit:
                    floor_ms = hit[0]["valu"] * 3.6 / (1024 * 2.4e9) * 1e3
                    issue = {"wave_instructions": hit[0]["valu"], "clocks_per_instruction": 3.6,
                             "simds": 1024, "clock_ghz": 2.4, "floor_ms": floor_ms,
                             "frac_alone": floor_ms / times[dom],
                             "active_lanes": hit[0].get("active_lanes"),
                             "lds_bank_conflict_share": hit[0].get("lds_bank_conflict_share")}
        except (OSError, KeyError, ValueError, NameError):
            issue = None
        achieved = alg[dom] / (times[dom] * 1e-3) / 1e9
        # `bound` names the roofline the harness prices the kernel against (these are
        # byte / integer paths: HBM); `limited_by` is what the counters say holds it
        # back in fact
        if achieved / 8000.0 >= 0.5:
            limited = "hbm bandwidth"
        elif issue is not None and issue["frac_alone"] >= 0.6:
            limited = "VALU issue (%.0f %% of the instruction-issue floor)" % (100 * issue["frac_alone"])
        elif issue is not None:
            limited = ("latency: the launch lasts as long as its longest wave - dependent LDS / "
                       "L2 round trips of one wave, %.0f %% of the VALU-issue floor, %.0f %% of "
                       "the lanes active" % (100 * issue["frac_alone"],
                                             100 * (issue.get("active_lanes") or 0)))
        else:
            limited = "latency / instruction issue (no counter profile for these kernel sources)"
        pass_bytes = sum(alg[k] for k in times if k in alg)
        roof = {"bound": "hbm", "limited_by": limited, "kernel": dom, "achieved": achieved,
                "peak": 8000.0,
                "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                "valu_issue": issue,
                # the whole pass: the algorithmic bytes of all its kernels over the time a
                # pass takes with the chains in flight
                "whole_pass": {"algorithmic_bytes": pass_bytes,
                               "gbps": pass_bytes / (dt / args.steps) / 1e9,
                               "frac": pass_bytes / (dt / args.steps) / 1e9 / 8000.0},
                "algorithmic_bytes": alg[dom],
                "ms_per_launch": times[dom],
                # per launch with the device to itself (chain 0, others idle):
                # the kernel's own figure; `in_flight`: the same launch while the
                # other chains run (agrees with a kernel trace of this command)
                "measured": "alone", "in_flight": in_flight,
                "all_kernels_ms": {k: round(v, 4) for k, v in times.items()},
                # algorithmic GB/s of every kernel of the pass (the whole-picture
                # ones - deblock, pad, SSD - are the HBM-bound ones)
                "all_kernels_gbps": {k: round(alg[k] / (v * 1e-3) / 1e9, 1)
                                     for k, v in times.items() if v > 0}}

    cpu = None
    if rank == 0 and not multi and not args.no_cpu:
        cpu = cpu_baseline(args, clip, bd, border)
    decode = None
    if rank == 0 and not multi and not args.no_decode:
        # its own context: the process holds more streams than the runtime has hardware
        # queues by now (the chains', the main context's), streams share queues in
        # creation order, and the host-bound decoder on the main context's stream ran at
        # half its rate (tools/dbg/dec_rate2.py: 4500 -> 3500 pictures/s beside three
        # idle contexts with 4 queues, 4440 with GPU_MAX_HW_QUEUES=8 as main() sets it -
        # but this process holds more than eight streams by now)
        # ... so the chains' contexts (their streams) go first: the frame-pass figures are
        # taken
        for e in extra:
            e[0].close()
        extra = []
        dctx = api.Context(local_rank)
        dctx.use_own_stream()
        decode = stream_decode_figure(dctx, api)
        dctx.close()

    if rank == 0:
        value = args.steps / dt
        out = {
            "metric": "hot-path frame passes/s (TZ + sub-pel ME, MC, transform + %s + "
                      "dequant + inverse, deblock, pad, PSNR parts), bit-exact vs the "
                      "reference's classes" % ("RDOQ" if rdoq else "QuantFast"),
            "value": value, "unit": "frame passes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "gpu_ms_per_step_events": gpu_ms / args.steps if n_chains == 1 else None,
            "pictures_in_flight": n_chains,
            "tail_on_priority_stream": bool(tail_priority),
            "host_issue_ms_per_step": 1e3 * t_issued / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "rccl_world_size": rccl_world if multi else 1,
            # what this rank's exchanges move per picture, from the C++ plan
            # (xvc_shard_plan_traffic): {exchange: [RCCL operations, bytes sent]}
            "shard_exchange_per_picture": ({k: list(v) for k, v in runner.traffic().items()}
                                           if runner is not None else None),
            "shard_control": ("xvc_host_sharded_frame_pass (C++: frame-pass phases on row ranges "
                              "+ ncclSend / ncclRecv groups)" if runner is not None and
                              isinstance(runner.comm, sharded.NativeComm) else None) if multi else None,
            "psnr_y": psnr_y,
            "config": {"workload": "%dx%d yuv420p 30fps synthetic, QP %d, internal "
                                   "bitdepth 10, 16x16 CUs, TZ range 96, %s" %
                                   (W, H, args.qp,
                                    "RDOQ (RdoQuant::QuantRdo + CoeffSignHideRdo, "
                                    "picture-initial CABAC contexts)" if rdoq else
                                    "QuantFast"),
                       "cus_per_picture": fp.desc.n_cus_total,
                       "regime": ("steady state of open picture chains (every picture coded "
                                  "against the previous reconstruction, no key-picture refresh), "
                                  "reached by %d untimed settle passes per chain before the "
                                  "warmup" % (settle // n_chains)) if settle else
                                 "chains started from the original of the first picture",
                       "parallelism": (("two-queue" if pipelined else
                                        "single" if n_chains == 1 else
                                        "%d independent picture chains in flight" % n_chains)
                                       if not multi else
                                       "cu-row-shard%d x %d chains in flight, halo rows by %s" %
                                       (world, n_chains,
                                        "RCCL send/recv groups issued by libxvcgpu.so" if native
                                        else "torch.distributed (%s)" % backend))},
            "roofline": roof,
            "cpu_baseline": cpu if cpu is not None or world == 1 else quoted_cpu_baseline(),
            "stream_decode": decode,
        }
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

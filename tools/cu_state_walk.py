#!/usr/bin/env python3
"""The RD search of one real 1080p B picture (stream fixture c1, POC 2) walked CU state
by CU state in the reference's issue order on k contexts at once (k independent
pictures in flight, here k replays of the same picture on their own streams and
buffers): microseconds per state, entry-point calls and read-backs per state,
pictures per second.

    python tools/cu_state_walk.py [--clip c1] [--poc 2] [--states 6000] [--k 1,4,8,16]
                                  [--mode serial|chained|by_state|live|engine]

--mode engine: k chains through the execution engine (one launch per step kind and round);
ENGINE_THREADS = T engines on T host threads and streams (every T-th chain each),
ENGINE_STREAMS = streams of one engine."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def walk(api, clip, poc, n_states, ks, mode="serial", check=True, decoded=None, sp=None,
         threads_list=(), engine_threads=None, reps=3, engine_live=None):
    """decoded: ({poc: device picture}, width, height) of the clip when the caller holds
    it already; sp: a rd_serial.SerialPicture to re-use."""
    import rd_serial
    import stream_fixture as sf
    base = None
    if decoded is None:
        from test_gpu_me_calls import decode_stream
        base = api.Context(0)
        pics, w, h = decode_stream(base, sf.StreamFixture(clip))
    else:
        pics, w, h = decoded
    sp = sp or rd_serial.SerialPicture(api, clip, poc)
    n = min(n_states, len(sp.states))
    # the stretch walked: the one whose mix of state kinds is closest to the whole picture's
    # (the 1080p picture's first states hold most of its intra states - 58
    # TransformAndReconstruct calls each: its first 4000 states would read 2.3 x the picture's
    # mean time per state)
    first = sp.representative_start(n)
    out = {"clip": clip, "poc": poc, "states_in_picture": len(sp.states), "states_walked": n,
           "first_state_walked": int(first), "stretch": sp.summary(first, n),
           "mode": mode, "summary": sp.summary(), "chains": {}}
    if mode == "engine":
        # k chains on ONE context: every round the chains' next steps grouped by kind, one
        # launch per kind (xvc_host_cs_run_programs_engine); each chain walks its own
        # stretch of the picture, so the steps do not line up
        lists = rd_serial.ref_lists_of(clip, poc)
        for k in ks:
            ectx = api.Context(0)
            ectx.use_own_stream()
            runs = [rd_serial.ChainedRun(api, ectx, sp, pics, w, h, lists) for _ in range(k)]
            # ENGINE_THREADS engines on as many host threads and streams, every T-th chain each;
            # with one thread ENGINE_STREAMS deals a round's groups over that many streams
            threads = int(engine_threads or os.environ.get("ENGINE_THREADS", "1"))
            # ENGINE_LIVE: the chains a live encoder could issue (a wait wherever the host's
            # entropy coder decides) instead of one chain per visit of a CU position
            elive = bool(int(os.environ.get("ENGINE_LIVE", "0"))) if engine_live is None else bool(engine_live)
            extra = [api.Context(0) for _ in range(max(int(os.environ.get("ENGINE_STREAMS", "1")), threads) - 1)]
            for c in extra:
                c.use_own_stream()
            firsts = [sp.position_start(c * (len(sp.states) - n - 64) // max(k - 1, 1)) for c in range(k)]
            rd_serial.ChainedRun.run_engine(runs[:min(k, 2)], firsts[:min(k, 2)], min(n, 300), streams=extra, live=elive)   # warm-up
            walls, stats = [], None
            for _ in range(reps):
                t0 = time.time()
                stats = rd_serial.ChainedRun.run_engine(runs, firsts, n, streams=extra, threads=min(threads, k), live=elive)
                walls.append(time.time() - t0)
            # (the wall time holds the Python loop that records the programs; the engine's own
            # clock starts when the first step is issued)
            entry = {"us_per_cu_state_aggregate": 1e6 * stats.seconds / stats.states,
                     "states_per_s": stats.states / stats.seconds,
                     "pictures_per_s": stats.states / stats.seconds / len(sp.states),
                     "launches_per_state": stats.api_calls / stats.states,
                     "round_trips_per_state": stats.round_trips / stats.states,
                     "states": int(stats.states)}
            if check:
                runs_c = runs[:min(k, 4)]
                rd_serial.ChainedRun.run_engine(runs, firsts, n, verify=True, streams=extra, threads=min(threads, k),
                                                live=elive)
                ok = True
                for r, f in zip(runs_c, firsts):
                    res = r.check(f, n, searches=False)
                    res.update(r.check_chained(f, n))
                    ok = ok and all(v[1] == 0 for v in res.values())
                entry["matches_reference"] = ok
            out["chains"][str(k)] = entry
            entry["streams"] = 1 + len(extra)
            entry["host_threads"] = min(threads, k)
            entry["live"] = elive
            for r in runs:
                r.destroy()
            for c in extra:
                c.close()
            ectx.close()
        if base is not None:
            for p in pics.values():
                p.destroy()
            base.close()
        return out
    for k in ks:
        ctxs = [api.Context(0) for _ in range(k)]
        for c in ctxs:
            c.use_own_stream()
        if mode == "serial":
            runs = [rd_serial.SerialRun(api, c, sp, pics, w, h) for c in ctxs]
        else:
            lists = rd_serial.ref_lists_of(clip, poc)
            runs = [rd_serial.ChainedRun(api, c, sp, pics, w, h, lists) for c in ctxs]
        stats = [None] * k

        def go(r, count, verify=False):
            if mode == "serial":
                return r.run_serial(first, count)
            # timed without the read-back of the composed prediction jobs, which only the
            # check below wants (an encoder takes the motion from the pass results)
            # modes: "chained" = a chain per visit of a CU position; "by_state" = a chain per
            # state; "live" = the chains a live encoder could issue (rd_serial.program)
            return r.run_chained(first, count, by_position=(mode == "chained"), verify=verify,
                                 live=(mode == "live"))

        def work(i):
            stats[i] = go(runs[i], n)

        for r in runs[:1]:          # warm-up (module load, scratch allocation)
            go(r, min(n, 200))
        if mode != "serial":        # the programs are recorded before the clock starts
            for r in runs:
                r.prepare(first, n, mode == "chained", False, mode == "live")
        # the median of three runs: with k host threads a run's rate depends on how
        # the threads and the streams' queues fall (0.28 - 0.57 pictures/s at k = 4)
        walls = []
        for _ in range(3 if k > 1 else 1):
            th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            t0 = time.time()
            for t in th:
                t.start()
            for t in th:
                t.join()
            walls.append(time.time() - t0)
        wall = sorted(walls)[len(walls) // 2]
        s0 = stats[0]
        done = sum(s.states for s in stats)
        per_pic = len(sp.states) / max(s0.states + s0.skipped, 1)      # scale to the whole picture
        entry = {
            "us_per_cu_state": 1e6 * sum(s.seconds for s in stats) / done,
            "api_calls_per_state": s0.api_calls / s0.states,
            "states_per_chain": s0.states / max(s0.round_trips, 1),
            "round_trips_per_state": s0.round_trips / s0.states,
            "states_per_s": done / wall,
            "pictures_per_s": done / wall / (s0.states * per_pic),
            "pictures_per_s_runs": [done / x / (s0.states * per_pic) for x in walls],
            "us_by_kind": {name: 1e6 * s0.seconds_by_kind[i] / max(s0.states_by_kind[i], 1)
                           for i, name in enumerate(("merge_rank", "eval", "inter", "motion_only", "intra"))},
        }
        if mode == "chained" and k > 1 and threads_list:
            # T host threads, each driving k / T chains interleaved
            entry["threads_x_chains"] = {}
            for T in threads_list:
                if T >= k or k % T:
                    continue
                groups = [runs[i::T] for i in range(T)]
                out_t = [None] * T

                def work_t(i):
                    out_t[i] = rd_serial.ChainedRun.run_interleaved(groups[i], first, n)
                th = [threading.Thread(target=work_t, args=(i,)) for i in range(T)]
                t0 = time.time()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                wall_t = time.time() - t0
                done_t = sum(x.states for x in out_t)
                entry["threads_x_chains"]["%dx%d" % (T, k // T)] = {
                    "states_per_s": done_t / wall_t,
                    "pictures_per_s": done_t / wall_t / (s0.states * per_pic)}
        if mode in ("chained", "live") and k > 1:
            # the same k chains driven by ONE host thread, a chain issued while the
            # others execute (xvc_host_cs_run_programs_interleaved)
            walls_i = []
            for _ in range(3):
                t0 = time.time()
                si = rd_serial.ChainedRun.run_interleaved(runs, first, n, mode == "chained", mode == "live")
                walls_i.append(time.time() - t0)
            wall_i = sorted(walls_i)[1]
            entry["one_thread"] = {"states_per_s": si.states / wall_i,
                                   "pictures_per_s": si.states / wall_i / (s0.states * per_pic)}
            if check:
                ri = runs[0].check(first, n, searches=False)
                entry["one_thread"]["matches_reference"] = all(v[1] == 0 for v in ri.values())
        if check:
            if mode != "serial":
                go(runs[-1], n, verify=True)
            res = runs[-1].check(first, n, searches=(mode == "serial"))
            if mode != "serial":
                res.update(runs[-1].check_chained(first, n))
            entry["matches_reference"] = all(v[1] == 0 for v in res.values())
            entry["compared"] = {a: v[0] for a, v in res.items()}
        out["chains"][str(k)] = entry
        for r in runs:
            r.destroy()
        for c in ctxs:
            c.close()
    if base is not None:
        for p in pics.values():
            p.destroy()
        base.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--clip", default="c1")
    ap.add_argument("--poc", type=int, default=2)
    ap.add_argument("--states", type=int, default=6000)
    ap.add_argument("--k", default="1,4,8,16")
    ap.add_argument("--mode", default="serial")
    ap.add_argument("--threads", default="", help="also: T threads x k/T interleaved chains, e.g. 2,4")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the comparison (and its second, verifying run): for kernel traces")
    a = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    from xvc_amd import api
    print(json.dumps(walk(api, a.clip, a.poc, a.states, [int(x) for x in a.k.split(",")], a.mode,
                          check=not a.no_check,
                          threads_list=[int(x) for x in a.threads.split(",") if x])))

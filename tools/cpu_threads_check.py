"""How the oracle frame pass scales with OpenMP threads on this host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, oracle_frame
from xvc_amd import pipeline, synth
W, H, bd, BL = 1920, 1080, 10, 128
xo = ol.Lib("xo")
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, BL if c == 0 else BL // 2, mode="edge")) for c, p in enumerate(planes)]
desc = pipeline.FrameDescriptors(W, H, 32)
ref, o = pad(clip.frame(0)), pad(clip.frame(1))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
for t in (1, 4, 16, 64, 256):
    t0 = time.time(); oracle_frame.frame_pass(desc, bd, o, ref, BL, lib=xo, threads=t); print("threads", t, "%.3f s" % (time.time() - t0))

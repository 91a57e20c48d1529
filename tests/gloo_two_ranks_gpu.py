"""Helper for test_gpu_parity.test_sharded_two_ranks_one_gpu: the product
row-sharded pass (GpuEngine + TorchComm) as TWO processes that share the one
GPU of a test box, exchanging device tensors through the gloo backend (RCCL
refuses two ranks on one device).  Everything but the transport is what runs
on a multi-GPU node.  Prints OK per rank."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch
    import torch.distributed as dist
    import oracle_frame
    import oracle_lib as ol
    from test_sharded import assert_valid_rows_equal
    from xvc_amd import api, pipeline, sharded, synth
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (352, 288)
    bd, qp, bl = 10, 32, api.BORDER_LUMA

    def padded(planes):
        return [np.ascontiguousarray(np.pad(p, bl >> (c > 0), mode="edge"))
                for c, p in enumerate(planes)]

    ctx = api.Context(0)
    clip = synth.SyntheticClip(w, h, bd)
    s = sharded.make_gpu_sharded(ctx, w, h, bd, qp, rank, world, dev, dist, own_stream=True)
    s.e.pictures[0].upload(padded(clip.frame(0)), bl)
    O = ctx.picture(w, h, bd)
    xo = ol.Lib("xo")
    desc = pipeline.FrameDescriptors(w, h, qp)
    ref = padded(clip.frame(0))
    for n in ((1, 2) if w > 1000 else (1, 2, 3)):
        orig = padded(clip.frame(n))
        O.upload(orig, bl)
        with torch.cuda.stream(s.e.stream):
            s.run(O, (n - 1) % 2, n % 2, n - 1)
        torch.cuda.synchronize()
        got_ssd = s.total_ssd()
        rec, _, _, _, ssd = oracle_frame.frame_pass(desc, bd, orig, ref, bl, n - 1, lib=xo,
                                                    threads=2)
        got = s.e.pictures[n % 2].download(bl)
        assert_valid_rows_equal(s, got, rec, h, (world, n))
        assert got_ssd == ssd, (got_ssd, ssd)
        ref = rec
    dist.barrier()
    dist.destroy_process_group()
    print("OK rank %d" % rank)


if __name__ == "__main__":
    main()

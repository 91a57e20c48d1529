"""Helper for test_gpu_parity.test_sharded_rccl_single_rank: the product
multi-GPU path (GpuEngine + TorchComm over the RCCL backend) as a one-rank
job on the one GPU a test box has.  Prints OK when the sharded frame pass and
its all-reduced PSNR parts equal the oracle's."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    import oracle_frame
    import oracle_lib as ol
    from xvc_amd import api, pipeline, sharded, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    w, h, bd, qp, bl = 208, 112, 10, 32, api.BORDER_LUMA

    def padded(planes):
        return [np.ascontiguousarray(np.pad(p, bl >> (c > 0), mode="edge"))
                for c, p in enumerate(planes)]

    ctx = api.Context(0)
    clip = synth.SyntheticClip(w, h, bd)
    s = sharded.make_gpu_sharded(ctx, w, h, bd, qp, 0, 1, dev, dist)
    s.e.pictures[0].upload(padded(clip.frame(0)), bl)
    # a second chain on its own stream, context and process group (bench.py --chains)
    ctx2, ts = api.Context(0), torch.cuda.Stream(device=dev)
    with torch.cuda.stream(ts):
        s2 = sharded.make_gpu_sharded(ctx2, w, h, bd, qp, 0, 1, dev, dist,
                                      group=dist.new_group())
    s2.e.pictures[0].upload(padded(clip.frame(0)), bl)
    O2 = ctx2.picture(w, h, bd)
    O = ctx.picture(w, h, bd)
    xo = ol.Lib("xo")
    desc = pipeline.FrameDescriptors(w, h, qp)
    ref = padded(clip.frame(0))
    for n in (1, 2):
        orig = padded(clip.frame(n))
        O.upload(orig, bl)
        s.run(O, (n - 1) % 2, n % 2, n - 1)
        O2.upload(orig, bl)
        with torch.cuda.stream(ts):
            s2.run(O2, (n - 1) % 2, n % 2, n - 1)
            ssd2 = s2.total_ssd()
        got_ssd = s.total_ssd()
        dist.barrier()
        torch.cuda.synchronize()
        rec, _, _, _, ssd = oracle_frame.frame_pass(desc, bd, orig, ref, bl, n - 1, lib=xo)
        got = s.e.pictures[n % 2].download(bl)
        assert all(np.array_equal(got[c], rec[c]) for c in range(3)), n
        assert got_ssd == ssd, (got_ssd, ssd)
        got2 = s2.e.pictures[n % 2].download(bl)
        assert all(np.array_equal(got2[c], rec[c]) for c in range(3)), n
        assert ssd2 == ssd
        ref = rec
    # the exchange itself, rank 0 to rank 0 (RCCL copies): row slabs of picture 0
    # into other rows of picture 1 and a CU metadata row, once packed (one
    # all_to_all_single with staging copies), once as point-to-point operations.
    # Catches tensor types the backend rejects (16-bit integers) and layout slips.
    for packed in (True, False):
        comm = sharded.TorchComm(dist, 0, 1, packed=packed)
        e = s.e
        e.cu_mem[:] = torch.arange(e.cu_mem.numel(), device=dev).to(torch.uint8)
        before = [m.clone() for m in e.mem] + [e.cu_mem.clone()]
        sends = [(0, e.row_slab(0, 0, 16, 20)), (0, e.row_slab(0, 1, 8, 10)),
                 (0, e.row_slab(0, 2, 8, 10)), (0, e.cu_slab(0, e.cus_per_row)),
                 (0, e.row_slab(0, 0, 33, 36))]
        recvs = [(0, e.row_slab(1, 0, 48, 52)), (0, e.row_slab(1, 1, 30, 32)),
                 (0, e.row_slab(1, 2, 2, 4)), (0, e.cu_slab(2 * e.cus_per_row, e.cus_per_row)),
                 (0, e.row_slab(1, 0, 97, 100))]
        exp = [t.clone() for _, t in sends]
        comm.exchange(sends, recvs, e.make_copier)
        comm.exchange(sends, recvs, e.make_copier)   # cached plan
        torch.cuda.synchronize()
        for (_, r), x in zip(recvs, exp):
            assert torch.equal(r, x), packed
        # nothing else moved
        touched = torch.zeros_like(e.mem[1].view(torch.uint8), dtype=torch.bool)
        for (_, r) in recvs[:3] + recvs[4:]:
            o = r.data_ptr() - e.mem[1].data_ptr()
            touched[o:o + r.numel()] = True
        same = e.mem[1].view(torch.uint8) == before[1].view(torch.uint8)
        assert bool((same | touched).all()), packed
        assert torch.equal(e.mem[0], before[0])
        e.mem[1].copy_(before[1])
        e.cu_mem.copy_(before[2])
        torch.cuda.synchronize()
    # the product path: libxvcgpu.so's own communicator and the C++ shard engine
    # (xvc_host_sharded_frame_pass: frame-pass phases on row ranges, RCCL groups
    # between them - none with one rank -, the PSNR parts all-reduced natively)
    ctx3 = api.Context(0)
    comm3 = api.Comm(ctx3, api.comm_unique_id(), 1, 0)
    s3 = sharded.make_gpu_sharded(ctx3, w, h, bd, qp, 0, 1, dev, dist, own_stream=True,
                                  native_comm=comm3)
    assert isinstance(s3.comm, sharded.NativeComm) and s3.e.one_call_phases
    assert ctx3.lib.xvcgpu_comm_world(comm3.h) == 1
    s3.e.pictures[0].upload(padded(clip.frame(0)), bl)
    O3 = ctx3.picture(w, h, bd)
    ref = padded(clip.frame(0))
    for n in (1, 2):
        orig = padded(clip.frame(n))
        O3.upload(orig, bl)
        with torch.cuda.stream(s3.e.stream):
            s3.run(O3, (n - 1) % 2, n % 2, n - 1)
            ssd3 = s3.total_ssd()
        ctx3.sync()
        rec, _, _, _, ssd = oracle_frame.frame_pass(desc, bd, orig, ref, bl, n - 1, lib=xo)
        got = s3.e.pictures[n % 2].download(bl)
        assert all(np.array_equal(got[c], rec[c]) for c in range(3)), ("native", n)
        assert ssd3 == ssd, (ssd3, ssd)
        ref = rec
    assert s3.traffic() == {"halo": (0, 0), "gather": (0, 0)}
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()

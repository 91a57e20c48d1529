"""Host cost of one batched P2P group through torch.distributed on RCCL, as a
function of the number of operations in it (one rank sending to itself)."""
import os, sys, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for n in (1, 2, 4, 8, 12):
    src = [torch.zeros(40000, dtype=torch.uint8, device=dev) for _ in range(n)]
    dst = [torch.zeros(40000, dtype=torch.uint8, device=dev) for _ in range(n)]
    ops = [dist.P2POp(dist.isend, t, 0) for t in src] + [dist.P2POp(dist.irecv, t, 0) for t in dst]
    try:
        for _ in range(5):
            for r in dist.batch_isend_irecv(ops): r.wait()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            for r in dist.batch_isend_irecv(ops): r.wait()
        host = (time.perf_counter() - t) / 200
        torch.cuda.synchronize()
        total = (time.perf_counter() - t) / 200
        print("%2d send + %2d recv: host %.1f us per group, %.1f us incl. device" % (n, n, host * 1e6, total * 1e6))
    except Exception as e:
        print(n, "failed:", str(e)[:200]); break
for name, fn in (("all_gather_into_tensor 64 KB", lambda a, b: dist.all_gather_into_tensor(b, a)),
                 ("all_to_all_single 64 KB", lambda a, b: dist.all_to_all_single(b, a)),
                 ("all_reduce 16 B", lambda a, b: dist.all_reduce(a[:16]))):
    a = torch.zeros(65536, dtype=torch.uint8, device=dev); b = torch.zeros(65536, dtype=torch.uint8, device=dev)
    for _ in range(5): fn(a, b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200): fn(a, b)
    host = (time.perf_counter() - t) / 200
    torch.cuda.synchronize()
    print("%-30s host %.1f us per call" % (name, host * 1e6))
# the product exchange (xvc_amd/sharded.py TorchComm): the 8 + 8 slabs of a 1080p
# halo exchange (4 luma rows + 2 x 2 chroma rows + one CU metadata row to each
# side), rank 0 to itself, packed into one collective vs a P2P operation each
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xvc_amd import api, sharded
ctx = api.Context(0)
s = sharded.make_gpu_sharded(ctx, 1920, 1080, 10, 32, 0, 1, dev, dist)
e = s.e
def slabs(idx, ya):
    return [(0, t) for y in (ya, ya + 64) for t in
            (e.row_slab(idx, 0, y, y + 4), e.row_slab(idx, 1, y // 2, y // 2 + 2),
             e.row_slab(idx, 2, y // 2, y // 2 + 2), e.cu_slab((y // 16) * e.cus_per_row, e.cus_per_row))]
for packed in (True, False):
    comm = sharded.TorchComm(dist, 0, 1, packed=packed)
    sends, recvs = slabs(0, 128), slabs(1, 512)
    for _ in range(5): comm.exchange(sends, recvs, e.make_copier)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200): comm.exchange(sends, recvs, e.make_copier)
    host = (time.perf_counter() - t) / 200
    torch.cuda.synchronize()
    total = (time.perf_counter() - t) / 200
    print("halo exchange 8 + 8 slabs, %s: host %.1f us, %.1f us incl. device" % (
        "packed  " if packed else "per slab", host * 1e6, total * 1e6))
dist.destroy_process_group()

"""tools/exp_protocol.py — host-side overhead of the N > 1 step (prescan, all_gather, carry combine,
rescan, all_reduce) against the plain scan, on one GPU with an RCCL group of size 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
nbytes = (16 << 30) // 330 * 330
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
buf = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, nbytes)
cap = nbytes // 300 + 16
rs = torch.empty(cap, dtype=torch.int64, device=dev)
gather_in = torch.zeros(7, dtype=torch.int64, device=dev)
gather_out = [torch.zeros(7, dtype=torch.int64, device=dev)]
counts = torch.zeros(2, dtype=torch.int64, device=dev)
def plain():
    ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), cap)
def proto():
    nn, ns, back0 = ctx.shard_prescan(buf.data_ptr(), nbytes)
    gather_in.copy_(torch.tensor([nbytes, nn, ns] + back0, dtype=torch.int64), non_blocking=False)
    dist.all_gather(gather_out, gather_in)
    rows = torch.stack(gather_out).cpu().numpy()
    ctx.rescan_launch(True, None, rs.data_ptr(), cap)
    s, c, st = ctx.scan_finish()
    counts[0] = s.n_records; counts[1] = 0
    dist.all_reduce(counts)
for name, fn in (("plain", plain), ("protocol", proto), ("plain", plain), ("protocol", proto)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print("%s: %.3f ms per step" % (name, dt), flush=True)
dist.destroy_process_group()

"""tools/exp_protocol.py — overhead of the N > 1 step against the plain scan, on one GPU with an RCCL group of size 1:
the host recipe (prescan, all_gather, carry combine on the host, rescan, all_reduce: two host hops) and the device
recipe (fqh_shard_prescan_launch, all_gather, fqh_shard_rescan_launch, all_reduce: one wait at the end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
nbytes = (16 << 30) // 330 * 330
torch.cuda.set_stream(torch.cuda.Stream(device=dev))   # (a real stream: the library replaces the null stream by one of its own)
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
W = pkg.SHARD_WORDS
words = torch.zeros(W, dtype=torch.int64, device=dev)
allw = torch.zeros(W, dtype=torch.int64, device=dev)
def proto_dev():
    ctx.shard_prescan_launch(buf.data_ptr(), nbytes, words.data_ptr())
    dist.all_gather_into_tensor(allw, words)
    ctx.shard_rescan_launch(True, allw.data_ptr(), 1, 0, rs.data_ptr(), cap, counts.data_ptr())
    dist.all_reduce(counts)
    s, c, st = ctx.scan_finish()
    assert s.n_records == nbytes // 330
for name, fn in (("plain", plain), ("host recipe", proto), ("device recipe", proto_dev), ("plain", plain), ("host recipe", proto), ("device recipe", proto_dev)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print("%s: %.3f ms per step" % (name, dt), flush=True)
dist.destroy_process_group()

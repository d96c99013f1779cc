"""tools/exp_long_trace.py — the long-read leg of bench.py alone (for rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
import bench
pkg = g.load_package()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_spin_wait(20000)
print(bench.long_read_leg(pkg, torch, dev, ctx, read_len=int(os.environ.get("READ_LEN", "5000"))))

"""Where does the time go when one rank brings up RCCL on a fresh box?  (diagnostic; run under torch.distributed.run)"""
import os, time, torch, torch.distributed as dist
t0 = time.time()
def T(msg): print(f"[{time.time() - t0:7.2f}s] {msg}", flush=True)
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
x = torch.ones(1024, device=dev); torch.cuda.synchronize(); T("cuda ready")
dist.init_process_group("nccl", device_id=dev); T("init_process_group")
y = torch.empty_like(x); dist.all_gather_into_tensor(y, x); torch.cuda.synchronize(); T("first all_gather")
dist.barrier(); torch.cuda.synchronize(); T("barrier")
dist.all_reduce(x); torch.cuda.synchronize(); T("all_reduce")
dist.destroy_process_group(); T("destroyed")

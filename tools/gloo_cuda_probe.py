"""probe: does torch's gloo all_gather_into_tensor take CUDA tensors on this box (two ranks sharing cuda:0)?"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
x = torch.full((8,), rank + 1, dtype=torch.int32, device=dev)
out = torch.zeros((world * 8,), dtype=torch.int32, device=dev)
try:
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print("rank", rank, "ok", out.tolist())
except Exception as e:
    print("rank", rank, "FAILED", repr(e)[:300])
dist.barrier()
dist.destroy_process_group()

"""One-rank RCCL sanity check of the collectives bench.py uses (gather, reduce, all_reduce, barrier) on the GPU box:
the multi-rank nccl path cannot run on a one-GPU box (two ranks cannot share a device under RCCL)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.arange(12, dtype=torch.float32, device="cuda").view(3, 4)
out = [torch.zeros_like(t)]
dist.gather(t, gather_list=out, dst=0)
assert torch.equal(out[0], t)
r = t.clone(); dist.reduce(r, dst=0); dist.all_reduce(r); dist.barrier()
import sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sunvolumerender_amd import dist as sd
asm = sd.FrameAssembler(37, 5, 8, 0, 1)
hdr = torch.arange(37 * 5 * 3, dtype=torch.float32, device="cuda")
assert torch.equal(asm.assemble(hdr).reshape(-1), hdr)
print("nccl 1-rank gather / reduce / all_reduce / barrier ok")
dist.destroy_process_group()

"""RCCL API rehearsal on one GPU: a one-rank `nccl` process group runs the same calls the data-parallel
path makes (broadcast of parameters, async AVG all-reduce on 64 MB views of the gradient arena, wait,
barrier) -- checks that RCCL accepts them as issued; results must be unchanged with one rank."""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from fcd_amd import FCDSolver
from fcd_amd.parallel import GradSync
from fcd_amd.synth import synth_batch
s = FCDSolver(None, image_size=64, g_conv_dim=16, d_conv_dim=16, batch_size=4, precision="fp32x3").to(dev)
gs = GradSync()
gs.world = 2                      # take the multi-rank code path; with one rank AVG leaves values unchanged
gs.broadcast_params([s.G, s.D])
s.grad_sync = gs
batch, _ = synth_batch(4, 64, 10, seed=1); b = tuple(t.to(dev) for t in batch)
for i in range(3):
    d_loss, g_loss = s.train_step(b, force_g=True)
torch.cuda.synchronize()
flat = s.arenas[0].grad
flat.normal_()
ref = flat.clone()
gs.allreduce_flat(flat)
torch.cuda.synchronize()
assert torch.equal(flat, ref), "one-rank AVG all-reduce changed the buffer"
dist.barrier()
print("rccl single-rank rehearsal ok:", float(d_loss), float(g_loss), "backend", dist.get_backend())
dist.destroy_process_group()

"""RCCL API rehearsal on one GPU: a one-rank `nccl` process group runs the same calls the data-parallel
path makes (broadcast of parameters and buffers, async AVG all-reduce on spans of the gradient arena issued
from inside backward by the autograd thread, the deferred D update, barrier) -- checks that RCCL accepts
them as issued (results must be unchanged with one rank) and prints what the call sequence costs a full-size
step beside the plain single-GPU step."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from fcd_amd import FCDSolver
from fcd_amd.parallel import GradSync
from fcd_amd.synth import synth_batch
full = len(sys.argv) > 1 and sys.argv[1] == "full"
kw = {} if full else dict(image_size=64, g_conv_dim=16, d_conv_dim=16, batch_size=4)
nb, size = (16, 128) if full else (4, 64)
batch, _ = synth_batch(nb, size, 10, seed=1); b = tuple(t.to(dev) for t in batch)


def run(sync, steps):
    torch.manual_seed(3)
    s = FCDSolver(None, precision="fp32x3", **kw).to(dev)
    if sync:
        gs = GradSync()
        gs.world = 2                  # take the multi-rank code path; with one rank AVG leaves values unchanged
        gs.broadcast_params([s.G, s.D])
        s.grad_sync = gs
    for _ in range(3):
        s.train_step(b, force_g=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        d_loss, g_loss = s.train_step(b, force_g=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    early = {k: r[1].fired_early for k, r in s._reducers.items()}
    w = torch.cat([p.detach().reshape(-1) for p in list(s.G.parameters()) + list(s.D.parameters())])
    return s, dt, early, w, float(d_loss), float(g_loss)


steps = 20 if full else 3
s0, t_plain, _, w0, *_ = run(False, steps)
s1, t_sync, early, w1, d_loss, g_loss = run(True, steps)
assert torch.equal(w0, w1), "one-rank data-parallel steps changed the weights"
assert early.get(0, 0) > 0 and early.get(1, 0) > 0, early
s1.sync_running_stats()
flat = s1.arenas[0].grad
flat.normal_()
ref = flat.clone()
s1.grad_sync.allreduce_flat(flat)
torch.cuda.synchronize()
assert torch.equal(flat, ref), "one-rank AVG all-reduce changed the buffer"
dist.barrier()
print(f"rccl single-rank rehearsal ok ({'full size' if full else 'small'}): losses {d_loss:.4f} {g_loss:.4f}, backend {dist.get_backend()}, "
      f"rccl {'.'.join(map(str, torch.cuda.nccl.version()))}; spans sent from inside backward per step (D, G) = "
      f"{early.get(0)}, {early.get(1)}; step {t_plain * 1e3:.2f} ms plain, {t_sync * 1e3:.2f} ms with the collectives issued "
      f"(one rank: no wire time, only the issue cost and RCCL's kernels)")
dist.destroy_process_group()

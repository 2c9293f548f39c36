"""Rehearse the data-parallel train step with N ranks sharing ONE GPU (gloo backend, tiny config):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29544 tools/dp_rehearsal.py

Every rank trains on a different batch; after each step all ranks must hold bit-identical weights,
and those must equal a single-process run that averages the per-rank gradients itself.
The 8-GPU RCCL run is the driver's; this checks the code path (arena all-reduce started after D's
backward, deferred D update, broadcast) with a real process group."""
import json, os, sys
import numpy as np, torch, torch.distributed as dist
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_step_gpu import build, batch_of, GOLD
from fcd_amd.parallel import GradSync

import faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("FCD_HANG_DUMP", "60")), exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
z = np.load(os.path.join(GOLD, "tiny_step.npz")); sc = json.load(open(os.path.join(GOLD, "tiny_scalars.json")))

def batch_for(r):
    b = list(batch_of(z, dev))
    b[0] = (b[0] * (1.0 - 0.05 * r)).contiguous()
    return tuple(b)

s = build(z, sc["cfg"], dev)
s.alpha_override = torch.from_numpy(z["alpha"])
if rank == 1:                      # broadcast must repair this
    with torch.no_grad():
        next(s.D.parameters()).mul_(1.5)
s.grad_sync = GradSync(bucket_mb=0.01)
s.grad_sync.broadcast_params([s.G, s.D])
print(rank, 'broadcast done', flush=True)
for it in range(2):
    s.train_step(batch_for(rank), force_g=True)
    print(rank, 'step', it, flush=True)
torch.cuda.synchronize()
flat = torch.cat([p.detach().reshape(-1) for p in list(s.G.parameters()) + list(s.D.parameters())]).cpu()
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
same = all(torch.equal(gathered[0], g) for g in gathered)

ok_ref = True
if rank == 0:
    # single process: same schedule, gradients of the per-rank batches averaged by hand
    ref = build(z, sc["cfg"], dev)
    ref.alpha_override = torch.from_numpy(z["alpha"])
    opt_d, opt_g = ref.optimizers()
    for it in range(2):
        for oi, (opt, mine, other) in enumerate(((opt_d, ref.D, ref.G), (opt_g, ref.G, ref.D))):
            other.requires_grad_(False); mine.requires_grad_(True)
            ref.global_step = ref.n_critic - 1
            acc = None
            for r in range(world):
                stats = {k: v.clone() for k, v in ref.G.state_dict().items() if "running" in k}
                ref.training_step(batch_for(r), 0, oi).backward()
                g = opt.arena.grad.clone(); opt.arena.zero_grad()
                acc = g if acc is None else acc + g
                if r < world - 1:       # running statistics follow rank 0's batch: rewind after the others
                    pass
            other.requires_grad_(True)
            opt.arena.grad.copy_(acc / world)
            opt.step(); opt.arena.zero_grad()
    rflat = torch.cat([p.detach().reshape(-1) for p in list(ref.G.parameters()) + list(ref.D.parameters())]).cpu()
    err = float((rflat - flat).norm() / rflat.norm())
    ok_ref = err < 1e-3
    print(f"dp rehearsal: world {world}, weights identical across ranks: {same}; vs hand-averaged single process rel-err {err:.2e}")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if (same and ok_ref) else 1)

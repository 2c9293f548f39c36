"""Data-parallel train steps with N ranks sharing ONE GPU (gloo backend, tiny config); started by
tests/test_parallel_gpu.py (or by hand):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29544 tests/dp_worker.py

Every rank trains on a different batch with the full data-parallel schedule of FCDSolver.train_step:
flat arenas, gradient spans sent to the all-reduce from inside the last backward pass
(parallel.BucketedReducer), D's update deferred behind the generator forwards.  Checked:

 1. after every step all ranks hold bit-identical weights;
 2. the same steps with the all-reduce issued after backward (no early sends) end in the same bits --
    an early send of an incomplete span would differ;
 3. they equal a single-process run that averages the per-rank gradients by hand (Adam's first steps
    are ~lr*sign(g), so an element at rounding level may flip: compared to 1e-3 in the weights and to
    1e-6 in the averaged gradients of the first D step);
 4. spans really went out from inside backward (fired_early > 0) for both networks;
 5. sync_running_stats() leaves every rank with rank 0's running statistics, and the validation
    confusion matrix is summed over ranks.
The multi-GPU RCCL run is the driver's; this checks the code path with a real process group."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from test_step_gpu import build, batch_of, GOLD      # noqa: E402
from fcd_amd.parallel import GradSync                  # noqa: E402

import faulthandler                                    # noqa: E402
faulthandler.dump_traceback_later(int(os.environ.get("FCD_HANG_DUMP", "100")), exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
z = np.load(os.path.join(GOLD, "tiny_step.npz"))
sc = json.load(open(os.path.join(GOLD, "tiny_scalars.json")))
STEPS = 3


def batch_for(r):
    b = list(batch_of(z, dev))
    b[0] = (b[0] * (1.0 - 0.05 * r)).contiguous()
    return tuple(b)


def flat_weights(s):
    return torch.cat([p.detach().reshape(-1) for p in list(s.G.parameters()) + list(s.D.parameters())]).cpu()


def run(bucket_overlap):
    s = build(z, sc["cfg"], dev)
    s.alpha_override = torch.from_numpy(z["alpha"])
    if rank == 1:                      # broadcast must repair this
        with torch.no_grad():
            next(s.D.parameters()).mul_(1.5)
    s.grad_sync = GradSync(bucket_mb=0.01)
    s.bucket_mb = (0.004, 0.002)       # a few spans per network even at the tiny size
    s.bucket_overlap = bucket_overlap
    s.grad_sync.broadcast_params([s.G, s.D])
    early = [0, 0]
    first_d_grad = None
    for it in range(STEPS):
        if it == 0:
            # the first D step by hand, to look at the averaged gradient before Adam consumes it
            s._run_optimizer(batch_for(rank), 0, 0, True)
            s._flush_pending_keep = True
            handle, opt, step = s._pending
            s.grad_sync.finish(handle)
            first_d_grad = opt.arena.grad.clone().cpu()
            s._pending = ([], opt, step)
            s.global_step = s.n_critic - 1
            s._run_optimizer(batch_for(rank), 0, 1, True)
            s._flush_pending()
            s.global_step = 1
        else:
            s.train_step(batch_for(rank), force_g=(it != 1))      # middle step: D update only
        for oi in (0, 1):
            red = s._reducers.get(oi)
            if red is not None:
                early[oi] += red[1].fired_early
    torch.cuda.synchronize()
    return s, flat_weights(s), early, first_d_grad


s, flat, early, dgrad = run(True)
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
same = all(torch.equal(gathered[0], g) for g in gathered)
_, flat_late, early_late, dgrad_late = run(False)
same_late = torch.equal(flat, flat_late) and torch.equal(dgrad, dgrad_late)
fired = early[0] > 0 and early[1] > 0 and early_late == [0, 0]

# running statistics: per-rank until synchronised, then rank 0's everywhere
stats = torch.cat([v.reshape(-1) for k, v in s.G.state_dict().items() if "running_" in k]).cpu()
g2 = [torch.empty_like(stats) for _ in range(world)]
dist.all_gather(g2, stats)
differed = not torch.equal(g2[0], g2[1])
s.sync_running_stats()
stats2 = torch.cat([v.reshape(-1) for k, v in s.G.state_dict().items() if "running_" in k]).cpu()
stats_ok = differed and torch.equal(stats2, g2[0])
# validation counts are summed over ranks
x = batch_for(rank)[0]
target = torch.full((x.shape[0], x.shape[2], x.shape[3]), 1 + rank, dtype=torch.uint8, device=dev)
s.G.eval()
s.validation_step((x, target))
cm = s.validation_epoch_end()["confusion_matrix"]
cm_ok = int(cm.sum()) == world * target.numel() and int(cm[0].sum()) == target.numel() == int(cm[1].sum())

ok_ref = True
if rank == 0:
    # single process: same schedule, gradients of the per-rank batches averaged by hand
    ref = build(z, sc["cfg"], dev)
    ref.alpha_override = torch.from_numpy(z["alpha"])
    opt_d, opt_g = ref.optimizers()
    gerr = None
    for it in range(STEPS):
        for oi, (opt, mine, other) in enumerate(((opt_d, ref.D, ref.G), (opt_g, ref.G, ref.D))):
            if oi == 1 and it == 1:
                continue
            other.requires_grad_(False)
            mine.requires_grad_(True)
            ref.global_step = ref.n_critic - 1
            acc = None
            for r in range(world):
                loss = ref.training_step(batch_for(r), 0, oi)
                with opt.arena.accumulating():
                    loss.backward()
                g = opt.arena.grad.clone()
                opt.arena.zero_grad()
                acc = g if acc is None else acc + g
            other.requires_grad_(True)
            opt.arena.grad.copy_(acc / world)
            if it == 0 and oi == 0:
                gerr = float((opt.arena.grad.cpu() - dgrad).norm() / dgrad.norm())
            opt.step()
            opt.arena.zero_grad()
    rflat = flat_weights(ref)
    err = float((rflat - flat).norm() / rflat.norm())
    ok_ref = err < 1e-3 and gerr < 1e-6
    print(f"dp worker: world {world}, weights identical across ranks: {same}; early sends == late sends: {same_late}; "
          f"spans sent from inside backward (D, G): {early}; first D-step averaged gradient vs hand-averaged {gerr:.2e}; "
          f"weights vs hand-averaged single process {err:.2e}; running stats synchronised: {stats_ok}; "
          f"validation counts summed: {cm_ok}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if (same and same_late and fired and ok_ref and stats_ok and cm_ok) else 1)

"""world_size-2 gloo test of the gradient mean all-reduce used for data parallelism."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcd_amd.parallel import GradSync
    torch.manual_seed(0)
    shapes = [(64, 11, 7, 7), (64,), (128, 64, 4, 4), (5,), (300, 300)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    gen = torch.Generator().manual_seed(100 + rank)
    for p in params:
        p.grad = torch.randn(p.shape, generator=gen)
    local = [p.grad.clone() for p in params]
    sync = GradSync(bucket_mb=0.05)
    sync.allreduce(params)
    # expected: mean over ranks, recomputed from the per-rank seeds
    exp = [torch.zeros(s) for s in shapes]
    for r in range(world):
        g2 = torch.Generator().manual_seed(100 + r)
        for e in exp:
            e += torch.randn(e.shape, generator=g2) / world
    ok = all(torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(params, exp))
    # broadcast makes rank 1 equal rank 0
    lin = torch.nn.Linear(4, 4)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    sync.broadcast_params([lin])
    ok = ok and bool((lin.weight == 1.0).all())
    # arena path: the flat gradient buffer itself is reduced, in bucket-sized views
    from fcd_amd.arena import ParamArena
    ps = [torch.nn.Parameter(torch.full(s, float(i))) for i, s in enumerate(shapes)]
    ar = ParamArena(ps)
    ok = ok and ar.intact() and all(bool((p == float(i)).all()) for i, p in enumerate(ps))
    gen = torch.Generator().manual_seed(100 + rank)
    for p_ in ps:
        p_.grad.copy_(torch.randn(p_.shape, generator=gen))
    sync.allreduce_flat(ar.grad)
    ok = ok and all(torch.allclose(p_.grad, e, atol=1e-6) for p_, e in zip(ps, exp)) and ar.intact()
    ok = ok and float(ar.grad.abs().sum()) > 0 and ar.flat.numel() % 64 == 0
    out[rank] = ok and not all(torch.equal(a, p.grad) for a, p in zip(local, params))
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]

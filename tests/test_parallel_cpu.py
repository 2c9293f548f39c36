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
    # diagnostics of the gradient exchange (bench.py prints them for N > 1): every finish() is timed per tag, the
    # version of a parameter written by broadcast_params moves (packed operands keyed on it are rebuilt), and the
    # reducer describes its spans
    v0 = lin.weight._version
    sync.broadcast_params([lin])
    ok = ok and lin.weight._version > v0
    sync.timing = []
    for p_ in ps:
        p_.grad.fill_(float(rank))
    sync.allreduce_flat(ar.grad, "D")
    sync.finish(sync.allreduce_flat_async(ar.grad), "G")
    ex = sync.exposed_ms()
    ok = ok and set(ex) == {"D", "G"} and all(v["calls"] == 1 and v["stream_wait_ms"] >= 0 and v["bytes"] == ar.grad.numel() * 4
                                              for v in ex.values()) and sync.timing == []
    ok = ok and all(bool((p_.grad == (world - 1) / 2).all()) for p_ in ps)       # (the arena's padding stays 0)
    desc = sync.bucketed(ar, 0.05).describe()
    ok = ok and abs(sum(desc["spans_mb"]) * (1 << 20) - ar.grad.numel() * 4) < 0.02 * (1 << 20) * len(desc["spans_mb"]) + 1 \
        and sum(desc["params_per_span"]) == len(ps)
    out[rank] = ok and not all(torch.equal(a, p.grad) for a, p in zip(local, params))
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]


def _reducer_worker(rank, world, port, out):
    """BucketedReducer on CPU tensors over gloo: spans go out as soon as their last parameter is
    reported in the final pass; results equal the plain mean; too many passes raise."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcd_amd.arena import ParamArena
    from fcd_amd.parallel import GradSync
    shapes = [(40, 3), (7,), (300, 10), (64,), (50, 50), (5,)]
    ps = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    ar = ParamArena(ps)
    sync = GradSync(bucket_mb=1.0)
    red = sync.bucketed(ar, bucket_mb=2000 * 4 / (1 << 20))       # spans of >= 2000 elements, built from the end
    ok = len(red.spans) >= 3 and red.spans[0][1] == ar.flat.numel() and red.spans[-1][0] == 0
    ok = ok and all(red.spans[i][0] == red.spans[i + 1][1] for i in range(len(red.spans) - 1))

    def fill(seed_base, scale=1.0):
        gen = torch.Generator().manual_seed(seed_base + rank)
        for p_ in ps:
            p_.grad.add_(torch.randn(p_.shape, generator=gen) * scale)

    def expected(seed_bases):
        exp = [torch.zeros(s) for s in shapes]
        for r in range(world):
            for sb in seed_bases:
                g2 = torch.Generator().manual_seed(sb + r)
                for e in exp:
                    e += torch.randn(e.shape, generator=g2) / world
        return exp

    # two passes; only the second may send.  Pass 1 reports nothing final.
    red.begin(2)
    with ar.accumulating():
        first = red.pass_begin()
        fill(100)
        for p_ in reversed(ps):
            red.param_ready(p_)              # ignored: not the final pass
        sent_in_first = len(red.handle)
        final = red.pass_begin()
        fill(200)
        fired_after = []
        for p_ in reversed(ps):
            red.param_ready(p_)
            fired_after.append(red.fired_early)
    sync.finish(red.finish_rest())
    ok = ok and (not first) and final and sent_in_first == 0
    ok = ok and fired_after[-1] == len(red.spans) and fired_after[0] < fired_after[-1]   # spread over the pass
    exp = expected([100, 200])
    ok = ok and all(torch.allclose(p_.grad, e, atol=1e-6) for p_, e in zip(ps, exp)) and ar.intact()
    # a pass outside accumulating() never sends early (autograd would deliver the gradients later)
    ar.zero_grad()
    red.begin(1)
    early_outside = red.pass_begin()
    fill(300)
    for p_ in ps:
        red.param_ready(p_)
    n_early = red.fired_early
    sync.finish(red.finish_rest())
    ok = ok and not early_outside and n_early == 0
    ok = ok and all(torch.allclose(p_.grad, e, atol=1e-6) for p_, e in zip(ps, expected([300])))
    # forward-order reporting (the GP's backward-of-backward walks D upwards) still covers every span
    ar.zero_grad()
    red.begin(1)
    with ar.accumulating():
        red.pass_begin()
        fill(400)
        for p_ in ps:
            red.param_ready(p_)
    sync.finish(red.finish_rest())
    ok = ok and all(torch.allclose(p_.grad, e, atol=1e-6) for p_, e in zip(ps, expected([400])))
    # more passes than declared: spans may already be gone -> loud failure
    red.begin(1)
    with ar.accumulating():
        red.pass_begin()
        try:
            red.pass_begin()
            raised = False
        except RuntimeError:
            raised = True
    sync.finish(red.finish_rest())
    ok = ok and raised
    # buffers: rank 0's values everywhere; counts: summed
    bn = torch.nn.BatchNorm1d(6)
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 1))
        bn.running_var.fill_(float(10 * (rank + 1)))
    sync.broadcast_buffers([bn])
    ok = ok and bool((bn.running_mean == 1.0).all()) and bool((bn.running_var == 10.0).all())
    cm = torch.tensor([[1, 2], [3, 4]], dtype=torch.int64) * (rank + 1)
    sync.sum_counts(cm)
    ok = ok and cm.tolist() == [[3, 6], [9, 12]]
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_bucketed_reducer_world2_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_reducer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_patch_loader_ranks_partition_one_permutation():
    """world_size > 1: every rank draws the same seed+epoch permutation, so the rank-strided slices
    are disjoint and cover the pool (DistributedSampler semantics), and epochs differ."""
    from fcd_amd.data_loader import PatchLoader

    class _Pool:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def gather(self, pos, aug, ret_mask):
            return pos.clone()

    n, world = 37, 2
    seen = []
    for epoch_loaders in [[PatchLoader(_Pool(n), batch_size=5, mode="train", rank=r, world_size=world, seed=11,
                                       generator=torch.Generator().manual_seed(1000 + r)) for r in range(world)]]:
        for _ in range(2):
            per_rank = [torch.cat(list(iter(ld))) for ld in epoch_loaders]
            allpos = torch.cat(per_rank)
            assert allpos.numel() == 38 and set(allpos.tolist()) == set(range(n))     # one wrap-around duplicate
            assert len(set(per_rank[0].tolist()) & set(per_rank[1].tolist())) <= 1
            seen.append(per_rank[0].tolist())
    assert seen[0] != seen[1]

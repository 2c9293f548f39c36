"""CPU suite: host logic of the HIP path (geometry descriptors, bucketing, state-dict layout),
the C-ABI library (loads, exports every symbol of include/fcd_hip.h) and loud failure
without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fcd_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fcd_hip.h")).read()
    declared = set(re.findall(r"\b(fcd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in fcd_hip.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    loaded = _lib.load()
    assert loaded.fcd_abi_version() == 2


def test_descriptor_structs_match_header_layout():
    from fcd_amd import _lib
    # sizeof(FcdConvDesc) / sizeof(FcdWgradDesc) of include/fcd_hip.h (gcc, x86-64): 33 ints + padding + 4 pointers;
    # 16 ints + 2 pointers
    assert ctypes.sizeof(_lib.ConvDesc) == 168 and _lib.ConvDesc.out_amax.offset == 160
    assert ctypes.sizeof(_lib.WgradDesc) == 80
    assert _lib.ConvDesc.a_amax.offset == 144 and _lib.ConvDesc.w_amax.offset == 152
    assert _lib.WgradDesc.g_amax.offset == 64 and _lib.WgradDesc.x_amax.offset == 72


def test_workspace_queries_run_without_gpu():
    from fcd_amd import _lib, ops
    lib = _lib.load()
    g = ops.ConvGeom(1024, 512, 4, 2, 1)
    d = g.fwd_desc(16, 8, 8)                      # D main.0.0: M = 256, needs split-K
    assert lib.fcd_conv_workspace_bytes(ctypes.byref(d)) > 0
    d2 = ops.ConvGeom(256, 256, 3, 1, 1).fwd_desc(16, 32, 32)
    assert lib.fcd_conv_workspace_bytes(ctypes.byref(d2)) == 0
    w = ops.ConvGeom(256, 256, 3, 1, 1).wgrad_desc(16, 32, 32)
    assert lib.fcd_wgrad_workspace_bytes(ctypes.byref(w)) >= 256 * 9 * 256 * 4
    assert lib.fcd_instnorm_workspace_bytes(16, 1024, 256) > 0
    # the 7x7 / 10-output-channel weight gradient takes the halo kernel: one slab per (image, row
    # range, 64-column strip) = 32 * 1 * 2 splits of [10][49 * 64] floats; an explicit split-K or a
    # channel count that is not a multiple of 16 falls back to the generic kernel's plan
    h = ops.ConvGeom(10, 64, 7, 1, 3).wgrad_desc(32, 128, 128)
    assert lib.fcd_wgrad_workspace_bytes(ctypes.byref(h)) == 64 * 10 * 49 * 64 * 4
    h.splitk = 5
    assert lib.fcd_wgrad_workspace_bytes(ctypes.byref(h)) == 5 * 10 * 49 * 64 * 4
    h8 = ops.ConvGeom(10, 8, 7, 1, 3).wgrad_desc(2, 32, 32)
    assert lib.fcd_wgrad_workspace_bytes(ctypes.byref(h8)) % (10 * 49 * 8 * 4) == 0
    # bad descriptors are rejected with a message, not a crash
    bad = g.fwd_desc(16, 8, 8)
    bad.Cin = 510
    rc = lib.fcd_conv_gather(ctypes.byref(bad), 16, 16, None, None, None, 16, None, 0, None)
    assert rc == -1 and b"Cin" in lib.fcd_last_error()


def test_conv_geometry():
    from fcd_amd import ops
    g = ops.ConvGeom(128, 64, 4, 2, 1)
    assert g.small_hw(128, 128) == (64, 64) and g.big_hw(64, 64) == (128, 128)
    d = g.adj_desc(2, 64, 64)
    assert d.nphase == 4 and d.KH == 2 and d.ostride == 2 and (d.GH, d.GW) == (64, 64)
    assert [d.pad_h[i] for i in range(4)] == [1, 1, 0, 0] and [d.ooff_w[i] for i in range(4)] == [0, 1, 0, 1]
    g7 = ops.ConvGeom(64, 11, 7, 1, 3)
    assert g7.IP == 12 and g7.OP == 64
    d = g7.adj_desc(2, 20, 24)
    assert d.nphase == 1 and d.pad_h[0] == 3 and (d.OH, d.OW) == (20, 24) and d.Cout == 11 and d.CoutP == 12
    w = g7.wgrad_desc(2, 20, 24)
    assert (w.GH, w.GW, w.O, w.I, w.IP) == (20, 24, 64, 11, 12)
    with pytest.raises(NotImplementedError):
        ops.ConvGeom(8, 8, 3, 2, 1).adj_desc(1, 4, 4)


def test_state_dict_layout_matches_reference_keys():
    """Key names/shapes listed in SURVEY.md section 8(b) (reference checkpoints load unchanged)."""
    from fcd_amd import Discriminator, Generator
    from oracle import fcd_oracle as O
    G = Generator(64, 1, 6, 10)
    ref = O.synth_generator_state(64, 1, 6, 10)
    sd = G.state_dict()
    assert set(sd) == set(ref) and all(sd[k].shape == ref[k].shape for k in ref)
    assert sum(p.numel() for p in G.parameters()) == 8461888
    D = Discriminator(128, 64, 1, 6, 10)
    refd = O.synth_discriminator_state(128, 64, 1, 6, 10)
    sdd = D.state_dict()
    assert list(sdd) == list(refd) and all(sdd[k].shape == refd[k].shape for k in refd)
    assert sum(p.numel() for p in D.parameters()) == 44736448
    G.load_state_dict(ref)
    D.load_state_dict(refd)
    # default init: U(+-1/sqrt(fan_in)) like torch's conv default
    w = Generator(8, 1, 1, 10).main[0].weight
    assert float(w.abs().max()) <= 1 / (11 * 49) ** 0.5 + 1e-7


def test_product_path_fails_loudly_on_cpu():
    from fcd_amd import Discriminator, Generator
    G = Generator(8, 1, 1, 10)
    with pytest.raises(RuntimeError, match="HIP path"):
        G(torch.zeros(1, 10, 16, 16), torch.zeros(1, 1))
    with pytest.raises(RuntimeError, match="HIP path"):
        Discriminator(16, 8, 1, 2, 10)(torch.zeros(1, 10, 16, 16))
    # options without a HIP path are rejected at construction, not silently emulated
    with pytest.raises(NotImplementedError):
        Generator(8, 1, 1, 10, activation="gelu")
    with pytest.raises(NotImplementedError):
        Discriminator(16, 8, 1, 2, 10, activation="gelu")
    with pytest.raises(NotImplementedError):
        Discriminator(16, 8, 1, 2, 10, use_attention=True)
    # interm_non_act keeps the reference's key layout (Sequential(act, conv): conv at index 1)
    assert "features.0.1.weight" in Discriminator(16, 8, 1, 2, 10, interm_non_act=True).state_dict()


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "fcd_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("oracle/", ""), fn


def test_bucket_plan():
    from fcd_amd.parallel import plan_buckets
    assert plan_buckets([10, 10, 10, 50, 5], 25) == [[0, 1], [2], [3], [4]]
    assert plan_buckets([], 10) == []
    assert plan_buckets([100], 10) == [[0]]


def test_on_before_batch_transfer_and_oracle_patch_batch():
    """Host half of the input pipeline (data_loader.py:56-71) and the numpy restatement of the
    batch assembly: flips are involutions, a shift moves pixels and fills with nodata."""
    import numpy as np
    from fcd_amd.data_loader import on_before_batch_transfer
    from oracle import fcd_oracle as O
    batch = {"image": torch.zeros(5, 10, 8, 8), "label": torch.tensor([0., 1., 1., 0., 1.]), "mask": torch.zeros(5, 8, 8)}
    out = on_before_batch_transfer(batch, generator=torch.Generator().manual_seed(3))
    assert len(out) == 6
    x, c_org, c_trg, l_org, l_trg, m = out
    perm = torch.randperm(5, generator=torch.Generator().manual_seed(3))
    assert l_org.shape == (5, 1) and torch.equal(l_trg, l_org[perm]) and torch.equal(c_trg, l_trg)
    assert len(on_before_batch_transfer({"image": batch["image"], "label": batch["label"]})) == 5

    rng = np.random.default_rng(0)
    images = rng.integers(0, 65536, size=(3, 6, 5, 4), dtype=np.uint16)
    masks = rng.integers(0, 3, size=(3, 6, 5), dtype=np.uint8)
    idx = np.array([2, 0])
    x0, m0 = O.patch_batch(images, idx, None, masks)
    assert x0.shape == (2, 4, 6, 5) and x0.dtype == np.float32
    assert np.array_equal(x0[0], np.transpose(O.normalize_u16(images[2]), (2, 0, 1)))
    xh, mh = O.patch_batch(images, idx, np.array([[1, 0, 0], [2, 0, 0]]), masks)
    assert np.array_equal(xh[0], x0[0][:, :, ::-1]) and np.array_equal(xh[1], x0[1][:, ::-1, :])
    assert np.array_equal(mh[0], m0[0][:, ::-1]) and np.array_equal(mh[1], m0[1][::-1, :])
    xs, ms = O.patch_batch(images, idx, np.array([[0, 2, -1], [0, 0, 0]]), masks)
    assert np.array_equal(xs[0][:, 2:, :-1], x0[0][:, :-2, 1:])
    assert (xs[0][:, :2, :] == -1.0).all() and (xs[0][:, :, -1] == -1.0).all()
    assert (ms[0][:2] == 0).all() and np.array_equal(ms[0][2:, :-1], m0[0][:-2, 1:])


def test_wgrad_pairing_protocol():
    """ops.WgradPairing (host logic of the one-launch weight gradients): within one backward() call the node that runs
    first leaves its item, the second takes it; an item nobody takes is launched by itself when the call ends; a
    mismatching taker leaves it alone; drain() launches a key's leftovers at once; a new call starts clean."""
    import torch
    from fcd_amd.ops import WgradPairing
    pairing, log = WgradPairing(), []

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, tag, mode):
            ctx.tag, ctx.mode = tag, mode
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            if ctx.mode == "drain":
                assert pairing.pending("w")
                pairing.drain("w")
                assert not pairing.pending("w")
                return g, None, None
            got = pairing.take("w", lambda it: ctx.mode != "mismatch") if ctx.mode != "leave-only" else None
            if got is not None:
                log.append(("pair", got, ctx.tag))
            else:
                pairing.leave("w", ctx.tag, lambda t=ctx.tag: log.append(("alone", t)))
            return g, None, None

    def run(modes):
        log.clear()
        x = torch.ones(3, requires_grad=True)
        y = x
        for i, m in enumerate(modes):          # node i+1 consumes node i: autograd runs the LAST one first
            y = Node.apply(y, i, m)
        y.sum().backward()
        assert pairing.empty()
        return list(log)

    assert run(["any", "any"]) == [("pair", 1, 0)]                                   # second pass takes the first's pair
    assert run(["any"]) == [("alone", 0)]                                            # a lone pass: flushed at the end
    assert run(["any", "any", "any"]) == [("pair", 2, 1), ("alone", 0)]              # the third finds nothing, leaves, is flushed
    assert run(["mismatch", "any"]) == [("alone", 1), ("alone", 0)]                  # refused by the taker: both go alone
    assert run(["any", "drain", "any"]) == [("alone", 2), ("alone", 0)]              # drained before the last node runs
    assert not pairing.pending("w") and pairing.take("w", lambda it: True) is None   # outside backward: nothing

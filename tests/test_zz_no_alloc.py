"""fcd_instnorm_stats_noalloc (include/fcd_hip.h): the InstanceNorm statistics with the column partials
in the library's built-in device array -- no workspace, no allocation inside the call.  Checked through
the C ABI against fcd_instnorm_stats (workspace variant, same kernels): identical bits, on two calls in a
row, the second on a smaller input (the array keeps the first call's contents), and beyond the maximum
size the call is refused."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, L, z, want_var):
    n, h, w, c = z.shape
    dev = z.device
    outs = []
    for noalloc in (False, True):
        mean = torch.full((n, c), float("nan"), device=dev)
        rstd = torch.full((n, c), float("nan"), device=dev)
        varu = torch.full((n, c), float("nan"), device=dev) if want_var else None
        st = torch.cuda.current_stream().cuda_stream
        if noalloc:
            rc = lib.fcd_instnorm_stats_noalloc(L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(varu), n, h * w, c, 1e-5, st)
        else:
            nbytes = lib.fcd_instnorm_workspace_bytes(n, h * w, c)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.fcd_instnorm_stats(L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(varu), n, h * w, c, 1e-5,
                                        ws.data_ptr(), nbytes, st)
        L.check(rc, "instnorm_stats" + ("_noalloc" if noalloc else ""))
        torch.cuda.synchronize()
        outs.append((mean, rstd, varu))
    return outs


def test_noalloc_stats_match_workspace_variant(dev):
    from fcd_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    # first call: the largest InstanceNorm of the baseline G step (paired batch of 32, 64 channels at 128x128)
    z1 = (torch.randn(32, 128, 128, 64, generator=g) * 1.7 + 0.3).to(dev)
    # second call, smaller and ragged: fewer samples, fewer chunks, another channel count
    z2 = (torch.randn(3, 37, 29, 12, generator=g) * 0.5 - 2.0).to(dev)
    for z, want_var in ((z1, True), (z2, False), (z1[:5].contiguous(), True)):
        (m0, r0, v0), (m1, r1, v1) = _both(lib, L, z, want_var)
        assert torch.isfinite(m1).all() and torch.isfinite(r1).all()
        assert torch.equal(m0, m1) and torch.equal(r0, r1)
        if want_var:
            assert torch.equal(v0, v1)
        ref = z.double().mean(dim=(1, 2))
        assert float((m1.double() - ref).abs().max()) <= 1e-5


def test_noalloc_rejects_oversize(dev):
    from fcd_amd import _lib as L
    lib = L.load()
    n, hw, c = 64, 64 * 64, 512          # 64 x 16 chunks x 512 channels x 16 B = 8 MiB of partials
    assert lib.fcd_instnorm_workspace_bytes(n, hw, c) > (4 << 20)
    z = torch.zeros(1, device=dev)        # never touched: the size check comes first
    out = torch.zeros(n * c, device=dev)
    rc = lib.fcd_instnorm_stats_noalloc(L.ptr(z), L.ptr(out), L.ptr(out), None, n, hw, c, 1e-5,
                                        torch.cuda.current_stream().cuda_stream)
    assert rc == -2                       # FCD_EWORKSPACE
    assert b"built-in scratch" in lib.fcd_last_error()

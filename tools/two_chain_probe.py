"""Is there time in running the residual-block chain as two half-batch chains on two streams?  Twelve
[3x3 256->256 convolution + InstanceNorm statistics + apply] layers on 32 x 32 images, scaled-fp16 precision:
(a) 32 samples on one stream (as the step runs a paired call), (b) two chains of 16 samples on two streams, started
together, (c) the same with the second chain started half a layer late, (d) 16 samples alone (half the chip idle).
python tools/two_chain_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
torch.manual_seed(0)
g = ops.ConvGeom(256, 256, 3, 1, 1)
NL = 12
wts = [torch.randn(256, 256, 3, 3, device=dev) / 48 for _ in range(NL)]
was = [ops.absmax(w) for w in wts]
wfs = [ops.pack_weight_fwd_h2(w, g, a) for w, a in zip(wts, was)]
gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def chain(x, amx):
    a = x
    for li in range(NL):
        d = g.fwd_desc(a.shape[0], 32, 32, precision=3)
        ops.set_amax(d, amx[li], was[li])
        amx[li + 1].zero_()
        z, mean, rstd, varu, y = ops.conv_gather_stats(d, a, wfs[li], apply=(gamma, beta, 0.0, None, amx[li + 1]))
        a = y
    return a


def amx_for(x):
    amx = torch.zeros((NL + 1, ops.AMAX_SLOTS), device=dev)
    ops.absmax(x, out=amx[0])
    return amx


x32 = torch.randn(32, 32, 32, 256, device=dev)
xa, xb = x32[:16].contiguous(), x32[16:].contiguous()
a32, aa, ab = amx_for(x32), amx_for(xa), amx_for(xb)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    chain(x32, a32)


def two(offset):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        chain(xa, aa)
    with torch.cuda.stream(s2):
        if offset:
            torch.cuda._sleep(int(offset))
        chain(xb, ab)
    cur.wait_stream(s1); cur.wait_stream(s2)


def half():
    chain(xa, aa)


def time_it(fn, n=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(2):
    print(f"(a) 32 samples, one stream      : {time_it(one):8.1f} us per 12 layers", flush=True)
    print(f"(b) 2 x 16 samples, two streams : {time_it(lambda: two(0)):8.1f} us", flush=True)
    print(f"(c) the same, second chain late : {time_it(lambda: two(120000)):8.1f} us  (offset 120 k cycles of s_sleep)", flush=True)
    print(f"(d) 16 samples alone            : {time_it(half):8.1f} us", flush=True)

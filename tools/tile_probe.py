import os, sys, torch, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def run(prec):
    from fcd_amd import _lib as L, ops
    dev = torch.device('cuda:0')
    g = ops.ConvGeom(256, 256, 3, 1, 1)
    n, h, w = 16, 32, 32
    x = torch.randn(n, h, w, 256, device=dev); wt = torch.randn(256, 256, 3, 3, device=dev) / 48
    wf = ops.pack_weight_fwd(wt, g)
    for sk in (1, 2, 3, 4):
        d = g.fwd_desc(n, h, w, precision=prec); d.splitk = sk
        out = ops.conv_gather(d, x, wf)
        for _ in range(3): ops.conv_gather(d, x, wf, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): ops.conv_gather(d, x, wf, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(os.environ.get('FCD_TILE'), 'prec', prec, 'splitk', sk, 'us', round(ms * 1e3, 1), 'TF', round(19.327 / ms, 1), flush=True)
if __name__ == '__main__':
    run(int(sys.argv[1]))

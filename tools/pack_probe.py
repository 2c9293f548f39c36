"""Time and bits of the BF16X3 weight packs on the discriminator's largest layer (4x4 s2, 1024 -> 2048)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for O, I in ((2048, 1024), (1024, 512), (128, 64)):
    g = ops.ConvGeom(O, I, 4, 2, 1)
    w = torch.randn(O, I, 4, 4, device=dev)
    for name, fn in (("fwd_x3", ops.pack_weight_fwd_x3), ("adj_x3", ops.pack_weight_adj_x3),
                     ("fwd", ops.pack_weight_fwd), ("adj", ops.pack_weight_adj)):
        out = fn(w, g)
        for _ in range(3): fn(w, g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn(w, g)
        e1.record(); torch.cuda.synchronize()
        sig = hashlib.sha256((out.view(torch.int16) if out.dtype == torch.bfloat16 else out).cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"{O}x{I} {name}: {e0.elapsed_time(e1) * 100:.1f} us sha {sig}", flush=True)

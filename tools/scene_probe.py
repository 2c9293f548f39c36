"""BASELINE.json configs[3]: mask of a tiled synthetic Landsat-8 scene (uint16, 10 bands, nodata border),
batch 64, everything on the device: tiling + normalisation, G eval forward, difference map, float64
stitch, nodata zeroing, float64 threshold."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.scene import generate_scene_mask
dev = torch.device('cuda:0')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 1900)
rng = np.random.default_rng(4)
scene = rng.integers(1, 65536, size=(h, w, 10), dtype=np.uint16)
scene[:64] = 0; scene[:, :48] = 0
torch.manual_seed(1)
s = FCDSolver(None, precision='fp32x3').to(dev)
s.G.eval()
diff, mask = generate_scene_mask(scene, s, threshold=0.02, batch_size=64)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n):
    diff, mask = generate_scene_mask(scene, s, threshold=0.02, batch_size=64)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
tiles = -(-h // 128) * -(-w // 128)
print(f"scene {h}x{w}: {dt * 1e3:.1f} ms per scene incl. the 76 MB upload and the mask download, {tiles} tiles -> "
      f"{tiles / dt:.0f} tiles/s, {h * w / dt / 1e6:.1f} Mpixel/s; mask sum {int(np.asarray(mask).sum())}", flush=True)

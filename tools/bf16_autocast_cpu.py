"""What does a bf16-operand FCD step look like in PyTorch itself?  The CPU oracle's D and G steps under
torch.autocast(device_type="cpu", dtype=torch.bfloat16) (convolutions in bf16, the rest by autocast's own
policy) against the float64 oracle: per-loss error and per-tensor gradient rel-err.  Runs in the build
container (no GPU); writes profiles/r2_bf16_autocast_cpu_<config>.json.

    python tools/bf16_autocast_cpu.py tiny|small|full"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fcd_oracle as O

CFG = {"tiny": dict(batch=2, size=32, bands=10, c_dim=1, g_conv_dim=8, d_conv_dim=8, g_repeat=2, d_repeat=3),
       "small": dict(batch=3, size=64, bands=10, c_dim=1, g_conv_dim=16, d_conv_dim=16, g_repeat=3, d_repeat=5),
       "full": dict(batch=16, size=128, bands=10, c_dim=1, g_conv_dim=64, d_conv_dim=64, g_repeat=6, d_repeat=6)}
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = CFG[name]
torch.set_num_threads(os.cpu_count())
gs = O.synth_generator_state(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
ds = O.synth_discriminator_state(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"])
batch, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])
gs64, ds64 = O.to_dtype(gs, torch.float64), O.to_dtype(ds, torch.float64)
b64 = tuple(t.double() for t in batch)
t0 = time.time()
_, dl64, dg64 = O.d_step(gs64, ds64, b64, alpha.double(), new_stats={})
_, gl64, gg64 = O.g_step(gs64, ds64, b64, new_stats={})
_, dl32, dg32 = O.d_step(gs, ds, batch, alpha, new_stats={})
_, gl32, gg32 = O.g_step(gs, ds, batch, new_stats={})
with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
    _, dlb, dgb = O.d_step(gs, ds, batch, alpha, new_stats={})
    _, glb, ggb = O.g_step(gs, ds, batch, new_stats={})


def rel(a, b):
    return float((a.double() - b).norm() / (b.norm() + 1e-300))


out = {"config": cfg, "seconds": None, "losses": {}, "grad_relerr_vs_fp64": {"d": {}, "g": {}}}
for tag, l64, l32, lb in (("d", dl64, dl32, dlb), ("g", gl64, gl32, glb)):
    for k in l64:
        den = max(1.0, abs(float(l64[k])))
        out["losses"][k] = {"fp64": float(l64[k]), "fp32_err": abs(float(l32[k]) - float(l64[k])) / den,
                            "bf16_autocast_err": abs(float(lb[k]) - float(l64[k])) / den}
for tag, g64, g32, gb in (("d", dg64, dg32, dgb), ("g", gg64, gg32, ggb)):
    for k in g64:
        out["grad_relerr_vs_fp64"][tag][k] = {"fp32": rel(g32[k], g64[k]), "bf16_autocast": rel(gb[k], g64[k])}
out["seconds"] = round(time.time() - t0, 1)
for tag in ("d", "g"):
    v = [e["bf16_autocast"] for e in out["grad_relerr_vs_fp64"][tag].values()]
    out[f"{tag}_grad_bf16_autocast_relerr_min_median_max"] = [min(v), sorted(v)[len(v) // 2], max(v)]
out["loss_bf16_autocast_err_max"] = max(e["bf16_autocast_err"] for e in out["losses"].values())
json.dump(out, open(os.path.join(ROOT, "profiles", f"r2_bf16_autocast_cpu_{name}.json"), "w"), indent=1)
print(name, "losses worst", out["loss_bf16_autocast_err_max"], "D grads (min, median, max)", out["d_grad_bf16_autocast_relerr_min_median_max"],
      "G grads", out["g_grad_bf16_autocast_relerr_min_median_max"], "seconds", out["seconds"])

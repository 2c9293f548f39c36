#!/usr/bin/env python3
"""FCD train images/sec: one full G+D step (D update with gradient penalty + G update, both
Adam steps) on synthetic 128x128 10-band patches, batch 16 per GPU.

    python bench.py --gpus N --steps K --warmup W [--precision fp32|bf16]

One rank per GPU (torch.distributed / RCCL when N > 1, launched by torch.distributed.run).
Rank 0 prints ONE JSON line; `value` is whole-job images/sec with inputs resident in HBM.
Beside it: `roofline` of the dominant kernel (the 256->256 3x3 residual-block convolution,
MFMA-bound) timed live with HIP events on the launch stream, and `cpu_baseline`: the CPU
oracle (a port of the reference's step) timed on this box's host cores on a bounded sample.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASE = dict(batch=16, size=128, bands=10, c_dim=1, conv_dim=64, g_repeat=6, d_repeat=6)
# algorithmic work per image of one G+D step: 12+1 G passes, 15 D passes
GFLOP_PER_IMG = 293.72
# MI355X_MICROARCH.md dense MFMA peaks: fp32 matrix 157.3 TF; the 16-bit forms run at 16 x that rate (fp16 = bf16)
PEAK_TFLOPS = {"fp32": 157.3, "fp32x3": 157.3, "fp32x2": 157.3, "bf16": 2516.8, "fp16": 2516.8}


def is_dominant(d):
    """Tag for ops.KernelTimer.  'dominant': the hottest launch, 3x3 s1 256->256 conv at 32x32 on the
    G step's paired batch of 2 x 16 samples (residual blocks, forward and dgrad).  'symbol': every other
    launch that runs the same kernel symbol -- rocprofv3's per-kernel average is over all of these.  In
    fp32x3 that is the 256x128-tile pre-split gather kernel (more than 64 output channels, 32-channel
    K slices, a grid of at least 256 such tiles: csrc/conv_igemm.hip gather_big_tile); in the other
    precisions the 128x128-tile kernel (Cout > 64 and more than 64 output pixels per phase)."""
    if (d.Cin == 256 and d.Cout == 256 and d.KH == 3 and d.KW == 3 and d.IH == 32 and d.IW == 32
            and d.N == 2 * BASE["batch"] and d.nphase == 1):
        return "dominant"      # the G step runs the generator on paired batches (2 x 16 samples per call)
    m = d.N * d.GH * d.GW
    if d.precision in (2, 3, 4):  # FCD_PREC_BF16X3 / FCD_PREC_F16X2 / FCD_PREC_F16
        big = -(-m // 256) * -(-d.Cout // 128) * d.nphase >= 256
        return "symbol" if (d.Cout > 64 and d.Cin % 32 == 0 and big) else None
    if d.Cout > 64 and m > 64:
        return "symbol"
    return None


def engine_clock_mhz(dev):
    """Current engine clock of `dev` from sysfs (amdgpu's pp_dpm_sclk marks the current level with '*'),
    matched by PCI address; None when the file is not there or not readable."""
    import glob
    try:
        p = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        for d in glob.glob("/sys/class/drm/card*/device"):
            if os.path.basename(os.path.realpath(d)) != bdf:
                continue
            for line in open(os.path.join(d, "pp_dpm_sclk")).read().splitlines():
                if line.strip().endswith("*"):
                    return int(line.split(":")[1].strip().split("Mhz")[0])
    except Exception:
        pass
    return None


def _sysfs_device_dir(dev):
    import glob
    p = torch.cuda.get_device_properties(dev)
    bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    for d in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(d)) == bdf:
            return d
    return None


class SysfsSampler:
    """Engine clock (pp_dpm_sclk, the level marked '*') and socket power (hwmon power1_average / power1_input,
    microwatts) of `dev`, read every `period` seconds by a side thread while the timed region runs.  The reads
    block in the kernel (GIL released); nothing is sampled when the files are not there."""

    def __init__(self, dev, period=0.02):
        import glob
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.sclk = self.power = None
        try:
            d = _sysfs_device_dir(dev)
            if d is not None:
                if os.access(os.path.join(d, "pp_dpm_sclk"), os.R_OK):
                    self.sclk = os.path.join(d, "pp_dpm_sclk")
                for name in ("power1_average", "power1_input"):
                    hit = glob.glob(os.path.join(d, "hwmon", "hwmon*", name))
                    if hit and os.access(hit[0], os.R_OK):
                        self.power = hit[0]
                        break
        except Exception:
            pass
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        mhz = watts = None
        try:
            if self.sclk:
                for line in open(self.sclk).read().splitlines():
                    if line.strip().endswith("*"):
                        mhz = int(line.split(":")[1].strip().split("Mhz")[0])
            if self.power:
                watts = int(open(self.power).read().strip()) / 1e6
        except Exception:
            pass
        return mhz, watts

    def _run(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            mhz, watts = self._read()
            self.samples.append((round((time.perf_counter() - t0) * 1e3, 1), mhz, watts))
            self._stop.wait(self.period)

    def start(self):
        if self.sclk or self.power:
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join()
        mhz = [m for _, m, _ in self.samples if m is not None]
        watts = [w for _, _, w in self.samples if w is not None]
        return {"samples": len(self.samples), "period_ms": self.period * 1e3,
                "sclk_mhz": {"min": min(mhz), "median": sorted(mhz)[len(mhz) // 2], "max": max(mhz)} if mhz else None,
                "socket_watts": {"min": round(min(watts), 1), "median": round(sorted(watts)[len(watts) // 2], 1),
                                 "max": round(max(watts), 1)} if watts else None,
                "trace_ms_mhz_w": [(t, m, None if w is None else round(w)) for t, m, w in self.samples][:64]}


def step_stats(step_ms):
    s = sorted(step_ms)
    return {"step_ms_min": round(s[0], 3), "step_ms_median": round(s[len(s) // 2], 3), "step_ms_max": round(s[-1], 3)}


def pmc_traffic(label, symbol):
    """HBM-side bytes per launch of kernel `symbol` from the rocprofv3 --pmc passes committed under profiles/
    (tools/prof_traffic.sh -> tools/pmc_traffic.py): FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md:
    the counter tallies 128-byte requests at 64 bytes) + WRITE_SIZE, both in KB.  The kernel is looked up by the
    symbol this run really launched (fcd_last_conv_kernel); if the profile does not hold that symbol -- the kernel
    changed since the profile was taken -- there is no figure, and the reason is reported instead of a stale one."""
    path = os.path.join(ROOT, "profiles", "r3_traffic.json")
    try:
        table = json.load(open(path))[label]
    except (OSError, KeyError, ValueError) as e:
        return None, f"no traffic profile for {label!r} in profiles/r3_traffic.json ({type(e).__name__})"
    hit = table.get(symbol)
    if hit is None:
        return None, (f"profiles/r3_traffic.json[{label!r}] has no kernel {symbol!r} (it holds {sorted(table)}): "
                      "re-run tools/prof_traffic.sh for the current kernels")
    return int(2 * hit["fetch_size_kb"] * 1024 + hit["write_size_kb"] * 1024), \
        f"{hit['files'][0]}, {hit['files'][1]}: FETCH_SIZE {hit['fetch_size_kb']:.0f} KB x 2 + WRITE_SIZE {hit['write_size_kb']:.0f} KB"


def dominant_kernel_roofline(precision, dev, in_step_us=None, in_step_launches=0, iters=20, symbol=None, overlapped_us=None):
    """The hottest kernel: 3x3 s1 256->256 conv at 32x32 on 32 samples (the G step pairs the reference's
    B=16 generator calls; forward form; dgrad runs the same kernel with flipped taps).  FLOPs per launch
    = 2*32*Cin*Cout*9*H*W.  `achieved` uses the
    average duration of this launch inside the timed training steps (HIP events around each of its
    launches on the launch stream); the back-to-back loop below is reported beside it."""
    from fcd_amd import _lib as L, ops
    g = ops.ConvGeom(256, 256, 3, 1, 1)
    n, h, w = 2 * BASE["batch"], 32, 32
    x = torch.randn(n, h, w, 256, device=dev)
    wt = torch.randn(256, 256, 3, 3, device=dev) / 48.0
    prec = {"fp32": L.PREC_F32, "fp32x3": L.PREC_BF16X3, "fp32x2": L.PREC_F16X2, "bf16": L.PREC_BF16, "fp16": L.PREC_F16}[precision]
    d = g.fwd_desc(n, h, w, precision=prec)
    fmt = ops.weight_format(d)
    if fmt == 2:
        amax_w, amax_x = ops.absmax(wt), ops.absmax(x)
        ops.set_amax(d, amax_x, amax_w)
        wf = ops.pack_weight_fwd_h2(wt, g, amax_w)
    else:
        wf = ops.pack_weight_fwd_x3(wt, g) if fmt == 1 else ops.pack_weight_fwd(wt, g)
    out = ops.conv_gather(d, x, wf)
    for _ in range(3):
        ops.conv_gather(d, x, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.conv_gather(d, x, wf, out=out)
    e1.record()
    torch.cuda.synchronize()
    loop_us = e0.elapsed_time(e1) / iters * 1e3
    # engine clock while this kernel runs back to back: queue ~1 s of launches, read sysfs twice while
    # they drain (valid only if the queue is still busy afterwards)
    for _ in range(int(1.0e6 / max(loop_us, 20.0))):
        ops.conv_gather(d, x, wf, out=out)
    e1.record()        # (the enqueue loop itself blocks on the full queue for a few 100 ms: the card is warm by now)
    clk = [engine_clock_mhz(dev)]
    time.sleep(0.05)
    clk.append(engine_clock_mhz(dev))
    sclk = None if (e1.query() or None in clk) else sum(clk) // 2
    torch.cuda.synchronize()
    ms = (in_step_us if in_step_us else loop_us) * 1e-3
    flops = 2.0 * n * 256 * 256 * 9 * h * w
    achieved = flops / (ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[precision]
    launched = L.load().fcd_last_conv_kernel().decode()
    traffic, traffic_src = (pmc_traffic("resblock_conv_x3_b32", launched) if precision in ("fp32x3", "fp32x2")
                            else (None, "traffic profile taken for the default kernel only"))
    extra = {}
    if precision == "fp32x2":
        # three fp16 MFMAs per fp32 product group: the scheme's own ceiling is the fp16 peak / 3
        peak = round(PEAK_TFLOPS["bf16"] / 3, 1)
        extra = {"executed_fp16_tflops": round(3 * achieved, 1), "fp32_mfma_peak": PEAK_TFLOPS["fp32"],
                 "frac_of_the_fp32x3_ceiling": round(achieved / (PEAK_TFLOPS["bf16"] / 6), 4),
                 "note": "algorithmic fp32 FLOPs; peak = dense fp16 MFMA peak / 3 (three fp16 MFMAs per fp32 product group: "
                         "operands as two fp16 pieces of the power-of-two-scaled values); the fp32x3 scheme of round 1 "
                         "(six bf16 MFMAs) had 416.7, the exact-fp32 MFMA pipe peaks at 157.3",
                 "sustained_mfma_rate": "tools/mfma_f16_probe.hip (random operands, two waves per SIMD, long launches): fp16 MFMAs "
                                        "alone sustain 0.62 of the nominal 2.5 PFLOP/s on this card (power), 0.53 with this "
                                        "kernel's 8 ds_read_b128 per 12 MFMAs beside them: 517 / 440 TFLOP/s in fp32x2 terms"}
    if precision == "fp32x3":
        # six bf16 MFMAs per fp32 product group: the scheme's own ceiling is the bf16 peak / 6
        peak = round(PEAK_TFLOPS["bf16"] / 6, 1)
        extra = {"executed_bf16_tflops": round(6 * achieved, 1), "fp32_mfma_peak": PEAK_TFLOPS["fp32"],
                 "note": "algorithmic fp32 FLOPs; peak = dense bf16 MFMA peak / 6 (six bf16 MFMAs per fp32 product group); "
                         "the exact-fp32 MFMA pipe peaks at 157.3",
                 "sustained_bf16_mfma_rate": "tools/mfma_bf16_probe.hip: bf16 MFMAs alone, random operands, sustain 0.76 of the "
                                             "nominal 2.5 PFLOP/s on this card (power), i.e. 317 TFLOP/s in fp32x3 terms"}
    if symbol is not None:
        # all launches of the same kernel symbol in the instrumented steps (what rocprofv3 --stats averages)
        cnt, us, fl = symbol
        extra["kernel_symbol_all_shapes"] = {
            "launches_per_step": cnt // 2, "avg_us_per_launch": round(us / max(1, cnt), 2),
            "achieved": round(fl / (us * 1e-6) / 1e12, 2), "frac": round(fl / (us * 1e-6) / 1e12 / peak, 4),
            "note": "every launch of this kernel symbol in 2 training steps (all layer shapes; "
                    "launches with fused InstanceNorm statistics include their ~6 us finalize kernel): compare "
                    "with the per-kernel average of profiles/*_kernel_stats.csv"}
    if overlapped_us:
        ach_ov = flops / (overlapped_us * 1e-6) / 1e12
        extra["overlapped"] = {
            "us_per_launch": round(overlapped_us, 2), "achieved": round(ach_ov, 2), "frac": round(ach_ov / peak, 4),
            "note": "the same launches in two steps as `value` runs them: weight gradients on a second HIP stream, so the "
                    "data-gradient half of these launches shares the CUs with a weight-gradient kernel (longer launches, "
                    "shorter step); kernel durations of such a step overlap and do not add up to the step time"}
    return {**extra, "bound": "mfma", "kernel": launched + " 3x3 256->256 @32x32, 32 samples per launch (two paired B=16 calls)",
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "us_per_launch": round(ms * 1e3, 2), "launches_timed": in_step_launches,
            "timed": "HIP events around every launch of this kernel in 2 training steps run right after the timed "
                     "region with the weight-gradient side stream off, i.e. every launch alone on the device (an event "
                     "pair per launch slows the step ~14 %, so they stay out of `value`)" if in_step_us
            else "back-to-back loop", "us_per_launch_back_to_back_loop": round(loop_us, 2),
            "gflop_per_launch": round(flops / 1e9, 3),
            "engine_clock_mhz_under_kernel": sclk, "engine_clock_mhz_nominal": 2400,
            "frac_at_measured_clock": round(achieved / peak * 2400.0 / sclk, 4) if sclk else None,
            "clock_note": "sysfs pp_dpm_sclk while this kernel runs back to back; the peaks above assume 2400 MHz, "
                          "so frac at the clock the card actually holds is frac * 2400 / engine_clock_mhz_under_kernel",
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": int(2 * n * h * w * 256 * 4 + 256 * 256 * 9 * (4 if precision == "fp32x2" else 6)),
            "traffic_unit": "HBM-side bytes per launch (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, looked up by the launched "
                            "kernel's symbol); algorithmic: input + output fp32 once + the weight planes (three bf16, or two fp16 under fp32x2)"}


def hbm_kernel_roofline(dev, iters=20):
    """The largest bandwidth-bound kernel pair: InstanceNorm apply (+ReLU) on the (16,128,128,64)
    activation: 1 read + 1 write = 2 x 67.1 MB algorithmic bytes per launch."""
    from fcd_amd import ops
    n, h, w, c = BASE["batch"], 128, 128, 64
    z = torch.randn(n, h, w, c, device=dev)
    mean, rstd, _ = ops.instnorm_stats(z)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    out = ops.instnorm_apply(z, mean, rstd, gamma, beta, 0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.instnorm_apply(z, mean, rstd, gamma, beta, 0.0, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = 2.0 * z.numel() * 4
    traffic, traffic_src = pmc_traffic("in_apply", "in_apply_kernel")
    return {"bound": "hbm", "kernel": "in_apply_kernel (16,128,128,64) fp32", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 8e12, 4),
            "us_per_launch": round(ms * 1e3, 2), "mbytes_per_launch": round(nbytes / 1e6, 1),
            "traffic": traffic, "traffic_source": traffic_src}


def inference_rate(solver, dev, iters=5):
    """BASELINE.json configs[3]-style inference: G in eval mode + difference map + threshold at
    B=64 (generate_masks' inner loop), inputs resident in HBM."""
    from fcd_amd.synth import synth_batch
    batch, _ = synth_batch(64, BASE["size"], BASE["bands"], seed=7)
    x = batch[0].to(dev)
    was_training = solver.G.training
    solver.G.eval()
    with torch.no_grad():
        solver.difference_and_mask(x, 0.027)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            solver.difference_and_mask(x, 0.027)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    solver.G.train(was_training)
    return {"metric": "generate_masks images/sec (G eval + difference map + threshold, B=64)",
            "value": round(64 * iters / dt, 1), "unit": "images/sec", "ms_per_batch": round(dt / iters * 1e3, 2)}


def inference_scene_rate(solver, dev, iters=3):
    """BASELINE.json configs[3] at its stated size: the mask of a tiled synthetic Landsat-8 scene
    (2000 x 1900 x 10 uint16, nodata border), 128x128 patches, batch 64, G in eval mode -- host array in,
    host raster + mask out: upload, per-tile nodata flags (one copy), tiling + normalisation, G, difference
    map, float64 stitch, nodata zeroing and float64 threshold on the device, download."""
    from fcd_amd.scene import generate_scene_mask
    from fcd_amd.synth import synth_scene
    h, w = 2000, 1900
    scene = synth_scene(h, w, BASE["bands"], seed=4, border=48)
    was_training = solver.G.training
    solver.G.eval()
    generate_scene_mask(scene, solver, threshold=0.02, batch_size=64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _, mask = generate_scene_mask(scene, solver, threshold=0.02, batch_size=64)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    solver.G.train(was_training)
    tiles = -(-h // BASE["size"]) * -(-w // BASE["size"])
    return {"workload": f"generate_masks over a tiled synthetic scene {h}x{w}x{BASE['bands']} uint16, 128x128 patches, batch 64 "
                        "(BASELINE.json configs[3]); host scene in, host float64 raster + uint8 mask out",
            "ms_per_scene": round(dt * 1e3, 2), "tiles": tiles, "tiles_per_sec": round(tiles / dt, 1),
            "mpixel_per_sec": round(h * w / dt / 1e6, 1), "mask_popcount": int(mask.sum())}


def input_pipeline_rate(dev, iters=20):
    """Input pipeline: a training batch gathered from a uint16 patch pool
    resident in HBM (random indices, random flips): 20 B read + 40 B written per pixel."""
    import numpy as np
    from fcd_amd import ops
    n, b = 2048, BASE["batch"]
    rng = np.random.default_rng(5)
    images = torch.from_numpy(rng.integers(0, 65536, size=(n, BASE["size"], BASE["size"], BASE["bands"]),
                                           dtype=np.uint16)).to(dev)
    index = torch.from_numpy(rng.integers(0, n, size=(iters, b))).to(dev)
    aug = torch.from_numpy(np.concatenate([rng.integers(0, 4, size=(iters, b, 1)), np.zeros((iters, b, 2), np.int64)],
                                          axis=2).astype(np.int32)).to(dev)
    ops.patch_batch(images, index[0], aug[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        ops.patch_batch(images, index[i], aug[i])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = b * BASE["size"] * BASE["size"] * BASE["bands"] * 6.0
    return {"bound": "hbm", "kernel": "patch_batch_kernel (16 of 2048 patches, 128x128x10 uint16 -> NCHW fp32, flips)",
            "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(nbytes / (ms * 1e-3) / 8e12, 4), "us_per_launch": round(ms * 1e3, 2),
            "images_per_sec": round(b / (ms * 1e-3), 0)}


def cpu_baseline(timed_steps=3):
    """CPU oracle (port of the reference step: oracle/fcd_oracle.py) on the host cores: full G+D
    steps (D-step + G-step + both Adam updates) at B=16, fp32 -- one warm-up step, then
    `timed_steps` timed ones (about 10-15 s of CPU work in all)."""
    from oracle import fcd_oracle as O
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    gs = O.synth_generator_state(BASE["conv_dim"], BASE["c_dim"], BASE["g_repeat"], BASE["bands"])
    ds = O.synth_discriminator_state(BASE["size"], BASE["conv_dim"], BASE["c_dim"], BASE["d_repeat"], BASE["bands"])
    batch, alpha = O.synth_batch(BASE["batch"], BASE["size"], BASE["bands"])
    dm = {k: torch.zeros_like(v) for k, v in ds.items()}
    dv = {k: torch.zeros_like(v) for k, v in ds.items()}
    gm = {k: torch.zeros_like(v) for k, v in gs.items()}
    gv = {k: torch.zeros_like(v) for k, v in gs.items()}

    def one_step(step):
        ns = {}
        _, _, dg = O.d_step(gs, ds, batch, alpha, new_stats=ns)
        for k, g in dg.items():
            ds[k], dm[k], dv[k] = O.adam_update(ds[k], g, dm[k], dv[k], step)
        _, _, gg = O.g_step(gs, ds, batch, new_stats=ns)
        for k, g in gg.items():
            gs[k], gm[k], gv[k] = O.adam_update(gs[k], g, gm[k], gv[k], step)
        gs.update(ns)

    one_step(1)
    t0 = time.perf_counter()
    for i in range(timed_steps):
        one_step(2 + i)
    dt = (time.perf_counter() - t0) / timed_steps
    # inference leg (fcd_solver.py:802-807, 762): G eval + difference map + fp32 threshold on a batch of 64
    xb, _ = O.synth_batch(64, BASE["size"], BASE["bands"], seed=7)
    with torch.no_grad():
        O.threshold_mask(O.difference_map(gs, xb[0][:8], False), 0.027)          # warm-up
        t1 = time.perf_counter()
        O.threshold_mask(O.difference_map(gs, xb[0], False), 0.027)
        dti = time.perf_counter() - t1
    return {"value": round(BASE["batch"] / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{timed_steps} full G+D steps (D-step+GP, G-step, Adam x2) after 1 warm-up step, B=16, "
                      "128x128x10, fp32, torch CPU kernels",
            "seconds_per_step": round(dt, 2),
            "inference": {"value": round(64 / dti, 2), "unit": "images/sec", "cores": cores, "kind": "port",
                          "sample": "one batch of 64: G eval forward + difference map + fp32 threshold "
                                    "(oracle difference_map / threshold_mask), torch CPU kernels",
                          "seconds_per_batch": round(dti, 2)}}


def extra_leg(res, key, fn):
    """One of the line's extra fields (never `value`): a failure in it is reported in the field and on stderr
    instead of costing the whole line."""
    try:
        res[key] = fn()
    except Exception as e:      # noqa: BLE001 -- the headline fields above are already measured
        import traceback
        traceback.print_exc(file=sys.stderr)
        res[key] = {"error": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("FCD_PRECISION", "fp32x2"), choices=["fp32", "fp32x3", "fp32x2", "fp16", "bf16"],
                    help="fp32x2 (default): fp32-faithful convolutions on the 16-bit matrix pipe: the 3x3 residual-block "
                         "layers take their operands as two fp16 pieces of the power-of-two-scaled fp32 values (three MFMAs "
                         "per product group), every other layer as in fp32x3; fp32x3: every fp32 operand split into three "
                         "bf16 pieces (six MFMAs); both: fp32 accumulate, same parity bars as fp32, error vs fp64 not "
                         "larger; fp32: exact-fp32 MFMA everywhere; bf16: operands rounded to bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("FCD_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend: nccl (= RCCL over xGMI); gloo only to rehearse the "
                         "multi-rank code path on a box with fewer GPUs than ranks")
    ap.add_argument("--quick", action="store_true",
                    help="only the timed steps and the headline fields (no roofline / inference / CPU legs)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by hand: start N ranks (one per GPU) BEFORE this process touches a GPU,
        # relay rank 0's JSON line and leave with the launcher's status.  (The driver starts the ranks itself
        # with `python -m torch.distributed.run ... bench.py --gpus N`; then WORLD_SIZE is set and this is skipped.)
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    if os.environ.get("FCD_HANG_DUMP"):           # debugging aid: stack of every thread after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["FCD_HANG_DUMP"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}",
              file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the FCD hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from fcd_amd import FCDSolver, _lib
    from fcd_amd.parallel import GradSync
    from fcd_amd.synth import synth_batch
    _lib.check(_lib.load().fcd_init(dev_index), "fcd_init")

    torch.manual_seed(8888)
    solver = FCDSolver(None, batch_size=BASE["batch"], precision=args.precision).to(dev)
    if world > 1:
        solver.grad_sync = GradSync()
        solver.grad_sync.broadcast_params([solver.G, solver.D])
    # per-rank batch (weak scaling: 16 images per GPU), resident in HBM before timing
    batch, _ = synth_batch(BASE["batch"], BASE["size"], BASE["bands"], seed=8888 + rank)
    batch = tuple(t.to(dev) for t in batch)

    def step():
        solver.train_step(batch, force_g=True)

    from fcd_amd import ops
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    # timed region: K steps between two synchronisations.  Beside the wall clock: one HIP event per step (GPU
    # time of every single step), the wall time of the enqueue loop alone (host cost), and the engine clock /
    # socket power sampled from sysfs by a side thread -- so a slow line says WHY it is slow.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_ms, gc_pauses = [], []

    def gc_watch(phase, info, _t=[0.0]):          # interpreter garbage collections inside the timed region
        if phase == "start":
            _t[0] = time.perf_counter()
        else:
            gc_pauses.append({"generation": info["generation"], "ms": round((time.perf_counter() - _t[0]) * 1e3, 2),
                              "in_step": len(host_ms)})
    gc.callbacks.append(gc_watch)
    sampler = SysfsSampler(dev).start()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        th = time.perf_counter()
        step()
        marks[i + 1].record()
        host_ms.append((time.perf_counter() - th) * 1e3)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    clocks = sampler.stop()
    gc.callbacks.remove(gc_watch)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    # The dominant kernel's duration inside real training steps: HIP events around each of its
    # launches in two more steps of the same loop.  Kept out of the timed region above because an
    # event pair per launch (108 per step) costs the step ~14 %.
    # Since round 3 the weight gradients run on a second stream (ops.SideWork), so in the step as timed above half of
    # this kernel's launches (the data gradients) share the CUs with a weight gradient and take longer while the step
    # gets shorter.  A kernel's roofline is about the kernel: `roofline.frac` comes from two steps with the second
    # stream off (every launch alone on the device, as in rounds 1-2); the same two steps with it on are reported
    # beside it (`roofline.overlapped`).
    timer, timer_ov = ops.KernelTimer(is_dominant), ops.KernelTimer(is_dominant)
    side = (solver.G.wgrad_side_stream, solver.D.wgrad_side_stream)
    if world == 1:
        solver.G.wgrad_side_stream = solver.D.wgrad_side_stream = False
        ops.KERNEL_TIMER = timer
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        solver.G.wgrad_side_stream, solver.D.wgrad_side_stream = side
        ops.KERNEL_TIMER = timer_ov
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.KERNEL_TIMER = None
    dist_info = {"world_size": 1, "backend": None}
    if dist is not None:
        t = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        early = [r[1].fired_early for _, r in sorted(solver._reducers.items())]
        spans = {"DG"[k]: r[1].describe() for k, r in sorted(solver._reducers.items())}
        # What the all-reduce costs this run (rank 0's view; after the timed region): two short windows of the same
        # loop, gradient spans sent from inside backward (the default) and only after it, with HIP events around every
        # wait for a collective -- `allreduce_exposed_ms` is the part of the gradient exchange that compute did not hide
        overlap = {}
        nw = max(3, min(10, args.steps))
        for mode in ("on", "off"):
            solver.bucket_overlap = mode == "on"
            solver.grad_sync.timing = []
            step()
            torch.cuda.synchronize()
            dist.barrier()
            solver.grad_sync.exposed_ms()
            tw = time.perf_counter()
            for _ in range(nw):
                step()
            torch.cuda.synchronize()
            dist.barrier()
            dtw = time.perf_counter() - tw
            ex = solver.grad_sync.exposed_ms()
            # per step: D's wait and G's wait (one finish each per step)
            overlap[mode] = {"ms_per_step": round(dtw / nw * 1e3, 3),
                             "allreduce_exposed_ms": {k: v["stream_wait_ms"] for k, v in sorted(ex.items())},
                             "allreduce_host_ms": {k: v["host_ms"] for k, v in sorted(ex.items())},
                             "allreduce_mbytes": {k: round(v["bytes"] / 1e6, 1) for k, v in sorted(ex.items())}}
        solver.bucket_overlap = True
        solver.grad_sync.timing = None
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None,
                     "grad_allreduce": "bucketed, issued from inside the last backward pass of each network "
                                       f"(spans of {solver.bucket_mb[0]:g} MB for D, {solver.bucket_mb[1]:g} MB for G; D's update deferred "
                                       "behind the generator forwards of the G step)",
                     "spans_sent_from_inside_backward_last_step": {"D": early[0] if early else 0,
                                                                   "G": early[1] if len(early) > 1 else 0},
                     "spans": spans,
                     "allreduce_exposed_ms": overlap["on"]["allreduce_exposed_ms"],
                     "bucket_overlap": overlap,
                     "ranks_per_visible_gpu": -(-world // max(1, ndev))}

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = BASE["batch"] * world * args.steps / dt
        res = {"metric": "FCD train images/sec (G+D step, 128x128 multispectral)", "value": round(value, 2),
               "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"fp32": "fp32", "bf16": "bf16 operands, fp32 accumulate",
                         "fp16": "scaled fp16 operands (one piece, 11 bits) in the generator, bf16 operands in the "
                                 "discriminator, fp32 accumulate: the reduced-precision variant, never the headline",
                         "fp32x3": "fp32 (conv operands as 3 bf16 pieces on the bf16 MFMA pipe, fp32 accumulate; "
                                   "fp32-faithful: same parity bars as exact fp32)",
                         "fp32x2": "fp32 (conv operands on the 16-bit MFMA pipe, fp32 accumulate: generator layers as 2 fp16 "
                                   "pieces of the power-of-two-scaled values, 3 MFMAs per product group; discriminator as 3 bf16 "
                                   "pieces, 6 MFMAs; fp32-faithful: measured error vs fp64 below the exact-fp32 MFMA chain's, "
                                   "same parity bars)"}[args.precision],
               "data": "synthetic (numpy PCG64 uint16 patches -> [-1,1], nodata rectangle; random-init weights)",
               "config": {"workload": "FCD G+D train step incl. gradient penalty, batch 16/GPU, 128x128x10, "
                                      "6 res blocks, D x6 (BASELINE.json configs[1])",
                          "global_batch": BASE["batch"] * world, "parallelism": f"dp{world}",
                          "precision": args.precision, **dist_info},
               "step_tflops": round(GFLOP_PER_IMG * BASE["batch"] / (ms * 1e-3) / 1e3, 2),
               # why the line reads what it reads: GPU time of every timed step (one HIP event per step), the host's
               # share (wall time of the enqueue loop alone: when it is close to ms_per_step x steps the host, not
               # the GPU, sets the pace), and the card's clock / power during the timed region
               **step_stats(step_ms), "step_ms": [round(v, 2) for v in step_ms],
               "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
               "host_step_ms": [round(v, 2) for v in host_ms], "gc_pauses_in_timed_region": gc_pauses,
               "timed_region": clocks}
        if args.quick:
            print(json.dumps(res), flush=True)
            if dist is not None:
                dist.barrier()
                dist.destroy_process_group()
            return
        if world == 1:
            # the same loop over 200 more steps (~4-5 s): what the card sustains once clocks and power have settled
            n_sus, win = 200, 20
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_sus // win + 1)]
            sampler = SysfsSampler(dev, period=0.1).start()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            ev[0].record()
            for i in range(n_sus):
                step()
                if (i + 1) % win == 0:
                    ev[(i + 1) // win].record()
            torch.cuda.synchronize()
            dts = time.perf_counter() - ts
            sus_clk = sampler.stop()
            res["sustained"] = {"steps": n_sus, "ms_per_step": round(dts / n_sus * 1e3, 3),
                                "value": round(BASE["batch"] * n_sus / dts, 2), "unit": "images/sec",
                                "ms_per_step_by_window_of_20": [round(ev[i].elapsed_time(ev[i + 1]) / win, 2)
                                                                for i in range(n_sus // win)],
                                "sclk_mhz": sus_clk["sclk_mhz"], "socket_watts": sus_clk["socket_watts"]}
        extra_leg(res, "roofline", lambda: dominant_kernel_roofline(
            args.precision, dev, timer.mean_us("dominant") if timer.pairs else None, timer.count("dominant"),
            symbol=timer.total() if timer.pairs else None,
            overlapped_us=timer_ov.mean_us("dominant") if timer_ov.pairs and any(side) else None))
        if world == 1:
            # the same step with the reference's literal call sequence: G(x_real, c_trg) computed in the D step AND
            # again in the G step, and the D(x_reconst_id) forward whose outputs nothing reads (fcd_solver.py:291, 316, 358)
            solver.share_generator_forward = solver.skip_unused_d_forward = False
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dtq = time.perf_counter() - tq
            solver.share_generator_forward = solver.skip_unused_d_forward = True
            res["reference_call_sequence"] = {
                "value": round(BASE["batch"] * args.steps / dtq, 2), "unit": "images/sec",
                "ms_per_step": round(dtq / args.steps * 1e3, 3),
                "note": "`value` shares one generator call between the D step and the G step (the reference computes "
                        "G(x_real, c_trg) twice per step with identical weights and inputs) and leaves out a discriminator "
                        "forward whose outputs the reference discards; same losses, gradients, weights and running "
                        "statistics (tests/test_step_gpu.py::test_shared_generator_forward_equals_reference_sequence). "
                        "This field times the step with both switched off: 13 generator and 15 discriminator passes "
                        "as in BASELINE.md's FLOP convention; `value` executes 12 and 14"}
        if args.precision in ("fp32x3", "fp32x2") and world == 1:
            # the same step with exact-fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32), for reference
            solver.set_precision("fp32")
            for _ in range(20):     # also rides out the clock transient ~1-2 s into a run (tools/sustain_probe.py)
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            solver.set_precision(args.precision)
            res["exact_fp32_mfma"] = {"value": round(BASE["batch"] * args.steps / dt1, 2), "unit": "images/sec",
                                      "ms_per_step": round(dt1 / args.steps * 1e3, 3)}
            if args.precision == "fp32x2":
                # round 1's split scheme everywhere (three bf16 pieces, six MFMAs per product group), for reference
                solver.set_precision("fp32x3")
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t3
                solver.set_precision(args.precision)
                res["fp32x3_variant"] = {"value": round(BASE["batch"] * args.steps / dt3, 2), "unit": "images/sec",
                                         "ms_per_step": round(dt3 / args.steps * 1e3, 3)}
            # the reduced-precision ("16-bit MFMA") variant of BASELINE.json configs[1]: scaled fp16 operands, ONE MFMA
            # per product group in the generator (11 significant bits per operand; bf16 has 8), bf16 operands in the
            # discriminator; an extra field, never `value`.  It replaces round 2's bf16 variant, which kept fp32 operands
            # in LDS and was no faster than fp32x2.
            solver.set_precision("fp16")
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            solver.check_overflow()
            solver.set_precision(args.precision)
            res["fp16_variant"] = {"value": round(BASE["batch"] * args.steps / dt2, 2), "unit": "images/sec",
                                   "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                                   "dtype": "generator: fp16 operands scaled by a power of two per tensor (11 significant bits), one "
                                            "v_mfma_f32_*_f16 per product group; discriminator: operands rounded to bf16; fp32 "
                                            "accumulate, fp32 activations in HBM",
                                   "parity": "per kernel <= 1e-3 of float64 (tests/test_conv_gpu.py, the `fp16` ids); losses within "
                                             "1e-2 of fp64 at full size, gradients inside the envelope of torch's own 16-bit "
                                             "autocast (tests/test_full_gpu.py::test_full_fp16_variant)"}
        extra_leg(res, "roofline_hbm", lambda: hbm_kernel_roofline(dev))
        extra_leg(res, "inference", lambda: inference_rate(solver, dev))
        extra_leg(res, "inference_scene", lambda: inference_scene_rate(solver, dev))
        extra_leg(res, "input_pipeline", lambda: input_pipeline_rate(dev))
        if world == 1 and not args.no_cpu_baseline:
            extra_leg(res, "cpu_baseline", cpu_baseline)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

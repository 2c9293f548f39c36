# The full-size fp32x2 G step's parity record under kernel switches that change nothing but the order of fp32 sums somewhere:
# how far the worst per-tensor ratios (HIP error / reference-fp32 error, norm-wise and element-wise) move.
# usage: bash tools/parity_lottery.sh   (inside gpurun)
cd $GRAFT_REPO_ROOT
for sw in NONE FCD_NO_S2D FCD_NO_PP FCD_NO_WGRAD4 FCD_NO_IN1F FCD_NO_HALO2 FCD_NO_IN1P; do
  env $sw=1 timeout -k 10 200 python -m pytest tests/test_full_gpu.py -q -k "test_full_g_step and fp32x2 and not arena" --timeout 180 --timeout-method thread > /dev/null 2>&1
  python3 - "$sw" <<'PY'
import json, sys
r = json.load(open("gpurun_out/full_parity_g_fp32x2.json"))
worst_n = worst_e = (0, "")
for n, e in r["errors_vs_fp64_oracle"].items():
    for k, v in e.items():
        worst_n = max(worst_n, (v["hip"] / max(1e-4 / 1.25, v["ref_fp32"]), n))
        worst_e = max(worst_e, (v["hip_maxabs"] / max(1e-4 / 2.5, v["ref_fp32_maxabs"]), n))
print(f"{sys.argv[1]:14s} norm-wise worst {worst_n[0]:.2f} ({worst_n[1]}), element-wise worst {worst_e[0]:.2f} ({worst_e[1]})", flush=True)
PY
done

"""A/B of the InstanceNorm kernels between builds of the library (FCD_HIP_LIB per child process, builds alternate):
python3 tools/ab_in.py libfcd_old.so libfcd_hip.so"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, FCD_HIP_LIB=os.path.join(root, "fcd_amd", lib))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "in_probe.py")], env=env, capture_output=True, text=True, timeout=300)
        print(rnd, lib, flush=True)
        print(out.stdout if out.stdout.strip() else out.stderr[-400:], flush=True)

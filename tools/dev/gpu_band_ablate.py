"""Developer tool (GPU box, NHW_DEV build): time of the fused front with every band of k_front_image ended after phase i -- the cumulative cost of the phases.
usage: gpu_band_ablate.py [quality]"""
import os, subprocess, sys
q = sys.argv[1] if len(sys.argv) > 1 else "20"
names = ['full', 'load+colour', 'chroma vertical', 'contrast', 'entry states', 'replay', 'pair rules', 'horizontal']
prev = None
for i, nm in enumerate(names):
    env = dict(os.environ)
    if i: env["NHW_BAND_STOP"] = str(i)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gpu_q_timing.py"), q], env=env, capture_output=True, text=True).stdout
    front = out.split("'front_ms': ")[1].split(",")[0] if "'front_ms': " in out else out[-200:]
    print(f"stop after {nm:16s} front_ms {front}", flush=True)

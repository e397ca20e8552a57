"""Developer tool (GPU box, NHW_DEV build): time of the fused front with every band ended after phase i -- the cumulative cost of the phases."""
import os, subprocess, sys
names = ['full', 'load', 'contrast+entry', 'replay', 'pairs', 'pass1', 'vertical']
for i, nm in enumerate(names):
    env = dict(os.environ)
    if i: env["NHW_BAND_STOP"] = str(i)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gpu_q_timing.py"), "20"], env=env, capture_output=True, text=True).stdout
    front = out.split("'front_ms': ")[1].split(",")[0] if "'front_ms': " in out else out[-200:]
    print(f"stop after {nm:16s} front_ms {front}", flush=True)

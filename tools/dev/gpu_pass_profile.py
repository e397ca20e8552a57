"""Developer tool (GPU box): per-pass wall-clock inside the wave-per-image phases.
Needs a library built with NHW_PROFILE=1 (python -c 'from nhwcodec_amd.build import build; build(True)')."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nhwcodec_amd
import torch

NAMES = ["tag_l2", "dequant_sim p1", "apply_tags", "precompensate", "res4+emit_ll2", "ll_code_luma", "restore", "dequant_sim p0",
         "Y19/Y20", "tag_small_runs", "Y22 classify", "Y23 code", "Y24+Y25 poslists", "Y26", "Y27 clean", "Y28 quantise", "Y29 hq", "Y30/31 scan+rewrite",
         "ll_code_chroma", "packetise", "container", "chroma marks+emit", "quantise_chroma", "pack: hist", "pack: ratio+rank", "pack: count", "pack: write", "chroma marks only", "chroma copy+stage"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
enc = nhwcodec_amd.Encoder(0, max_batch=n)
bgr = enc.synth_device(n, 1234)
enc.encode_device(bgr, q)
torch.cuda.synchronize()
acc = np.zeros(64, np.float64)
pick = [(i * 251 + 17) % n for i in range(32)]
for i in pick:
    buf = np.zeros(64, np.uint64)
    assert enc.lib.nhw_debug_read(enc.h, 55, i, ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(512)) == 0
    acc += buf
acc /= len(pick)
tot = acc.sum()
for k, nm in enumerate(NAMES):
    print(f"{nm:24s} {acc[k] / 100e3:9.3f} ms  {100 * acc[k] / tot:5.1f}%")   # wall_clock64: 100 MHz
for k in range(len(NAMES), 64):                                               # finer stamps inside a pass (Y31: 40 clear, 41 selection, 42 / 43 rewrites 1 / 2, 44 rewrite 3; Y25: 45 the two sweeps over the tag plane, 46 .. 48 the lists' packing)
    if acc[k]: print(f"stamp {k:2d}                 {acc[k] / 100e3:9.3f} ms")
t = enc.timing()
print("timing ms", {f: round(getattr(t, f), 2) for f, _ in t._fields_})

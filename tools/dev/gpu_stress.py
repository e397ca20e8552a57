"""Developer tool (GPU box): encode the same resident batch many times and report images whose output changes
between runs (races), with the workspace buffers that differ.  usage: gpu_stress.py [runs] [batch] [quality]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nhwcodec_amd

Q = 65536
BUFS = {"JPEG": (0, 8 * Q), "PROC": (1, 8 * Q), "CPROC": (5, 2 * Q), "LL1": (6, 2 * Q), "L2SAVE": (7, 2 * Q), "CLL1": (8, Q // 2), "KEEP": (10, 4 * Q), "FIRST": (11, 2 * Q),
        "SCAN": (17, 6 * Q), "LLBYTES": (18, 24832), "LLFULL": (19, 16384), "EXW": (20, 4096), "LLCOMP": (21, 32768), "LLWORD": (22, 16384), "LLMEM": (23, 32768),
        "RES4": (24, 4096), "R1LIST": (30, 4096), "R3LIST": (33, 4096), "R5LIST": (36, 4096), "PACKET": (46, 65536), "BOOK1": (47, 512), "META": (54, 120)}
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
q = int(sys.argv[3]) if len(sys.argv) > 3 else 20
e = nhwcodec_amd.Encoder(0, max_batch=n)
bgr = e.synth_device(n, seed_base=1000)
out = e.alloc_out(n)
idx = torch.arange(nhwcodec_amd.OUT_STRIDE, device="cuda")[None, :]
w = (idx % 251 + 1).to(torch.int64)

def read(img):
    res = {}
    for name, (b, nb) in BUFS.items():
        a = np.zeros(nb, np.uint8)
        rc = e.lib.nhw_debug_read(e.h, b, img, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(nb))
        if rc == 0:
            res[name] = a
    return res

def digests():
    e.encode_device(bgr, q, out)
    torch.cuda.synchronize()
    o, sizes, status = out
    per = torch.empty(n, dtype=torch.int64, device="cuda")
    for a in range(0, n, 512):
        b = min(n, a + 512)
        m = torch.where(idx < sizes[a:b, None].to(torch.int64), o[a:b], torch.zeros_like(o[a:b])).to(torch.int64)
        per[a:b] = (m * w).sum(1) + sizes[a:b].to(torch.int64) * 1000003
    return per.clone(), sizes.clone()

ref, rsz = digests()
events = 0
for k in range(runs):
    d, sz = digests()
    diff = torch.nonzero(d != ref).flatten().tolist()
    if not diff:
        continue
    events += 1
    print(f"run {k}: {len(diff)} images differ from run 0", diff[:12], flush=True)
    dumps = {i: read(i) for i in diff[:3]}
    d2, _ = digests()
    for i, bad in dumps.items():
        if int(d2[i]) != int(ref[i]):
            print(f"  image {i}: differs again, skipping"); continue
        good = read(i)
        for name in BUFS:
            if name in bad and not np.array_equal(bad[name], good[name]):
                at = np.nonzero(bad[name] != good[name])[0]
                print(f"  image {i}: {name} differs at {len(at)} bytes, first {at[:6].tolist()} last {int(at[-1])}  bad {bad[name][at[:4]].tolist()} good {good[name][at[:4]].tolist()}")
print("events:", events, "of", runs)

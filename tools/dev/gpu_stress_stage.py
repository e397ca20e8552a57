"""Developer tool (GPU box): run the pipeline up to each debug stage many times over one resident batch and report
the first stage whose workspace planes are not reproducible.  usage: gpu_stress_stage.py [runs] [batch] [quality] [first] [last]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import nhwcodec_amd

Q = 65536
BUFS = {"JPEG": (0, 8 * Q), "PROC": (1, 8 * Q), "LL1": (6, 2 * Q), "CJPEG": (4, 2 * Q), "CPROC": (5, 2 * Q), "PU": (2, Q), "PV": (3, Q)}
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
q = int(sys.argv[3]) if len(sys.argv) > 3 else 20
first = int(sys.argv[4]) if len(sys.argv) > 4 else 1
last = int(sys.argv[5]) if len(sys.argv) > 5 else 16
e = nhwcodec_amd.Encoder(0, max_batch=n)
bgr = e.synth_device(n, seed_base=1000)
out = e.alloc_out(n)
if os.environ.get("NHW_FORCE_FB"):
    e.lib.nhw_debug_front_fallback.argtypes = [ctypes.c_void_p, ctypes.c_int]
    e.lib.nhw_debug_front_fallback(e.h, 1)
e.lib.nhw_debug_hash.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

def hashes():
    res = {}
    for name, (b, nb) in BUFS.items():
        h = torch.zeros(n, dtype=torch.int64, device="cuda")
        assert e.lib.nhw_debug_hash(e.h, b, nb, n, h.data_ptr(), e._stream()) == 0
        res[name] = h
    torch.cuda.synchronize()
    return res

for stage in range(first, last + 1):
    e.lib.nhw_debug_stop_after(e.h, stage)
    e.encode_device(bgr, q, out)
    ref = hashes()
    bad = 0
    for k in range(runs):
        e.encode_device(bgr, q, out)
        h = hashes()
        for name in BUFS:
            d = torch.nonzero(h[name] != ref[name]).flatten().tolist()
            if d:
                bad += 1
                if bad < 4: print(f"stage {stage} run {k}: {name} differs for {len(d)} images {d[:10]}", flush=True)
    print(f"stage {stage}: {bad} irreproducible plane digests in {runs} runs", flush=True)

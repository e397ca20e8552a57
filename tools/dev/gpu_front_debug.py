"""Developer tool: the front launch group of chosen (image, quality) pairs against the oracle's stage functions, with the positions that differ.
usage: python tools/dev/gpu_front_debug.py q seed [seed ...]   (seed: int = synthetic seed, or class:seed)"""
import ctypes, sys
import numpy as np
import torch
import nhwcodec_amd
from oracle.oraclepy import Oracle
from oracle.harness import class_image

q = int(sys.argv[1])
orc = Oracle()
imgs = []
for a in sys.argv[2:]:
    if ":" in a:
        k, s = a.split(":"); imgs.append(class_image(k, int(s)))
    else:
        imgs.append(orc.synth(int(a)))
e = nhwcodec_amd.Encoder(0, max_batch=len(imgs))
if "--force" in sys.argv: pass
e.lib.nhw_debug_stop_after(e.h, 4 if q < 22 else 3)
e.encode_device(torch.from_numpy(np.stack(imgs)).cuda(), q)
torch.cuda.synchronize()

def rd(buf, i, nbytes):
    out = np.empty(nbytes, np.uint8)
    assert e.lib.nhw_debug_read(e.h, buf, i, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(nbytes)) == 0
    return out.view(np.int16)

def report(name, got, want, shape):
    bad = np.argwhere(got.reshape(shape) != want.reshape(shape))
    if len(bad):
        print(f"  {name}: {len(bad)} differ; first {bad[:8].tolist()} rows {sorted(set(bad[:, 0].tolist()))[:20]} cols {sorted(set(bad[:, 1].tolist()))[:20]}")
        r, c = bad[0]
        print(f"     got {got.reshape(shape)[r, c]} want {want.reshape(shape)[r, c]}")
    else:
        print(f"  {name}: equal")

for i, im in enumerate(imgs):
    print("image", i)
    y, u, v = orc.color(im, q)
    report("U", rd(2, i, 65536).view(np.uint8), u.ravel(), (256, 256))
    report("V", rd(3, i, 65536).view(np.uint8), v.ravel(), (256, 256))
    if q < 22:
        y = orc.prefilter(y, q)
    oj, op, ok = orc.analysis(y, 512, 512, 0, keep=True)
    report("proc (transposed level-1 plane: row = column c, col = ky / 256+ky)", rd(1, i, 8 * 65536), op, (512, 512))
    report("LL", rd(0, i, 8 * 65536).reshape(512, 512)[:256, :256], oj.reshape(512, 512)[:256, :256], (256, 256))
    report("ll1", rd(6, i, 2 * 65536), oj.reshape(512, 512)[:256, :256].ravel(), (256, 256))
    if q >= 22:
        report("keep", rd(10, i, 4 * 65536), ok, (256, 512))
    import os
    if os.environ.get("NHW_FRONT_DUMP") and q < 22:
        got = rd(14, i, 8 * 65536).reshape(512, 512)[1:511]
        report("pre-filtered luma (rows 1..510; +1 for the image row)", got, y.reshape(512, 512)[1:511], (510, 512))

"""Developer tool (GPU box, NHW_DEV build): time of the decoder's final kernel with every band ended after phase i."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import torch, nhwcodec_amd as na
    n = 4096
    enc = na.Encoder(0, n); img = enc.synth_device(n, 7); out, sizes, status = enc.encode_device(img, 20); torch.cuda.synchronize(); enc.close(); del img
    dec = na.Decoder(0, n)
    offs = torch.arange(n, dtype=torch.int64, device="cuda") * na.OUT_STRIDE
    pix = torch.empty((n, 512, 512, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3): dec.decode_device(out, offs, sizes, pix)
    torch.cuda.synchronize()
    t = dec.timing()
    print(f"recon_ms {t.recon_ms:.3f} total {t.total_ms:.3f}")
else:
    for i, nm in enumerate(['full', 'stage A + chroma', 'first direction', 'corr + smoothing', 'second direction', 'colour']):
        env = dict(os.environ)
        if i: env["NHW_FINAL_STOP"] = str(i)
        out = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True).stdout
        print(f"stop after {nm:18s} {out.strip().splitlines()[-1] if out.strip() else '?'}", flush=True)

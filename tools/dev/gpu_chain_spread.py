"""Developer tool (GPU box, NHW_DEV build): how long each picture of a batch spends in k_low_chain (the pair machine of the q <= 16 pre-filter):
percentiles over the batch, and how well a cheap estimate from the picture predicts it.   usage: gpu_chain_spread.py [q ...]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nhwcodec_amd as na

CH_BYTES = ((510 * 255 + 255) // 256) * 256
B_KEEP = 10

def main(qs, n=4096):
    os.environ["NHW_LOW_PARTS"] = "1"
    enc = na.Encoder(0, n)
    img = enc.synth_device(n, 0)
    out = enc.alloc_out(n)
    g = img[:, ::4, ::4, 1].to(torch.float32)                      # a cheap estimate: mean absolute Laplacian of the subsampled green plane
    lap = (4 * g[:, 1:-1, 1:-1] - g[:, :-2, 1:-1] - g[:, 2:, 1:-1] - g[:, 1:-1, :-2] - g[:, 1:-1, 2:]).abs().mean(dim=(1, 2)).cpu().numpy()
    for q in qs:
        for _ in range(2): enc.encode_device(img, q, out)
        torch.cuda.synchronize()
        ticks = np.zeros(n)
        buf = np.zeros(CH_BYTES, np.uint8)
        for i in range(n):
            enc.lib.nhw_debug_read(enc.h, B_KEEP, i, ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(CH_BYTES))
            ticks[i] = buf[-4:].view(np.uint32)[0] * 64.0
        ms = ticks / 2.1e6
        pc = np.percentile(ms, [0, 10, 25, 50, 75, 90, 95, 99, 100])
        print(f"q{q}: picture time in k_low_chain, ms: min {pc[0]:.2f} p10 {pc[1]:.2f} p25 {pc[2]:.2f} median {pc[3]:.2f} p75 {pc[4]:.2f} p90 {pc[5]:.2f} p95 {pc[6]:.2f} p99 {pc[7]:.2f} max {pc[8]:.2f}; mean {ms.mean():.2f}")
        order = np.argsort(ms)
        r = np.corrcoef(np.argsort(np.argsort(lap)), np.argsort(np.argsort(ms)))[0, 1]
        print(f"   rank correlation with the Laplacian estimate {r:.3f}; of the slowest 10 % the estimate's top 10 % holds {np.isin(order[-n // 10:], np.argsort(lap)[-n // 10:]).mean() * 100:.0f} %, its top 25 % {np.isin(order[-n // 10:], np.argsort(lap)[-n // 4:]).mean() * 100:.0f} %")
        halves = [ms[order[:n // 2]].max(), ms[order[n // 2:]].max()]
        print(f"   sorted into halves: the light half's slowest {halves[0]:.2f} ms, the heavy half's {halves[1]:.2f} ms; into quarters: " + ", ".join(f"{ms[order[k * n // 4:(k + 1) * n // 4]].max():.2f}" for k in range(4)))

if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [10, 8, 1])

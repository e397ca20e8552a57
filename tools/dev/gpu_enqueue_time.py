"""Developer tool (GPU box): how long the HOST takes to queue a batch (nhw_enc_batch_device returns when everything is queued), against the batch's time on the device.
usage: gpu_enqueue_time.py [quality]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nhwcodec_amd as na
q = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 4096
enc = na.Encoder(0, n); img = enc.synth_device(n, 0); out = enc.alloc_out(n)
for _ in range(3): enc.encode_device(img, q, out)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): enc.encode_device(img, q, out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"q{q}: host queues a batch in {(t1 - t0) / K * 1e3:.2f} ms; {K} batches done after {(t2 - t0) / K * 1e3:.2f} ms each; the device's own total {enc.timing().total_ms:.2f} ms")
t0 = time.perf_counter()
for _ in range(K):
    enc.encode_device(img, q, out); torch.cuda.synchronize()
print(f"   one at a time (sync behind each): {(time.perf_counter() - t0) / K * 1e3:.2f} ms")

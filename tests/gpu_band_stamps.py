import sys, ctypes, numpy as np
sys.path.insert(0,'/root/repo')
import nhwcodec_amd, torch
n=4096
e=nhwcodec_amd.Encoder(0,max_batch=n)
b=e.synth_device(n,1)
e.lib.nhw_debug_stop_after(e.h,4)
for _ in range(2): e.encode_device(b,20)
torch.cuda.synchronize()
st=np.zeros(16,np.uint64); e.lib.nhw_debug_stamps(ctypes.c_void_p(st.ctypes.data))
names=['load','contrast+entry','replay','pairs','pass1','vertical','LL store']
for i,nm in enumerate(names): print(f"{nm:16s} {(int(st[i+1])-int(st[i]))/100.0:8.2f} us")
print('total', (int(st[7])-int(st[0]))/100.0)
print('big pair groups per image', int(st[15]) / (2.0 * n), 'of', 510 * 64)

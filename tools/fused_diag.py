"""Times d2pc_process_mono_device with the fused persistent kernel (callback_fused=1) against the two launches, 16 x 4K u8;
with D2PC_LIBRARY_VARIANT=<name> a diagnosis build (make variant DEFS=-DD2PC_FUSED_DIAG_NO_REPROJ: filter chunks only,
-DD2PC_FUSED_NO_STAGGER, -DD2PC_FUSED_MCHUNK=n).  GPU box only."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch
def t(fn, iters=5, rounds=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(rounds):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1)/iters*1e3)
    return float(np.median(ts))
torch.cuda.set_stream(torch.cuda.Stream())
s=torch.cuda.current_stream().cuda_stream
w,h,n=3840,2160,16
ctx=d2pc.Context(q=d2pc.make_q())
raw=torch.randint(0,256,(n,h,w),dtype=torch.uint8,device="cuda")
b=DeviceBatch(ctx,n,h,w,dtype=torch.uint8)
for fused in (1,0):
    ctx.set_tuning("callback_fused",fused)
    print(os.environ.get("D2PC_LIBRARY_VARIANT","base"), "fused" if fused else "two launches", round(t(lambda: ctx.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, w,h,w,w*h,n,11,0.125,b.points.data_ptr(),None,b.stride,b.counts.data_ptr(),s)),1), "us", flush=True)
print("median alone", round(t(lambda: ctx.median_roi_device(raw.data_ptr(), w,h,w,w*h,n,b.disp.data_ptr(),w,w*h,11,s)),1))

import os,sys,torch
sys.path[:0]=['.','mp-reid_amd']
from mpreid import ops, synth
f,_=synth.clustered_features(20000,768,3.0,seed=1234)
ft=torch.from_numpy(f).cuda()
buf=torch.empty((20000,20000),dtype=torch.float32,device='cuda')
def t(fn,reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
print("fp16 fast 20k x 20k:", round(t(lambda: ops.euclidean_distance(ft,ft,mode=ops.GEMM_F16_FAST,out=buf)),4),"ms")

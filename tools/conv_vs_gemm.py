import ctypes as C, os, sys, torch
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'mp-reid_amd')]
from mpreid import _lib, ops
L=_lib.load(); dev=_lib.require_gpu(); s=_lib.stream_ptr()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
for (B,H,W,cin,cout) in [(256,16,8,512,2048),(256,16,8,2048,512),(256,16,8,1024,2048),(256,32,16,256,1024),(256,32,16,1024,256),(256,64,32,64,256),(256,64,32,256,64)]:
    M=B*H*W
    act=(torch.rand((B,H,W,cin),device=dev)*2-1).half()
    wgt=((torch.rand((max(cout,128),cin),device=dev)*2-1)*0.05).half()
    bias=torch.randn(max(cout,128),device=dev)
    t_conv=timeit(lambda: ops.conv_f16_nhwc(act, wgt, bias, cout, 1, relu=True))
    out=torch.empty((M,max(cout,128)),device=dev,dtype=torch.float16)
    a2=act.view(M,cin)
    if cout%128==0 and M%128==0 and cin%64==0:
        t_gemm=timeit(lambda: _lib.check(L.mpreid_gemm_f16_nt_ex(C.c_void_p(a2.data_ptr()),C.c_void_p(wgt.data_ptr()),C.c_void_p(out.data_ptr()),C.c_void_p(bias.data_ptr()),M,cout,cin,1,s),"g"))
    else: t_gemm=float('nan')
    fl=2.0*M*cout*cin
    print(f"M={M} cin={cin} cout={cout}: conv {t_conv*1e3:7.1f} us {fl/t_conv/1e9:6.0f} TF | gemm(auto) {t_gemm*1e3:7.1f} us {fl/t_gemm/1e9:6.0f} TF")

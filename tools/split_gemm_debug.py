import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch
from mpreid import _lib
L = _lib.load(); dev = _lib.require_gpu()
def pair(x, scale=1.0):
    x = x.contiguous(); y = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=torch.float16, device=x.device)
    _lib.check(L.mpreid_split_pack_f32(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], float(scale), C.c_void_p(y.data_ptr()), _lib.stream_ptr()), "p")
    return y
for (n, k, epi) in [(768, 768, 11), (768, 3072, 11), (2304, 768, 10), (768, 768, 10), (3072, 768, 12)]:
    gen = torch.Generator(device="cpu").manual_seed(11)
    mb, ms = 16384, 256
    a = (torch.rand((mb, k), generator=gen) * 2 - 1).to(dev)
    w = ((torch.rand((n, k), generator=gen) * 2 - 1) * 0.05).to(dev)
    e = 9 - int(np.floor(np.log2(float(w.abs().max()))))
    a2, w2 = pair(a), pair(w, 2.0 ** e)
    bias = torch.randn(n, generator=gen).to(dev)
    init = torch.randn((mb, n), generator=gen).to(dev)
    if epi == 12:
        ob = torch.zeros((mb, 2 * n), dtype=torch.float16, device=dev); osm = torch.zeros((ms, 2 * n), dtype=torch.float16, device=dev)
    else:
        ob, osm = init.clone(), init[:ms].clone()
    for x, o in ((a2, ob), (a2[:ms].contiguous(), osm)):
        _lib.check(L.mpreid_gemm_f16_split_nt(C.c_void_p(x.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_void_p(o.data_ptr()),
                                              C.c_void_p(bias.data_ptr()), x.shape[0], n, k, float(2.0 ** -e), epi, _lib.stream_ptr()), "g")
    torch.cuda.synchronize()
    ref = a.double() @ w.double().T + bias.double()
    if epi == 11: ref = init.double() + ref
    if epi == 12:
        ref = ref * torch.sigmoid(1.702 * ref)
        gb = ob[:, :n].double() + ob[:, n:].double(); gs = osm[:, :n].double() + osm[:, n:].double()
    else:
        gb, gs = ob.double(), osm.double()
    eb = (gb - ref).abs(); es = (gs - ref[:ms]).abs()
    print(f"N={n} K={k} epi={epi}: big max err {float(eb.max()):.3e} (rel {float((gb-ref).norm()/ref.norm()):.2e})  small max err {float(es.max()):.3e}  equal={torch.equal(ob[:ms], osm)}")
    bad = (eb > 1e-4).nonzero()
    if len(bad):
        r = bad[:, 0].cpu().numpy(); c = bad[:, 1].cpu().numpy()
        print("   bad count", len(bad), "rows%256 hist", np.bincount(r % 256, minlength=256).reshape(16, 16).sum(1), " cols%64 hist", np.bincount(c % 64, minlength=64).reshape(4,16).sum(1), "first", bad[:5].tolist())
        i, j = int(bad[0, 0]), int(bad[0, 1])
        print("   got", float(gb[i, j]), "ref", float(ref[i, j]), "init", float(init[i, j]) if epi != 12 else None, "bias", float(bias[j]))
# ---- epi 11 forensic: which operand is wrong in the big kernel?
n, k = 768, 768
gen = torch.Generator(device="cpu").manual_seed(11)
mb = 16384
a = (torch.rand((mb, k), generator=gen) * 2 - 1).to(dev)
w = ((torch.rand((n, k), generator=gen) * 2 - 1) * 0.05).to(dev)
e = 9 - int(np.floor(np.log2(float(w.abs().max()))))
a2, w2 = pair(a), pair(w, 2.0 ** e)
for case in ("x=0,bias=0", "x=rowcol,bias=0", "x=0,bias=col"):
    bias = torch.zeros(n, device=dev)
    init = torch.zeros((mb, n), device=dev)
    if case == "x=rowcol,bias=0":
        init = (torch.arange(mb, device=dev)[:, None] * 1000.0 + torch.arange(n, device=dev)[None, :]).float()
    if case == "x=0,bias=col":
        bias = torch.arange(n, device=dev).float() + 1000
    ob = init.clone()
    _lib.check(L.mpreid_gemm_f16_split_nt(C.c_void_p(a2.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_void_p(ob.data_ptr()),
                                          C.c_void_p(bias.data_ptr()), mb, n, k, float(2.0 ** -e), 11, _lib.stream_ptr()), "g")
    torch.cuda.synchronize()
    acc = a.double() @ w.double().T
    ref = init.double() + acc + bias.double()
    d = (ob.double() - ref)
    bad = (d.abs() > 1e-3).nonzero()
    print(case, "bad", len(bad))
    for t in bad[:6].tolist():
        i, j = t
        print("   at", (i, j), "got-acc =", float(ob[i, j].double() - acc[i, j]), " expected x+bias =", float(init[i, j] + bias[j]), " got-x-bias =", float(ob[i, j].double() - init[i, j] - bias[j]), "acc", float(acc[i, j]))

import sys, os, numpy as np, torch
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'mp-reid_amd')]
from mpreid import ops, synth
from oracle import oracle as orc
n,nq,d,k1,k2=1200,1199,64,50,15
f,_=synth.clustered_features(n,d,1.0,seed=n+k1,per_id=20)
want,orank,ovc,ovq=orc.re_ranking(f[:nq],f[nq:],k1,k2,0.3,debug=True)
got,st,rank,vc,vq=ops.re_ranking(torch.from_numpy(f[:nq]),torch.from_numpy(f[nq:]),k1,k2,0.3,debug=True)
got=got.cpu().numpy()
print("rank eq",np.array_equal(rank,orank),"vc eq",np.array_equal(vc,ovc),"vq eq",np.array_equal(vq,ovq), "out eq", np.array_equal(got,want), "ndiff", (got!=want).sum(), st['vqe_nnz'], int(ovq.sum()))
bad=np.nonzero((got!=want)[:,0])[0]; print(bad[:20], len(bad))

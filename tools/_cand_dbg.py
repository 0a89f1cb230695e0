import sys, os, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, os.path.join(R,"mp-reid_amd")]
from mpreid import ops, synth
f,_=synth.clustered_features(20000,768,3.0,seed=1234)
ft=torch.from_numpy(f).cuda()
for _ in range(4):
    try:
        ops.re_ranking(ft[:4000],ft[4000:],50,15,0.3,algo=ops.RERANK_SPARSE)
    except Exception as e:
        pass
torch.cuda.synchronize()

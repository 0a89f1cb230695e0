import sys
sys.path[:0]=['/root/repo','/root/repo/mp-reid_amd','/root/repo/tests']
import numpy as np, torch
from mpreid import ops, synth
from oracle import oracle as orc
from conftest import map_noise_envelope
torch.set_num_threads(32)
for n_ids, per_id in ((128,4),(128,8)):
    x, pid = synth.identity_images(n_ids, per_id, 0.5, grid=(8, 4))
    sd = orc.rn50_calibrate_bn(synth.rn50_state_dict(synth.RN50, seed=11), synth.RN50, x[:64], selective=1.0)
    f_or = np.concatenate([orc.rn50_features(sd, synth.RN50, x[s:s + 32]) for s in range(0, len(pid), 32)])
    f64 = np.concatenate([orc.rn50_features(sd, synth.RN50, x[s:s + 32], dtype="float64") for s in range(0, len(pid), 32)])
    print("oracle fp32 vs fp64 graph: rel", np.linalg.norm(f_or-f64)/np.linalg.norm(f64))
    n=len(pid); nq=n//4
    fo = orc.l2_normalize(f_or)
    fo64 = orc.l2_normalize(f64.astype(np.float32))
    for prec in ("split","fp32"):
        enc = ops.Rn50Encoder(synth.RN50, sd, (256, 128), precision=prec)
        f = torch.cat([enc(torch.from_numpy(x[s:s + 256])) for s in range(0, n, 256)])
        rel=float(np.linalg.norm(f.cpu().numpy() - f_or) / np.linalg.norm(f_or))
        rel64=float(np.linalg.norm(f.cpu().numpy() - f64) / np.linalg.norm(f64))
        fn = ops.l2_normalize(f)
        for rr in (False, True):
            d_or = orc.re_ranking(fo[:nq], fo[nq:], 20, 6, 0.3) if rr else orc.euclidean_distance(fo[:nq], fo[nq:])
            cmc_o, map_o = orc.eval_func(d_or, pid[:nq], pid[nq:])
            d64 = orc.re_ranking(fo64[:nq], fo64[nq:], 20, 6, 0.3) if rr else orc.euclidean_distance(fo64[:nq], fo64[nq:])
            cmc64, map64 = orc.eval_func(d64, pid[:nq], pid[nq:])
            d = ops.re_ranking(fn[:nq], fn[nq:], 20, 6, 0.3)[0] if rr else ops.euclidean_distance(fn[:nq], fn[nq:])
            cmc, mAP = orc.eval_func(d.cpu().numpy(), pid[:nq], pid[nq:])
            print(n, prec, "rr",rr, "rel %.2e (vs fp64 %.2e) mAP_o %.5f dmAP %.2e dR1 %.2e | oracle32 vs oracle64: dmAP %.2e | hip vs oracle64 %.2e"%(rel,rel64,map_o,abs(mAP-map_o),abs(cmc[0]-cmc_o[0]),abs(map_o-map64),abs(mAP-map64)), flush=True)
        del enc
    for rr in (False, True):
        for relp in (2e-6, 7e-6):
            env = map_noise_envelope(orc, f_or, relp, pid, nq, rr, 20, 6, seeds=4)
            print(n, "envelope rr",rr,"rel",relp, env, flush=True)

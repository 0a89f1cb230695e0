#!/usr/bin/env python3
"""Would FEWER fp16 products per multiply-add hold north_star's 1e-4 mAP bound?  

The `split` encoder mode carries every operand of every linear layer as an fp16 pair x = hi + lo and runs three products
(hi.hi' + lo.hi' + hi.lo') per multiply-add on the fp16 matrix cores: 3x the matrix work of the fp16 mode -- the whole gap
between roofline.frac 0.16 and 0.48.  This tool measures, on the 'spread' set of tests/test_gpu_map_parity.py (128 ids x 8
images, weights ~ N(0, 0.05^2)), what the features and the metrics would be with cheaper mixes.  It is a NUMERICAL EMULATION:
the ViT-B/16 graph of oracle/oracle.py evaluated with torch in float64 on the GPU, the operands of the chosen linear layers
rounded the way the kernels round them (hi = fp16(x), lo = fp16(x - hi); weights scaled by a power of two per matrix so that the
largest entry sits in [2^9, 2^10)), the products accumulated exactly (float64) -- i.e. ONLY the operand-rounding error of each
mix, without the fp32 accumulation noise (~1e-6) every real kernel adds on top.  Everything else (LayerNorm, softmax, GELU, the
attention products, residual stream) is exact.  The reference is the same graph with unrounded operands.

    python tools/precision_mix_study.py            # prints a markdown table
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mpreid import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = "cuda"
F64 = torch.float64


def pair(x):
    hi = x.to(torch.float16).to(F64)
    lo = (x - hi).to(torch.float16).to(F64)
    return hi, lo


def wpair(w):
    e = 9 - int(np.floor(np.log2(float(w.abs().max()))))      # largest entry into [2^9, 2^10)
    sc = 2.0 ** e
    hi, lo = pair(w * sc)
    return hi / sc, lo / sc


def mm(a, w, mode):
    """a [..., K] @ w[N, K]^T with the operands rounded per `mode`:
    exact | 3 (hi.hi' + lo.hi' + hi.lo') | a2 (activations split, weights single) | w2 (weights split, activations single) | 1 (fp16)"""
    if mode == "exact":
        return a @ w.t()
    ah, al = pair(a)
    wh, wl = wpair(w)
    out = ah @ wh.t()
    if mode in ("f8c", "f8a", "f8w"):   # cross terms with fp8 (OCP e4m3) operands: lo scaled by a power of two into e4m3's range, hi rounded to 4 bits
        def f8(x, per_row=True):
            amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300) if per_row else x.abs().max()
            sc = torch.exp2(torch.floor(torch.log2(256.0 / amax)))          # largest entry into [128, 256) <= 448
            return (x * sc).to(torch.float32).to(torch.float8_e4m3fn).to(F64) / sc
        if mode == "f8c":
            return out + f8(al) @ f8(wh).t() + f8(ah) @ f8(wl).t()
    if mode in ("m6w", "m6c", "m6s", "m8c"):
        # round 6: the cross terms on the BLOCK-SCALED matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4):
        # MX operands = 32 consecutive k share one power-of-two scale (E8M0), elements fp6 e2m3 (4 significant bits, 4x the fp16
        # rate on the data sheet) or fp8 e4m3 (2x).  Two SLICES per operand (x ~ s1 + s2, s2 = MX(x - s1) with its own block
        # scales: ~8 significant bits relative to the block maximum) and the three products s1.t1 + s2.t1 + s1.t2 per cross term.
        def mx(x, fmt):
            K = x.shape[-1]
            xb = x.reshape(*x.shape[:-1], K // 32, 32)
            amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
            emax = 2 if fmt == "fp6" else 7
            sc = torch.exp2(torch.floor(torch.log2(amax)) - emax)
            y = xb / sc
            if fmt == "fp6":    # e2m3: steps 1/8 below 2, 1/4 in [2, 4), 1/2 in [4, 7.5]; saturates at 7.5
                mag = y.abs().clamp_max(7.5)
                step = torch.where(mag < 2, 0.125, torch.where(mag < 4, 0.25, 0.5)).to(F64)
                q = torch.sign(y) * torch.round(mag / step) * step
            else:
                q = y.to(torch.float32).to(torch.float8_e4m3fn).to(F64)
            return (q * sc).reshape(x.shape)

        def two(x, fmt):
            s1 = mx(x, fmt)
            return s1, mx(x - s1, fmt)

        def cross2(u, v, fmt):      # u . v^T with both operands as two MX slices, the s2.t2 product dropped
            u1, u2 = two(u, fmt)
            v1, v2 = two(v, fmt)
            return u1 @ v1.t() + u2 @ v1.t() + u1 @ v2.t()
        if mode == "m6w":    # hi.lo' on two fp6 slices, lo.hi' on fp16
            return out + al @ wh.t() + cross2(ah, wl, "fp6")
        if mode == "m6c":    # both cross terms on two fp6 slices
            return out + cross2(al, wh, "fp6") + cross2(ah, wl, "fp6")
        if mode == "m6s":    # both cross terms on ONE fp6 slice per operand
            return out + mx(al, "fp6") @ mx(wh, "fp6").t() + mx(ah, "fp6") @ mx(wl, "fp6").t()
        return out + mx(al, "fp8") @ mx(wh, "fp8").t() + mx(ah, "fp8") @ mx(wl, "fp8").t()   # m8c: one MX fp8 slice
    if mode == "f8a":   # lo.hi' on fp8, hi.lo' on fp16
        return out + f8(al) @ f8(wh).t() + ah @ wl.t()
    if mode == "f8w":   # hi.lo' on fp8, lo.hi' on fp16
        return out + al @ wh.t() + f8(ah) @ f8(wl).t()
    if mode in ("3", "a2"):
        out = out + al @ wh.t()
    if mode in ("3", "w2"):
        out = out + ah @ wl.t()
    return out


def vit(sd, cfg, imgs, modes):
    """modes: dict layer kind -> mode for 'patch', 'qkv', 'out', 'fc1', 'fc2'"""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV, F64)  # noqa: E731
    x = T(imgs)
    B = x.shape[0]
    w, heads, p, s = cfg["width"], cfg["heads"], cfg["patch"], cfg["stride"]
    dh = w // heads
    cols = F.unfold(x, kernel_size=p, stride=s)
    tok = mm(cols.transpose(1, 2), T(sd["conv1.weight"]).reshape(w, -1), modes["patch"])
    cls = T(sd["class_embedding"]).expand(B, 1, w).clone()
    x = torch.cat([cls, tok], dim=1) + T(sd["positional_embedding"])
    x = F.layer_norm(x, (w,), T(sd["ln_pre.weight"]), T(sd["ln_pre.bias"]), 1e-5)
    L = x.shape[1]
    for i in range(cfg["layers"]):
        b = f"transformer.resblocks.{i}"
        h = F.layer_norm(x, (w,), T(sd[b + ".ln_1.weight"]), T(sd[b + ".ln_1.bias"]), 1e-5)
        qkv = mm(h, T(sd[b + ".attn.in_proj_weight"]), modes["qkv"]) + T(sd[b + ".attn.in_proj_bias"])
        q, k, v = qkv.split(w, dim=2)
        q = q.reshape(B, L, heads, dh).transpose(1, 2)
        k = k.reshape(B, L, heads, dh).transpose(1, 2)
        v = v.reshape(B, L, heads, dh).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(2, 3)) * (dh ** -0.5), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, L, w)
        x = x + mm(a, T(sd[b + ".attn.out_proj.weight"]), modes["out"]) + T(sd[b + ".attn.out_proj.bias"])
        h = F.layer_norm(x, (w,), T(sd[b + ".ln_2.weight"]), T(sd[b + ".ln_2.bias"]), 1e-5)
        h = mm(h, T(sd[b + ".mlp.c_fc.weight"]), modes["fc1"]) + T(sd[b + ".mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)
        x = x + mm(h, T(sd[b + ".mlp.c_proj.weight"]), modes["fc2"]) + T(sd[b + ".mlp.c_proj.bias"])
    x12 = F.layer_norm(x[:, 0], (w,), T(sd["ln_post.weight"]), T(sd["ln_post.bias"]), 1e-5)
    return torch.cat([x12, x12 @ T(sd["proj"])], dim=1)


def features(sd, cfg, x, modes, bs=64):
    with torch.no_grad():
        return torch.cat([vit(sd, cfg, x[s:s + bs], modes) for s in range(0, x.shape[0], bs)]).cpu().numpy()


def metrics(f, pid, nq):
    fo = orc.l2_normalize(f.astype(np.float32))
    out = []
    for rr in (False, True):
        d = orc.re_ranking(fo[:nq], fo[nq:], 50, 15, 0.3) if rr else orc.euclidean_distance(fo[:nq], fo[nq:])
        cmc, mAP = orc.eval_func(d, pid[:nq], pid[nq:])
        out.append((float(mAP), float(cmc[0])))
    return out


def main():
    cfg = synth.VIT_B16
    x, pid = synth.identity_images(128, 8, 0.4)
    sd = synth.vit_state_dict(cfg, seed=7, std=0.05)
    nq = len(pid) // 5
    kinds = ("patch", "qkv", "out", "fc1", "fc2")
    allm = lambda m: {k: m for k in kinds}  # noqa: E731
    only = os.environ.get("MIX_ONLY")   # e.g. MIX_ONLY="(ix),(x),(xi),(xii)": just those rows (+ the first)
    mixes = [
        ("three products everywhere (the `split` mode)", allm("3"), 3.0),
        ("(i) activations split, weights single fp16: hi.hi' + lo.hi'", allm("a2"), 2.0),
        ("(ii) weights split, activations single fp16: hi.hi' + hi.lo'", allm("w2"), 2.0),
        ("(iii-a) three on QKV + out-proj, two (activations split) on FC1 / FC2", dict(allm("3"), fc1="a2", fc2="a2"), 2.33),
        ("(iii-b) three on QKV + out-proj, two (weights split) on FC1 / FC2", dict(allm("3"), fc1="w2", fc2="w2"), 2.33),
        ("(iv) two (activations split) on QKV only, three elsewhere", dict(allm("3"), qkv="a2"), 2.78),
        ("(v) two (activations split) on out-proj only, three elsewhere", dict(allm("3"), out="a2"), 2.93),
        ("(vi) hi.hi' on fp16, BOTH cross terms on the fp8 matrix cores (e4m3 operands, per-row power-of-two scales; fp8 = 2x the fp16 rate)", allm("f8c"), 2.0),
        ("(vii) as (vi) but only lo.hi' (activations' lo) on fp8, hi.lo' on fp16", allm("f8a"), 2.5),
        ("(viii) as (vi) but only hi.lo' (weights' lo) on fp8, lo.hi' on fp16", allm("f8w"), 2.5),
        ("(ix) both cross terms on ONE block-scaled (MX, 32 k per scale) fp8 e4m3 slice per operand", allm("m8c"), 2.0),
        ("(x) both cross terms on ONE MX fp6 e2m3 slice per operand (fp6 = 4x the fp16 rate on the data sheet)", allm("m6s"), 1.5),
        ("(xi) hi.lo' on TWO MX fp6 slices per operand (3 fp6 products), lo.hi' on fp16", allm("m6w"), 2.75),
        ("(xii) both cross terms on TWO MX fp6 slices per operand (3 + 3 fp6 products)", allm("m6c"), 2.5),
        ("(xii-b) as (xii) on FC1 / FC2 only, three fp16 products elsewhere", dict(allm("3"), fc1="m6c", fc2="m6c"), 2.67),
        ("one product everywhere (the `fp16` mode)", allm("1"), 1.0),
    ]
    if only:
        mixes = [m for i, m in enumerate(mixes) if i == 0 or any(m[0].startswith(t + " ") for t in only.split(","))]
    f_ref = features(sd, cfg, x, allm("exact"))
    m_ref = metrics(f_ref, pid, nq)
    print(f"spread set: {len(pid)} images, {nq} queries; exact graph: mAP {m_ref[0][0]:.5f} / re-ranked {m_ref[1][0]:.5f}, "
          f"Rank-1 {m_ref[0][1]:.4f} / {m_ref[1][1]:.4f}\n")
    print("| operand mix of the linear layers | fp16 products per multiply-add (FLOP-weighted) | feature rel-L2 | |ΔmAP| Euclid | |ΔmAP| re-ranked | "
          "Rank-1 queries changed (Euclid / re-ranked) | holds 2e-5 and 1e-4? |")
    print("|---|---|---|---|---|---|---|")
    for name, modes, cost in mixes:
        f = features(sd, cfg, x, modes)
        rel = float(np.linalg.norm(f - f_ref) / np.linalg.norm(f_ref))
        m = metrics(f, pid, nq)
        dm = [abs(m[i][0] - m_ref[i][0]) for i in (0, 1)]
        dq = [round(abs(m[i][1] - m_ref[i][1]) * nq) for i in (0, 1)]
        ok = rel <= 2e-5 and max(dm) <= 1e-4 and max(dq) == 0
        print(f"| {name} | {cost:.2f} | {rel:.1e} | {dm[0]:.1e} | {dm[1]:.1e} | {dq[0]} / {dq[1]} | {'yes' if ok else 'NO'} |", flush=True)


if __name__ == "__main__":
    main()

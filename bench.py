#!/usr/bin/env python3
"""bench.py — MP-ReID evaluation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload market|synth|msmt17] [--rerank]

Launch: with --gpus N > 1 and no WORLD_SIZE in the environment this process stays GPU-free, starts N fresh child
processes (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) and relays
rank 0's JSON line; under `python -m torch.distributed.run` the ranks are the launcher's.  It never re-execs.

Workloads (BASELINE.json configs):
  market  (default; configs[1], with --rerank the configs[2] stand-in)  THE DROP-IN PATH: one step is one call of
          ``processor.do_inference(cfg, model, val_loader, num_query)`` on a ``make_model(cfg, ...)`` model -- the
          reference's test.py:58-62 -- over a query-then-gallery loader of 64-image fp32 batches that are resident in HBM:
          ViT-B/16 encode of 3368 query + 15913 gallery images PER GPU SHARD (grouped, two streams: mpreid/pipeline.py)
          -> R1_mAP_eval.compute(): L2-normalise -> all-gather of the query features -> [3368, 15913] exact distance
          block per GPU -> CMC / mAP ranking on the GPU -> matrix and features handed to the host.  Weak scaling
          (per-GPU gallery fixed).  extras: the same call fed by a HOST fp32 loader (the reference's loader type) and by a
          RawImageBatch loader (decoded uint8 images), PCIe inclusive.
  synth   (configs[3])  ONE 20 000-query x 80 000-gallery x 768 problem, gallery rows sharded over the N GPUs,
          all-gather of the query features, per-shard [20000, 80000/N] distance blocks; --rerank adds the row-sharded
          k-reciprocal re-ranking of the N = 100 000 problem.  Strong scaling (total work fixed).
  msmt17  (configs[4])  MSMT17 shape: encode 11 659 + 82 161 images sharded over the GPUs, distance blocks at
          D = 1280, row-sharded re-ranking, fp16-MFMA distance mode checked against the exact fp32 mode.  Strong.

One "step" = one full pass over the workload; inputs are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line with `roofline` (dominant kernel of the step, hipEvents on the launch stream), `roofline_all` (every stage
of the metric incl. the 20k x 20k feat-GEMM and the re-rank stages, N = 1 only) and `cpu_baseline` (the CPU oracle
on a bounded sample, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mp-reid_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

NQ, NG, H, W = 3368, 15913, 256, 128          # Market-1501 test split (datasets/market1501.py:24), vit_base.yml
MSMT_NQ, MSMT_NG = 11659, 82161                 # datasets/msmt17.py:21 (SURVEY.md §8d shape D)
SYN_NQ, SYN_NG, SYN_D = 20000, 80000, 768       # BASELINE configs[3] (SURVEY.md §8d shape C)
GFLOP_PER_IMG = 21.12                           # last block CLS-only (22.68 if computed for all tokens)
PEAK_F16_TFLOPS = 2500.0                        # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
PEAK_F32_TFLOPS = 157.3                         # fp32 MFMA
PEAK_HBM_GBS = 8000.0                           # HBM3E
METRIC = "gallery images/s encode + distmat+rerank ms, 20k×20k; mAP/Rank-1 parity"


# ----------------------------------------------------------------------------------------------------------------
# the ONE stdout line (<= 4 KB) and the extras file
# ----------------------------------------------------------------------------------------------------------------
EXTRAS_FILE = "bench_extras.json"               # written to the current directory by rank 0
MAX_LINE_BYTES = 4096
MAX_STR = 110                                   # (round 5's driver record clipped strings near 120 characters)
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "metric_20k", "roofline", "cpu_baseline", "rccl_ranks", "all_gather",
             "extras_file")
# the other two thirds of BASELINE.json's metric ("distmat + rerank ms, 20k x 20k; mAP / Rank-1 parity"): bare numbers
METRIC_20K_KEYS = ("distmat_exact_ms", "distmat_split3_ms", "distmat_split3_frac_exec", "distmat_f16_ms", "distmat_f16_frac",
                   "distmat_f16_2t_ms", "distmat_f16_2t_frac", "rerank_ms", "rerank_cpu_port_ms", "rerank_vs_cpu_x",
                   "rerank_vs_ref265s_x", "dist_err_exact", "dist_err_split3", "dist_err_f16", "feat_rel_l2_split",
                   "dmap_split", "dr1_split", "dmap_rr_split", "dr1_rr_split", "parity_images")
CONFIG_KEYS = ("workload", "images_per_step", "encoder_batch", "encoder_precision", "encoder_streams", "sharding",
               "rerank", "distance_mode")
ROOFLINE_KEYS = ("bound", "kernel", "shape", "achieved", "peak", "unit", "frac", "achieved_executed", "frac_executed",
                 "avg_launch_ms", "launches", "algorithmic_flop", "algorithmic_bytes", "traffic", "traffic_replayed",
                 "traffic_source", "measured")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample")
ALL_GATHER_KEYS = ("calls_per_step", "bytes_per_step", "ms_per_step", "gb_per_s")


def _clip(v):
    """strings of the line are short by contract (<= MAX_STR characters); anything longer belongs in the extras file"""
    if isinstance(v, str) and len(v) > MAX_STR:
        return v[:MAX_STR - 3] + "..."
    return v


def _pick(d, keys):
    return None if d is None else {k: _clip(d[k]) for k in keys if k in d}


def format_line(res, extras_file=EXTRAS_FILE):
    """(line, extras): `line` is the ONE JSON line of the driver contract -- the contract's keys, `roofline` of the
    dominant kernel only, `cpu_baseline`, the RCCL figures when N > 1 -- at most MAX_LINE_BYTES bytes, every string at
    most MAX_STR characters; `extras` is everything else (`roofline_all`, `extras`, `gemm_classes`, `drop_in`,
    `reference_cpu`, full-length notes) for the extras file.  tests/test_bench_line.py holds the format."""
    line = {k: _clip(res[k]) for k in LINE_KEYS if k in res and k not in ("config", "roofline", "cpu_baseline", "all_gather", "metric_20k")}
    line["config"] = _pick(res.get("config") or {}, CONFIG_KEYS)
    line["roofline"] = _pick(res.get("roofline"), ROOFLINE_KEYS)
    if res.get("metric_20k"):
        line["metric_20k"] = _pick(res["metric_20k"], METRIC_20K_KEYS)
    line["cpu_baseline"] = _pick(res.get("cpu_baseline"), CPU_BASELINE_KEYS)
    if res.get("all_gather"):
        line["all_gather"] = _pick(res["all_gather"], ALL_GATHER_KEYS)
    line["extras_file"] = extras_file
    ordered = {k: line[k] for k in LINE_KEYS if k in line}
    text = json.dumps(ordered)
    for drop in (("roofline", "measured"), ("roofline", "traffic_source"), ("cpu_baseline", "sample"), ("config", "sharding")):
        if len(text.encode()) <= MAX_LINE_BYTES:
            break
        if ordered.get(drop[0]) and drop[1] in ordered[drop[0]]:     # never reached with the strings below; a guard, not a plan
            del ordered[drop[0]][drop[1]]
            text = json.dumps(ordered)
    assert len(text.encode()) <= MAX_LINE_BYTES, len(text)
    extras = {k: v for k, v in res.items() if k not in ("metric",)}
    return text, extras


def emit(res, extras_file=EXTRAS_FILE):
    """rank 0: write the extras file (never fatal), then print the line -- the last thing on stdout"""
    text, extras = format_line(res, extras_file)
    try:
        with open(extras_file, "w") as fh:
            json.dump(extras, fh, indent=1)
    except OSError as e:
        print(f"bench.py: could not write {extras_file}: {e}", file=sys.stderr)
    sys.stdout.flush()
    print(text, flush=True)


def visible_gpus():
    """number of GPUs this process may use, WITHOUT initialising HIP (torch.cuda.device_count() reads the topology only
    on this image; the parent of the self-launch must stay GPU-free)"""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def check_gpu_count(n, quiet=False):
    """--gpus N with fewer than N devices visible: exit 2 with a clear message instead of letting ranks share a device
    silently (a 'scaling curve' on one GPU).  MPREID_DIST_BACKEND=gloo (the tests' one-GPU staging of several ranks)
    and MPREID_ALLOW_SHARED_GPU=1 opt out."""
    if os.environ.get("MPREID_DIST_BACKEND", "nccl") == "gloo" or os.environ.get("MPREID_ALLOW_SHARED_GPU") == "1":
        return
    have = visible_gpus()
    if n > have:
        if quiet:
            sys.exit(2)
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = "
              f"{os.environ.get('HIP_VISIBLE_DEVICES')!r} / {os.environ.get('ROCR_VISIBLE_DEVICES')!r}); refusing to run "
              f"several RCCL ranks on one device", file=sys.stderr)
        sys.exit(2)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("market", "synth", "msmt17"), default="market")
    ap.add_argument("--batch", type=int, default=508,
                    help="images per encoder call; 508*129 tokens = 256 row tiles of 256 rows = whole waves of "
                         "tiles on 256 CUs (reference yml: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child runs of the dominant GEMM class, "
                         "~40 s): replay the committed figure of profiles/ instead, labelled traffic_replayed")
    ap.add_argument("--cpu-images", type=int, default=192,
                    help="images of the cpu_baseline's oracle encode (~25 images/s on 32 host threads: ~8 s; round 4 extrapolated from 24)")
    ap.add_argument("--small", action="store_true", help="debug: 1/16 of the workload")
    ap.add_argument("--rerank", action="store_true",
                    help="also run k-reciprocal re-ranking (k1=50, k2=15, lambda=0.3) in every step "
                         "(rows sharded over the ranks when N > 1)")
    ap.add_argument("--rerank-algo", choices=("exact", "split3"), default="exact",
                    help="re-ranking: bit-parity mode (default) or RERANK_SPARSE_SPLIT3 (the blend term's distance rows from "
                         "the fp16 matrix cores: ranks and sparse vectors identical, outputs within 1e-6)")
    ap.add_argument("--dist-mode", choices=("exact", "f16", "split3"), default="exact",
                    help="arithmetic of the distance GEMM blocks: exact fp32 MFMA (parity mode), one-pass fp16 "
                         "(speed mode, |err| ~1e-4), 3-term fp16 split (|err| <= 1e-6)")
    ap.add_argument("--encoder-precision", choices=("split", "fp16", "fp32"), default="split",
                    help="arithmetic of the encoder's linear layers / attention products: split (default) = fp16 operand "
                         "PAIRS hi + lo, three products per multiply-add on the fp16 matrix cores, fp32-grade features: the "
                         "mode that meets the 1e-4 mAP bound; fp16 = single fp16 operands (fastest, ~4e-4 feature error, "
                         "does not meet the bound on hard data); fp32 = exact fp32 matrix instruction")
    ap.add_argument("--host-concat", choices=("auto", "on", "off"), default="auto",
                    help="gather the per-shard distance blocks into one pinned host matrix on rank 0 inside every step "
                         "(north_star); auto = on for the market workload (214 MB per GPU shard), off for synth / msmt17 "
                         "(6.4 / 3.8 GB matrices: the step would measure PCIe)")
    ap.add_argument("--group", type=int, default=0,
                    help="images per encoder call of the do_inference pipeline (market workload; 0 = the model's encode_group: 508 "
                         "images = one sweep of 256-row tiles over 256 CUs)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the encoder batches alternate on (HBM-bound phases of one batch overlap "
                         "MFMA phases of the other)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------
# self-launch (parent process: no torch.cuda call, no HIP library loaded)
# ----------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_children(n, timeout_s=3000):
    """Start n ranks of this script as fresh processes and relay rank 0's JSON line.  The parent never touches the
    GPU, so nothing that has initialised HIP is ever exec-ed or forked.  All children are polled: as soon as one exits
    non-zero (or the overall timeout passes) the others are terminated -- a rank that died after the rendezvous would
    otherwise leave rank 0 inside a collective for ever -- and the parent exits 1."""
    import tempfile
    import threading
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            failed = "a rank exited with a non-zero code"
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed = f"timeout after {timeout_s} s"
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        killer = threading.Timer(10.0, lambda: [p.kill() for p in procs if p.poll() is None])
        killer.start()
        for p in procs:
            p.wait()
        killer.cancel()
    codes = [p.returncode for p in procs]
    out0.seek(0)
    line = None
    for ln in out0.read().splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if failed or any(codes) or line is None:
        print(f"bench.py: child ranks exited with {codes}" + (f" ({failed})" if failed else ""), file=sys.stderr)
        sys.exit(1)
    print(line, flush=True)


# ----------------------------------------------------------------------------------------------------------------
# helpers (rank processes)
# ----------------------------------------------------------------------------------------------------------------
def make_images(n, seed, device, chunk=1024):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, 3, H, W), dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        out[s:e] = torch.randn((e - s, 3, H, W), generator=g, device=device).clamp_(-1.0, 1.0)
    return out


def make_features(n, d, sigma, seed, device, per_id=20):
    """clustered unit-norm features (SURVEY.md §8d recipe) generated on the device: identical on every rank"""
    import torch
    from mpreid import ops
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_ids = max(1, n // per_id)
    cent = torch.randn((n_ids, d), generator=g, device=device)
    pid = torch.randint(0, n_ids, (n,), generator=g, device=device)
    x = cent[pid] + sigma * torch.randn((n, d), generator=g, device=device)
    return ops.l2_normalize(x), pid


def timed_ms(fn, reps, warm=1):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cpu_baseline(n_img):
    """CPU oracle (kind 'port') on a bounded sample of the same workload: all host cores, and one thread.  Returns
    (cpu_baseline dict, parity sample): the encode leg's images are synthetic identity images under the `spread` weights of
    tests/test_gpu_map_parity.py, so the oracle features it has just timed also serve as the checker of parity_deltas()."""
    import numpy as np  # noqa: F401
    import torch
    from mpreid import synth
    from oracle import oracle as orc
    cores = min(os.cpu_count() or 1, 32)   # more threads than this only add contention for these sizes
    torch.set_num_threads(cores)
    per_id = 8 if n_img >= 64 else 4
    n_ids = max(2, n_img // per_id)
    n_img = n_ids * per_id
    sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=0.05)
    imgs, pid = synth.identity_images(n_ids, per_id, 0.4)
    orc.vit_features(sd, synth.VIT_B16, imgs[:2])  # warm
    t0 = time.perf_counter()
    f_or = []
    for s0 in range(0, n_img, 64):      # (the reference's batch size: TEST.IMS_PER_BATCH 64)
        f_or.append(orc.vit_features(sd, synth.VIT_B16, imgs[s0:s0 + 64]))
    t_enc = time.perf_counter() - t0
    f_or = np.concatenate(f_or)
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    orc.vit_features(sd, synth.VIT_B16, imgs[:3])
    t_enc1 = (time.perf_counter() - t0) / 3
    torch.set_num_threads(cores)
    f, _ = synth.clustered_features(NQ + NG, 1280, 3.5, seed=2)     # the distance leg at FULL size (no extrapolation)
    orc.euclidean_distance(f[:64], f[NQ:NQ + 1024])
    t0 = time.perf_counter()
    orc.euclidean_distance(f[:NQ], f[NQ:])
    t_dist = time.perf_counter() - t0
    del f
    total = (NQ + NG) / n_img * t_enc + t_dist
    # re-rank leg: the oracle at the bench's own N = 20 000 (nq 4000, D 768, k1 50, k2 15) on all cores; one thread
    # on an N = 6000 sample (the dense N x N passes make it ~quadratic in N)
    f, _ = synth.clustered_features(20000, 768, 3.0, seed=1234)
    t0 = time.perf_counter()
    orc.re_ranking(f[:4000], f[4000:], 50, 15, 0.3)
    t_rr = time.perf_counter() - t0
    code = ("import sys,time;sys.path[:0]=[%r,%r];from mpreid import synth;from oracle import oracle as orc;"
            "f,_=synth.clustered_features(6000,768,3.0,seed=1234);t=time.perf_counter();"
            "orc.re_ranking(f[:1200],f[1200:],50,15,0.3);print(time.perf_counter()-t)" %
            (ROOT, os.path.join(ROOT, "mp-reid_amd")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True,
                       text=True)
    t_rr1 = float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
    base = {"value": round((NQ + NG) / total, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle ViT f32 {n_img} img {t_enc:.1f}s (scaled to {NQ + NG}) + euclid {NQ}x{NG}x1280 {t_dist:.2f}s; "
                      f"re-rank N=20000 {t_rr:.1f}s",
            "sample_long": f"oracle ViT-B/16 fp32 (torch CPU) on {n_img} images: {t_enc:.2f} s, extrapolated linearly to "
                           f"{NQ}+{NG} images; oracle euclid {NQ}x{NG}x1280 at full size: {t_dist:.2f} s; oracle re-rank "
                           f"N=20000 (nq 4000, D 768, k1 50, k2 15) on {cores} threads: {t_rr:.2f} s; on 1 thread at "
                           f"N=6000: {t_rr1 if t_rr1 is None else round(t_rr1, 2)} s",
            "encode_images_per_s": round(n_img / t_enc, 3), "encode_images_per_s_1thread": round(1.0 / t_enc1, 3),
            "rerank_s": round(t_rr, 3), "rerank_n": 20000, "rerank_s_1thread_n6000": t_rr1}
    return base, {"imgs": imgs, "pid": pid, "sd": sd, "f_or": f_or}


def parity_deltas(ops, sample, precision="split"):
    """north_star's parity clause, measured live on the cpu_baseline leg's sample (the oracle here is the CHECKER): the HIP
    encoder's features against the oracle's (relative L2), |dmAP| / |dRank-1| of the two pipelines with and without
    re-ranking (k1 20, k2 6: the sample is a few hundred images), and the largest distance-entry error of the three
    distance modes against the oracle's exact chain.  The asserted versions: tests/test_gpu_map_parity.py (1024 / 4096
    images), tests/test_gpu_distance.py."""
    import numpy as np
    import torch
    from mpreid import synth
    from oracle import oracle as orc
    imgs, pid, f_or = sample["imgs"], sample["pid"], sample["f_or"]
    n = len(pid)
    nq = max(1, n // 5)
    enc = ops.VitEncoder(synth.VIT_B16, sample["sd"], (H, W), precision=precision)
    f = enc(torch.from_numpy(imgs))
    del enc
    out = {"parity_images": n,
           "feat_rel_l2_split": float(np.linalg.norm(f.cpu().numpy() - f_or) / np.linalg.norm(f_or))}
    fo = orc.l2_normalize(f_or)
    fn = ops.l2_normalize(f)
    k1, k2 = (20, 6) if n >= 64 else (4, 2)
    for key, rr in (("", False), ("_rr", True)):
        d_o = orc.re_ranking(fo[:nq], fo[nq:], k1, k2, 0.3) if rr else orc.euclidean_distance(fo[:nq], fo[nq:])
        d_h = ops.re_ranking(fn[:nq], fn[nq:], k1, k2, 0.3)[0] if rr else ops.euclidean_distance(fn[:nq], fn[nq:])
        cmc_o, map_o = orc.eval_func(d_o, pid[:nq], pid[nq:])
        cmc_h, map_h = orc.eval_func(d_h.cpu().numpy(), pid[:nq], pid[nq:])
        out[f"dmap{key}_split"] = float(abs(map_h - map_o))
        out[f"dr1{key}_split"] = float(abs(float(cmc_h[0]) - float(cmc_o[0])))
    # distance entries (north_star: within 1e-5 fp32): the three modes on the oracle's own normalised features
    fot = torch.from_numpy(fo).to(f.device)
    want = orc.euclidean_distance(fo[:nq], fo[nq:])
    for key, mode in (("exact", ops.GEMM_F32_EXACT), ("split3", getattr(ops, "GEMM_F16_SPLIT3", None)), ("f16", ops.GEMM_F16_FAST)):
        if mode is not None:
            got = ops.euclidean_distance(fot[:nq], fot[nq:], mode=mode).cpu().numpy()
            out[f"dist_err_{key}"] = float(np.abs(got - want).max())
    return out


# (live_gemm_traffic, rerank_roofline, extras, drop_in_extras: tools/bench_extras.py)


# ----------------------------------------------------------------------------------------------------------------
# the drop-in path: make_model -> do_inference -> R1_mAP_eval.compute()   (market workload)
# ----------------------------------------------------------------------------------------------------------------
class ValLoader:
    """Query-then-gallery validation loader with the reference's batch tuple (datasets/make_dataloader.py:39-43):
    (img, pids, camids, camids_tensor, viewids_tensor, img_paths), `per_batch` samples per batch.  `batches` holds the images of
    THIS rank's samples (global positions `index`, ascending) batch by batch: fp32 tensors in HBM (headline), pageable
    fp32 host tensors (the reference's loader type) or RawImageBatch lists of decoded uint8 images."""

    def __init__(self, batches, index, n_total, pids, camids):
        import torch
        self.batches, self.index, self.n = batches, list(index), n_total
        self.meta = []
        lo = 0
        for b in batches:
            k = len(b)
            idx = self.index[lo:lo + k]
            cams = tuple(int(camids[i]) for i in idx)
            self.meta.append((tuple(int(pids[i]) for i in idx), cams, torch.tensor(cams, dtype=torch.int64),
                              torch.zeros(k, dtype=torch.int64), tuple("synthetic/%07d.jpg" % i for i in idx)))
            lo += k
        assert lo == len(self.index)

    def __iter__(self):
        for img, m in zip(self.batches, self.meta):
            yield (img,) + m

    def __len__(self):
        return len(self.batches)

    def shard(self, indices):
        """processor.shard_val_loader asks for this rank's samples: exactly the ones this loader holds"""
        assert list(indices) == self.index, "ValLoader holds another shard"
        return iter(self)


def market_cfg(a, nq, ng_total, rerank):
    from config import cfg_base
    cfg = cfg_base.clone()
    cfg.defrost()
    cfg.merge_from_list(["DATASETS.SYNTH_QUERY", nq, "DATASETS.SYNTH_GALLERY", ng_total, "TEST.RE_RANKING", bool(rerank),
                         "MODEL.ENCODER_PRECISION", a.encoder_precision, "TEST.DISTANCE_MODE", a.dist_mode,
                         "TEST.RERANK_ALGO", a.rerank_algo, "TEST.IMS_PER_BATCH", 64])
    cfg.freeze()
    return cfg


def quiet_do_inference(cfg, model, loader, nq):
    """do_inference with the evaluator's progress prints dropped: stdout carries the ONE JSON line and stderr stays a
    handful of lines (round 4's ~90 repeated evaluator lines on stderr drowned the line in the driver's capture)"""
    import contextlib
    from processor.processor import do_inference
    with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
        return do_inference(cfg, model, loader, nq)


def timed_do_inference(cfg, model, loader, nq, reps, pipeline_env=None):
    """wall seconds per call of do_inference (whole call: encode pipeline + compute() + host hand-over), after one warm call"""
    import torch
    old = os.environ.get("MPREID_PIPELINE")
    if pipeline_env is not None:
        os.environ["MPREID_PIPELINE"] = pipeline_env
    try:
        quiet_do_inference(cfg, model, loader, nq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            quiet_do_inference(cfg, model, loader, nq)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    finally:
        if pipeline_env is not None:
            if old is None:
                os.environ.pop("MPREID_PIPELINE", None)
            else:
                os.environ["MPREID_PIPELINE"] = old


# ----------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------
def run_rank(a):
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    import bench_extras as BX   # tools/bench_extras.py: the secondary legs (extras file)
    from mpreid import _lib, distributed as D, ops, synth
    BX.bind(sys.modules[__name__])

    rank, world, local = D.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    local = local % max(torch.cuda.device_count(), 1)   # gloo staging / MPREID_ALLOW_SHARED_GPU: ranks share devices (main() checked)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = _lib.load()
    mode = {"exact": ops.GEMM_F32_EXACT, "f16": ops.GEMM_F16_FAST, "split3": getattr(ops, "GEMM_F16_SPLIT3", None)}[a.dist_mode]
    assert mode is not None, "--dist-mode split3 needs the 3-term split GEMM"
    comm = {"bytes": 0, "ms": 0.0, "calls": 0}

    def gather_rows(x, n_total):
        """(bytes and event-timed duration are recorded inside mpreid.distributed: D.comm_stats)"""
        return D.all_gather_rows(x, n_total)

    nstreams = max(1, a.streams)
    div = 16 if a.small else 1
    wl = a.workload
    enc = None
    side, encs = [], []
    if wl == "msmt17":
        enc = ops.VitEncoder(synth.VIT_B16, synth.vit_state_dict(synth.VIT_B16, seed=7), (H, W),
                             precision=a.encoder_precision)
        side = [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]
        encs = [enc] + [enc.clone_for_stream(f"vit{i + 1}") for i in range(nstreams - 1)]

    def encode_all(imgs, feats):
        main = torch.cuda.current_stream()
        for st in side:
            st.wait_stream(main)
        n_local = imgs.shape[0]
        for bi, s in enumerate(range(0, n_local, a.batch)):
            e = min(n_local, s + a.batch)
            k = bi % len(encs)
            if k == 0:
                encs[0](imgs[s:e], out=feats[s:e])
            else:
                with torch.cuda.stream(side[k - 1]):
                    encs[k](imgs[s:e], out=feats[s:e])
        for st in side:
            main.wait_stream(st)

    market = None
    if wl == "market":
        # weak scaling: every rank owns a whole Market-1501-sized gallery shard + 1/world of the queries.  The step is the
        # drop-in call itself: do_inference on a make_model model over a loader of HBM-resident 64-image batches
        from model.make_model import make_model
        nq, ng = NQ // div, NG // div
        q_lo, q_hi = D.shard_range(nq, rank, world)
        nq_local, ng_local = q_hi - q_lo, ng
        ng_total = world * ng
        scaling = "weak"
        imgs = make_images(nq_local + ng_local, 1234 + rank, dev)
        images_per_step = nq + ng_total
        cfg = market_cfg(a, nq, ng_total, a.rerank)
        model = make_model(cfg, num_class=751, camera_num=6, view_num=1)
        model.to("cuda")
        rng = np.random.default_rng(1234)
        pids_all = rng.integers(0, max(ng_total // 21, 1), size=nq + ng_total)    # ~21 gallery images per identity (Market-1501)
        cams_all = rng.integers(0, 6, size=nq + ng_total)
        g_lo, g_hi = D.shard_range(ng_total, rank, world)
        index = list(range(q_lo, q_hi)) + list(range(nq + g_lo, nq + g_hi))
        loader = ValLoader(list(torch.split(imgs, 64)), index, nq + ng_total, pids_all, cams_all)
        market = {"cfg": cfg, "model": model, "loader": loader}
        feat_dim = 1280
        enc = model   # (truthy: the workload has an encoder)
    elif wl == "msmt17":
        nq, ng_total = MSMT_NQ // div, MSMT_NG // div
        q_lo, q_hi = D.shard_range(nq, rank, world)
        g_lo, g_hi = D.shard_range(ng_total, rank, world)
        nq_local, ng_local = q_hi - q_lo, g_hi - g_lo
        scaling = "strong"
        imgs = make_images(nq_local + ng_local, 1234 + rank, dev)
        feat_dim = enc.feat_dim
        images_per_step = nq + ng_total
    else:  # synth: features are the input (BASELINE configs[3] "768-d feats")
        nq, ng_total = SYN_NQ // div, SYN_NG // div
        q_lo, q_hi = D.shard_range(nq, rank, world)
        g_lo, g_hi = D.shard_range(ng_total, rank, world)
        nq_local, ng_local = q_hi - q_lo, g_hi - g_lo
        scaling = "strong"
        allf, _ = make_features(nq + ng_total, SYN_D, 3.0, 1234, dev)
        fq_local = allf[q_lo:q_hi].clone()
        fg_local = allf[nq + g_lo: nq + g_hi].clone()
        del allf
        feat_dim = SYN_D
        images_per_step = nq + ng_total
    if wl == "msmt17":
        feats = torch.empty((nq_local + ng_local, feat_dim), dtype=torch.float32, device=dev)
    if wl != "market":
        block = torch.empty((nq, ng_local), dtype=torch.float32, device=dev)
    rr_holder = {}

    host_concat = a.host_concat == "on"   # (market: the hand-over to the host is part of R1_mAP_eval.compute())
    pipe_env = f"streams={nstreams}" + (f",group={a.group}" if a.group > 0 else "")
    os.environ["MPREID_PIPELINE"] = pipe_env

    def step():
        if market is not None:
            r1, _ = quiet_do_inference(market["cfg"], market["model"], market["loader"], nq)
            rr_holder["rank1"] = float(r1)
            return
        if wl == "synth":
            fq, fg = fq_local, fg_local
        else:
            encode_all(imgs, feats)
            fn = ops.l2_normalize(feats)
            fq, fg = fn[:nq_local], fn[nq_local:]
        qf = gather_rows(fq, nq)                       # RCCL all-gather of the query features (N > 1)
        ops.euclidean_distance(qf, fg, mode=mode, out=block)
        if a.rerank:
            gf_all = gather_rows(fg, ng_total)          # every rank needs all features for its rows of the N x N problem
            rr = D.re_ranking_sharded(qf, gf_all, 50, 15, 0.3,   # this rank's final_dist[q_lo:q_hi, nq:]
                                      algo=ops.RERANK_SPARSE_SPLIT3 if a.rerank_algo == "split3" else ops.RERANK_AUTO)
            assert rr.shape[1] == ng_total
            rr_holder["rr"] = rr
        # north_star: "per-shard distance blocks concatenated on the host": the column blocks (and the re-ranked row
        # blocks) travel as tensors to rank 0 (RCCL gather over xGMI) and land in ONE pinned host matrix; at N = 1 this is
        # the D2H copy of the matrix R1_mAP_eval.compute() hands to its caller
        if host_concat:
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            full = D.gather_column_blocks_to_host(block, dst=0, reuse_buffer=True)
            if a.rerank:
                D.gather_row_blocks_to_host(rr_holder["rr"], dst=0, reuse_buffer=True)
            comm["host_ms"] = comm.get("host_ms", 0.0) + (time.perf_counter() - t_h) * 1e3
            if rank == 0:
                assert full.shape == (nq, ng_total)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    comm.update(host_ms=0.0)
    D.comm_stats_reset(timing=True)
    instrument_live = nstreams == 1 or enc is None   # event pairs on one stream also span other streams' kernels
    L.mpreid_profile_reset()
    if instrument_live:
        L.mpreid_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    L.mpreid_profile_enable(0)
    comm_ms = sum(e0.elapsed_time(e1) for e0, e1 in D.comm_stats["events"])
    comm_bytes, comm_calls = D.comm_stats["bytes"], D.comm_stats["calls"]
    D.comm_stats_reset()
    host_concat_ms = comm.get("host_ms", 0.0) / max(a.steps, 1)
    if not instrument_live:
        # roofline leg: one more pass of the same work on ONE stream with per-launch hipEvents
        saved = (list(side), list(encs))
        side[:], encs[:] = [], [enc]
        os.environ["MPREID_PIPELINE"] = "streams=1"
        L.mpreid_profile_enable(1)
        step()
        fence()
        L.mpreid_profile_enable(0)
        os.environ["MPREID_PIPELINE"] = pipe_env
        side[:], encs[:] = saved
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if market is not None:
        assert np.isfinite(rr_holder["rank1"])
    else:
        assert torch.isfinite(block).all()

    if rank == 0:
        ents = (_lib.ProfileEntry * 48)()
        n_ent = L.mpreid_profile_query(ents, 48)
        classes, other = [], []
        for i in range(min(n_ent, 48)):
            e = ents[i]
            avg_ms = e.total_ms / max(e.launches, 1)
            if e.epilogue >= 100:   # HBM-bound encoder kernels: flops_total holds algorithmic bytes
                other.append({"kernel": {100: "layernorm_kernel", 101: "attention_kernel / attention_split_kernel",
                                         102: "eval_rank_kernel"}.get(e.epilogue),
                              "rows": e.m, "launches": e.launches, "avg_ms": round(avg_ms, 4), "total_ms": round(e.total_ms, 2),
                              "cls_tail": bool(e.n == 1) if e.epilogue == 101 else None,
                              "gb_per_launch": round(e.flops_total / max(e.launches, 1) / 1e9, 4),
                              "gbps": round(e.flops_total / e.total_ms / 1e6, 1) if e.total_ms > 0 else None,
                              "class_id": e.epilogue})
                continue
            split = e.epilogue in (10, 11, 12, 13)
            big = e.m % 256 == 0 and e.n % 256 == 0 and e.m * e.n >= 128 * 65536
            classes.append({"kernel": ("gemm_f16_big_kernel<%s>" if big else "gemm_f16_kernel<%s>") %
                                      _lib.GEMM_EPILOGUE_NAMES.get(e.epilogue, e.epilogue),
                            "M": e.m, "N": e.n, "K": (e.k // 2 if split else e.k), "operand_pairs": split,
                            "launches": e.launches, "avg_ms": round(avg_ms, 4),
                            "total_ms": round(e.total_ms, 2), "gflop_per_launch": round(e.flops_total / max(e.launches, 1) / 1e9, 2),
                            "tflops": round(e.flops_total / e.total_ms / 1e9, 1) if e.total_ms > 0 else None,
                            "epilogue_id": e.epilogue})
        top = classes[0] if classes else None
        roof = None
        traffic = None
        traffic_src = None
        traffic_replayed = None
        if top and world == 1 and not a.small and not a.no_live_traffic:
            traffic, traffic_src = BX.live_gemm_traffic(top["epilogue_id"], top["N"], top["K"])
            traffic_replayed = False if traffic is not None else None
            live_note = traffic_src
        else:
            live_note = "not attempted (N > 1, --small or --no-live-traffic)"
        if top and traffic is None:
            for fn in ("r06_gemm_pmc_traffic.json", "r05_gemm_pmc_traffic.json", "r04_gemm_pmc_traffic.json"):
                try:  # HBM bytes per launch of the dominant kernel: committed rocprofv3 PMC passes (profiles/)
                    tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    key = f"{_lib.GEMM_EPILOGUE_NAMES.get(top['epilogue_id'])}:{top['N']}:{top['K']}"
                    traffic = tj["classes"].get(key, {}).get("hbm_bytes_per_launch")
                    if traffic is not None:
                        traffic_src = f"replayed: profiles/{fn} (rocprofv3 --pmc, separate FETCH / WRITE passes)"
                        traffic_replayed = True
                        break
                except Exception:
                    traffic = None
        if top:
            step_ms = dt / a.steps * 1e3
            sp = top["operand_pairs"]
            mult = 3.0 if sp else 1.0   # executed fp16 products per multiply-add of the algorithm
            alg_tf = top["tflops"] / mult
            roof = {"bound": "mfma", "kernel": top["kernel"], "shape": [top["M"], top["N"], top["K"]],
                    "achieved": round(alg_tf, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(alg_tf / PEAK_F16_TFLOPS, 4),
                    "avg_launch_ms": top["avg_ms"], "launches": top["launches"],
                    "algorithmic_flop": 2 * top["M"] * top["N"] * top["K"],
                    "traffic": traffic, "traffic_replayed": traffic_replayed,
                    "traffic_source": traffic_src if traffic is not None else None,
                    "traffic_live_note": live_note,
                    "measured": "hipEvents per launch on its stream, " +
                                ("inside the timed region" if instrument_live else
                                 "1-stream pass of the same step right after the timed region"),
                    # ---- below: extras file only ----
                    "algorithmic_gflop_per_launch": round(top["gflop_per_launch"] / mult, 2),
                    "flop_convention": "2*M*N*K per launch (SURVEY.md section 8d): ALGORITHMIC flops" +
                                       ("; the kernel executes three fp16 products per multiply-add (operand pairs hi + lo: "
                                        "hi.hi' + lo.hi' + hi.lo'), reported as achieved_executed / frac_executed" if sp else ""),
                    "traffic_note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, same kernel and shape at M=65536",
                    "measured_note": ("inside the timed region" if instrument_live else
                                      "in a single-stream pass of the same step right after the timed region "
                                      "(the timed region alternates batches over %d streams)" % nstreams),
                    "all_gemm_tflops": round(sum(c["tflops"] * c["total_ms"] for c in classes) /
                                             max(sum(c["total_ms"] for c in classes), 1e-9), 1),
                    "gemm_share_of_step": round(sum(c["total_ms"] for c in classes) /
                                                (dt * 1e3 if instrument_live else step_ms), 3)}
            if sp:
                # `achieved` / `frac` follow SURVEY 8(d) (2*M*N*K); the executed pair counts what the matrix cores do (three
                # fp16 products per multiply-add of the fp32-grade algorithm)
                roof["achieved_executed"] = top["tflops"]
                roof["frac_executed"] = round(top["tflops"] / PEAK_F16_TFLOPS, 4)
                roof["frac_of_sustainable_1250TF"] = round(top["tflops"] / 1250.0, 4)
            try:   # matrix-pipe counters of the same kernel class (committed rocprofv3 --pmc pass, tools/collect_profiles.sh)
                mfn = next(f for f in ("r06_gemm_pmc_mfma.json", "r05_gemm_pmc_mfma.json", "r04_gemm_pmc_mfma.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
                mj = json.load(open(os.path.join(ROOT, "profiles", mfn)))
                key = f"{_lib.GEMM_EPILOGUE_NAMES.get(top['epilogue_id'])}:{top['N']}:{top['K']}"
                if key in mj.get("classes", {}):
                    roof["mfma_busy"] = mj["classes"][key]
                    roof["mfma_busy_source"] = "profiles/" + mfn
            except Exception:
                pass
        desc_long = {
            "market": "Market-1501 shape on MI355X (BASELINE configs[1]) through the drop-in API: one step = "
                      "processor.do_inference(cfg, make_model(cfg, ...), val_loader, num_query) -- ViT-B/16 encode of "
                      f"{nq} query + {NG // div} gallery 3x256x128 images per GPU shard (seeded random init; 64-image fp32 "
                      "loader batches resident in HBM), R1_mAP_eval.compute(): L2-normalise, all-gather query features, "
                      f"euclidean distmat [{nq} x {NG // div}] per GPU ({a.dist_mode}), CMC/mAP ranking, matrix + "
                      "features handed to the host",
            "msmt17": f"MSMT17 shape (BASELINE configs[4]): ViT-B/16 encode of {nq} query + {ng_total} gallery "
                      f"images sharded over {world} GPU(s), L2-normalise, all-gather query features, euclidean "
                      f"distmat [{nq} x {ng_total}/{world}] at D=1280 ({a.dist_mode})",
            "synth": f"synthetic {nq}-query x {ng_total}-gallery x {SYN_D} features (BASELINE configs[3]), gallery rows "
                     f"sharded over {world} GPU(s), all-gather query features, euclidean distmat "
                     f"[{nq} x {ng_total}/{world}] per GPU ({a.dist_mode})",
        }[wl] + (", plus k-reciprocal re-ranking (k1=50, k2=15, lambda=0.3) of all queries against the whole gallery, "
                 "rows sharded over the GPUs" + (" [RERANK_SPARSE_SPLIT3: blend-term distances from the fp16 matrix cores, "
                                                   "outputs within 1e-6 of the bit-parity mode]" if a.rerank_algo == "split3" else "")
                 if a.rerank else ", no re-rank")
        desc = {   # the line's strings are <= MAX_STR characters (format_line); the long form goes to the extras file
            "market": f"Market-1501 configs[{2 if a.rerank else 1}]: do_inference ViT-B/16 {nq}q+{NG // div}g/GPU",
            "msmt17": f"MSMT17 shape (configs[4]): ViT-B/16 {nq}q+{ng_total}g over {world} GPU(s), euclid D=1280",
            "synth": f"synthetic {nq}x{ng_total}x{SYN_D} feats (configs[3]), gallery rows over {world} GPU(s), euclid",
        }[wl] + (", rerank" if a.rerank else ", no rerank") + ("; HBM-resident loader" if wl == "market" else "")
        res = {
            "metric": METRIC,
            "value": round(images_per_step * a.steps / dt, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None,
            "dtype": ({"split": "f16 pairs hi+lo (3 MFMA products per multiply-add, f32-grade), f32 acc; ",
                       "fp16": "f16 MFMA operands, f32 acc; ", "fp32": "f32 (exact f32 MFMA); "}[a.encoder_precision]
                      if enc is not None else "") +
                     {"exact": "distmat f32 exact", "f16": "distmat f16 one pass", "split3": "distmat 3-term f16 split"}[a.dist_mode],
            "dtype_note": ({"split": "encoder: f16 operand PAIRS (hi+lo, 3 products per multiply-add) on the f16 MFMA, f32 "
                                     "accumulate/residual/softmax -- fp32-grade features, meets |dmAP|,|dR1| <= 1e-4 "
                                     "(tests/test_gpu_map_parity.py)",
                            "fp16": "encoder: single f16 MFMA operands, f32 accumulate/residual -- feature error ~4e-4, does NOT "
                                    "meet the 1e-4 mAP bound on hard data",
                            "fp32": "encoder: all-f32 (exact f32 MFMA) -- meets the 1e-4 mAP bound"}[a.encoder_precision]
                           if enc is not None else None),
            "data": "synthetic",
            "config": {"workload": desc, "workload_long": desc_long, "images_per_step": images_per_step,
                       "rerank": bool(a.rerank), "distance_mode": a.dist_mode,
                       "encoder_batch": ((a.group or market["model"].encode_group) if market else a.batch) if enc else None,
                       "encoder_precision": a.encoder_precision if enc else None,
                       "encoder_streams": nstreams if enc else None,
                       "sharding": f"gallery rows over {world} GPU(s); 1 all-gather of query feats; blocks concatenated on rank 0's host"},
            "host_concat": None if not host_concat else {"ms_per_step": round(host_concat_ms, 3),
                            "bytes_per_step": int(4 * nq * ng_total * (2 if a.rerank else 1)),
                            "note": "inside the timed step: per-shard distance blocks -> rank 0 (tensor gather over xGMI when "
                                    "N > 1) -> one pinned host matrix [nq, ng]; wall time on rank 0 incl. the wait for the "
                                    "device to finish the step's kernels is excluded (the step is synchronised first)"},
            "roofline": roof, "gemm_classes": classes, "other_encoder_kernels": other,
            "reference_cpu": {"note": "the REFERENCE's own Python on 8 host threads of the build container (BASELINE.md section 2); "
                                      "cpu_baseline below is the C/torch ORACLE port on this box, not the reference",
                              "encode_images_per_s": 8.2, "euclid_distmat_market_s": 0.65, "re_ranking_market_s": 265.0},
        }
        if enc is not None:
            res["encode_tflops_algorithmic"] = round(images_per_step * a.steps * GFLOP_PER_IMG / dt / 1e3, 1)
        else:
            res["distmat_tflops_algorithmic"] = round(2.0 * nq * ng_total * SYN_D * a.steps / dt / 1e12, 1)
            res["unit_note"] = "images/s = (query + gallery feature rows) per second through distmat (+ re-rank)"
        if world > 1:
            res["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0   # 0: gloo staging (tests)
            res["all_gather"] = {"calls_per_step": comm_calls / max(a.steps, 1),
                                 "bytes_per_step": int(comm_bytes / max(a.steps, 1)),
                                 "ms_per_step": round(comm_ms / max(a.steps, 1), 3),
                                 "gb_per_s": round(comm_bytes / max(comm_ms, 1e-9) / 1e6, 2),
                                 "note": "bytes = gathered output per rank (RCCL all_gather_into_tensor over xGMI); "
                                         "hipEvents on the launch stream of rank 0"}
        if wl == "msmt17" or (wl == "synth" and a.dist_mode == "exact"):
            # fp16-MFMA distance mode against the exact fp32 mode on this rank's first 2048 x 8192 block
            qf_s = (ops.l2_normalize(feats)[:min(nq_local, 2048)] if wl != "synth" else fq_local[:2048])
            gf_s = (ops.l2_normalize(feats)[nq_local:nq_local + 8192] if wl != "synth" else fg_local[:8192])
            ex = ops.euclidean_distance(qf_s, gf_s)
            f16 = ops.euclidean_distance(qf_s, gf_s, mode=ops.GEMM_F16_FAST)
            chk = {"fp16_one_pass_max_abs_err": float((ex - f16).abs().max())}
            if hasattr(ops, "GEMM_F16_SPLIT3"):
                s3 = ops.euclidean_distance(qf_s, gf_s, mode=ops.GEMM_F16_SPLIT3)
                chk["fp16_split3_max_abs_err"] = float((ex - s3).abs().max())
            res["tolerance_check_vs_exact_fp32"] = chk
        parity = {}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"], sample = cpu_baseline(a.cpu_images)
            parity = parity_deltas(ops, sample, a.encoder_precision if enc is not None else "split")
            res["parity_sample"] = dict(parity, note="oracle (CPU checker) vs HIP on the cpu_baseline leg's identity images, "
                                                     "spread weights (std 0.05), nq = n/5, re-rank k1 20 k2 6; "
                                                     "asserted at 1024 / 4096 images in tests/test_gpu_map_parity.py")
            del sample
        if market is not None:
            from processor.processor import do_inference as _di
            res["drop_in"] = {"call": "processor.do_inference(cfg, model.make_model.make_model(cfg, ...), val_loader, num_query)",
                              "loader_batch": 64, "rank1": rr_holder.get("rank1"),
                              "pipeline": dict(getattr(_di, "last_pipeline_stats", {}))}
        if world == 1 and not a.no_extras and not a.small:
            drop = {}
            if market is not None:
                drop = BX.drop_in_extras(a, market["cfg"], market["model"], nq, ng_total, pids_all, cams_all, dev)
                market["loader"] = None
                del loader
            else:
                del block
            if wl != "synth":
                del imgs
            if wl == "msmt17":
                del feats
            ops.release_workspaces()
            torch.cuda.empty_cache()
            res["extras"], roofs = BX.extras(ops, dev, with_widened=(wl == "market"))
            res["extras"].update(drop)
            if drop:
                res["extras"]["do_inference_fp32_loader_over_headline"] = round(drop["do_inference_images_per_s_fp32_loader"] / res["value"], 4)
                res["extras"]["do_inference_raw_loader_over_headline"] = round(drop["do_inference_images_per_s_raw_loader"] / res["value"], 4)
                res["config"]["workload"] += "; host-fed x%.2f" % res["extras"]["do_inference_fp32_loader_over_headline"]
            ex = res["extras"]
            rr_ms = ex["rerank_N20000_nq4000_d768_k50_15_ms"]
            m20 = {"distmat_exact_ms": ex["feat_gemm_20kx20k_d768_fp32exact_ms"],
                   "distmat_split3_ms": ex.get("feat_gemm_20kx20k_d768_split3_ms"),
                   "distmat_split3_frac_exec": (round(ex["feat_gemm_20kx20k_d768_split3_executed_tflops"] / PEAK_F16_TFLOPS, 4)
                                                if "feat_gemm_20kx20k_d768_split3_executed_tflops" in ex else None),
                   "distmat_f16_ms": ex["feat_gemm_20kx20k_d768_fp16_ms"],
                   "distmat_f16_frac": ex["feat_gemm_20kx20k_d768_fp16_frac_of_peak"],
                   "distmat_f16_2t_ms": ex["feat_gemm_20kx20k_d768_fp16_two_tensors_ms"],
                   "distmat_f16_2t_frac": ex["feat_gemm_20kx20k_d768_fp16_two_tensors_frac_of_peak"],
                   "rerank_ms": rr_ms, "rerank_vs_ref265s_x": round(265e3 / rr_ms, 0)}
            if res.get("cpu_baseline"):
                m20["rerank_cpu_port_ms"] = round(res["cpu_baseline"]["rerank_s"] * 1e3, 0)
                m20["rerank_vs_cpu_x"] = round(res["cpu_baseline"]["rerank_s"] * 1e3 / rr_ms, 0)
            res["metric_20k"] = m20
            res["metric_20k_note"] = (
                "the rest of BASELINE.json's metric, N = 20 000 clustered features, D = 768: all-pairs stored distance matrix in the "
                "exact fp32 chain (bit-parity mode), the 3-term fp16 split (<= 1e-6) and one-pass fp16 (~1e-4; `frac` = 2*N*N*D "
                "over the 2.5 PF fp16 peak; `_2t` = two separate tensors, every tile computed), re-ranking nq 4000 / k1 50 / k2 15 "
                "(bit-identical to the oracle) against the oracle port on this box's host cores and the reference's 265 s "
                "(BASELINE.md: its own Python, 8 threads, N = 19 281); parity deltas: parity_sample")
            enc_roofs = [dict(roof, stage="encoder dominant GEMM class")] if roof else []
            for o in other:   # the HBM-bound encoder kernels of the timed (or single-stream) pass
                if o["gbps"]:
                    enc_roofs.append({"stage": "eval_func ranking (R1_mAP_eval.compute(), matrix resident)" if o["class_id"] == 102 else
                                      "encoder." + ("layernorm" if o["class_id"] == 100 else
                                                    ("attention (CLS tile only, last block)" if o["cls_tail"] else "attention")),
                                      "kernel": o["kernel"], "bound": "hbm", "achieved": o["gbps"], "peak": PEAK_HBM_GBS,
                                      "unit": "GB/s", "frac": round(o["gbps"] / PEAK_HBM_GBS, 4),
                                      "algorithmic_bytes": int(o["gb_per_launch"] * 1e9), "avg_launch_ms": o["avg_ms"],
                                      "traffic": None})
            res["roofline_all"] = enc_roofs + roofs
        if parity:
            res["metric_20k"] = dict(res.get("metric_20k") or {}, **{k: (float("%.3g" % v) if isinstance(v, float) else v)
                                                                      for k, v in parity.items()})
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    launched = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if a.gpus > 1:                     # before anything initialises HIP or joins a rendezvous: every rank exits 2,
        check_gpu_count(a.gpus, quiet=int(os.environ.get("LOCAL_RANK", "0")) != 0)   # local rank 0 says why
    if a.gpus > 1 and not launched:
        launch_children(a.gpus)
        return
    run_rank(a)


if __name__ == "__main__":
    main()

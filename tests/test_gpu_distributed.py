"""Real multi-process execution of the N > 1 path on ONE GPU: two ranks (separate processes, a real default
process group) share cuda:0.  RCCL refuses two ranks on one device, so the collectives go through
MPREID_DIST_BACKEND=gloo (host-staged) -- the partition, padding / trimming of ragged shards, the int16 -> uint8 view
of the fp16 bit patterns, the global-max all-reduce and the four phases of re_ranking_sharded all run exactly as with
RCCL.  Bar: byte-equality with the 1-rank result (and the oracle).

Replaces nn.DataParallel in the reference's do_inference (processor/processor.py:178-182)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "mp-reid_amd")]
    from mpreid import distributed as D, ops, synth
    import torch.distributed as dist
    rank, world, local = D.init_from_env()
    assert world == {world} and dist.get_backend() == "gloo"
    torch.cuda.set_device(0)
    out_dir = sys.argv[1]
    res = {{}}
    for ci, (n, nq, d, k1, k2, lam) in enumerate({cases!r}):
        f, _ = synth.clustered_features(n, d, 2.5, seed=1000 + n + k1, per_id=10)
        ft = torch.from_numpy(f).cuda()
        q, g = ft[:nq], ft[nq:]
        # (1) sharded distance matrix: this rank's gallery shard -> column block -> host concatenation
        g_lo, g_hi = D.shard_range(n - nq, rank, world)
        q_lo, q_hi = D.shard_range(nq, rank, world)
        qf = D.all_gather_rows(q[q_lo:q_hi].contiguous(), nq)   # ragged shards: padded, gathered, trimmed
        assert torch.equal(qf, q)
        block = ops.euclidean_distance(qf, g[g_lo:g_hi].contiguous())
        full = D.gather_column_blocks_to_host(block, dst=0)
        # (2) row-sharded re-ranking over the real process group
        rr = D.re_ranking_sharded(q, g, k1, k2, lam)
        assert rr.shape == (q_hi - q_lo, n - nq)
        rr_full = D.gather_row_blocks_to_host(rr, dst=0)
        if rank == 0:
            res[f"dist{{ci}}"] = full
            res[f"rr{{ci}}"] = rr_full
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), **res)
    dist.barrier(); dist.destroy_process_group()
""")

CASES = [(1500, 300, 256, 50, 15, 0.3), (901, 7, 64, 20, 6, 0.3), (700, 101, 128, 10, 1, 0.5),
         (2600, 500, 128, 50, 15, 0.3)]   # the last one is large enough for the SPARSE sharded phases


def _spawn(script, args, world, tmp_path, extra_env=None):
    from conftest import run_ranks
    run_ranks([sys.executable, str(script)] + args, world, 900,
              dict(MPREID_DIST_BACKEND="gloo", OMP_NUM_THREADS="4", **(extra_env or {})), local_rank=lambda r: 0)


@pytest.mark.parametrize("world", [2, 3])
def test_real_ranks_sharded_rerank_and_distmat_equal_single_rank(tmp_path, world):
    import torch
    from mpreid import ops, synth
    from oracle import oracle as orc
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, world=world, cases=CASES))
    _spawn(script, [str(tmp_path)], world, tmp_path)
    got = np.load(tmp_path / "out.npz")
    for ci, (n, nq, d, k1, k2, lam) in enumerate(CASES):
        f, _ = synth.clustered_features(n, d, 2.5, seed=1000 + n + k1, per_id=10)
        ft = torch.from_numpy(f).cuda()
        single_d = ops.euclidean_distance(ft[:nq], ft[nq:]).cpu().numpy()
        assert np.array_equal(got[f"dist{ci}"], single_d), ci
        single_rr, _ = ops.re_ranking(ft[:nq], ft[nq:], k1, k2, lam)
        assert np.array_equal(got[f"rr{ci}"], single_rr.cpu().numpy()), ci
        assert np.array_equal(got[f"rr{ci}"], orc.re_ranking(f[:nq], f[nq:], k1, k2, lam)), ci


@pytest.mark.parametrize("args", [["--workload", "market", "--rerank"], ["--workload", "synth", "--rerank"],
                                  ["--workload", "msmt17"]])
def test_bench_self_launches_two_ranks(args):
    """`python bench.py --gpus 2` without a launcher: the parent spawns the ranks itself and relays ONE JSON line"""
    env = dict(os.environ, MPREID_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--small", "--steps", "1",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline"] + args, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    assert len(lines[0].encode()) <= 4096                      # the N-rank line obeys the same size contract
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["all_gather"]["bytes_per_step"] > 0
    assert set(j["all_gather"]) == {"calls_per_step", "bytes_per_step", "ms_per_step", "gb_per_s"}
    assert j["rccl_ranks"] == 0                                # gloo staging: NOT an RCCL run, and the line says so
    assert j["cpu_baseline"] is None and "roofline" in j and j["extras_file"] == "bench_extras.json"
    assert j["scaling"] == ("weak" if args[1] == "market" else "strong")


def test_bench_under_the_drivers_launcher_two_ranks(tmp_path):
    """the driver's multi-GPU form, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` (two gloo-staged ranks on the one GPU here): rank 0 prints the ONE
    line, the others nothing, exit code 0"""
    from conftest import free_port
    env = dict(os.environ, MPREID_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--small"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0].encode()) <= 4096
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["value"] > 0
    assert j["rccl_ranks"] == 0 and j["all_gather"]["calls_per_step"] >= 1 and j["cpu_baseline"] is None
    assert "metric_20k" not in j or j["metric_20k"] is None or isinstance(j["metric_20k"], dict)


# ---- the evaluator and do_inference behind the reference API under WORLD_SIZE > 1 -------------------------------------
# Reference: processor/processor.py:178-182 goes multi-device inside do_inference (nn.DataParallel) and
# utils/metrics.py:110-134 returns ONE distmat from compute().  Here: one R1_mAP_eval instance per rank; rank 0's 7-tuple
# must equal the single-process one byte for byte.
EVAL_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "mp-reid_amd")]
    from mpreid import distributed as D, synth
    from utils.metrics import R1_mAP_eval
    import torch.distributed as dist
    rank, world, local = D.init_from_env()
    torch.cuda.set_device({device})
    res = {{}}
    for ci, (n, nq, d, rerank) in enumerate({cases!r}):
        f, pid = synth.clustered_features(n, d, 2.5, seed=77 + n, per_id=6, normalize=False)
        cam = synth.labels_for(n)
        q_lo, q_hi = D.shard_range(nq, rank, world)
        g_lo, g_hi = D.shard_range(n - nq, rank, world)
        idx = list(range(q_lo, q_hi)) + list(range(nq + g_lo, nq + g_hi))      # this rank's samples, global order
        ev = R1_mAP_eval(nq, max_rank=50, feat_norm='yes', reranking=rerank)
        ev.reset()
        for s in range(0, len(idx), 64):                                     # loader-sized update() calls
            sel = idx[s:s + 64]
            ev.update((torch.from_numpy(f[sel]).cuda(), tuple(int(p) for p in pid[sel]), tuple(int(c) for c in cam[sel])))
        cmc, mAP, distmat, pids, camids, qf, gf = ev.compute()
        cmc2, mAP2 = ev.compute()[:2]                                        # compute() may be called again
        assert np.array_equal(cmc, cmc2) and mAP == mAP2
        if rank == 0:
            res.update({{f"cmc{{ci}}": cmc, f"map{{ci}}": np.float64(mAP), f"dist{{ci}}": distmat, f"pids{{ci}}": np.asarray(pids),
                        f"cams{{ci}}": np.asarray(camids), f"qf{{ci}}": qf.numpy(), f"gf{{ci}}": gf.numpy()}})
        else:
            assert distmat is None and qf.shape[0] == nq
    if rank == 0:
        np.savez(os.path.join(sys.argv[1], "eval.npz"), **res)
    dist.barrier(); dist.destroy_process_group()
""")

EVAL_CASES = [(900, 150, 192, False), (2600, 500, 128, True), (643, 41, 64, True), (700, 5, 64, False)]


@pytest.mark.parametrize("world", [2, 3])
def test_r1_map_eval_sharded_equals_single_process(tmp_path, world):
    import torch
    from mpreid import synth
    from utils.metrics import R1_mAP_eval
    script = tmp_path / "eval_worker.py"
    script.write_text(EVAL_WORKER.format(root=ROOT, cases=EVAL_CASES, device="0"))
    _spawn(script, [str(tmp_path)], world, tmp_path)
    _check_eval_npz(tmp_path / "eval.npz")


def _check_eval_npz(path):
    import torch
    from mpreid import synth
    from utils.metrics import R1_mAP_eval
    got = np.load(path)
    for ci, (n, nq, d, rerank) in enumerate(EVAL_CASES):
        f, pid = synth.clustered_features(n, d, 2.5, seed=77 + n, per_id=6, normalize=False)
        cam = synth.labels_for(n)
        ev = R1_mAP_eval(nq, max_rank=50, feat_norm='yes', reranking=rerank)
        ev.reset()
        ev.update((torch.from_numpy(f).cuda(), tuple(int(p) for p in pid), tuple(int(c) for c in cam)))
        cmc, mAP, distmat, pids, camids, qf, gf = ev.compute()
        assert np.array_equal(got[f"cmc{ci}"], cmc) and got[f"cmc{ci}"].dtype == cmc.dtype, ci
        assert float(got[f"map{ci}"]) == float(mAP), ci
        assert np.array_equal(got[f"dist{ci}"], distmat) and got[f"dist{ci}"].dtype == np.float32, ci
        assert np.array_equal(got[f"pids{ci}"], np.asarray(pids)) and np.array_equal(got[f"cams{ci}"], np.asarray(camids)), ci
        assert np.array_equal(got[f"qf{ci}"], qf.numpy()) and np.array_equal(got[f"gf{ci}"], gf.numpy()), ci


@pytest.mark.parametrize("rerank", ["False", "True"])
def test_test_py_under_two_ranks_matches_single_process(tmp_path, rerank):
    """python -m torch.distributed.run ... test.py: do_inference shards the loader by index range, every rank returns the
    single-process (Rank-1, Rank-5) and rank 0 logs the same mAP line"""
    code = ("import sys, os; sys.path.insert(0, {pkg!r}); os.chdir({pkg!r}); import test as T; "
            "r = T.main(['--config_file', '', 'DATASETS.SYNTH_QUERY', '24', 'DATASETS.SYNTH_GALLERY', '131', "
            "'DATASETS.SYNTH_IDS', '6', 'TEST.IMS_PER_BATCH', '16', 'TEST.RE_RANKING', {rr!r}, 'MODEL.SIE_CAMERA', 'True']); "
            "print('RESULT', float(r[0]), float(r[1]))").format(pkg=os.path.join(ROOT, "mp-reid_amd"), rr=rerank)

    def run(world):
        from conftest import run_ranks
        if world == 1:   # a plain process: no rank variables at all
            env = dict(os.environ, MPREID_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:]
            return [r.stdout]
        return run_ranks([sys.executable, "-c", code], world, 900, dict(MPREID_DIST_BACKEND="gloo", OMP_NUM_THREADS="4"),
                         local_rank=lambda r: 0, capture_dir=tmp_path)

    single = run(1)[0]
    multi = run(2)
    res = [ln for ln in single.splitlines() if ln.startswith("RESULT")]
    assert len(res) == 1
    for o in multi:
        assert [ln for ln in o.splitlines() if ln.startswith("RESULT")] == res
    pick = lambda o: [ln.split("transreid.test INFO: ")[1] for ln in o.splitlines() if "mAP:" in ln or "CMC curve" in ln]   # noqa: E731
    assert pick(single) == pick(multi[0]) and len(pick(single)) == 4 and pick(multi[1]) == []


RCCL_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "mp-reid_amd")]
    import torch.distributed as dist
    from mpreid import distributed as D, ops, synth
    from utils.metrics import R1_mAP_eval
    rank, world, local = D.init_from_env("nccl") if int(os.environ["WORLD_SIZE"]) > 1 else (0, 1, 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)      # ONE rank on the real RCCL backend
    assert dist.get_backend() == "nccl" and D.sharded_active()
    res = {{}}
    for ci, (n, nq, d, rerank) in enumerate({cases!r}):
        f, pid = synth.clustered_features(n, d, 2.5, seed=77 + n, per_id=6, normalize=False)
        cam = synth.labels_for(n)
        ev = R1_mAP_eval(nq, max_rank=50, feat_norm='yes', reranking=rerank)
        ev.reset()
        ev.update((torch.from_numpy(f).cuda(), tuple(int(p) for p in pid), tuple(int(c) for c in cam)))
        cmc, mAP, distmat, pids, camids, qf, gf = ev.compute()
        res.update({{f"cmc{{ci}}": cmc, f"map{{ci}}": np.float64(mAP), f"dist{{ci}}": distmat, f"qf{{ci}}": qf.numpy(), f"gf{{ci}}": gf.numpy()}})
    np.savez(os.path.join(sys.argv[1], "rccl.npz"), **res)
    dist.barrier(); dist.destroy_process_group()
""")


def test_sharded_evaluator_on_the_rccl_backend_one_rank(tmp_path):
    """RCCL refuses two ranks per device, so the multi-rank tests above stage their collectives through gloo.  This one runs
    the SAME sharded code path -- all_gather_into_tensor, the all_to_all_single of column_to_row_blocks, the tensor gather
    into the pinned host matrix, the CSR all-gather of the sparse rows, the evaluator's metadata exchange -- on the real
    `nccl` (= RCCL) backend with a one-rank group (MPREID_DIST_FORCE_COLLECTIVES=1 disables the single-process shortcuts):
    device tensors, dtypes and split sizes as an 8-GPU run issues them.  Bar: the single-process result byte for byte."""
    import torch
    from mpreid import synth
    from utils.metrics import R1_mAP_eval
    cases = [(900, 150, 192, False), (2600, 500, 128, True)]
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.format(root=ROOT, cases=cases))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(__import__("conftest").free_port()), MPREID_DIST_FORCE_COLLECTIVES="1")
    env.pop("MPREID_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, str(script), str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "rccl.npz")
    for ci, (n, nq, d, rerank) in enumerate(cases):
        f, pid = synth.clustered_features(n, d, 2.5, seed=77 + n, per_id=6, normalize=False)
        cam = synth.labels_for(n)
        ev = R1_mAP_eval(nq, max_rank=50, feat_norm='yes', reranking=rerank)
        ev.reset()
        ev.update((torch.from_numpy(f).cuda(), tuple(int(p) for p in pid), tuple(int(c) for c in cam)))
        cmc, mAP, distmat, pids, camids, qf, gf = ev.compute()
        assert np.array_equal(got[f"cmc{ci}"], cmc) and float(got[f"map{ci}"]) == float(mAP), ci
        assert np.array_equal(got[f"dist{ci}"], distmat), ci
        assert np.array_equal(got[f"qf{ci}"], qf.numpy()) and np.array_equal(got[f"gf{ci}"], gf.numpy()), ci


@pytest.mark.parametrize("dim", [0, 1])
def test_concat_parts_to_host_ragged_device_pieces(dim):
    """what rank 0 does with the pieces an RCCL gather delivered: ragged, padded DEVICE tensors -> one contiguous device
    matrix -> ONE contiguous D2H copy into the pinned host matrix (dim = 1 used to copy into strided host views)"""
    import torch
    from mpreid import distributed as D
    from test_distributed_cpu import _ragged_parts
    parts, sizes, other, want = _ragged_parts(dim, torch.device("cuda", 0))
    got = D._concat_parts_to_host(parts, sizes, dim, other)
    assert got.shape == want.shape and np.array_equal(got, want)
    view = D._concat_parts_to_host(parts, sizes, dim, other, reuse_buffer=True)
    assert np.array_equal(view, want)


# ---- the reference's OWN test.py shape under a launcher: no process group is initialised by the caller ------------------------
# reference test.py:10-65 parses the config, sets CUDA_VISIBLE_DEVICES = cfg.MODEL.DEVICE_ID (:39), builds loader and model,
# load_param(TEST.WEIGHT) and calls do_inference -- it never touches torch.distributed.  This script is that flow, written
# against the drop-in modules; processor.do_inference must bring the ranks up by itself (mpreid.distributed.ensure_group_from_env).
REF_SHAPED_TEST_PY = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {pkg!r}); os.chdir({pkg!r})
    from config import cfg_base as cfg
    from datasets.make_dataloader import make_dataloader
    from model.make_model import make_model
    from processor.processor import do_inference
    from utils.logger import setup_logger
    cfg.merge_from_list(sys.argv[1:])
    cfg.freeze()
    logger = setup_logger("transreid", cfg.OUTPUT_DIR, if_train=False)
    os.environ['CUDA_VISIBLE_DEVICES'] = cfg.MODEL.DEVICE_ID
    train_loader, train_loader_normal, val_loader, num_query, num_classes, camera_num, view_num = make_dataloader(cfg)
    model = make_model(cfg, num_class=num_classes, camera_num=camera_num, view_num=view_num)
    model.load_param(cfg.TEST.WEIGHT)
    r = do_inference(cfg, model, val_loader, num_query)
    print('RESULT', float(r[0]), float(r[1]))
""")
REF_OPTS = ['DATASETS.SYNTH_QUERY', '24', 'DATASETS.SYNTH_GALLERY', '131', 'DATASETS.SYNTH_IDS', '6', 'TEST.IMS_PER_BATCH', '16',
            'MODEL.SIE_CAMERA', 'True']


def _ref_shaped_setup(tmp_path):
    """(script path, weight path): the harness above and a checkpoint in the reference's layout (a state dict)"""
    import torch
    from config import cfg_base
    from model.make_model import make_model
    cfg = cfg_base.clone()
    cfg.defrost()
    cfg.merge_from_list(REF_OPTS + ['MODEL.INIT_SEED', '11'])
    cfg.freeze()
    w = tmp_path / "weights.pth"
    torch.save(make_model(cfg, num_class=6, camera_num=6, view_num=1).state_dict(), w)
    script = tmp_path / "ref_shaped_test.py"
    script.write_text(REF_SHAPED_TEST_PY.format(pkg=os.path.join(ROOT, "mp-reid_amd")))
    return script, w


def _result_lines(text):
    return [ln for ln in text.splitlines() if ln.startswith("RESULT")]


def _metric_lines(text):
    return [ln.split("transreid.test INFO: ")[1] for ln in text.splitlines() if "mAP:" in ln or "CMC curve" in ln]


@pytest.mark.parametrize("rerank", ["False", "True"])
def test_reference_shaped_test_py_needs_no_explicit_init(tmp_path, rerank):
    """two gloo-staged ranks (one GPU) through the reference-shaped harness: do_inference initialises the group itself, shards
    the loader and every rank prints the single-process (Rank-1, Rank-5); rank 0 alone logs the metric lines"""
    from conftest import run_ranks
    script, w = _ref_shaped_setup(tmp_path)
    cmd = [sys.executable, str(script)] + REF_OPTS + ['TEST.WEIGHT', str(w), 'TEST.RE_RANKING', rerank, 'MODEL.DEVICE_ID', "('0')"]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MPREID_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    single = r.stdout
    multi = run_ranks(cmd, 2, 900, dict(MPREID_DIST_BACKEND="gloo", OMP_NUM_THREADS="4"), local_rank=lambda r_: 0,
                      capture_dir=tmp_path)
    assert len(_result_lines(single)) == 1
    for o in multi:
        assert _result_lines(o) == _result_lines(single)


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs >= 2 visible GPUs: real RCCL ranks (one per device) cannot share a device")
def test_real_rccl_ranks_end_to_end(tmp_path):
    """Switches itself on the day >= 2 devices are visible (every other multi-rank test of this suite is gloo-staged, one-rank
    nccl or emulated: RCCL with P > 1 has never run in the builder's environment).  P = min(8, devices) FRESH ranks per leg,
    started by a launcher that has not touched a GPU; a failing rank exits non-zero and takes the leg down; nothing re-execs.
      1. the reference-shaped test.py under `python -m torch.distributed.run` (MODEL.DEVICE_ID lists the devices): every
         rank prints the single-process result;
      2. R1_mAP_eval sharded over real RCCL ranks: rank 0's 7-tuple == the single-process compute() byte for byte
         (ragged shards, with and without re-ranking);
      3. bench.py --gpus P --workload synth --rerank (1/16 size): one line, rccl_ranks == P, all-gather figures present."""
    from conftest import free_port, run_ranks
    P = min(8, _visible_gpus())
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MPREID_DIST_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # 1. reference-shaped harness
    script, w = _ref_shaped_setup(tmp_path)
    opts = REF_OPTS + ['TEST.WEIGHT', str(w), 'TEST.RE_RANKING', 'True',
                       'MODEL.DEVICE_ID', "('%s')" % ",".join(str(i) for i in range(P))]
    r1 = subprocess.run([sys.executable, str(script)] + opts, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                        timeout=900)
    assert r1.returncode == 0, r1.stdout[-2000:]
    rP = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}",
                         "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)] + opts,
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    assert rP.returncode == 0, rP.stdout[-3000:]
    assert len(_result_lines(r1.stdout)) == 1 and _result_lines(rP.stdout) == _result_lines(r1.stdout) * P
    assert _metric_lines(rP.stdout) == _metric_lines(r1.stdout)          # rank 0 alone logs, and logs the same lines
    # 2. the evaluator's 7-tuple over real RCCL ranks
    worker = tmp_path / "eval_worker.py"
    worker.write_text(EVAL_WORKER.format(root=ROOT, cases=EVAL_CASES, device="local"))
    run_ranks([sys.executable, str(worker), str(tmp_path)], P, 1800, dict(OMP_NUM_THREADS="4"))
    _check_eval_npz(tmp_path / "eval.npz")
    # 3. the bench's own launcher
    rb = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(P), "--workload", "synth", "--rerank",
                         "--small", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], env=env,
                        cwd=str(tmp_path), capture_output=True, text=True, timeout=1800)
    assert rb.returncode == 0, rb.stderr[-3000:]
    lines = [ln for ln in rb.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == P and j["rccl_ranks"] == P and j["value"] > 0 and j["all_gather"]["bytes_per_step"] > 0

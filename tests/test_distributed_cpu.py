"""world_size-2 gloo test of the N > 1 path's host logic (runs on CPU, no GPU).

The GPU kernels are replaced by the oracle here (this is a test of the partition / all-gather /
host concatenation, not of the kernels): the sharded result must equal the 1-rank result bit for bit."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "mp-reid_amd")]
    from mpreid import distributed as D, synth
    from oracle import oracle as orc
    rank, world, local = D.init_from_env("gloo")
    nq, ng, d = 37, 101, 48
    f, _ = synth.clustered_features(nq + ng, d, 2.0, seed=3, per_id=5)
    q_lo, q_hi = D.shard_range(nq, rank, world)
    g_lo, g_hi = D.shard_range(ng, rank, world)
    # every rank "encodes" its query slice and its gallery shard, then normalises
    qf_local = torch.from_numpy(orc.l2_normalize(f[q_lo:q_hi] * 3.0))
    gf_local = orc.l2_normalize(f[nq + g_lo: nq + g_hi] * 3.0)
    qf = D.all_gather_rows(qf_local, nq)
    assert qf.shape == (nq, d)
    block = torch.from_numpy(orc.euclidean_distance(qf.numpy(), gf_local))
    full = D.gather_column_blocks_to_host(block, dst=0)
    # the exchange step of the sharded evaluation: column blocks -> row blocks, then sharded CMC / mAP.  The GPU ranking
    # kernel is replaced by a host restatement here: this is a test of the partition, the collectives and the reduction
    rows = D.column_to_row_blocks(block, nq, D.shard_sizes(ng, world))
    import utils.metrics as M
    def rows_host(dist_rows, q_pids, g_pids, max_rank):
        d = dist_rows.numpy()
        hits, ap, nv = np.zeros(max_rank, np.float32), [], 0
        for i in range(d.shape[0]):
            order = np.argsort(d[i], kind="stable")
            m = (g_pids[order] == q_pids[i])
            if not m.any():
                continue
            nv += 1
            c = np.cumsum(m)
            hits += (c[:max_rank] > 0).astype(np.float32)
            pos = np.nonzero(m)[0]
            ap.append((np.arange(1, pos.size + 1) / (pos + 1.0)).sum() / pos.size)
        return hits, np.asarray(ap, np.float64), nv
    M._eval_rows_device = rows_host
    rng = np.random.default_rng(5)
    pid = rng.integers(0, 9, size=nq + ng)
    pid[3] = 1000    # a query without any match (skipped by eval_func)
    cmc, mAP = M.eval_func_sharded(rows, pid[q_lo:q_hi], pid[nq:], max_rank=50)
    rows_full = D.gather_row_blocks_to_host(rows, dst=0)
    if rank == 0:
        np.savez(sys.argv[1], full=full, rows_full=rows_full, cmc=cmc, mAP=mAP, pid=pid)
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
""")


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_distmat_and_eval_match_single_rank(tmp_path, world):
    from mpreid import distributed as D, synth
    from oracle import oracle as orc
    assert D.shard_sizes(10, 4) == [3, 3, 2, 2] and D.shard_range(10, 3, 4) == (8, 10)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    out = tmp_path / "full.npz"
    from conftest import run_ranks
    run_ranks([sys.executable, str(script), str(out)], world, 300, dict(OMP_NUM_THREADS="2"))
    nq, ng, d = 37, 101, 48
    f, _ = synth.clustered_features(nq + ng, d, 2.0, seed=3, per_id=5)
    want = orc.euclidean_distance(orc.l2_normalize(f[:nq] * 3.0), orc.l2_normalize(f[nq:] * 3.0))
    got = np.load(out)
    assert got["full"].shape == (nq, ng) and np.array_equal(got["full"], want)
    assert np.array_equal(got["rows_full"], want)        # column blocks -> row blocks -> host: the same matrix
    # sharded CMC / mAP == the unsharded host eval_func, bit for bit (integer hit counts, AP lists in query order)
    import utils.metrics as M
    pid = got["pid"]
    cmc, mAP = M.eval_func(want, pid[:nq], pid[nq:], None, None)
    assert np.array_equal(got["cmc"], cmc) and abs(float(got["mAP"]) - mAP) <= 1e-15


def test_bench_parent_launches_ranks_without_touching_the_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent spawns 2 fresh rank processes itself (it must not
    assert on WORLD_SIZE, import the HIP library or re-exec).  Without a GPU the ranks fail loudly and the parent
    reports that with exit code 1 instead of hanging or printing a bogus line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check of the launcher's failure path (the success path is tests/test_gpu_distributed.py)")
    env = dict(os.environ, MPREID_ALLOW_SHARED_GPU="1")   # (without it `--gpus 2` exits 2 up front: tests/test_bench_line.py)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--small", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 1
    assert "child ranks exited with" in r.stderr and "AssertionError: --gpus" not in r.stderr
    assert r.stdout.strip() == ""


def test_sharded_loader_paths_cover_every_sample_once():
    """processor.shard_val_loader: a rank's shard = its query range then its gallery range, in order, for the three loader
    kinds (a loader that shards itself, a torch DataLoader -> Subset, any other iterable -> filtered); the union over the
    ranks is every sample exactly once (host logic only: no GPU, no process group -- rank / world are passed explicitly)."""
    import torch
    from datasets.make_dataloader import SyntheticValLoader
    from mpreid import distributed as D
    from processor.processor import _ShardedLoader

    nq, ng, batch, world = 10, 37, 8, 3
    syn = SyntheticValLoader(nq, ng, 5, (32, 16), batch, 3)
    full = list(syn)
    img_all = torch.cat([b[0] for b in full])
    pid_all = [p for b in full for p in b[1]]
    path_all = [p for b in full for p in b[5]]

    class Plain:                       # an iterable without .shard / .dataset: the filter path
        def __iter__(self):
            return iter(full)

    class DS(torch.utils.data.Dataset):   # a map-style dataset behind a real DataLoader: the Subset path
        def __len__(self):
            return nq + ng

        def __getitem__(self, i):
            return img_all[i], int(pid_all[i]), 0, 0, path_all[i]

    def collate(batch_):
        imgs, pids, cams, views, paths = zip(*batch_)
        return (torch.stack(imgs), pids, cams, torch.tensor(cams), torch.tensor(views), paths)

    dl = torch.utils.data.DataLoader(DS(), batch_size=batch, shuffle=False, collate_fn=collate)
    seen = {"synthetic": [], "plain": [], "dataloader": []}
    for rank in range(world):
        q_lo, q_hi = D.shard_range(nq, rank, world)
        g_lo, g_hi = D.shard_range(ng, rank, world)
        idx = list(range(q_lo, q_hi)) + list(range(nq + g_lo, nq + g_hi))
        for name, ld in (("synthetic", syn), ("plain", Plain()), ("dataloader", dl)):
            got = list(_ShardedLoader(ld, idx))
            gi = torch.cat([b[0] for b in got])
            assert torch.equal(gi, img_all[idx]), (name, rank)
            assert [p for b in got for p in b[1]] == [pid_all[i] for i in idx], (name, rank)
            assert [p for b in got for p in b[5]] == [path_all[i] for i in idx], (name, rank)
            assert all(len(b[1]) == b[3].shape[0] == b[4].shape[0] for b in got)
            seen[name] += idx
    for name, ids in seen.items():
        assert sorted(ids) == list(range(nq + ng)), name


def test_rank_launcher_terminates_survivors_when_a_rank_dies(tmp_path):
    """the tests' rank launcher (conftest.run_ranks; bench.py's launch_children follows the same rule): a rank that dies
    must not leave the others waiting in a collective -- they are terminated and the failure is reported at once"""
    import time
    from conftest import run_ranks
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(120)\n")
    t0 = time.time()
    with pytest.raises(AssertionError, match="rank 1 exited with code 3"):
        run_ranks([sys.executable, str(script)], 3, 100)
    assert time.time() - t0 < 30
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nprint('rank', os.environ['RANK'], os.environ['MASTER_ADDR'])\n")
    outs = run_ranks([sys.executable, str(ok)], 2, 60, capture_dir=tmp_path)
    assert [o.strip() for o in outs] == ["rank 0 127.0.0.1", "rank 1 127.0.0.1"]


def _ragged_parts(dim, device):
    import torch
    sizes = [5, 0, 3, 4]
    other = 7
    mx = max(sizes)
    g = torch.Generator().manual_seed(3)
    shape = (other, mx) if dim == 1 else (mx, other)
    parts = [torch.randn(shape, generator=g).to(device) for _ in sizes]
    want = np.concatenate([p.cpu().numpy()[:, :n] if dim == 1 else p.cpu().numpy()[:n] for p, n in zip(parts, sizes)], axis=dim)
    return parts, sizes, other, want


@pytest.mark.parametrize("dim", [0, 1])
def test_concat_parts_to_host_ragged_cpu(dim):
    """the host concatenation of padded ragged pieces (mpreid.distributed._concat_parts_to_host), host tensors"""
    from mpreid import distributed as D
    parts, sizes, other, want = _ragged_parts(dim, "cpu")
    got = D._concat_parts_to_host(parts, sizes, dim, other)
    assert got.shape == want.shape and np.array_equal(got, want)


ENSURE_WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "mp-reid_amd")]
    import torch.distributed as dist
    from mpreid import distributed as D
    # the REFERENCE-shaped caller: nothing here initialises a process group (reference test.py:39,65 never does)
    assert not dist.is_initialized()
    rank, world = D.ensure_group_from_env()
    assert dist.is_initialized() and dist.get_backend() == "gloo"
    assert (rank, world) == (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])) == D.rank_world()
    assert D.ensure_group_from_env() == (rank, world)          # idempotent: a second do_inference call re-uses the group
    import torch
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)
    assert int(t) == world * (world + 1) // 2
    dist.barrier(); dist.destroy_process_group()
""")


def test_do_inference_initialises_the_group_itself_under_a_launcher(tmp_path):
    """The reference's test.py never calls init_process_group; started unchanged under torch.distributed.run its ranks reach
    do_inference with WORLD_SIZE / RANK / LOCAL_RANK set and no group.  processor.do_inference's first step
    (mpreid.distributed.ensure_group_from_env) initialises it: two gloo ranks here, no explicit init in the caller."""
    script = tmp_path / "ensure_worker.py"
    script.write_text(ENSURE_WORKER.format(root=ROOT))
    from conftest import run_ranks
    run_ranks([sys.executable, str(script)], 2, 300, dict(OMP_NUM_THREADS="2", MPREID_DIST_BACKEND="gloo"))


def test_ranks_that_share_one_visible_device_are_refused(monkeypatch):
    """WORLD_SIZE = 8 with MODEL.DEVICE_ID '0' (the shipped YAMLs; the reference's test.py turns it into
    CUDA_VISIBLE_DEVICES): every rank would see ONE device.  RCCL cannot run them and a silent fallback would evaluate
    everything eight times on one GPU: a RuntimeError that names the fix instead.  Without WORLD_SIZE nothing happens."""
    import torch
    from mpreid import distributed as D
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert D.ensure_group_from_env() == (0, 1)
    monkeypatch.setenv("WORLD_SIZE", "8"), monkeypatch.setenv("RANK", "3"), monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("MPREID_DIST_BACKEND", "nccl")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match=r"sees 1 HIP device.*MODEL\.DEVICE_ID"):
        D.ensure_group_from_env()
    assert not torch.distributed.is_initialized()

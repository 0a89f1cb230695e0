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
    port = str(29700 + os.getpid() % 200)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, MPREID_DIST_BACKEND="gloo", OMP_NUM_THREADS="4", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)] + args, env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0


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
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["all_gather"]["bytes_per_step"] > 0
    assert j["scaling"] == ("weak" if args[1] == "market" else "strong")

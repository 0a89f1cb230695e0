import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mp-reid_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def map_noise_envelope(orc, feats, rel, pid, nq, rerank, k1, k2, seeds=6):
    """How far do mAP / Rank-1 move when `feats` carry a RANDOM error of relative size `rel` (Gaussian, relative L2 = rel)?
    The largest |dmAP| and |dRank-1| over `seeds` perturbations, through the oracle's normalise -> distance / re-ranking ->
    eval_func pipeline.  The image -> mAP tests hold an encoder whose measured feature error is `rel` to a small multiple
    of this envelope: on nearly parallel features (the "degenerate" sets) a 1e-6 feature error legitimately moves mAP by
    1e-4 ... 1e-3, and the envelope says how much WITHOUT relying on one lucky realisation of the reference's own rounding."""
    base = orc.l2_normalize(np.asarray(feats, np.float32))
    d0 = orc.re_ranking(base[:nq], base[nq:], k1, k2, 0.3) if rerank else orc.euclidean_distance(base[:nq], base[nq:])
    cmc0, map0 = orc.eval_func(d0, pid[:nq], pid[nq:])
    scale = rel * float(np.linalg.norm(feats)) / np.sqrt(feats.size)
    dmap = dr1 = 0.0
    for s in range(seeds):
        rng = np.random.default_rng(1000 + s)
        f = (np.asarray(feats, np.float64) + scale * rng.standard_normal(feats.shape)).astype(np.float32)
        fn = orc.l2_normalize(f)
        d = orc.re_ranking(fn[:nq], fn[nq:], k1, k2, 0.3) if rerank else orc.euclidean_distance(fn[:nq], fn[nq:])
        cmc, mp = orc.eval_func(d, pid[:nq], pid[nq:])
        dmap, dr1 = max(dmap, abs(mp - map0)), max(dr1, abs(float(cmc[0]) - float(cmc0[0])))
    return dmap, dr1


def free_port():
    """a port the kernel just handed out (bound to port 0, then released)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_ranks(cmd, world, timeout, extra_env=None, local_rank=lambda r: r, capture_dir=None):
    """Start `world` ranks of `cmd` (fresh child processes, rendezvous on 127.0.0.1), poll ALL of them, and as soon as one
    exits non-zero -- or the deadline passes -- terminate the others (a rank stuck in a collective with a dead peer
    would otherwise hang the test until the backend's own timeout, or forever with gloo).  Asserts that every rank
    returned 0.  capture_dir: every rank's stdout goes to a file there and the texts are returned (rank order)."""
    import subprocess
    import time
    port = str(free_port())
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank(r)), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, **(extra_env or {}))
        fh = open(os.path.join(str(capture_dir), "rank%d_of_%d.out" % (r, world)), "w+") if capture_dir else None
        files.append(fh)
        procs.append(subprocess.Popen(cmd, env=env, stdout=fh))

    def texts():
        out = []
        for fh in files:
            fh.seek(0)
            out.append(fh.read())
            fh.close()
        return out
    deadline = time.time() + timeout
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = "rank %d exited with code %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                return texts() if capture_dir else None
            if time.time() > deadline:
                failed = "timeout after %d s (exit codes so far: %s)" % (timeout, codes)
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
    raise AssertionError(failed)

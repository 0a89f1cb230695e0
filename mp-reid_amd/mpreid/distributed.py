"""One-process-per-GPU evaluation helpers (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).

Partition: the gallery is sharded row-wise across ranks; every rank also encodes a
1/P slice of the queries; the L2-normalised query features are exchanged with ONE all-gather
([nq, 1280] fp32 = 17 MB at Market-1501 scale, ~2 MB per link on the fully connected xGMI mesh);
each rank then owns the [nq, ng_local] column block of the distance matrix, and rank 0 (or the
caller) concatenates the blocks on the host.  No floating-point reduction crosses ranks, so the
result does not depend on the number of ranks.
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import List, Tuple

import torch
import torch.distributed as dist


# The collectives go through _pg(): torch.distributed's default process group -- or, in the tests only, an object with the
# same functions installed for the calling thread by use_group() (tests/emulated_group.py runs P virtual ranks as threads
# of ONE process on ONE GPU and delivers the collectives by device copies, so that the branches only a multi-rank RCCL run
# reaches -- ragged all_to_all_single splits, padded gathers, device-tensor all-gathers -- execute with P > 1).
_tls = threading.local()


def _pg():
    return getattr(_tls, "group", None) or dist


@contextlib.contextmanager
def use_group(group):
    """route this thread's collectives through `group` (same functions as torch.distributed: get_rank, get_world_size,
    get_backend, is_initialized, all_gather, all_gather_into_tensor, all_to_all_single, gather, all_reduce,
    all_gather_object, barrier)"""
    old = getattr(_tls, "group", None)
    _tls.group = group
    try:
        yield group
    finally:
        _tls.group = old


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MPREID_DIST_BACKEND=gloo: debugging aid (several ranks sharing one GPU, CPU-staged collectives)
            backend = os.environ.get("MPREID_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, local


def ensure_group_from_env(logger=None) -> Tuple[int, int]:
    """(rank, world).  Called by the drop-in ``do_inference``: the REFERENCE's test.py (test.py:39,65) never initialises a
    process group -- it goes multi-device inside do_inference (nn.DataParallel, processor/processor.py:178-182).  Started
    unchanged under ``python -m torch.distributed.run --nproc-per-node P test.py ...`` every rank arrives here with
    WORLD_SIZE / RANK / LOCAL_RANK set and no group: initialise it (RCCL; MPREID_DIST_BACKEND=gloo for staging) and bind the
    rank to cuda:LOCAL_RANK, instead of silently evaluating everything P times on one device.  A rank that cannot see one
    device per local rank raises -- the reference sets CUDA_VISIBLE_DEVICES = cfg.MODEL.DEVICE_ID (test.py:39; '0' in the
    shipped YAMLs): pass MODEL.DEVICE_ID "('0,1,2,3,4,5,6,7')" (INTEGRATION.md section C)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return rank_world()
    backend = os.environ.get("MPREID_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    ndev = torch.cuda.device_count()
    if backend == "nccl":
        if ndev < local_world or local >= ndev:
            raise RuntimeError(
                f"WORLD_SIZE={world} (LOCAL_RANK={local} of {local_world} on this node) but this process sees {ndev} HIP "
                f"device(s) (CUDA_VISIBLE_DEVICES={os.environ.get('CUDA_VISIBLE_DEVICES')!r}, HIP_VISIBLE_DEVICES="
                f"{os.environ.get('HIP_VISIBLE_DEVICES')!r}): one process per GPU needs one visible device per local rank and "
                "RCCL cannot run several ranks on one device.  The reference's test.py sets CUDA_VISIBLE_DEVICES = "
                "cfg.MODEL.DEVICE_ID: pass MODEL.DEVICE_ID \"('0,1,...')\" listing every GPU of the node")
        torch.cuda.set_device(local)
    elif ndev:
        torch.cuda.set_device(local % ndev)    # gloo staging: ranks may share a device
    rank, world, _ = init_from_env(backend)
    import atexit

    def _teardown():   # the caller (the reference's test.py) knows nothing about the group: leave no RCCL communicator behind
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
    atexit.register(_teardown)
    if logger is not None:
        logger.info("process group initialised by do_inference: rank {} of {} on cuda:{} ({})".format(
            rank, world, torch.cuda.current_device() if ndev else "-", backend))
    return rank, world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) slice of range(n) owned by `rank`; sizes differ by at most one"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


#: accounting of the row all-gathers (the data-path collective of the evaluation): bytes gathered per rank, calls and --
#: while comm_stats["timing"] is set -- hipEvent pairs around each call on its launch stream (bench.py reads them)
comm_stats = {"bytes": 0, "calls": 0, "events": [], "timing": False}


def comm_stats_reset(timing: bool = False):
    comm_stats.update(bytes=0, calls=0, events=[], timing=bool(timing))


def all_gather_rows(x: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate the row shards of every rank (shard sizes from shard_sizes(n_total, world)).
    Ragged shards are padded to the largest one for the collective and trimmed afterwards."""
    if _single():
        return x
    if x.dtype == torch.int16:   # fp16 bit patterns: neither NCCL/RCCL nor gloo has a 16-bit integer type
        x2 = x.contiguous().view(torch.uint8)
        return all_gather_rows(x2, n_total).view(torch.int16)
    timed = comm_stats["timing"] and x.is_cuda
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    out = _all_gather_rows(x, n_total)
    if timed:
        e1.record()
        comm_stats["events"].append((e0, e1))
    comm_stats["bytes"] += out.numel() * out.element_size()
    comm_stats["calls"] += 1
    return out


def _all_gather_rows(x: torch.Tensor, n_total: int) -> torch.Tensor:
    world = _pg().get_world_size()
    sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    assert x.shape[0] == sizes[_pg().get_rank()], (x.shape, sizes)
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    if _pg().get_backend() == "gloo" and x.is_cuda:   # debug backend: stage through the host
        host = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(world)]
        _pg().all_gather(host, pad.cpu())
        out = torch.cat(host, dim=0).to(x.device)
    else:
        out = torch.empty((world * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _pg().all_gather_into_tensor(out, pad)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def rank_world() -> Tuple[int, int]:
    """(rank, world) of the default process group; (0, 1) without one"""
    if _pg().is_initialized():
        return _pg().get_rank(), _pg().get_world_size()
    return 0, 1


def _single() -> bool:
    """True when the single-process shortcuts apply.  MPREID_DIST_FORCE_COLLECTIVES=1 (tests) sends a ONE-rank process group
    through every collective instead, so that the RCCL branches (device tensors, all_to_all_single, gather, byte views) run
    on one GPU -- RCCL refuses two ranks per device, so this is the only way to execute them without a second GPU."""
    if not _pg().is_initialized():
        return True
    return _pg().get_world_size() == 1 and os.environ.get("MPREID_DIST_FORCE_COLLECTIVES") != "1"


def sharded_active() -> bool:
    """the evaluator / loader take their multi-rank path"""
    return not _single()


_pinned = {}


def _pinned_matrix(rows: int, cols: int, dtype, tag: str) -> torch.Tensor:
    """grow-only page-locked host buffer (D2H copies into pageable memory go through a staging copy)"""
    need = rows * cols
    buf = _pinned.get((tag, dtype))
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1), dtype=dtype)
        if torch.cuda.is_available():
            buf = buf.pin_memory()
        _pinned[(tag, dtype)] = buf
    return buf[:need].view(rows, cols)


def _gather_blocks_to_host(block: torch.Tensor, dim: int, dst: int, reuse_buffer: bool = False):
    """The host concatenation (north_star: "per-shard distance blocks concatenated on the host"): every rank's block
    travels as a TENSOR (RCCL gather over xGMI to `dst`'s HBM; host tensors with the gloo debug backend) and `dst` copies the
    pieces straight into ONE page-locked host matrix -- no pickling, no per-piece numpy temporaries.  Blocks may be ragged
    along `dim` (padded to the largest for the collective).  Returns the matrix on `dst` (None elsewhere): a fresh numpy
    array the caller owns, or -- reuse_buffer=True -- a view of the pinned matrix itself, valid until the next call."""
    rank, world = rank_world()
    if _single():
        host = _pinned_matrix(block.shape[0], block.shape[1], block.dtype, f"cat{dim}")
        host.copy_(block, non_blocking=block.is_cuda)
        if block.is_cuda:
            torch.cuda.current_stream().synchronize()
        return host.numpy() if reuse_buffer else host.numpy().copy()
    other = 1 - dim
    sz = torch.tensor([block.shape[dim]], dtype=torch.int64, device=block.device if _pg().get_backend() != "gloo" else "cpu")
    sizes = [torch.empty_like(sz) for _ in range(world)]
    _pg().all_gather(sizes, sz)
    sizes = [int(t.item()) for t in sizes]
    mx = max(max(sizes), 1)
    shape = list(block.shape)
    shape[dim] = mx
    staged = _pg().get_backend() == "gloo"
    pad = torch.zeros(shape, dtype=block.dtype, device="cpu" if staged else block.device)
    pad.narrow(dim, 0, block.shape[dim]).copy_(block)
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    _pg().gather(pad, parts, dst=dst)
    if rank != dst:
        return None
    return _concat_parts_to_host(parts, sizes, dim, block.shape[other], reuse_buffer)


def _concat_parts_to_host(parts, sizes, dim: int, other_len: int, reuse_buffer: bool = False):
    """the gathered (padded) pieces -> ONE page-locked host matrix.  Device pieces are first laid side by side in one
    contiguous DEVICE matrix (a strided D2D copy per piece, HBM speed) so that a single contiguous D2H copy lands in the
    pinned matrix: a non_blocking copy_ into a strided view of a CPU tensor would make torch allocate a pageable
    temporary per piece and finish with a CPU-to-CPU copy (the ragged column-block case, dim = 1)."""
    total = sum(sizes)
    dtype = parts[0].dtype
    rows, cols = (other_len, total) if dim == 1 else (total, other_len)
    host = _pinned_matrix(rows, cols, dtype, f"cat{dim}")
    on_dev = parts[0].is_cuda
    dst = torch.empty((rows, cols), dtype=dtype, device=parts[0].device) if on_dev else host
    lo = 0
    for p, n in zip(parts, sizes):
        if n:
            dst.narrow(dim, lo, n).copy_(p.narrow(dim, 0, n))
        lo += n
    if on_dev:
        assert dst.is_contiguous() and host.is_contiguous()
        host.copy_(dst, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    return host.numpy() if reuse_buffer else host.numpy().copy()


def gather_column_blocks_to_host(block: torch.Tensor, dst: int = 0, reuse_buffer: bool = False):
    """Host-side concatenation of the per-rank [nq, ng_local] distance blocks (gallery shards = column blocks).
    Returns the full numpy matrix on `dst`, None elsewhere."""
    return _gather_blocks_to_host(block, 1, dst, reuse_buffer)


def column_to_row_blocks(block: torch.Tensor, nq: int, ng_sizes: List[int]) -> torch.Tensor:
    """The one exchange step of the sharded evaluation without re-ranking: every rank holds the COLUMN block
    [nq, ng_local] of the distance matrix (its gallery shard) and the ranking statistics need whole ROWS.  One
    all-to-all (RCCL over xGMI; each rank sends (P-1)/P of its block) turns the column blocks into the row blocks
    [nq_local, ng] of shard_range(nq, rank, world): rank s sends rows [q_lo_r, q_hi_r) of its block to rank r."""
    rank, world = rank_world()
    if _single():
        return block
    assert block.shape == (nq, ng_sizes[rank]), (block.shape, nq, ng_sizes)
    q_sizes = shard_sizes(nq, world)
    nql = q_sizes[rank]
    ng = sum(ng_sizes)
    out = torch.empty((nql, ng), dtype=block.dtype, device=block.device)
    if _pg().get_backend() == "gloo":   # debug backend (CPU tests, several ranks on one GPU): all-gather, keep own rows
        mx = max(max(ng_sizes), 1)
        pad = torch.zeros((nq, mx), dtype=block.dtype)
        pad[:, :block.shape[1]] = block.cpu()
        parts = [torch.empty_like(pad) for _ in range(world)]
        _pg().all_gather(parts, pad)
        q_lo, q_hi = shard_range(nq, rank, world)
        lo = 0
        for p, n in zip(parts, ng_sizes):
            out[:, lo:lo + n] = p[q_lo:q_hi, :n].to(block.device)
            lo += n
        return out
    send = block.contiguous().view(-1)
    recv = torch.empty(nql * ng, dtype=block.dtype, device=block.device)
    _pg().all_to_all_single(recv, send, output_split_sizes=[nql * n for n in ng_sizes],
                           input_split_sizes=[q * ng_sizes[rank] for q in q_sizes])
    lo = off = 0
    for n in ng_sizes:
        out[:, lo:lo + n] = recv[off:off + nql * n].view(nql, n)
        lo += n
        off += nql * n
    return out


# ----------------------------------------------------------------------------------------------
# Row-sharded k-reciprocal re-ranking (SURVEY.md §8e, row "Re-rank")
#
#   phase 1  rows [r_lo, r_hi) of D = all-pairs distance, row maxima, first max(k1+1, k2) neighbours
#            -> ALL-GATHER rank table [N][KR] int32 (+ one column: the float bits of the row maxima)
#   phase 2  k-reciprocal expansion -> sparse V rows of the local rows
#            -> ALL-GATHER V (CSR transport: row counts + the nnz entries of 6 bytes, all_gather_sparse_rows)
#   phase 3  local query expansion of the local rows (skipped when k2 == 1)
#            -> ALL-GATHER V_qe
#   phase 4  queries sharded nq/P: distance rows of the local queries over the gallery columns; inverted index of V_qe
#            (gallery rows) built BY COLUMN SHARD (round 5: rank r counts and fills the columns shard_range(N, r, P) only)
#            -> ALL-GATHER column counts [N] u32, the packed index pieces (4 B per entry, contiguous in rank order) and the
#               rows of the chunk-boundary table; then Jaccard + blend of the local queries over the assembled index
#            -> each rank owns final_dist[q_lo:q_hi, nq:]; row blocks are concatenated on the host
#
# Phases 1-2 have a SPARSE form (default when N >= 2048 and max(k1+1, k2) <= 64): no [N/P][N] distance block per rank --
# fp16 candidate GEMM of the local rows against all columns, certified exact refinement, on-the-fly distances in the
# expansion (csrc/rerank.hip); a rank whose rows cannot be certified uses the dense form by itself.
#
# Every row is produced by the same instruction sequence as in the single-GPU call and no floating-point
# reduction crosses ranks, so the result is bit-identical for any number of ranks
# (tests/test_gpu_rerank.py::test_sharded_rerank_is_rank_count_independent).
# ----------------------------------------------------------------------------------------------
def _rr_ptr(t):
    import ctypes as C
    return C.c_void_p(0 if t is None else t.data_ptr())


class _RerankShard:
    """state of one (real or virtual) rank"""

    def __init__(self, feat_all, norms_all, nq, k1, k2, lam, rank, world, algo=0):
        from . import _lib
        # _lib.RERANK_AUTO / RERANK_DENSE / RERANK_SPARSE for phases 1-2 of this rank; RERANK_SPARSE_SPLIT3: sparse phases
        # and the blend term's distance rows of phase 4 from the fp16 matrix cores (outputs within 1e-6)
        self.split3_rows = algo == _lib.RERANK_SPARSE_SPLIT3
        self.algo = _lib.RERANK_SPARSE if self.split3_rows else algo
        self.sparse = False
        self.L = _lib.load()
        self.lib = _lib
        self.feat, self.norms = feat_all, norms_all
        self.N, self.d = feat_all.shape
        self.nq, self.k1, self.lam = nq, k1, lam
        self.k2 = min(k2, self.N)   # numpy clamps initial_rank[i, :k2]; np.mean averages the clamped row count
        self.KR = max(min(k1 + 1, self.N), self.k2)
        self.r_lo, self.r_hi = shard_range(self.N, rank, world)
        self.q_lo, self.q_hi = shard_range(nq, rank, world)
        self.rows = self.r_hi - self.r_lo
        self.ld = (self.N + 63) // 64 * 64
        self.dev = feat_all.device
        self.vcap = self.L.mpreid_rr_vcap(self.N, k1)
        self.rowmax_all = None   # [N] fp32 once the extended rank table has been gathered

    def phase1(self):
        t, dev = torch, self.dev
        self.rowmax = t.empty(max(self.rows, 1), dtype=t.float32, device=dev)
        rank_local = t.empty((self.rows, self.KR), dtype=t.int32, device=dev)
        lib = self.lib
        eligible = self.N >= 2048 and self.KR <= 64 and self.rows > 0
        if self.algo == lib.RERANK_SPARSE and not eligible:
            raise RuntimeError("the sparse phases need N >= 2048 and max(k1+1, k2) <= 64")
        if eligible and self.algo != lib.RERANK_DENSE:
            # sparse phase 1: no [rows][N] distance block; a rank whose rows cannot be certified falls back to the
            # dense phases by itself (same bits either way)
            self.rankd = t.empty((self.rows, self.KR), dtype=t.float32, device=dev)
            wsb = self.L.mpreid_rr_sparse_workspace_bytes(self.N, self.d, self.rows, self.KR)
            ws = t.empty(wsb, dtype=t.uint8, device=dev)
            rc = self.L.mpreid_rr_neighbours_sparse(_rr_ptr(self.feat), _rr_ptr(self.norms), self.N, self.d, self.r_lo,
                                                    self.rows, self.KR, _rr_ptr(rank_local), _rr_ptr(self.rowmax),
                                                    _rr_ptr(self.rankd), _rr_ptr(ws), ws.numel(), lib.stream_ptr())
            del ws
            if rc == 0:
                self.sparse = True
                return rank_local
            if rc != lib.ERR_RETRY_DENSE or self.algo == lib.RERANK_SPARSE:
                lib.check(rc, "mpreid_rr_neighbours_sparse")
        self.D = t.empty((max(self.rows, 1), self.ld), dtype=t.float32, device=dev)
        if self.rows:
            self.lib.check(self.L.mpreid_rr_dist_rows(_rr_ptr(self.feat), _rr_ptr(self.norms), self.N, self.d, self.r_lo,
                                                      self.rows, _rr_ptr(self.D), self.ld, _rr_ptr(self.rowmax),
                                                      _rr_ptr(rank_local), self.KR, self.lib.stream_ptr()),
                           "mpreid_rr_dist_rows")
        return rank_local

    def phase1_ext(self):
        """phase 1 + the row maxima as one more int32 column (their float bits): the all-gather of the rank table then
        carries them to the ranks that own those rows as QUERIES in phase 4 -- no fourth collective"""
        rank_local = self.phase1()
        rm = self.rowmax[:self.rows].contiguous().view(torch.int32).reshape(self.rows, 1)
        return torch.cat([rank_local, rm], dim=1).contiguous()

    @staticmethod
    def split_ext(ext, kr):
        """gathered [N][KR + 1] table -> (rank table [N][KR], row maxima [N] fp32)"""
        return ext[:, :kr].contiguous(), ext[:, kr].contiguous().view(torch.float32)

    def phase2(self, rank_all):
        t, dev = torch, self.dev
        self.rank_all = rank_all
        self.vcnt = t.zeros(self.rows, dtype=t.int32, device=dev)
        self.vidx = t.empty((max(self.rows, 1), self.vcap), dtype=t.int32, device=dev)
        self.vval = t.empty((max(self.rows, 1), self.vcap), dtype=t.int16, device=dev)
        if self.rows and self.sparse:
            scratch = t.empty(self.L.mpreid_rr_krecip_scratch_bytes(self.N), dtype=t.uint8, device=dev)
            self.lib.check(self.L.mpreid_rr_krecip_sparse(_rr_ptr(self.feat), _rr_ptr(self.norms), self.N, self.d,
                                                          _rr_ptr(self.rowmax), _rr_ptr(rank_all), _rr_ptr(self.rankd),
                                                          self.k1, self.KR, self.r_lo, self.rows, _rr_ptr(self.vcnt),
                                                          _rr_ptr(self.vidx), _rr_ptr(self.vval), _rr_ptr(scratch),
                                                          self.lib.stream_ptr()), "mpreid_rr_krecip_sparse")
        elif self.rows:
            scratch = t.empty(self.L.mpreid_rr_krecip_scratch_bytes(self.N), dtype=t.uint8, device=dev)
            self.lib.check(self.L.mpreid_rr_krecip(_rr_ptr(self.D), self.ld, self.N, _rr_ptr(self.rowmax),
                                                   _rr_ptr(rank_all), self.k1, self.KR, self.r_lo, self.rows,
                                                   _rr_ptr(self.vcnt), _rr_ptr(self.vidx), _rr_ptr(self.vval),
                                                   _rr_ptr(scratch), self.lib.stream_ptr()), "mpreid_rr_krecip")
        return int(self.vcnt.max().item()) if self.rows else 0

    def _pack(self, cnt, idx, val, width):
        t = torch
        oi = t.empty((self.rows, width), dtype=t.int32, device=self.dev)
        ov = t.empty((self.rows, width), dtype=t.int16, device=self.dev)
        if self.rows:
            self.lib.check(self.L.mpreid_rr_pack_rows(_rr_ptr(cnt), _rr_ptr(idx), _rr_ptr(val), self.rows, idx.shape[1],
                                                      width, _rr_ptr(oi), _rr_ptr(ov), self.lib.stream_ptr()),
                           "mpreid_rr_pack_rows")
        return oi, ov

    def pack_v(self, width):
        return (self.vcnt,) + self._pack(self.vcnt, self.vidx, self.vval, width)

    def phase3_count(self, vcnt_all, vidx_all, vval_all):
        t = torch
        self.V = (vcnt_all, vidx_all, vval_all)
        self.ucnt = t.zeros(self.rows, dtype=t.int32, device=self.dev)
        if self.rows:
            self.lib.check(self.L.mpreid_rr_qe_count(self.N, _rr_ptr(self.rank_all), self.KR, self.k2, self.r_lo, self.rows,
                                                     _rr_ptr(vcnt_all), _rr_ptr(vidx_all), vidx_all.shape[1],
                                                     _rr_ptr(self.ucnt), self.lib.stream_ptr()), "mpreid_rr_qe_count")
        return int(self.ucnt.max().item()) if self.rows else 0

    def phase3_fill(self, qcap):
        t = torch
        vcnt_all, vidx_all, vval_all = self.V
        qcnt = t.zeros(self.rows, dtype=t.int32, device=self.dev)
        qidx = t.zeros((self.rows, qcap), dtype=t.int32, device=self.dev)
        qval = t.zeros((self.rows, qcap), dtype=t.int16, device=self.dev)
        if self.rows:
            self.lib.check(self.L.mpreid_rr_qe_fill(self.N, _rr_ptr(self.rank_all), self.KR, self.k2, self.r_lo, self.rows,
                                                    _rr_ptr(vcnt_all), _rr_ptr(vidx_all), _rr_ptr(vval_all),
                                                    vidx_all.shape[1], qcap, _rr_ptr(qcnt), _rr_ptr(qidx), _rr_ptr(qval),
                                                    self.lib.stream_ptr()), "mpreid_rr_qe_fill")
        return qcnt, qidx, qval

    def phase4_rows(self):
        """exact (or split3) distance rows of this rank's queries over the gallery columns + their row maxima"""
        t, dev = torch, self.dev
        qrows = self.q_hi - self.q_lo
        self.D = None  # the row block of phase 1 (dense phases) is no longer needed
        if qrows == 0:
            self.dq = self.rmq = None
            return
        dq = t.empty((qrows, self.ld), dtype=t.float32, device=dev)
        if self.rowmax_all is not None:
            # the row maxima travelled with the rank table: only the GALLERY columns of the query rows are needed (the
            # blend of final_dist[:nq, nq:]), exact or -- RERANK_SPARSE_SPLIT3 -- from the fp16 matrix cores
            from . import ops
            rmq = self.rowmax_all[self.q_lo:self.q_hi].contiguous()
            ops.euclidean_distance(self.feat[self.q_lo:self.q_hi], self.feat[self.nq:],
                                   mode=ops.GEMM_F16_SPLIT3 if self.split3_rows else ops.GEMM_F32_EXACT, out=dq,
                                   col_offset=self.nq)
        else:
            rmq = t.empty(qrows, dtype=t.float32, device=dev)
            self.lib.check(self.L.mpreid_rr_dist_rows(_rr_ptr(self.feat), _rr_ptr(self.norms), self.N, self.d, self.q_lo,
                                                      qrows, _rr_ptr(dq), self.ld, _rr_ptr(rmq), None, 0,
                                                      self.lib.stream_ptr()), "mpreid_rr_dist_rows")
        self.dq, self.rmq = dq, rmq

    def phase4(self, qcnt_all, qidx_all, qval_all):
        """phase 4 with the WHOLE inverted index built on this rank (one rank, or n * stride >= 2^32)"""
        t, dev = torch, self.dev
        qrows = self.q_hi - self.q_lo
        ng = self.N - self.nq
        out = t.empty((qrows, ng), dtype=t.float32, device=dev)
        self.phase4_rows()
        if qrows == 0:
            return out
        nnz = int(qcnt_all.sum().item())
        ccnt = t.empty(self.N + 1, dtype=t.int32, device=dev)
        cptr = t.empty(self.N + 1, dtype=t.int64, device=dev)
        crow = t.empty(max(nnz, 1), dtype=t.int32, device=dev)
        cval = t.empty(max(nnz, 1), dtype=t.int16, device=dev)
        chist = t.empty(self.L.mpreid_rr_jaccard_hist_bytes(self.N), dtype=t.uint8, device=dev)
        self.lib.check(self.L.mpreid_rr_jaccard(self.N, self.nq, self.q_lo, qrows, _rr_ptr(self.dq), self.ld, _rr_ptr(self.rmq),
                                                _rr_ptr(qcnt_all), _rr_ptr(qidx_all), _rr_ptr(qval_all),
                                                qidx_all.shape[1], float(self.lam), _rr_ptr(ccnt), _rr_ptr(cptr),
                                                _rr_ptr(crow), _rr_ptr(cval), _rr_ptr(chist), _rr_ptr(out), ng,
                                                self.lib.stream_ptr()), "mpreid_rr_jaccard")
        self.dq = self.rmq = None
        return out

    # -- phase 4 with the index build sharded by column range (include/mpreid.h: mpreid_rr_csc_*) ---------------------------
    def index_shardable(self, qstride: int) -> bool:
        return self.N * int(qstride) < (1 << 32)

    def phase4_count(self, qcnt_all, qidx_all, world, rank):
        """step 1: entries per column of this rank's column shard -> [cols] int32 (u32 bit patterns)"""
        t, dev = torch, self.dev
        self.c_lo, self.c_hi = shard_range(self.N, rank, world)
        self.col_rank = rank
        self.ccnt = t.zeros(self.N + 1, dtype=t.int32, device=dev)
        self.chist = t.empty(self.L.mpreid_rr_jaccard_hist_bytes(self.N), dtype=t.uint8, device=dev)
        self.lib.check(self.L.mpreid_rr_csc_count(self.N, self.nq, _rr_ptr(qcnt_all), _rr_ptr(qidx_all), qidx_all.shape[1],
                                                  self.c_lo, self.c_hi, _rr_ptr(self.chist), _rr_ptr(self.ccnt),
                                                  self.lib.stream_ptr()), "mpreid_rr_csc_count")
        return self.ccnt[self.c_lo:self.c_hi]

    def phase4_fill(self, ccnt_all, qcnt_all, qidx_all, qval_all, world):
        """step 2: global column pointers, this rank's piece of the packed index (at its global position in a buffer of the
        whole index's size) and its rows of the boundary table.  Returns (piece [entries] int32, hb rows [cols][nchunks + 1]
        int32, piece boundaries of all ranks: python list of world + 1 offsets)."""
        t, dev = torch, self.dev
        total = int(qcnt_all[self.nq:].sum().item())     # indexed entries = entries of the gallery rows
        nch = self.L.mpreid_rr_csc_chunks(self.N, self.nq)
        if nch <= 0:      # (negative: the library's argument error -- no indexed row, nq >= n)
            self.lib.check(nch, "mpreid_rr_csc_chunks")
        nb1 = nch + 1
        self.cptr = t.empty(self.N + 1, dtype=t.int64, device=dev)
        self.cpk = t.empty(max(total, 1), dtype=t.int32, device=dev)
        self.hb = t.empty((self.N, nb1), dtype=t.int32, device=dev)
        scratch = t.zeros(self.N + 1, dtype=t.int32, device=dev)
        scratch[:self.N] = ccnt_all                  # (consumed by the scan)
        self.lib.check(self.L.mpreid_rr_csc_fill(self.N, self.nq, _rr_ptr(qcnt_all), _rr_ptr(qidx_all), _rr_ptr(qval_all),
                                                 qidx_all.shape[1], self.c_lo, self.c_hi, _rr_ptr(scratch),
                                                 _rr_ptr(self.chist), _rr_ptr(self.cptr), _rr_ptr(self.cpk), _rr_ptr(self.hb),
                                                 self.lib.stream_ptr()), "mpreid_rr_csc_fill")
        self.chist = self.ccnt = None
        edges = [shard_range(self.N, r, world)[0] for r in range(world)] + [self.N]
        self.bounds = self.cptr[t.tensor(edges, dtype=t.int64, device=dev)].cpu().tolist()   # the collective's sizes: one sync
        assert self.bounds[-1] == total, (self.bounds[-1], total)
        return self.cpk[self.bounds[self.col_rank]:self.bounds[self.col_rank + 1]], self.hb[self.c_lo:self.c_hi]

    def phase4_assemble(self, pieces, hb_all, world):
        """the gathered pieces (rank order; this rank's own may be None) into the index buffer, the gathered boundary table"""
        me = self.col_rank
        for r, piece in enumerate(pieces):
            if r != me and self.bounds[r + 1] > self.bounds[r]:
                self.cpk[self.bounds[r]:self.bounds[r + 1]] = piece[:self.bounds[r + 1] - self.bounds[r]]
        self.hb = hb_all

    def phase4_jaccard(self, qcnt_all, qidx_all, qval_all):
        """step 3: Jaccard + blend of this rank's queries over the assembled index"""
        t = torch
        qrows = self.q_hi - self.q_lo
        ng = self.N - self.nq
        out = t.empty((qrows, ng), dtype=t.float32, device=self.dev)
        if qrows:
            self.lib.check(self.L.mpreid_rr_jaccard_indexed(self.N, self.nq, self.q_lo, qrows, _rr_ptr(self.dq), self.ld,
                                                            _rr_ptr(self.rmq), _rr_ptr(qcnt_all), _rr_ptr(qidx_all),
                                                            _rr_ptr(qval_all), qidx_all.shape[1], float(self.lam),
                                                            _rr_ptr(self.cptr), _rr_ptr(self.cpk), _rr_ptr(self.hb),
                                                            _rr_ptr(out), ng, self.lib.stream_ptr()),
                           "mpreid_rr_jaccard_indexed")
        self.dq = self.rmq = self.cpk = self.hb = self.cptr = None
        return out


# ---- CSR transport of the sparse rows (V, V_qe) ------------------------------------------------------------------------
# The phases produce ELL rows (fixed stride); what crosses xGMI is the CSR payload: the row counts (4 B per row) and the
# nnz entries (4 B index + 2 B value) of every rank back to back, padded only to the LARGEST RANK PAYLOAD -- not every row
# to the globally longest row (round 2: 434 MB for V_qe at N = 100 000 against 6 B x nnz = 210 MB).  Every rank unpacks the
# payloads into ELL rows of the global maximum length for the next phase's kernels.
def _csr_pack(lib, L, cnt, idx, val):
    """ELL rows of one shard -> (nnz, idx_csr [nnz], val_csr [nnz])"""
    t = torch
    rows = cnt.shape[0]
    dev = cnt.device
    rowptr = t.empty(rows + 1, dtype=t.int64, device=dev)
    lib.check(L.mpreid_rr_rowptr(_rr_ptr(cnt), rows, _rr_ptr(rowptr), lib.stream_ptr()), "mpreid_rr_rowptr")
    nnz = int(rowptr[rows].item())
    ci = t.empty(max(nnz, 1), dtype=t.int32, device=dev)
    cv = t.empty(max(nnz, 1), dtype=t.int16, device=dev)
    if rows:
        lib.check(L.mpreid_rr_ell_to_csr(_rr_ptr(rowptr), _rr_ptr(idx), _rr_ptr(val), rows, idx.shape[1], _rr_ptr(ci), _rr_ptr(cv),
                                         lib.stream_ptr()), "mpreid_rr_ell_to_csr")
    return nnz, ci, cv


def _csr_unpack(lib, L, cnt_all, payloads, row_ranges):
    """cnt_all [N] + per-rank payloads [(idx_csr, val_csr)] in rank order -> ELL (idx_all [N][W], val_all [N][W]), W = global max"""
    import ctypes as C
    t = torch
    N = cnt_all.shape[0]
    dev = cnt_all.device
    W = max(int(cnt_all.max().item()) if N else 0, 1)
    rowptr = t.empty(N + 1, dtype=t.int64, device=dev)
    lib.check(L.mpreid_rr_rowptr(_rr_ptr(cnt_all), N, _rr_ptr(rowptr), lib.stream_ptr()), "mpreid_rr_rowptr")
    starts = rowptr[t.tensor([lo for lo, _ in row_ranges], dtype=t.int64, device=dev)].cpu().tolist()
    idx_all = t.zeros((N, W), dtype=t.int32, device=dev)
    val_all = t.zeros((N, W), dtype=t.int16, device=dev)
    for (lo, hi), off, (ci, cv) in zip(row_ranges, starts, payloads):
        if hi <= lo:
            continue
        # rowptr holds GLOBAL offsets; the rank's payload starts at global offset `off`: shift the payload base instead of
        # rewriting the offsets (the kernel only ever touches [off, off + nnz_rank))
        lib.check(L.mpreid_rr_csr_to_ell(C.c_void_p(rowptr.data_ptr() + 8 * lo), C.c_void_p(ci.data_ptr() - 4 * off),
                                         C.c_void_p(cv.data_ptr() - 2 * off), hi - lo, W,
                                         C.c_void_p(idx_all.data_ptr() + 4 * lo * W), C.c_void_p(val_all.data_ptr() + 2 * lo * W),
                                         lib.stream_ptr()), "mpreid_rr_csr_to_ell")
    return idx_all, val_all


def all_gather_sparse_rows(cnt, idx, val, n_total):
    """All-gather of sparse row shards (shard sizes shard_sizes(n_total, world)) with CSR transport.  Returns
    (cnt_all [N], idx_all [N][W], val_all [N][W], bytes moved per rank)."""
    from . import _lib as lib
    L = lib.load()
    rank, world = rank_world()
    cnt_all = all_gather_rows(cnt, n_total)
    nnz, ci, cv = _csr_pack(lib, L, cnt, idx, val)
    ranges = [shard_range(n_total, r, world) for r in range(world)]
    if _single():
        return (cnt_all,) + _csr_unpack(lib, L, cnt_all, [(ci, cv)], ranges) + (0,)
    staged = _pg().get_backend() == "gloo"
    sz = torch.tensor([nnz], dtype=torch.int64, device="cpu" if staged else cnt.device)
    sizes = [torch.empty_like(sz) for _ in range(world)]
    _pg().all_gather(sizes, sz)
    mx = max(max(int(x.item()) for x in sizes), 1)
    payloads = []
    for buf in (ci, cv.view(torch.uint8)):        # (int16 has no collective type: bytes)
        per = buf.numel() // max(ci.numel(), 1)    # elements of `buf` per entry (1 for idx, 2 bytes for val)
        pad = torch.zeros(mx * per, dtype=buf.dtype, device=buf.device)
        pad[: nnz * per] = buf[: nnz * per]
        if staged and pad.is_cuda:
            host = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(world)]
            _pg().all_gather(host, pad.cpu())
            out = torch.cat(host).to(buf.device)
        else:
            out = torch.empty(world * mx * per, dtype=buf.dtype, device=buf.device)
            _pg().all_gather_into_tensor(out, pad)
        payloads.append([out[r * mx * per: (r + 1) * mx * per] for r in range(world)])
    pl = [(payloads[0][r], payloads[1][r].view(torch.int16)) for r in range(world)]
    idx_all, val_all = _csr_unpack(lib, L, cnt_all, pl, ranges)
    return cnt_all, idx_all, val_all, 4 * n_total + 6 * world * mx


def all_gather_ragged(x: torch.Tensor, sizes: List[int]) -> List[torch.Tensor]:
    """All-gather of 1-D pieces whose lengths `sizes` every rank already knows (the index pieces of phase 4: contiguous
    column ranges of the packed inverted index).  Padded to the largest piece for the collective; returns per-rank views
    (rank order) of the gathered buffer."""
    rank, world = rank_world()
    assert x.dim() == 1 and len(sizes) == world and x.shape[0] == sizes[rank], (x.shape, sizes, rank)
    mx = max(max(sizes), 1)
    timed = comm_stats["timing"] and x.is_cuda
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    pad = torch.zeros(mx, dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    if _pg().get_backend() == "gloo" and x.is_cuda:
        host = [torch.empty(mx, dtype=x.dtype) for _ in range(world)]
        _pg().all_gather(host, pad.cpu())
        out = torch.cat(host).to(x.device)
    else:
        out = torch.empty(world * mx, dtype=x.dtype, device=x.device)
        _pg().all_gather_into_tensor(out, pad)
    if timed:
        e1.record()
        comm_stats["events"].append((e0, e1))
    comm_stats["bytes"] += sum(sizes) * x.element_size()
    comm_stats["calls"] += 1
    return [out[r * mx: r * mx + sizes[r]] for r in range(world)]


_side_streams: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """one side stream per device, created once (workspaces are keyed by stream: ops._workspace)"""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def _rr_prepare(qf, gf):
    from . import ops
    dev = ops._lib.require_gpu()
    feat = torch.cat([ops._dev_f32(qf, dev), ops._dev_f32(gf, dev)], dim=0).contiguous()
    return feat, ops.sqnorm(feat)


def re_ranking_sharded(qf_all, gf_all, k1, k2, lambda_value, algo=0):
    """Re-ranking with the rows of the N x N problem sharded over the ranks of the default process group
    (every rank passes the full, all-gathered query and gallery features).  Returns this rank's
    final_dist[q_lo:q_hi, nq:] block on the GPU; use gather_row_blocks_to_host() for the full matrix."""
    world = _pg().get_world_size() if _pg().is_initialized() else 1
    rank = _pg().get_rank() if _pg().is_initialized() else 0
    if _single():
        # one GPU: the single call (symmetric distance GEMM: half the tiles) gives the same bits as the phases below
        # (tests/test_gpu_rerank.py::test_sharded_rerank_is_rank_count_independent)
        from . import ops
        return ops.re_ranking(qf_all, gf_all, k1, k2, lambda_value, algo=algo)[0]
    feat, norms = _rr_prepare(qf_all, gf_all)
    N, nq = feat.shape[0], qf_all.shape[0]
    sh = _RerankShard(feat, norms, nq, int(k1), int(k2), float(lambda_value), rank, world, algo)

    def gmax(v):
        if _single():
            return v
        tt = torch.tensor([v], dtype=torch.int64, device=feat.device)
        _pg().all_reduce(tt, op=dist.ReduceOp.MAX)
        return int(tt.item())

    rank_all, sh.rowmax_all = sh.split_ext(all_gather_rows(sh.phase1_ext(), N), sh.KR)
    sh.phase2(rank_all)
    vc, vi, vv, _ = all_gather_sparse_rows(sh.vcnt, sh.vidx, sh.vval, N)          # CSR transport: counts + nnz entries
    if k2 != 1:
        qcap = max(gmax(sh.phase3_count(vc, vi, vv)), 1)
        qc, qi, qv = sh.phase3_fill(qcap)
        vc, vi, vv, _ = all_gather_sparse_rows(qc, qi, qv, N)
    if os.environ.get("MPREID_RR_FULL_INDEX") == "1" or not sh.index_shardable(vi.shape[1]):
        # n * row stride >= 2^32, or the documented switch back to round 4's form (set it IDENTICALLY ON EVERY RANK: the
        # column-sharded build below adds three collectives): every rank builds the whole index, no further exchange
        return sh.phase4(vc, vi, vv)
    # phase 4: the index build sharded by column range; three all-gathers (counts, packed pieces, boundary rows)
    # The exact distance rows of the local queries (matrix cores, ~3 ms at N = 100 000 / P = 8) run on a side stream beside
    # the index build and its three all-gathers (HBM / LDS / xGMI): the Jaccard stage is the first to need both.
    main = torch.cuda.current_stream(feat.device)
    side = _side_stream(feat.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        sh.phase4_rows()
        for t_ in (sh.dq, sh.rmq):
            if t_ is not None:
                t_.record_stream(main)     # allocated under the side stream, read (and released) on the main one
    ccnt_all = all_gather_rows(sh.phase4_count(vc, vi, world, rank).contiguous(), N)
    piece, hb_rows = sh.phase4_fill(ccnt_all, vc, vi, vv, world)
    sizes = [sh.bounds[r + 1] - sh.bounds[r] for r in range(world)]
    pieces = all_gather_ragged(piece.contiguous(), sizes)
    sh.phase4_assemble(pieces, all_gather_rows(hb_rows.contiguous(), N), world)
    main.wait_stream(side)
    return sh.phase4_jaccard(vc, vi, vv)


def re_ranking_virtual(qf_all, gf_all, k1, k2, lambda_value, world, algo=0, timings=None):
    """The same phases for `world` VIRTUAL ranks executed one after the other on the current GPU (no process
    group): the all-gathers become concatenations.  Used to test rank-count independence on one GPU.
    timings: optional dict that receives, per phase, the list of per-rank wall times in ms (synchronised) and the
    bytes each all-gather would move -- the compute side of a `world`-GPU run, measured on one GPU."""
    import time

    def timed(name, fn):
        if timings is None:
            return fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        timings.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return out

    feat, norms = _rr_prepare(qf_all, gf_all)
    nq = qf_all.shape[0]
    shards = [_RerankShard(feat, norms, nq, int(k1), int(k2), float(lambda_value), r, world, algo) for r in range(world)]
    rank_all, rowmax_all = _RerankShard.split_ext(torch.cat([timed("phase1_neighbours", s.phase1_ext) for s in shards], dim=0),
                                                  shards[0].KR)
    for s in shards:
        s.rowmax_all = rowmax_all
    from . import _lib as lib
    Lc = lib.load()
    N = feat.shape[0]
    ranges = [shard_range(N, r, world) for r in range(world)]

    def virtual_gather(parts):
        """the CSR all-gather of all_gather_sparse_rows with concatenation in place of the collective; bytes as on the wire"""
        cnt_all = torch.cat([p[0] for p in parts], dim=0)
        packed = [_csr_pack(lib, Lc, *p) for p in parts]
        idx_all, val_all = _csr_unpack(lib, Lc, cnt_all, [(ci, cv) for _, ci, cv in packed], ranges)
        return cnt_all, idx_all, val_all, 4 * N + 6 * world * max(max(n for n, _, _ in packed), 1)

    for s in shards:
        timed("phase2_krecip", lambda s=s: s.phase2(rank_all))
    vc, vi, vv, bv = virtual_gather([(s.vcnt, s.vidx, s.vval) for s in shards])
    gathers = {"rank_table": rank_all.numel() * 4 + rowmax_all.numel() * 4, "V": bv,
               "V_ell_round2": vc.numel() * 4 + vi.numel() * 6}
    if k2 != 1:
        qcap = max(max(timed("phase3_qe_count", lambda s=s: s.phase3_count(vc, vi, vv)) for s in shards), 1)
        fills = [timed("phase3_qe_fill", lambda s=s: s.phase3_fill(qcap)) for s in shards]
        vc, vi, vv, bq = virtual_gather(fills)
        gathers["V_qe"] = bq
        gathers["V_qe_ell_round2"] = vc.numel() * 4 + vi.numel() * 6
    if world > 1 and shards[0].index_shardable(vi.shape[1]) and os.environ.get("MPREID_RR_FULL_INDEX") != "1":
        # column-sharded index build: per virtual rank count -> (gather) -> fill -> (gather) -> Jaccard; the per-rank phase 4
        # time is the sum of its three steps
        def step(s, name, fn):
            return timed(f"phase4_{name}", fn)
        for s in shards:
            step(s, "rows", s.phase4_rows)
        cc = torch.cat([step(s, "index_count", lambda s=s, r=r: s.phase4_count(vc, vi, world, r)) for r, s in enumerate(shards)])
        filled = [step(s, "index_fill", lambda s=s: s.phase4_fill(cc, vc, vi, vv, world)) for s in shards]
        hb_all = torch.cat([f[1] for f in filled], dim=0)
        pieces = [f[0] for f in filled]
        for s in shards:
            s.phase4_assemble(pieces, hb_all, world)      # (on the wire: the ragged all-gather of the pieces)
        out = torch.cat([step(s, "jaccard", lambda s=s: s.phase4_jaccard(vc, vi, vv)) for s in shards], dim=0)
        gathers["index"] = cc.numel() * 4 + sum(p.numel() for p in pieces) * 4 + hb_all.numel() * 4
        if timings is not None:   # slowest rank of phase 4 = max over ranks of the sum of its steps
            keys = ("phase4_rows", "phase4_index_count", "phase4_index_fill", "phase4_jaccard")
            timings["phase4_jaccard_total"] = [sum(timings[k][r] for k in keys) for r in range(world)]
    else:
        out = torch.cat([timed("phase4_jaccard", lambda s=s: s.phase4(vc, vi, vv)) for s in shards], dim=0)
    if timings is not None:
        timings["all_gather_bytes"] = gathers
        timings["sparse_ranks"] = sum(1 for s in shards if s.sparse)
    return out


def gather_row_blocks_to_host(block: torch.Tensor, dst: int = 0, reuse_buffer: bool = False):
    """host-side concatenation of per-rank ROW blocks (sharded re-ranking)"""
    return _gather_blocks_to_host(block, 0, dst, reuse_buffer)

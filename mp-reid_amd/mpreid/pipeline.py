"""The encode loop of ``do_inference`` as a pipeline (reference processor/processor.py:187-198: ``img = img.to(device)``;
``feat = model(img, ...)``; ``evaluator.update(...)`` per 64-image loader batch).

The reference uploads one pageable fp32 batch, encodes it and only then touches the next one.  Here the loader is drained
by a STAGER thread that lays consecutive loader batches into groups of ``group`` images (the encoder's persistent GEMMs
want ~65 000 token rows per call: 508 images of 129 tokens) and uploads every group on a copy stream while earlier groups
are being encoded; the calling thread issues the encoder calls, alternating over ``streams`` HIP streams so that the
HBM-bound phases of one group overlap the matrix phases of the other.  A feature row does not depend on which images share
its encoder call (tests/test_gpu_vit.py: bit for bit), so the features are those of the plain loop, and the evaluator
still sees the loader's own batches, one at a time and in order.

Loader batch images, by type:
  * fp32 / uint8 tensors on the host -- the reference's loader type (datasets/make_dataloader.py:103-106, a DataLoader
    without pin_memory): ``stage='direct'`` (the default: equal or 1-3 % ahead in every measurement of round 4) hands the
    pageable memory to the HIP runtime batch by batch (the runtime pins or stages it itself; host-synchronous, which is
    harmless in the stager thread); ``stage='pinned'`` copies each batch into a page-locked group buffer (one of ``slots``)
    and issues ONE asynchronous H2D per group.  Already-pinned batches are always copied directly.
  * tensors already on the device: read in place when a group's batches are back-to-back slices of one allocation (a
    device-resident dataset cut into batches), else gathered into the group buffer by D2D copies on the copy stream.
    CONTRACT: a device batch must be complete with respect to the device's DEFAULT stream when the loader yields it (the
    stager thread orders its reads after that stream, the way a consumer of a DataLoader batch would); a loader that
    fills batches on another stream synchronises that stream, or waits for its event on the default stream, first.
  * ``RawImageBatch`` (decoded uint8 RGB images of ragged sizes): packed back to back into a pinned byte buffer, one H2D
    per group, then Resize + ToTensor + Normalize inside the model call (ops.PackedRawImages).
No host ``torch.cat``, no per-batch synchronisation; the host runs at most ``slots`` groups ahead of the device.
"""
from __future__ import annotations

import queue
import threading
from typing import Iterable, List, Optional

import numpy as np
import torch

from . import _lib, ops


class _Slot:
    """one group buffer: page-locked host bytes + device bytes (both grow-only) and the two events that recycle it"""

    def __init__(self, device):
        self.device = device
        self.pin: Optional[torch.Tensor] = None
        self.dev: Optional[torch.Tensor] = None
        self.meta_pin: Optional[torch.Tensor] = None     # offsets / hw of a raw group
        self.meta_dev: Optional[torch.Tensor] = None
        self.ready = torch.cuda.Event()                  # recorded on the copy stream: the group is in HBM
        self.consumed = torch.cuda.Event()               # recorded on an encode stream: the device buffer may be rewritten
        self.used = False

    def host(self, nbytes: int) -> torch.Tensor:
        if self.pin is None or self.pin.numel() < nbytes:
            self.pin = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        return self.pin

    def devbuf(self, nbytes: int) -> torch.Tensor:
        if self.dev is None or self.dev.numel() < nbytes:
            if self.dev is not None and self.used:
                self.consumed.synchronize()   # the old buffer goes back to the allocator: its last reader must be done
            self.dev = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
        return self.dev

    def meta(self, nbytes: int):
        if self.meta_pin is None or self.meta_pin.numel() < nbytes:
            self.meta_pin = torch.empty(nbytes * 2, dtype=torch.uint8).pin_memory()
            self.meta_dev = torch.empty(nbytes * 2, dtype=torch.uint8, device=self.device)
        return self.meta_pin, self.meta_dev


class _Group:
    __slots__ = ("slot", "count", "kind", "shape", "dtype", "pieces", "cams", "views", "raw_total", "raw_max_h", "error",
                 "dev_parts", "direct")

    def __init__(self, slot):
        self.slot, self.count, self.kind = slot, 0, None
        self.shape = self.dtype = None
        self.pieces: List[tuple] = []       # (loader batch index, rows of that batch, row offset inside the group)
        self.cams: List[torch.Tensor] = []
        self.views: List[torch.Tensor] = []
        self.raw_total, self.raw_max_h = 0, 0
        self.error = None
        self.dev_parts: List[torch.Tensor] = []   # kind "device": the loader's own tensors, in group order
        self.direct = None                        # ... and, when they turn out to be one contiguous range, a view of it


_END = object()

#: copy / encode streams per device, created once.  The encoders' workspaces are keyed by the stream they run on
#: (ops._workspace): new streams on every run() would leave two more encoder workspaces (GBs each) in that cache per
#: evaluation -- torch hands streams out round-robin from a pool of 32.  One run() at a time per device (one Python thread
#: drives the evaluation: SURVEY.md section 8b).
_STREAMS: dict = {}


def _pipeline_streams(device, n_encode: int):
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    ent = _STREAMS.setdefault(key, {"copy": None, "enc": []})
    if ent["copy"] is None:
        ent["copy"] = torch.cuda.Stream(device=dev)
    while len(ent["enc"]) < n_encode:
        ent["enc"].append(torch.cuda.Stream(device=dev))
    return ent["copy"], ent["enc"][:n_encode]


def _contiguous_view(parts, count, shape, dtype):
    """the tensors of `parts` as ONE tensor [count, *shape] when they are back-to-back slices of the same allocation, else None"""
    if not parts:
        return None
    first = parts[0]
    st = first.untyped_storage()
    at = first.data_ptr()
    for t in parts:
        if t.data_ptr() != at or t.untyped_storage().data_ptr() != st.data_ptr() or not t.is_contiguous():
            return None
        at += t.numel() * t.element_size()
    off = first.data_ptr() - st.data_ptr()
    if off % first.element_size() or at > st.data_ptr() + st.nbytes() or first.data_ptr() % 16:
        return None
    return torch.empty(0, dtype=dtype, device=first.device).set_(st, off // first.element_size(), (count,) + tuple(shape))


class EncodePipeline:
    """``for feat, batch in EncodePipeline(model, ...).run(loader)``: the loader's batches in order, each with the features
    of its images (a device tensor [b, D] valid on the CURRENT stream of the caller).

    model: called as ``model(x, cam_label=..., view_label=...)`` with x a device tensor [n, ...] (n <= group) or an
    ops.PackedRawImages; must run on the current HIP stream (mpreid's encoders do)."""

    def __init__(self, model, group: int = 508, sie_camera: bool = False, sie_view: bool = False, device=None,
                 streams: int = 2, slots: int = 3, stage: str = "direct"):
        assert stage in ("pinned", "direct") and streams >= 1 and slots >= 2 and group >= 1
        self.model, self.group = model, int(group)
        self.sie_camera, self.sie_view = bool(sie_camera), bool(sie_view)
        self.device = device or _lib.require_gpu()
        self.stage = stage
        self.n_streams, self.n_slots = streams, slots
        self.stats = {"groups": 0, "images": 0, "h2d_bytes": 0, "stage": stage}

    # ------------------------------------------------------------------------------------------------ stager thread
    def _stager(self, loader: Iterable, free_q: "queue.Queue", ready_q: "queue.Queue", meta_q: "queue.Queue",
                stop: threading.Event, copy_s: torch.cuda.Stream):
        try:
            torch.cuda.set_device(self.device)
            cur: Optional[_Group] = None

            def take_slot():
                while not stop.is_set():
                    try:
                        return free_q.get(timeout=0.05)
                    except queue.Empty:
                        continue
                return None

            def open_group(kind, shape, dtype):
                slot = take_slot()
                if slot is None:
                    return None
                g = _Group(slot)
                g.kind, g.shape, g.dtype = kind, shape, dtype
                if slot.used:
                    slot.ready.synchronize()             # the H2D that last read the pinned buffer has finished
                    copy_s.wait_event(slot.consumed)     # ... and the encoder call that read the device buffer
                return g

            def close_group(g):
                slot = g.slot
                with torch.cuda.stream(copy_s):
                    if g.kind == "host_pinned":
                        nb = g.count * int(np.prod(g.shape)) * g.dtype.itemsize
                        slot.devbuf(nb)[:nb].copy_(slot.pin[:nb], non_blocking=True)
                        self.stats["h2d_bytes"] += nb
                    elif g.kind == "device":
                        # Batches that are consecutive slices of ONE allocation (a device-resident dataset cut into batches) are
                        # read where they lie: no copy.  Anything else is gathered into the group buffer by D2D copies.
                        g.direct = _contiguous_view(g.dev_parts, g.count, g.shape, g.dtype)
                        if g.direct is None:
                            sb = int(np.prod(g.shape)) * g.dtype.itemsize
                            buf, at = slot.devbuf(self.group * sb), 0
                            for t in g.dev_parts:
                                nb = t.shape[0] * sb
                                buf[at:at + nb].copy_(t.reshape(-1).view(torch.uint8), non_blocking=True)
                                t.record_stream(copy_s)
                                at += nb
                            self.stats["d2d_bytes"] = self.stats.get("d2d_bytes", 0) + at
                        else:
                            self.stats["zero_copy_groups"] = self.stats.get("zero_copy_groups", 0) + 1
                            g.dev_parts = []
                    elif g.kind == "raw":
                        nb = g.raw_total
                        slot.devbuf(nb)[:nb].copy_(slot.pin[:nb], non_blocking=True)
                        mb = self.group * 16                 # offsets [group] int64 | hw [group][2] int32
                        slot.meta_dev[:mb].copy_(slot.meta_pin[:mb], non_blocking=True)
                        self.stats["h2d_bytes"] += nb + mb
                    slot.ready.record(copy_s)
                slot.used = True
                ready_q.put(g)

            bi = 0
            for batch in loader:
                if stop.is_set():
                    break
                img = batch[0]
                n = len(batch[1])
                meta_q.put((bi, n, batch))
                raw = isinstance(img, (list, tuple))
                if raw:
                    kind, shape, dtype = "raw", None, None
                    if n:
                        arrs, hw, _, _ = ops.raw_image_layout(img)
                else:
                    assert torch.is_tensor(img) and img.shape[0] == n, "loader batch: images must be a tensor [b, ...] or a list"
                    img = img.detach()
                    if not img.is_contiguous():
                        img = img.contiguous()
                    shape, dtype = tuple(img.shape[1:]), img.dtype
                    if img.is_cuda:
                        kind = "device"
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream())   # = the default stream in this thread (module docstring: CONTRACT)
                        copy_s.wait_event(ev)
                    else:
                        kind = "host_direct" if (self.stage == "direct" or img.is_pinned()) else "host_pinned"
                lo = 0
                while lo < n:
                    if cur is not None and (cur.kind != kind or cur.shape != shape or cur.dtype != dtype):
                        close_group(cur)    # a loader that changes its sample type mid-way: start a new group
                        cur = None
                    if cur is None:
                        cur = open_group(kind, shape, dtype)
                        if cur is None:
                            return
                    take = min(n - lo, self.group - cur.count)
                    slot = cur.slot
                    if raw:
                        part, phw = arrs[lo:lo + take], hw[lo:lo + take]
                        sizes = phw[:, 0].astype(np.int64) * phw[:, 1] * 3
                        tot = int(sizes.sum())
                        need = cur.raw_total + tot
                        if slot.pin is None or slot.pin.numel() < need:   # grow, keeping what the group already holds
                            old = slot.pin
                            slot.pin = None
                            new = slot.host(max(need, self.group * 32768))
                            if old is not None and cur.raw_total:
                                new[:cur.raw_total].copy_(old[:cur.raw_total])
                        mp, _ = slot.meta(self.group * 16)
                        offs = mp[:self.group * 8].view(torch.int64).numpy()
                        hws = mp[self.group * 8:self.group * 16].view(torch.int32).numpy().reshape(self.group, 2)
                        o = np.zeros(take, np.int64)
                        o[1:] = np.cumsum(sizes)[:-1]
                        offs[cur.count:cur.count + take] = o + cur.raw_total
                        hws[cur.count:cur.count + take] = phw
                        np.concatenate([a.reshape(-1) for a in part], out=slot.pin.numpy()[cur.raw_total:need])
                        cur.raw_total = need
                        cur.raw_max_h = max(cur.raw_max_h, int(phw[:, 0].max()))
                    else:
                        sb = int(np.prod(shape)) * dtype.itemsize
                        src = img[lo:lo + take].reshape(-1).view(torch.uint8)
                        if kind == "host_pinned":
                            slot.host(self.group * sb)[cur.count * sb:(cur.count + take) * sb].copy_(src)
                        elif kind == "device":   # decided when the group closes: read in place or gathered
                            cur.dev_parts.append(img[lo:lo + take])
                        else:   # pageable / pinned host memory straight to the device
                            with torch.cuda.stream(copy_s):
                                slot.devbuf(self.group * sb)[cur.count * sb:(cur.count + take) * sb].copy_(src, non_blocking=True)
                            self.stats["h2d_bytes"] += take * sb
                    cur.pieces.append((bi, lo, lo + take, cur.count))
                    if self.sie_camera:
                        cur.cams.append(batch[3][lo:lo + take])
                    if self.sie_view:
                        cur.views.append(batch[4][lo:lo + take])
                    cur.count += take
                    lo += take
                    if cur.count == self.group:
                        close_group(cur)
                        cur = None
                bi += 1
            if cur is not None and cur.count:
                close_group(cur)
            ready_q.put(_END)
        except BaseException as e:   # surfaces in the consuming thread
            g = _Group(None)
            g.error = e
            ready_q.put(g)

    # ------------------------------------------------------------------------------------------------ consumer
    def _input_of(self, g: _Group):
        slot = g.slot
        if g.kind == "raw":
            G = self.group
            offs = slot.meta_dev[:G * 8].view(torch.int64)[:g.count]
            hw = slot.meta_dev[G * 8:G * 16].view(torch.int32).view(G, 2)[:g.count]
            return ops.PackedRawImages(slot.dev[:g.raw_total], offs, hw, g.count, g.raw_max_h)
        if g.direct is not None:
            return g.direct
        nb = g.count * int(np.prod(g.shape)) * g.dtype.itemsize
        return slot.dev[:nb].view(g.dtype).view((g.count,) + g.shape)

    def run(self, loader: Iterable):
        dev = self.device
        main = torch.cuda.current_stream(dev)
        copy_s, enc_s = _pipeline_streams(dev, self.n_streams)
        # a model that builds its device encoder lazily (make_model) does so HERE, on the caller's stream: the weight
        # preparation kernels are then ordered before every encode stream by the wait_stream(main) below, not only before
        # the stream that happens to run the first group
        warm = getattr(self.model, "_get_encoder", None)
        if callable(warm):
            warm()
        free_q, ready_q, meta_q = queue.Queue(), queue.Queue(), queue.Queue()
        for _ in range(self.n_slots):
            free_q.put(_Slot(dev))
        stop = threading.Event()
        th = threading.Thread(target=self._stager, args=(loader, free_q, ready_q, meta_q, stop, copy_s), daemon=True,
                              name="mpreid-stager")
        for s in enc_s + [copy_s]:
            s.wait_stream(main)
        th.start()
        pending = {}          # loader batch index -> [n, batch, rows done, [(lo, feature piece)]]
        next_out = 0
        gi = 0
        width = 0             # feature dimension (for loader batches without images)

        def drain_meta():
            while not meta_q.empty():
                bi, n, batch = meta_q.get()
                pending[bi] = [n, batch, 0, []]

        def emit():
            nonlocal next_out
            while next_out in pending and pending[next_out][2] == pending[next_out][0]:
                n, batch, _, parts = pending.pop(next_out)
                parts.sort(key=lambda p: p[0])
                if not parts:
                    f = torch.empty((0, width), dtype=torch.float32, device=dev)
                else:
                    f = parts[0][1] if len(parts) == 1 else torch.cat([p[1] for p in parts], dim=0)
                next_out += 1
                yield f, batch
        try:
            while True:
                g = ready_q.get()
                if g is _END:
                    break
                if g.error is not None:
                    raise g.error
                s = enc_s[gi % len(enc_s)]
                gi += 1
                with torch.cuda.stream(s):
                    s.wait_event(g.slot.ready)
                    x = self._input_of(g)
                    if g.direct is not None:
                        x.record_stream(s)        # the loader's memory, read on this stream
                    cam = torch.cat(g.cams).to(dev, non_blocking=True) if self.sie_camera else None
                    view = torch.cat(g.views).to(dev, non_blocking=True) if self.sie_view else None
                    with torch.no_grad():
                        feat = self.model(x, cam_label=cam, view_label=view)
                    g.slot.consumed.record(s)
                    done = torch.cuda.Event()
                    done.record(s)
                free_q.put(g.slot)
                main.wait_event(done)
                feat.record_stream(main)
                self.stats["groups"] += 1
                self.stats["images"] += g.count
                width = feat.shape[1]
                drain_meta()                      # (the stager queues a batch's meta before any of its pieces)
                for (bi, lo, hi, off) in g.pieces:
                    ent = pending[bi]
                    ent[3].append((lo, feat[off:off + (hi - lo)]))
                    ent[2] += hi - lo
                yield from emit()
            drain_meta()                          # trailing loader batches without images carry no pieces
            yield from emit()
            assert not pending, "encode pipeline: loader batches left without features"
        finally:
            stop.set()
            th.join(timeout=30)
            if th.is_alive():   # error path only: the stager is blocked in the loader; whatever it still copies lands in
                import warnings   # slots nobody reads, but the copy stream must drain before their memory is reused
                warnings.warn("mpreid encode pipeline: the stager thread is still running 30 s after shutdown was requested")
                copy_s.synchronize()
            for s in enc_s + [copy_s]:
                main.wait_stream(s)

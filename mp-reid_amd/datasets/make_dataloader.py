"""``make_dataloader(cfg)`` with the reference's return tuple (datasets/make_dataloader.py:111):
(train_loader, train_loader_normal, val_loader, num_query, num_classes, cam_num, view_num).

Only the evaluation side exists here and only for DATASETS.NAMES == 'synthetic' (neither box has
Market-1501 / MSMT17 / MMMP; their directory parsers are out of scope).
The synthetic val set keeps the reference's conventions: query images first, then gallery
(:101); batches are (img, pids, camids, camids_tensor, viewids_tensor, img_paths) (:39-43); images
are float32 NCHW in the value range left by Normalize(mean=0.5, std=0.5).

DATASETS.SYNTH_RAW moves val_transforms (:57-61: Resize, ToTensor, Normalize) to the GPU: the loader then yields
a RawImageBatch -- the DECODED uint8 RGB images, any sizes, exactly what PIL hands to T.Resize -- and the model
resizes (bit-exact with Pillow), scales and normalises on the device (mpreid.ops.resize_bilinear_u8 +
VitEncoder.forward_u8).  A real dataset plugs in the same way: decode to uint8 HWC in the workers, collate with
``raw_val_collate_fn``.
"""
import numpy as np
import torch

from mpreid import synth


class RawImageBatch(list):
    """A batch of decoded uint8 [h, w, 3] RGB images of possibly different sizes (numpy arrays or tensors).
    ``model(batch, ...)`` runs val_transforms on the GPU.  ``.to(device)`` is a no-op returning self so that the
    reference's ``img = img.to(device)`` line in do_inference keeps working; the upload happens packed, through
    pinned staging buffers, inside the model call."""

    def to(self, *a, **k):
        return self

    @property
    def shape(self):
        return (len(self),)


def raw_val_collate_fn(batch):
    """val_collate_fn (datasets/make_dataloader.py:41-45) for datasets that yield decoded uint8 images"""
    imgs, pids, camids, viewids, img_paths = zip(*batch)
    return (RawImageBatch(imgs), pids, camids, torch.tensor(camids, dtype=torch.int64),
            torch.tensor(viewids, dtype=torch.int64), img_paths)


class SyntheticValLoader:
    def __init__(self, n_query, n_gallery, n_ids, hw, batch, seed, device="cpu", raw=False):
        self.n = n_query + n_gallery
        self.hw, self.batch, self.seed = hw, batch, seed
        rng = np.random.default_rng(seed)
        self.pids = rng.integers(0, n_ids, size=self.n)
        self.camids = rng.integers(0, 6, size=self.n)
        self.device = device
        self.raw = raw

    def _raw_images(self, s, e):
        """decoded-image stand-ins: uint8, heights/widths scattered around Market-1501's 128x64"""
        rng = np.random.default_rng(self.seed * 7919 + s)
        out = []
        for _ in range(s, e):
            h, w = int(rng.integers(96, 200)), int(rng.integers(48, 100))
            out.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        return RawImageBatch(out)

    def __len__(self):
        return (self.n + self.batch - 1) // self.batch

    def _batch(self, s, e):
        img = self._raw_images(s, e) if self.raw else \
            torch.from_numpy(synth.synthetic_images(e - s, self.hw[0], self.hw[1], seed=self.seed + s))
        pids = tuple(int(p) for p in self.pids[s:e])
        cams = tuple(int(c) for c in self.camids[s:e])
        return (img, pids, cams, torch.tensor(cams, dtype=torch.int64), torch.zeros(e - s, dtype=torch.int64),
                tuple(f"synthetic/{i:07d}.jpg" for i in range(s, e)))

    def __iter__(self):
        for s in range(0, self.n, self.batch):
            yield self._batch(s, min(self.n, s + self.batch))

    def shard(self, indices):
        """the samples `indices` (ascending global positions) in order: what one rank of a multi-GPU evaluation encodes.
        The images are a function of the unsharded batch they belong to, so the owning batches are generated whole and
        cut; batches without an owned sample are skipped."""
        from processor.processor import select_samples
        want = sorted(indices)
        i = 0
        for s in range(0, self.n, self.batch):
            e = min(self.n, s + self.batch)
            keep = []
            while i < len(want) and want[i] < e:
                keep.append(want[i] - s)
                i += 1
            if keep:
                yield select_samples(self._batch(s, e), keep)


def make_dataloader(cfg):
    if cfg.DATASETS.NAMES != "synthetic":
        raise NotImplementedError(
            f"dataset {cfg.DATASETS.NAMES!r}: directory parsers are out of scope of this build; "
            "use DATASETS.NAMES synthetic or feed R1_mAP_eval / do_inference your own loader")
    d = cfg.DATASETS
    val_loader = SyntheticValLoader(int(d.SYNTH_QUERY), int(d.SYNTH_GALLERY), int(d.SYNTH_IDS),
                                    tuple(cfg.INPUT.SIZE_TEST), int(cfg.TEST.IMS_PER_BATCH), int(d.SYNTH_SEED),
                                    raw=bool(d.get("SYNTH_RAW", False)))
    return None, None, val_loader, int(d.SYNTH_QUERY), int(d.SYNTH_IDS), 6, 1

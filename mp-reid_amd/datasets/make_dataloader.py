"""``make_dataloader(cfg)`` with the reference's return tuple (datasets/make_dataloader.py:111):
(train_loader, train_loader_normal, val_loader, num_query, num_classes, cam_num, view_num).

Only the evaluation side exists here and only for DATASETS.NAMES == 'synthetic' (neither box has
Market-1501 / MSMT17 / MMMP; their directory parsers are out of scope).
The synthetic val set keeps the reference's conventions: query images first, then gallery
(:101); batches are (img, pids, camids, camids_tensor, viewids_tensor, img_paths) (:39-43); images
are float32 NCHW in the value range left by Normalize(mean=0.5, std=0.5).
"""
import numpy as np
import torch

from mpreid import synth


class SyntheticValLoader:
    def __init__(self, n_query, n_gallery, n_ids, hw, batch, seed, device="cpu"):
        self.n = n_query + n_gallery
        self.hw, self.batch, self.seed = hw, batch, seed
        rng = np.random.default_rng(seed)
        self.pids = rng.integers(0, n_ids, size=self.n)
        self.camids = rng.integers(0, 6, size=self.n)
        self.device = device

    def __len__(self):
        return (self.n + self.batch - 1) // self.batch

    def __iter__(self):
        for s in range(0, self.n, self.batch):
            e = min(self.n, s + self.batch)
            img = torch.from_numpy(synth.synthetic_images(e - s, self.hw[0], self.hw[1], seed=self.seed + s))
            pids = tuple(int(p) for p in self.pids[s:e])
            cams = tuple(int(c) for c in self.camids[s:e])
            yield (img, pids, cams, torch.tensor(cams, dtype=torch.int64), torch.zeros(e - s, dtype=torch.int64),
                   tuple(f"synthetic/{i:07d}.jpg" for i in range(s, e)))


def make_dataloader(cfg):
    if cfg.DATASETS.NAMES != "synthetic":
        raise NotImplementedError(
            f"dataset {cfg.DATASETS.NAMES!r}: directory parsers are out of scope of this build; "
            "use DATASETS.NAMES synthetic or feed R1_mAP_eval / do_inference your own loader")
    d = cfg.DATASETS
    val_loader = SyntheticValLoader(int(d.SYNTH_QUERY), int(d.SYNTH_GALLERY), int(d.SYNTH_IDS),
                                    tuple(cfg.INPUT.SIZE_TEST), int(cfg.TEST.IMS_PER_BATCH), int(d.SYNTH_SEED))
    return None, None, val_loader, int(d.SYNTH_QUERY), int(d.SYNTH_IDS), 6, 1

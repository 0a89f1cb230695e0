"""``datasets/make_dataloader_uniprompt.py`` of the reference: same val loader as make_dataloader.py; the return
tuple is (train_loader_stage2, train_loader_stage1, val_loader, num_query, num_classes, cam_num, view_num)
(reference :118) -- the two training loaders are None here (evaluation only)."""
from .make_dataloader import RawImageBatch, SyntheticValLoader, make_dataloader, raw_val_collate_fn  # noqa: F401

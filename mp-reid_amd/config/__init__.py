"""Minimal stand-in for the reference's yacs config (yacs is not installed in this image).

``cfg_base`` / ``cfg`` expose the attribute paths the evaluation hot path reads
(reference config/defaults_base.py): MODEL.{NAME,STRIDE_SIZE,SIE_CAMERA,SIE_VIEW,SIE_COE,DEVICE_ID},
INPUT.{SIZE_TRAIN,SIZE_TEST,PIXEL_MEAN,PIXEL_STD}, TEST.{IMS_PER_BATCH,WEIGHT,NECK_FEAT,FEAT_NORM,RE_RANKING},
DATASETS.{NAMES,ROOT_DIR}, DATALOADER.NUM_WORKERS, OUTPUT_DIR, with merge_from_file / merge_from_list / freeze.
"""
from .node import CfgNode, make_defaults

cfg_base = make_defaults()
cfg = make_defaults()
__all__ = ["cfg_base", "cfg", "CfgNode"]

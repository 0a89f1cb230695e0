import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError("config is frozen")
        self[k] = v

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        return copy.deepcopy(self)

    def _merge(self, other):
        for k, v in other.items():
            if v is None:      # a section whose keys are all commented out (reference YAMLs have these)
                continue
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        opts = list(opts or [])
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode()) if p not in node else node[p]
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = val


def make_defaults():
    return CfgNode({
        "MODEL": {"NAME": "ViT-B-16", "DEVICE": "cuda", "DEVICE_ID": "0", "STRIDE_SIZE": [16, 16],
                  "SIE_CAMERA": False, "SIE_VIEW": False, "SIE_COE": 3.0, "NECK": "bnneck", "COS_LAYER": False,
                  "DIST_TRAIN": False, "INIT_SEED": 7, "ENCODER_PRECISION": "split"},
        "INPUT": {"SIZE_TRAIN": [256, 128], "SIZE_TEST": [256, 128], "PIXEL_MEAN": [0.5, 0.5, 0.5],
                  "PIXEL_STD": [0.5, 0.5, 0.5]},
        "DATASETS": {"NAMES": "synthetic", "ROOT_DIR": "", "EXP_SETTING": "", "SYNTH_QUERY": 3368,
                     "SYNTH_GALLERY": 15913, "SYNTH_IDS": 751, "SYNTH_SEED": 1234,
                     # SYNTH_RAW: the loader hands over DECODED uint8 images of ragged sizes (what PIL gives before
                     # val_transforms) and Resize + ToTensor + Normalize run on the GPU
                     "SYNTH_RAW": False},
        "DATALOADER": {"NUM_WORKERS": 0},
        "TEST": {"IMS_PER_BATCH": 64, "RE_RANKING": False, "WEIGHT": "", "NECK_FEAT": "before", "FEAT_NORM": "yes",
                 "DIST_MAT": "dist_mat.npy", "EVAL": False,
                 "DISTANCE_MODE": "exact", "RERANK_ALGO": "exact",   # not reference keys: see processor.do_inference
                 # Uni-Prompt evaluation (reference config/defaults.py:331-344)
                 "TTA_ENABLED": False, "TTPT": {"ENABLED": False, "LR": 0.001, "STEPS": 5, "TEMPERATURE": 0.07}},
        "OUTPUT_DIR": "",
    })

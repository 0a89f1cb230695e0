"""``setup_logger(name, save_dir, if_train)`` as used by the reference's entry scripts
(reference utils/logger.py:5-25): DEBUG-level logger to stdout and to train_log.txt / test_log.txt."""
import logging
import os
import sys


def setup_logger(name, save_dir, if_train):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(asctime)s %(name)s %(levelname)s: %(message)s")
    if not any(isinstance(h, logging.StreamHandler) and getattr(h, "_mpreid", False) for h in logger.handlers):
        sh = logging.StreamHandler(stream=sys.stdout)
        sh._mpreid = True
        sh.setLevel(logging.DEBUG)
        sh.setFormatter(fmt)
        logger.addHandler(sh)
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        fh = logging.FileHandler(os.path.join(save_dir, "train_log.txt" if if_train else "test_log.txt"), mode='w')
        fh.setLevel(logging.DEBUG)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger

#!/usr/bin/env python3
"""Evaluation entry point of the HIP build -- the counterpart of the reference's ``test.py`` (test.py:10-65):

    python test.py --config_file configs/person/vit_base.yml [KEY VALUE ...]

config (YAML + command-line overrides, frozen) -> logger ("transreid", <OUTPUT_DIR>/test_log.txt) -> val loader ->
model -> ``load_param(TEST.WEIGHT)`` -> ``do_inference`` (ViT-B/16 on the GPU, R1_mAP_eval with the HIP distance /
re-ranking kernels).  Differences from the reference, on purpose: TEST.WEIGHT may be empty (seeded random weights, for
smoke runs); MODEL.DEVICE_ID selects the HIP device through HIP_VISIBLE_DEVICES *before* anything touches the GPU;
the VehicleID 10-trial loop is not reproduced (VehicleID's parser is out of scope); ``main`` returns
(rank1, rank5) so that tests can call it in-process.  Multi-GPU: ``python -m torch.distributed.run --nnodes=1
--nproc-per-node P --master-addr 127.0.0.1 test.py --config_file ...`` runs one rank per GPU; every rank encodes its
shard and all of them return the same (rank1, rank5) as the single-GPU run.
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def build_parser():
    p = argparse.ArgumentParser(description="MP-ReID evaluation on MI355X (HIP build)")
    p.add_argument("--config_file", default=os.path.join("configs", "person", "vit_base.yml"), type=str,
                   help="path to config file ('' = built-in defaults only)")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER,
                   help="KEY VALUE pairs overriding config options, e.g. TEST.RE_RANKING True")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    from config import cfg_base
    cfg = cfg_base.clone()
    cfg.defrost()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()

    if cfg.OUTPUT_DIR:
        os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    if world > 1:
        # one process per GPU (python -m torch.distributed.run --nproc-per-node P test.py ...): the launcher's LOCAL_RANK
        # picks the device, the default process group (RCCL) carries the evaluator's collectives; replaces the
        # reference's single-process nn.DataParallel (processor/processor.py:178-182)
        import torch
        from mpreid import distributed as D
        rank, world, local = D.init_from_env()
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    else:
        # device selection has to happen before the HIP runtime initialises (the reference sets CUDA_VISIBLE_DEVICES here)
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(cfg.MODEL.DEVICE_ID))

    from utils.logger import setup_logger
    logger = setup_logger("transreid", cfg.OUTPUT_DIR if rank == 0 else "", if_train=False)
    logger.info(args)
    if args.config_file:
        logger.info("Loaded configuration file {}".format(args.config_file))
        with open(args.config_file) as fh:
            logger.info("\n" + fh.read())
    logger.info("Running with config:\n{}".format(cfg))

    from datasets.make_dataloader import make_dataloader
    from model.make_model import make_model
    from processor.processor import do_inference

    _, _, val_loader, num_query, num_classes, camera_num, view_num = make_dataloader(cfg)
    model = make_model(cfg, num_class=num_classes, camera_num=camera_num, view_num=view_num)
    if cfg.TEST.WEIGHT:
        model.load_param(cfg.TEST.WEIGHT)
    else:
        logger.info("TEST.WEIGHT is empty: evaluating the seeded random initialisation (MODEL.INIT_SEED)")
    res = do_inference(cfg, model, val_loader, num_query)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()

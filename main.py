#!/usr/bin/env python
"""Train / test entry point — same flow and flags as main.py of the reference (main.py:11-65).

    python main.py --config cfgs/PCN_models/AdaPoinTr.yaml --exp_name run0
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py \
        --launcher pytorch --sync_bn --config cfgs/PCN_models/AdaPoinTr.yaml --exp_name run0
"""
import os
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see imgadapointr_amd/__init__.py

import torch  # noqa: E402

from imgadapointr_amd.tools import builder
from imgadapointr_amd.tools.runner import run_net, test_net
from imgadapointr_amd.utils import dist_utils, misc, parser
from imgadapointr_amd.utils.config import get_config, log_args_to_file, log_config_to_file
from imgadapointr_amd.utils.logger import get_root_logger


class _ScalarLog:
    """Tiny stand-in for tensorboardX.SummaryWriter (not installed here): scalars go to a TSV."""

    def __init__(self, path):
        os.makedirs(path, exist_ok=True)
        self.fh = open(os.path.join(path, 'scalars.tsv'), 'a')

    def add_scalar(self, tag, value, step):
        self.fh.write(f'{tag}\t{step}\t{value}\n')

    def close(self):
        self.fh.close()


def _writer(path):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(path)
    except ImportError:
        return _ScalarLog(path)


def main():
    args = parser.get_args()
    args.use_gpu = torch.cuda.is_available()
    if args.launcher == 'none':
        args.distributed = False
        args.world_size = 1
        if args.use_gpu:                      # `--launcher none --local_rank N` runs on GPU N: make it the current device
            torch.cuda.set_device(args.local_rank)
    else:
        args.distributed = True
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist_utils.init_dist(args.launcher)
        _, args.world_size = dist_utils.get_dist_info()
    timestamp = time.strftime('%Y%m%d_%H%M%S', time.localtime())
    logger = get_root_logger(log_file=os.path.join(args.experiment_path, f'{timestamp}.log'), name=args.log_name)
    train_writer = val_writer = None
    if not args.test and args.local_rank == 0:
        train_writer = _writer(os.path.join(args.tfboard_path, 'train'))
        val_writer = _writer(os.path.join(args.tfboard_path, 'test'))
    config = get_config(args, logger=logger)
    config.dataset.train.others.bs = builder.resolve_batch_size(
        config.total_bs, args.world_size, args.distributed, getattr(args, 'total_bs', None), logger=logger)
    log_args_to_file(args, 'args', logger=logger)
    log_config_to_file(config, 'config', logger=logger)
    logger.info(f'Distributed training: {args.distributed}')
    if args.seed is not None:
        logger.info(f'Set random seed to {args.seed}, deterministic: {args.deterministic}')
        misc.set_random_seed(args.seed + args.local_rank, deterministic=args.deterministic)
    if args.test:
        test_net(args, config)
    else:
        run_net(args, config, train_writer, val_writer)


if __name__ == '__main__':
    main()

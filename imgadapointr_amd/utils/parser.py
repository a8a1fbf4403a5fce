"""Command line of main.py — same flags and derived paths as utils/parser.py:5-99 of the reference.

--config --gdanet_w --launcher{none,pytorch} --local_rank --num_workers --seed --deterministic
--sync_bn --output-dir --exp_name --start_ckpts --ckpts --val_freq --resume --wandb
--use_pretrained_mt --test --mode{easy,median,hard}; experiment directory
./experiments/<cfg-stem>/<cfg-parent>/<exp_name>[_mode] and .../TFBoard/<exp_name>.
One addition: --log_every (iterations between loss log lines / device->host reads, default 100).
"""
import argparse
import os
from pathlib import Path


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--config', type=str, default='./cfgs/PCN_models/AdaPoinTr.yaml', help='yaml config file')
    p.add_argument('--gdanet_w', type=str, default='./pretrained/GDANet_best_insiou_model.pth',
                   help='path to GDANet weights')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--num_workers', type=int, default=4)
    p.add_argument('--graph_train', action='store_true', default=False,
                   help='replay the train step from HIP graphs (one process; for the launch-bound batch-16 models)')
    p.add_argument('--total_bs', type=int, default=None,
                   help='batch over all ranks; overrides the YAML (the reference ignores its YAML in distributed mode)')
    p.add_argument('--seed', type=int, default=0, help='random seed')
    p.add_argument('--deterministic', action='store_true',
                   help='whether to set deterministic options for the conv backend.')
    p.add_argument('--sync_bn', action='store_true', default=False, help='whether to use sync bn')
    p.add_argument('--output-dir', type=str, default='inference_result', help='inference output directory')
    p.add_argument('--exp_name', type=str, default='example3', help='experiment name')
    p.add_argument('--start_ckpts', type=str, default=None, help='reload used ckpt path')
    p.add_argument('--ckpts', type=str, default='./pretrained/AdaPoinTr_PCN.pth', help='test used ckpt path')
    p.add_argument('--val_freq', type=int, default=1, help='test freq')
    p.add_argument('--resume', action='store_true', default=False,
                   help='autoresume training (interrupted by accident)')
    p.add_argument('--wandb', action='store_true', default=False, help='use wandb')
    p.add_argument('--use_pretrained_mt', action='store_true', default=False,
                   help='use pretrained modality transfer')
    p.add_argument('--test', action='store_true', default=False, help='test mode for certain ckpt')
    p.add_argument('--mode', choices=['easy', 'median', 'hard', None], default='easy',
                   help='difficulty mode for shapenet')
    p.add_argument('--log_every', type=int, default=100, help='iterations between loss log lines')
    return p


def get_args(argv=None):
    args = build_parser().parse_args(argv)
    if args.test and args.resume:
        raise ValueError('--test and --resume cannot be both activate')
    if args.resume and args.start_ckpts is not None:
        raise ValueError('--resume and --start_ckpts cannot be both activate')
    if args.test and args.ckpts is None:
        raise ValueError('ckpts shouldnt be None while test mode')
    if 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = str(args.local_rank)
    else:  # torch.distributed.run exports LOCAL_RANK instead of passing --local_rank
        args.local_rank = int(os.environ['LOCAL_RANK'])
    if args.test:
        args.exp_name = 'test_' + args.exp_name
    if args.mode is not None:
        args.exp_name = args.exp_name + '_' + args.mode
    stem, parent = Path(args.config).stem, Path(args.config).parent.stem
    args.experiment_path = os.path.join('./experiments', stem, parent, args.exp_name)
    args.tfboard_path = os.path.join('./experiments', stem, parent, 'TFBoard', args.exp_name)
    args.log_name = stem
    create_experiment_dir(args)
    return args


def create_experiment_dir(args):
    for path, what in ((args.experiment_path, 'experiment'), (args.tfboard_path, 'TFBoard')):
        if not os.path.exists(path):
            os.makedirs(path, exist_ok=True)
            print(f'Create {what} path successfully at {path}')

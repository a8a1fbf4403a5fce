"""Single-cloud completion CLI (interface of tools/inference.py of the reference, minus the image
rendering): `python -m imgadapointr_amd.tools.inference <config.yaml> <ckpt.pth> --pc file.pcd
--out_pc_root out/` writes out/<name>/fine.npy.  The input is up-sampled (or sub-sampled) to 2048
points by UpSamplePoints, as the reference does (inference.py:65-76)."""
import argparse
import os

import numpy as np
import torch

from ..datasets.data_transforms import Compose
from ..utils.config import cfg_from_yaml_file
from ..utils.io_module import IO
from . import builder


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('model_config', help='yaml config file')
    p.add_argument('model_checkpoint', help='pretrained weight')
    p.add_argument('--pc_root', type=str, default='', help='directory of input clouds')
    p.add_argument('--pc', type=str, default='', help='one input cloud')
    p.add_argument('--out_pc_root', type=str, default='', help='root of the output .npy files')
    p.add_argument('--device', default='cuda:0', help='device used for inference')
    args = p.parse_args(argv)
    assert args.out_pc_root != '', 'give --out_pc_root'
    assert (args.pc != '') or (args.pc_root != '')
    return args


def inference_single(model, pc_path, args, config, root=None):
    pc_file = os.path.join(root, pc_path) if root is not None else pc_path
    pc = IO.get(pc_file).astype(np.float32)
    shapenet = config.dataset.train._base_['NAME'] == 'ShapeNet'
    if shapenet:  # normalise to the unit sphere, inference.py:58-63
        centroid = np.mean(pc, axis=0)
        pc = pc - centroid
        m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
        pc = pc / m
    transform = Compose([{'callback': 'UpSamplePoints', 'parameters': {'n_points': 2048}, 'objects': ['input']},
                         {'callback': 'ToTensor', 'objects': ['input']}])
    x = transform({'input': pc})['input'].unsqueeze(0).to(args.device.lower())
    with torch.no_grad():
        dense = model(x)[-1].squeeze(0).cpu().numpy()     # `model` may be a GraphedForward (main): one HIP graph replay
    if shapenet:
        dense = dense * m + centroid
    target = os.path.join(args.out_pc_root, os.path.splitext(os.path.basename(pc_path))[0])
    os.makedirs(target, exist_ok=True)
    np.save(os.path.join(target, 'fine.npy'), dense)
    return dense


def main(argv=None):
    args = get_args(argv)
    config = cfg_from_yaml_file(args.model_config)
    model = builder.model_builder(config.model)
    builder.load_model(model, args.model_checkpoint)
    model.to(args.device.lower()).eval()
    if args.device.lower().startswith('cuda'):
        from ..utils.graph_runner import GraphedForward
        model = GraphedForward(model)                     # every cloud is up-sampled to 2048 points: one graph serves all
    if args.pc_root != '':
        for pc_file in sorted(os.listdir(args.pc_root)):
            inference_single(model, pc_file, args, config, root=args.pc_root)
    else:
        inference_single(model, args.pc, args, config)


if __name__ == '__main__':
    main()

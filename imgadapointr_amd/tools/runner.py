"""Train / validation driver (interface of tools/runner.py of the reference: run_net :28,
validate :307, test_net :460).

What is kept: the step semantics (forward -> get_loss -> backward -> [every step_per_update:
clip_grad_norm_(10) -> optimizer.step -> zero_grad], runner.py:186-215), per-epoch scheduler
steps, validation with 4 Chamfer losses + Metrics, best/last/epoch checkpoints with the
reference's dict layout, losses reported x1000.
What is different (MI355X-first, results unchanged):
  * one process per GPU with FlatDDP (utils/flat_ddp.py) instead of DistributedDataParallel +
    find_unused_parameters; single-GPU runs use the bare model (no nn.DataParallel wrapper);
  * no per-iteration `.item()` x2 + cuda.synchronize (runner.py:220,225): losses stay on the
    device in a small accumulator and are read back only when a log line is printed; the two
    scalar all-reduces per iteration are one packed all-reduce at log time;
  * datasets of the image / segmentation families are dispatched by the same name rules.
"""
import contextlib
import json
import os
import time

import torch
import torch.nn as nn

from .. import _native
from ..extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
from ..utils.AverageMeter import AverageMeter
from ..utils.flat_ddp import FlatDDP
from ..utils.graph_runner import GraphedForward, GraphedTrainStep
from ..utils.logger import get_logger, print_log
from ..utils.metrics import Metrics
from . import builder


def _unwrap(model):
    return model.module if hasattr(model, 'module') else model


def unpack_batch(dataset_name, data, device):
    """(partial, gt, extra_inputs) for the dataset families of runner.py:142-184."""
    if 'SegImgPCN' in dataset_name:
        return data[0].to(device), data[1].to(device), (data[2].to(device), data[3].to(device))
    if 'ImgPCN' in dataset_name or 'ViPC' in dataset_name:
        return data[0].to(device), data[1].to(device), (data[2].to(device),)
    if 'PCN' in dataset_name or dataset_name in ('Completion3D',) or 'ProjectShape' in dataset_name:
        return data[0].to(device), data[1].to(device), ()
    raise NotImplementedError(f'Train phase do not support {dataset_name}')


DEFER_FOLDS = os.environ.get("PTOPS_DEFER_FOLDS", "1") != "0"
FUSED_CLIP_ADAMW = os.environ.get("PTOPS_FUSED_ADAMW", "1") != "0"


class TrainStep:
    """One optimisation step of the hot path, shared by run_net and bench.py.

    forward + get_loss + backward + gradient averaging (FlatDDP) + clip + optimizer step."""

    def __init__(self, model, optimizer, grad_norm_clip=10.0, step_per_update=1):
        self.model = model
        self.net = _unwrap(model)
        self.optimizer = optimizer
        self.clip = grad_norm_clip
        self.step_per_update = step_per_update
        self.ddp = model if isinstance(model, FlatDDP) else None
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        self._every_param = list(self.net.parameters())
        self._num_iter = 0
        self.before_update = None       # optional callable(): runs when the gradients of an update are complete
        self.zero_grad()

    def zero_grad(self):
        if self.ddp is not None:
            self.ddp.zero_grad()
        else:
            # nn.Module.zero_grad(set_to_none=True) walks the module tree on every call (0.7 ms of host time per step for
            # the 330 parameters of AdaPoinTr); the parameter list is fixed for the life of this object
            for p in self._every_param:
                p.grad = None

    def __call__(self, partial, gt, extra=(), epoch=0, denoise_noise=None, _update=None):
        """`denoise_noise` ([B,64,3]) replaces the random jitter of the denoise seeds (fixture-driven parity runs).
        `_update` (utils/graph_runner.py::GraphedTrainStep, inside a stream capture) replaces clip + optimizer step."""
        if denoise_noise is None:
            ret = self.model(partial, *extra)
        else:
            ret = self.model(partial, *extra, denoise_noise=denoise_noise)
        sparse_loss, dense_loss = self.net.get_loss(ret, gt, epoch)
        if self.ddp is not None:            # average only the backward that completes an update
            self.ddp.set_sync(self._num_iter + 1 == self.step_per_update)
        # the first backward after zero_grad hands its gradients over unread (`.grad` is None), so the tiny fold launches
        # of the bias / LayerNorm gradients can be queued and run batched (include/ptops.h, deferred folds)
        with (_native.deferred_folds(self.params, partial.device) if (self._num_iter == 0 and DEFER_FOLDS and partial.is_cuda)
              else contextlib.nullcontext()):
            (sparse_loss + dense_loss).backward()
        self._num_iter += 1
        if self._num_iter == self.step_per_update:
            self._num_iter = 0
            if self.ddp is not None:
                self.ddp.finalize()
            if self.before_update is not None:
                self.before_update()
            if _update is not None:
                _update()
            elif hasattr(self.optimizer, "step_clipped") and FUSED_CLIP_ADAMW:
                self.optimizer.step_clipped(self.clip)          # clip + AdamW of all parameters, three launches
            else:
                torch.nn.utils.clip_grad_norm_([p for p in self.params if p.grad is not None], self.clip, norm_type=2)
                self.optimizer.step()
            self.zero_grad()
        return sparse_loss.detach(), dense_loss.detach()


def load_frozen_segmentator(base_model, weights_path, logger=None):
    """GDANet weights are loaded (strict=False, 'module.' stripped) and frozen, tools/runner.py:55-66 of
    the reference.  The reference fails without the file; here a missing file only logs a warning so
    that synthetic-data runs work, and the segmentator is frozen either way."""
    seg = base_model.base_model.segmentator
    if weights_path and os.path.exists(weights_path):
        state = torch.load(weights_path, map_location='cpu', weights_only=False)['model']
        seg.load_state_dict({k.replace('module.', ''): v for k, v in state.items()}, strict=False)
        print_log(f'Loaded segmentator weights from {weights_path}', logger=logger)
    else:
        print_log(f'[WARNING] segmentator weights {weights_path} not found: GDANet stays at its initialisation',
                  logger=logger)
    for p in seg.parameters():
        p.requires_grad = False


def wrap_model(base_model, args, logger=None):
    if getattr(args, 'distributed', False):
        if getattr(args, 'sync_bn', False):
            base_model = nn.SyncBatchNorm.convert_sync_batchnorm(base_model)
            print_log('Using Synchronized BatchNorm ...', logger=logger)
        base_model = FlatDDP(base_model)
        print_log('Using flat-bucket data parallel (one process per GPU, RCCL) ...', logger=logger)
    return base_model


def run_net(args, config, train_writer=None, val_writer=None):
    logger = get_logger(args.log_name)
    device = torch.device('cuda', args.local_rank) if args.use_gpu else torch.device('cpu')
    (train_sampler, train_dataloader) = builder.dataset_builder(args, config.dataset.train)
    (_, test_dataloader) = builder.dataset_builder(args, config.dataset.val)
    base_model = builder.model_builder(config.model).to(device)

    start_epoch, best_metrics, metrics = 0, None, None
    if args.resume:
        start_epoch, best_metrics = builder.resume_model(base_model, args, logger=logger)
        best_metrics = Metrics(config.consider_metric, best_metrics)
    elif args.start_ckpts is not None:
        builder.load_model(base_model, args.start_ckpts, logger=logger)

    dataset_name = config.dataset.train._base_.NAME
    if 'SegImgPCN' in dataset_name and hasattr(base_model.base_model, 'segmentator'):
        load_frozen_segmentator(base_model, getattr(args, 'gdanet_w', None), logger)

    base_model = wrap_model(base_model, args, logger)
    optimizer = builder.build_optimizer(base_model, config)
    ChamferDisL1, ChamferDisL2 = ChamferDistanceL1(), ChamferDistanceL2()
    if args.resume:
        builder.resume_optimizer(optimizer, args, logger=logger)
    scheduler = builder.build_scheduler(base_model, optimizer, config, last_epoch=start_epoch - 1)
    step = TrainStep(base_model, optimizer, getattr(config, 'grad_norm_clip', 10),
                     getattr(config, 'step_per_update', 1))
    if getattr(args, 'graph_train', False) and not args.distributed and device.type == 'cuda':
        # --graph_train: the whole step replayed from one HIP graph per (input shapes, epoch-dependent scalars); pays for
        # the launch-bound batch-16 image / segmentation models (utils/graph_runner.py); eager again by itself if its
        # capture-time check against the eager step fails
        step = GraphedTrainStep(step)
        print_log('Train steps are replayed from HIP graphs (--graph_train)', logger=logger)
    log_every = int(getattr(args, 'log_every', 100))

    for epoch in range(start_epoch, config.max_epoch + 1):
        if args.distributed and train_sampler is not None:
            train_sampler.set_epoch(epoch)
        base_model.train()
        epoch_start = batch_start = time.time()
        batch_time, data_time = AverageMeter(), AverageMeter()
        losses = AverageMeter(['SparseLoss', 'DenseLoss'])
        loss_acc = torch.zeros(3, device=device)  # sum sparse, sum dense, count — stays on device
        n_batches = len(train_dataloader)
        for idx, (taxonomy_ids, model_ids, data) in enumerate(train_dataloader):
            data_time.update(time.time() - batch_start)
            partial, gt, extra = unpack_batch(dataset_name, data, device)
            sparse_loss, dense_loss = step(partial, gt, extra, epoch)
            loss_acc += torch.stack([sparse_loss, dense_loss, torch.ones((), device=device)])
            batch_time.update(time.time() - batch_start)
            batch_start = time.time()
            if idx % log_every == 0 or idx == n_batches - 1:
                acc = loss_acc.clone()
                if args.distributed:
                    torch.distributed.all_reduce(acc)  # one packed collective per log line
                    if step.ddp is not None:
                        step.ddp.check_late_gradients()   # same error on every rank instead of a hang (flat_ddp.py)
                s, d, c = acc.tolist()
                # window mean, weighted by the iterations it covers: losses.avg() is the per-iteration epoch mean
                # of the reference (tools/runner.py:221-222), losses.val() the mean since the last log line
                losses.update([s / c * 1000, d / c * 1000], n=max(1, round(c / max(1, args.world_size if args.distributed else 1))))
                loss_acc.zero_()
                n_itr = epoch * n_batches + idx
                if train_writer is not None:
                    train_writer.add_scalar('Loss/Batch/Sparse', losses.val(0), n_itr)
                    train_writer.add_scalar('Loss/Batch/Dense', losses.val(1), n_itr)
                print_log('[Epoch %d/%d][Batch %d/%d] BatchTime = %.3f (s) DataTime = %.3f (s) Losses = %s lr = %.6f' %
                          (epoch, config.max_epoch, idx + 1, n_batches, batch_time.val(), data_time.val(),
                           ['%.4f' % l for l in losses.val()], optimizer.param_groups[0]['lr']), logger=logger)
        for item in (scheduler if isinstance(scheduler, list) else [scheduler]):
            item.step()
        if train_writer is not None:
            train_writer.add_scalar('Loss/Epoch/Sparse', losses.avg(0), epoch)
            train_writer.add_scalar('Loss/Epoch/Dense', losses.avg(1), epoch)
        print_log('[Training] EPOCH: %d EpochTime = %.3f (s) Losses = %s' %
                  (epoch, time.time() - epoch_start, ['%.4f' % l for l in losses.avg()]), logger=logger)

        if epoch % args.val_freq == 0:
            metrics, _, _ = validate(base_model, test_dataloader, epoch, ChamferDisL1, ChamferDisL2, val_writer,
                                     args, config, logger=logger)
            if metrics.better_than(best_metrics):
                best_metrics = metrics
                builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, 'ckpt-best', args,
                                        logger=logger)
        builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, 'ckpt-last', args, logger=logger)
        if (config.max_epoch - epoch) < 2:
            builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, f'ckpt-epoch-{epoch:03d}',
                                    args, logger=logger)
    if train_writer is not None and val_writer is not None:
        train_writer.close()
        val_writer.close()
    return best_metrics


def _synset_names():
    """ShapeNet synset id -> category name for the TEST RESULTS table (the reference opens ./cfgs/shapenet_synset_dict.json
    relative to the working directory, runner.py:428); the copy shipped next to this package serves other directories."""
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for path in (os.path.join('cfgs', 'shapenet_synset_dict.json'), os.path.join(here, 'cfgs', 'shapenet_synset_dict.json')):
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh)
    return {}


def validate(base_model, test_dataloader, epoch, ChamferDisL1, ChamferDisL2, val_writer, args, config, logger=None):
    print_log(f'[VALIDATION] Start validating epoch {epoch}', logger=logger)
    if isinstance(base_model, FlatDDP):
        base_model.sync_buffers()      # rank 0's BatchNorm running statistics on every rank (DDP: broadcast_buffers)
    base_model.eval()
    device = next(base_model.parameters()).device
    test_losses = AverageMeter(['SparseLossL1', 'SparseLossL2', 'DenseLossL1', 'DenseLossL2'])
    test_metrics = AverageMeter(Metrics.names())
    category_metrics = dict()
    n_samples = len(test_dataloader)
    dataset_name = config.dataset.val._base_.NAME
    loss_L1, loss_L2 = [], []
    # bs = 1 validation is launch-bound: replay the forward as one HIP graph per input shape (utils/graph_runner.py)
    forward = GraphedForward(base_model, enabled=(device.type == 'cuda' and getattr(args, 'graph_eval', True)))
    with torch.no_grad():
        for idx, (taxonomy_ids, model_ids, data) in enumerate(test_dataloader):
            taxonomy_id = taxonomy_ids[0] if isinstance(taxonomy_ids[0], str) else taxonomy_ids[0].item()
            partial, gt, extra = unpack_batch(dataset_name, data, device)
            ret = forward(partial, *extra)
            coarse_points, dense_points = ret[0], ret[-1]
            if dense_points.size(0) != gt.size(0):      # multi-view batch: V completions per ground-truth cloud
                gt = gt.repeat_interleave(dense_points.size(0) // gt.size(0), dim=0)
            vals = [ChamferDisL1(coarse_points, gt), ChamferDisL2(coarse_points, gt),
                    ChamferDisL1(dense_points, gt), ChamferDisL2(dense_points, gt)]
            vals += Metrics.get(dense_points, gt)
            packed = torch.stack([v.float().reshape(()) for v in vals])
            local_dense = packed[2:4].tolist()
            loss_L1.append(local_dense[0] * 1000)
            loss_L2.append(local_dense[1] * 1000)
            if args.distributed:  # 8 scalars, one collective (the reference issues 8)
                torch.distributed.all_reduce(packed)
                packed /= args.world_size
            packed = packed.tolist()
            test_losses.update([v * 1000 for v in packed[:4]])
            _metrics = packed[4:]
            for _taxonomy_id in taxonomy_ids:
                if _taxonomy_id not in category_metrics:
                    category_metrics[_taxonomy_id] = AverageMeter(Metrics.names())
                category_metrics[_taxonomy_id].update(_metrics)
            if (idx + 1) % max(n_samples, 1) == 0:
                print_log('Test[%d/%d] Taxonomy = %s Sample = %s Losses = %s Metrics = %s' %
                          (idx + 1, n_samples, taxonomy_id, model_ids[0], ['%.4f' % l for l in test_losses.val()],
                           ['%.4f' % m for m in _metrics]), logger=logger)
        for _, v in category_metrics.items():
            test_metrics.update(v.avg())
        print_log('[Validation] EPOCH: %d  Metrics = %s' % (epoch, ['%.4f' % m for m in test_metrics.avg()]),
                  logger=logger)

    names = _synset_names()
    print_log('============================ TEST RESULTS ============================', logger=logger)
    print_log('Taxonomy\t#Sample\t' + '\t'.join(test_metrics.items) + '\t#ModelName\t', logger=logger)
    for taxonomy_id in category_metrics:
        msg = f'{taxonomy_id}\t{category_metrics[taxonomy_id].count(0)}\t'
        msg += ''.join('%.3f \t' % v for v in category_metrics[taxonomy_id].avg())
        print_log(msg + names.get(taxonomy_id, '?') + '\t', logger=logger)
    print_log('Overall\t\t' + ''.join('%.3f \t' % v for v in test_metrics.avg()), logger=logger)

    if val_writer is not None:
        val_writer.add_scalar('Loss/Epoch/Sparse', test_losses.avg(0), epoch)
        val_writer.add_scalar('Loss/Epoch/Dense', test_losses.avg(2), epoch)
        for i, metric in enumerate(test_metrics.items):
            val_writer.add_scalar('Metric/%s' % metric, test_metrics.avg(i), epoch)
    return Metrics(config.consider_metric, test_metrics.avg()), loss_L1, loss_L2


def test_net(args, config):
    logger = get_logger(args.log_name)
    print_log('Tester start ... ', logger=logger)
    _, test_dataloader = builder.dataset_builder(args, config.dataset.test)
    device = torch.device('cuda', args.local_rank) if args.use_gpu else torch.device('cpu')
    base_model = builder.model_builder(config.model)
    builder.load_model(base_model, args.ckpts, logger=logger)
    base_model = base_model.to(device)
    if args.distributed:
        raise NotImplementedError('test_net runs on a single process, as in the reference')
    metrics, _, _ = validate(base_model, test_dataloader, 0, ChamferDistanceL1(), ChamferDistanceL2(), None, args,
                             config, logger=logger)
    return metrics

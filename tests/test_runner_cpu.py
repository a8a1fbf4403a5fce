"""run_net / validate / checkpoint round trip on CPU (oracle backend) with a tiny AdaPoinTr."""
import argparse
import os

import pytest
import torch

from oracle_backend import OracleBackend


def tiny_config():
    from imgadapointr_amd.utils.config import EasyDict
    blk = dict(embed_dim=48, num_heads=6, k=8, n_group=2, mlp_ratio=2.)
    ds = dict(NAME='SyntheticPCN', N_POINTS=512, N_PARTIAL=256, LENGTH=4)
    return EasyDict(
        optimizer=dict(type='AdamW', kwargs=dict(lr=1e-3, weight_decay=5e-4)),
        scheduler=dict(type='LambdaLR', kwargs=dict(decay_step=21, lr_decay=0.9, lowest_decay=0.02)),
        bnmscheduler=dict(type='Lambda', kwargs=dict(decay_step=21, bn_decay=0.5, bn_momentum=0.9, lowest_decay=0.01)),
        dataset=dict(train=dict(_base_=dict(ds), others=dict(subset='train', bs=2)),
                     val=dict(_base_=dict(ds, LENGTH=2), others=dict(subset='val')),
                     test=dict(_base_=dict(ds, LENGTH=2), others=dict(subset='test'))),
        model=dict(NAME='AdaPoinTr', num_query=32, num_points=512, center_num=[64, 32], global_feature_dim=64,
                   encoder_type='graph', decoder_type='fc',
                   encoder_config=dict(blk, depth=2, block_style_list=['attn-graph', 'attn'], combine_style='concat'),
                   decoder_config=dict(blk, depth=2, self_attn_block_style_list=['attn-graph', 'attn'],
                                       self_attn_combine_style='concat',
                                       cross_attn_block_style_list=['attn-graph', 'attn'],
                                       cross_attn_combine_style='concat')),
        total_bs=2, step_per_update=1, max_epoch=1, consider_metric='CDL1')


@pytest.fixture()
def oracle_backend(ref):
    from imgadapointr_amd import _native
    _native.set_backend_for_testing(OracleBackend())
    yield
    _native.set_backend_for_testing(None)


def make_args(tmp_path, **kw):
    a = argparse.Namespace(distributed=False, world_size=1, local_rank=0, use_gpu=False, num_workers=0,
                           resume=False, start_ckpts=None, ckpts=None, val_freq=1, sync_bn=False, log_every=1,
                           experiment_path=str(tmp_path), log_name='tiny', test=False)
    a.__dict__.update(kw)
    return a


def test_run_net_validate_and_resume(tmp_path, oracle_backend):
    from imgadapointr_amd.tools import builder, runner
    from imgadapointr_amd.utils.metrics import Metrics
    torch.manual_seed(0)
    cfg = tiny_config()
    args = make_args(tmp_path)
    best = runner.run_net(args, cfg)                     # epochs 0 and 1, validation + checkpoints
    assert isinstance(best, Metrics)
    for name in ('ckpt-best.pth', 'ckpt-last.pth', 'ckpt-epoch-000.pth', 'ckpt-epoch-001.pth'):
        assert os.path.exists(tmp_path / name), name
    ck = torch.load(tmp_path / 'ckpt-last.pth', weights_only=False)
    assert set(ck) == {'base_model', 'optimizer', 'epoch', 'metrics', 'best_metrics'}   # builder.py:132-138
    assert ck['epoch'] == 1 and set(ck['metrics']) == {'F-Score', 'CDL1', 'CDL2', 'EMDistance'}
    assert all(k.startswith(('base_model.', 'decode_head.', 'increase_dim.', 'reduce_map.')) for k in ck['base_model'])

    # resume picks up model, optimizer and epoch
    model = builder.model_builder(cfg.model)
    start_epoch, best_metrics = builder.resume_model(model, make_args(tmp_path, resume=True))
    assert start_epoch == 2 and 'CDL1' in best_metrics
    for k, v in model.state_dict().items():
        assert torch.equal(v, ck['base_model'][k])
    # load_model accepts both 'base_model' and 'model' keys and strips 'module.' (builder.py:149-154)
    torch.save({'model': {'module.' + k: v for k, v in ck['base_model'].items()}}, tmp_path / 'alt.pth')
    builder.load_model(builder.model_builder(cfg.model), str(tmp_path / 'alt.pth'))

    # test_net on the saved checkpoint
    m = runner.test_net(make_args(tmp_path, ckpts=str(tmp_path / 'ckpt-best.pth'), test=True), cfg)
    assert m.state_dict()['CDL1'] > 0


@pytest.mark.parametrize("model", [dict(NAME='PCN', num_pred=512, encoder_channel=1024),
                                   dict(NAME='FoldingNet', num_pred=529, encoder_channel=64),
                                   dict(NAME='TopNet', node_feature=8, num_pred=2048, encoder_channel=64, nlevels=4)])
def test_run_net_chamfer_baselines(tmp_path, oracle_backend, model):
    """The Adam + StepLR recipes of cfgs/PCN_models/{PCN,FoldingNet,TopNet}.yaml through the same run_net."""
    from imgadapointr_amd.tools import runner
    from imgadapointr_amd.utils.config import EasyDict
    torch.manual_seed(0)
    cfg = tiny_config()
    cfg.pop('bnmscheduler')
    cfg.optimizer = EasyDict(type='Adam', kwargs=dict(lr=1e-3, weight_decay=0))
    cfg.scheduler = EasyDict(type='StepLR', kwargs=dict(step_size=50, gamma=0.5))
    cfg.model = EasyDict(model)
    best = runner.run_net(make_args(tmp_path), cfg)
    assert best.state_dict()['CDL1'] > 0
    assert os.path.exists(tmp_path / 'ckpt-best.pth')


def test_metrics_f_score_and_cd(oracle_backend):
    from imgadapointr_amd.utils.metrics import Metrics
    torch.manual_seed(1)
    gt = torch.rand(2, 300, 3)
    vals = Metrics.get(gt.clone(), gt)
    assert abs(vals[0].item() - 1.0) < 1e-6 and vals[1].item() == 0 and vals[2].item() == 0 and vals[3].item() == 0
    far = gt + 10.0
    assert Metrics.get(far, gt)[0].item() == 0.0
    half = gt.clone()
    half[:, :150] += 10.0                     # half the predicted points are far away
    f = Metrics.get(half, gt)[0].item()       # precision 0.5, recall 0.5 -> F = 0.5
    assert abs(f - 0.5) < 1e-6
    a, b = Metrics('CDL1', {'CDL1': 5.0}), Metrics('CDL1', {'CDL1': 6.0})
    assert a.better_than(b) and not b.better_than(a) and a.better_than(None)
    c, d = Metrics('F-Score', {'F-Score': 0.9}), Metrics('F-Score', {'F-Score': 0.8})
    assert c.better_than(d)


def test_parser_paths(tmp_path, monkeypatch):
    from imgadapointr_amd.utils import parser
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv('LOCAL_RANK', raising=False)
    args = parser.get_args(['--config', 'cfgs/PCN_models/AdaPoinTr.yaml', '--exp_name', 'x', '--mode', 'hard'])
    assert args.experiment_path == os.path.join('./experiments', 'AdaPoinTr', 'PCN_models', 'x_hard')
    assert args.tfboard_path == os.path.join('./experiments', 'AdaPoinTr', 'PCN_models', 'TFBoard', 'x_hard')
    assert args.log_name == 'AdaPoinTr' and os.path.isdir(args.experiment_path)
    with pytest.raises(ValueError):
        parser.get_args(['--test', '--resume'])


def test_graphed_forward_device_rules():
    """GraphedForward keys its graphs by the inputs' device and never graphs CPU or mixed-device calls
    (utils/graph_runner.py; the capture / replay paths themselves are GPU tests)."""
    from imgadapointr_amd.utils.graph_runner import GraphedForward, input_device, signature

    class M(torch.nn.Module):
        def forward(self, x):
            return x + 1

    a = torch.zeros(2, 3)
    assert input_device([a]) is None and input_device([a, "x"]) is None and input_device([]) is None
    m0, m1 = torch.empty(2, 3, device="meta"), torch.empty(2, 3, device="meta", dtype=torch.float16)
    assert signature([m0]) != signature([m1])
    assert signature([m0])[0][2:] == ("meta", None) or signature([m0])[0][2] == "meta"   # device type and index are in the key
    gf = GraphedForward(M().eval())
    with torch.no_grad():
        assert torch.equal(gf(a), a + 1)                           # CPU input: plain eager call
    assert gf.replays == 0 and gf.enabled and not gf._cache


def test_average_meter_weighted_update_is_the_per_iteration_mean():
    """The runner reads its on-device loss accumulator once per log line and feeds the window mean with the window's
    iteration count: avg() must equal the reference's one-update-per-iteration mean (utils/AverageMeter.py)."""
    from imgadapointr_amd.utils.AverageMeter import AverageMeter
    vals = [float(v) for v in range(1, 12)]                 # 11 iterations, windows of 1, 5 and 5 (log_every = 5)
    per_iter, windowed = AverageMeter(['a']), AverageMeter(['a'])
    for v in vals:
        per_iter.update([v])
    for lo, hi in ((0, 1), (1, 6), (6, 11)):
        w = vals[lo:hi]
        windowed.update([sum(w) / len(w)], n=len(w))
    assert abs(windowed.avg(0) - per_iter.avg(0)) < 1e-12
    assert windowed.count(0) == 11 and windowed.val(0) == sum(vals[6:]) / 5


def test_clip_adamw_host_path_matches_torch():
    """ClipAdamW on CPU parameters (no device kernels) takes the stock path: clip_grad_norm_ + torch's AdamW step,
    same state keys, and its plain step() is torch's."""
    import torch
    from imgadapointr_amd.utils.clip_adamw import ClipAdamW
    torch.manual_seed(0)
    a = [torch.randn(5, 7, requires_grad=True), torch.randn(7, requires_grad=True)]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    ref = torch.optim.AdamW([{"params": [a[1]], "weight_decay": 0.0}, {"params": [a[0]], "weight_decay": 1e-2}], lr=1e-2)
    our = ClipAdamW([{"params": [b[1]], "weight_decay": 0.0}, {"params": [b[0]], "weight_decay": 1e-2}], lr=1e-2)
    for _ in range(3):
        for p, q in zip(a, b):
            g = torch.randn_like(p) * 50
            p.grad, q.grad = g.clone(), g.clone()
        torch.nn.utils.clip_grad_norm_(a, 10.0)
        ref.step()
        our.step_clipped(10.0)
    for p, q in zip(a, b):
        assert torch.equal(p, q)
        assert set(our.state[q]) == {"step", "exp_avg", "exp_avg_sq"}


def test_batch_size_of_the_image_yaml_under_several_ranks():
    """cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml carries the reference's placeholder `total_bs: 1` (its
    distributed loader hard-codes 64 per rank, builder.py:19).  A multi-rank launch of that YAML must get a loader, not an
    assertion: 64 per rank by default, `--total_bs` when given; non-divisible explicit batches stay an error."""
    from types import SimpleNamespace
    from imgadapointr_amd.tools import builder
    from imgadapointr_amd.utils.config import cfg_from_yaml_file
    from imgadapointr_amd.utils import parser
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cwd = os.getcwd()
    os.chdir(root)
    try:
        cfg = cfg_from_yaml_file("cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml")
    finally:
        os.chdir(cwd)
    assert cfg.total_bs == 1
    assert builder.resolve_batch_size(cfg.total_bs, 8, True) == builder.REFERENCE_DISTRIBUTED_BS == 64
    assert builder.resolve_batch_size(cfg.total_bs, 8, True, override=128) == 16
    assert builder.resolve_batch_size(cfg.total_bs, 1, False) == 1
    assert builder.resolve_batch_size(cfg.total_bs, 1, False, override=16) == 16
    assert builder.resolve_batch_size(256, 8, True) == 32
    with pytest.raises(AssertionError):
        builder.resolve_batch_size(12, 8, True)
    with pytest.raises(AssertionError):
        builder.resolve_batch_size(1, 8, True, override=12)
    assert parser.build_parser().parse_args(['--total_bs', '128']).total_bs == 128
    # and the loader a rank of a 2-rank job builds from it (synthetic dataset in place of the PCN tree)
    from imgadapointr_amd.utils.config import EasyDict
    ds = EasyDict(_base_=EasyDict(NAME='SyntheticPCN', N_POINTS=64, N_PARTIAL=32, LENGTH=256),
                  others=EasyDict(subset='train', bs=builder.resolve_batch_size(cfg.total_bs, 2, True)))
    import torch.distributed as dist
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29653', rank=0, world_size=1)
    try:
        sampler, loader = builder.dataset_builder(SimpleNamespace(distributed=True, num_workers=0), ds)
        assert loader.batch_size == 64 and sampler is not None
    finally:
        if created:
            dist.destroy_process_group()


def test_validate_and_metrics_match_the_reference(oracle_backend):
    import parity_checks as pc
    print(pc.check_validate("cpu"))


def test_inference_cli_matches_the_reference(tmp_path, oracle_backend):
    import parity_checks as pc
    print(pc.check_inference("cpu", tmp_path))


def test_graphed_train_step_falls_through_on_cpu_and_private_state_swaps_tables(oracle_backend):
    """`GraphedTrainStep` on CPU tensors is the eager `TrainStep` (no GPU, nothing to capture), and `_native.private_state()`
    swaps this module's workspace tables in and out without touching the global ones."""
    from imgadapointr_amd import _native
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.tools import builder, runner
    from imgadapointr_amd.utils.graph_runner import GraphedTrainStep
    torch.manual_seed(0)
    cfg = tiny_config()
    model = build_model_from_cfg(cfg.model).train()
    step = GraphedTrainStep(runner.TrainStep(model, builder.build_optimizer(model, cfg)))
    partial, gt = torch.rand(2, 256, 3) - 0.5, torch.rand(2, 512, 3) - 0.5
    before = [p.detach().clone() for p in model.parameters()]
    l0, l1 = step(partial, gt, (), 0)
    assert torch.isfinite(l0) and torch.isfinite(l1) and step.replays == 0 and step.enabled
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))          # the eager step ran
    assert step.optimizer is step.step.optimizer and step.ddp is None
    marker = object()
    _native._colsum_ws["marker"] = marker
    try:
        state = _native.private_state()
        with state:
            assert "marker" not in _native._colsum_ws
            _native._colsum_ws["inside"] = 1
        assert _native._colsum_ws.get("marker") is marker and "inside" not in _native._colsum_ws
        assert state.tables["_colsum_ws"] == {"inside": 1}
        with state:                                                   # the same tables come back on re-entry
            assert _native._colsum_ws == {"inside": 1}
    finally:
        _native._colsum_ws.pop("marker", None)


def test_no_gemm_tuning_guard_restores_the_tuning_state(monkeypatch):
    """`graph_runner._no_gemm_tuning` switches TunableOp's TUNING off for the duration of a capture and back on afterwards,
    only if it was on; host logic, checked against a stand-in for `torch.cuda.tunable`."""
    import types
    import torch
    from imgadapointr_amd.utils import graph_runner
    state = {"enabled": True, "tuning": True, "calls": []}
    fake = types.SimpleNamespace(is_enabled=lambda: state["enabled"], tuning_is_enabled=lambda: state["tuning"],
                                 tuning_enable=lambda v: (state.__setitem__("tuning", bool(v)), state["calls"].append(bool(v))))
    monkeypatch.setattr(torch.cuda, "tunable", fake, raising=False)
    with graph_runner._no_gemm_tuning():
        assert state["tuning"] is False
    assert state["tuning"] is True and state["calls"] == [False, True]
    try:
        with graph_runner._no_gemm_tuning():
            raise RuntimeError("capture failed")
    except RuntimeError:
        pass
    assert state["tuning"] is True                         # restored on the error path as well
    state.update(tuning=False, calls=[])
    with graph_runner._no_gemm_tuning():
        pass
    assert state["calls"] == [] and state["tuning"] is False   # was off: left alone
    state.update(enabled=False, tuning=True, calls=[])
    with graph_runner._no_gemm_tuning():
        pass
    assert state["calls"] == []                            # TunableOp not in use: left alone

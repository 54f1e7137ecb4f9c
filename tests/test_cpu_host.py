"""CPU: host logic, the C-ABI library's exports, state_dict compatibility, refusal of CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from heatnet_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "heatnet_hip.h")).read()
    declared = set(re.findall(r"\b(hn_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), name
    # (the three with a non-int return type are bound separately in _lib.load())
    assert declared - {"hn_strerror", "hn_version", "hn_weight_prep_table_bytes"} == set(_lib.SIGNATURES)
    assert ctypes.sizeof(_lib.HnWeightPrepItem) == 4 * 8 + 8 * 4 and lib.hn_weight_prep_table_bytes(2) > 2 * 772
    assert lib.hn_strerror(-4) == b"inconsistent shapes"
    assert ctypes.sizeof(_lib.HnConvDesc) == 14 * 4


def test_cpu_tensors_are_refused():
    from heatnet_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d(torch.zeros(1, 64, 4, 4), torch.zeros(64, 64, 1, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cross_entropy(torch.zeros(1, 13, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))


def test_state_dict_layout_matches_oracle_and_reference():
    from heatnet_amd.pspnet import PSPNet
    from heatnet_amd import conf_segnet
    from oracle import heatnet_ref as R
    for late in (True, False):
        mine = PSPNet(backend="resnet50", pretrained=False, late_fusion=late, in_channels=4)
        ref = R.RefPSPNet(backend="resnet50", late_fusion=late, in_channels=4)
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape for k in a)
    assert len(PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4).state_dict()) == 488
    m = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="pspnet", late_fusion=True)
    r = R.RefConfSegnet()
    assert list(m.state_dict()) == list(r.state_dict())
    assert sum(p.numel() for p in m.parameters()) == 75794006
    # conv weights are physically OHWI
    w = mine.feats.layer1[0].conv2.weight
    assert w.shape == (64, 64, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    # checkpoint interchange keeps logical values
    sd = {k: v.clone() for k, v in r.state_dict().items()}
    m.load_state_dict(sd)
    assert torch.equal(m.state_dict()["critics.0.conv1.weight"], sd["critics.0.conv1.weight"])
    assert m.critics[0].conv1.weight.is_contiguous(memory_format=torch.channels_last)


def test_caller_side_initialisers_hit_our_modules():
    """scripts/main.py:init_weights keys on isinstance(nn.Conv2d); cm/utils.py:weights_init_normal on class names."""
    from heatnet_amd.pspnet import PSPNet
    from heatnet_amd.conf_segnet import weights_init_normal
    net = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    # 23 bottlenecks x 3 + 2 stems + 5 projections + 4 pyramid + bottleneck + 3 decoder + final
    assert len(convs) == 85 and len(bns) == 79
    for m in net.modules():
        name = m.__class__.__name__
        if "Conv" in name or "BatchNorm2d" in name:
            assert hasattr(m, "weight")
    net.apply(weights_init_normal)
    assert abs(float(net.feats.conv1.weight.std()) - 0.02) < 0.005
    for m in convs:
        torch.nn.init.xavier_normal_(m.weight, 1.0)


def test_phase_switch_and_drop_in_module_paths():
    import models.build_net as top
    import models.confusion_maximization.models.build_net as cm
    from models.confusion_maximization.models.conf_segnet import conv_segnet
    from models.confusion_maximization.utils import calculate_ious
    from oracle import heatnet_ref as R
    assert set(top.models) >= {"resnet18", "resnet34", "resnet50", "resnet101", "resnet152"}
    assert callable(top.build_network) and callable(cm.build_network)
    m = conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="pspnet", late_fusion=True)
    m.setPhase("train_critic")
    assert not any(p.requires_grad for p in m.trgb_segnet.parameters())
    assert all(p.requires_grad for p in m.critics.parameters())
    m.setPhase("train_seg")
    assert all(p.requires_grad for p in m.trgb_segnet.parameters())
    assert not any(p.requires_grad for p in m.critics.parameters())
    with pytest.raises(NotImplementedError):
        conv_segnet(pretrained=False, arch="custom")  # (the default disc_arch 'resnet': critics the reference cannot train)
    # the reference's default arch: ResNeXt-50 + ASPP, same state_dict keys as the oracle pinned to the reference's module
    from models.confusion_maximization.models import trgb_segnet
    from oracle import resnext_ref
    assert callable(trgb_segnet.net_resnext50) and callable(trgb_segnet.net_resnext101) and callable(trgb_segnet.net_resnext152)
    c = conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="custom", late_fusion=True)
    ref_keys = set(resnext_ref.RefResNeXt(late_fusion=True).state_dict())
    assert set(c.trgb_segnet.state_dict()) == ref_keys
    r = np.random.RandomState(0)
    pred, tgt = r.randint(0, 13, (2, 30, 40)), r.randint(0, 14, (2, 30, 40))
    np.testing.assert_array_equal(calculate_ious(torch.from_numpy(pred), torch.from_numpy(tgt)),
                                  R.calculate_ious(pred, tgt))


def test_miou_host_arithmetic_against_golden(golden_dir):
    from heatnet_amd import miou
    g = np.load(os.path.join(golden_dir, "miou_cases.npz"))
    for case in range(3):
        pred, tgt = g["c%d_pred" % case].astype(np.int64), g["c%d_tgt" % case].astype(np.int64)
        conf = np.bincount((tgt * 14 + pred).reshape(-1), minlength=196).reshape(14, 14)
        np.testing.assert_array_equal(miou.ious_from_confusion(conf), g["c%d_ious" % case])
        iou, m = miou.iou_eval_value(conf, (12, 13))
        np.testing.assert_array_equal(iou, g["c%d_cm_iou" % case])
        assert m == float(g["c%d_cm_miou" % case])


def test_config_file_is_the_reference_file():
    import json
    conf = json.load(open(os.path.join(ROOT, "experiments", "heatnet_conf.json")))
    assert conf["optimizer"]["learning_rate"] == 1e-05 and conf["optimizer"]["batch_size"] == 30
    assert conf["network"]["arch"] == "resnext50" and conf["optimizer"]["clip"] == 0


def test_checkpoint_interchange_with_reference_format(tmp_path):
    """Reference-format checkpoints (`module.`-prefixed keys, {'epoch','state_dict',...}) load into our networks and
    ours load back into the oracle (= reference layout); snapshot files `<name>_<epoch>` go through build_network."""
    from heatnet_amd import checkpoint
    from heatnet_amd.pspnet import PSPNet
    from oracle import heatnet_ref as R
    from _seeded import seeded_state
    ref = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    state = seeded_state(ref, seed=3)
    ckpt = {"epoch": 7, "state_dict": {"module.trgb_segnet." + k: v for k, v in state.items()}, "best_iou": 0.5}
    path = tmp_path / "training_nc_irrgb_best.pth"
    torch.save(ckpt, str(path))
    mine = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
    checkpoint.init_model_renamed(mine, str(path), "module.trgb_segnet.", "")
    got = mine.state_dict()
    assert all(torch.equal(got[k], state[k]) for k in state)
    assert mine.feats.layer4[0].conv2.weight.is_contiguous(memory_format=torch.channels_last)
    # ours -> reference layout
    out = tmp_path / "ours.pth.tar"
    checkpoint.save_checkpoint({"epoch": 1, "state_dict": mine.state_dict(), "best_iou": 0.0}, True, str(out))
    back = torch.load(str(out))
    ref2 = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    ref2.load_state_dict(back["state_dict"])
    assert all(v.is_contiguous() for v in back["state_dict"].values() if torch.is_tensor(v))
    assert os.path.exists(str(out) + "_best.pth.tar")
    # snapshot convention of build_network: "<name>_<epoch>"
    from heatnet_amd import build_net
    snap = tmp_path / "PSPNet_12"
    torch.save(mine.state_dict(), str(snap))
    net = build_net.build_network(str(snap), "resnet50", in_channels=4, late_fusion=True)
    assert torch.equal(net.state_dict()["final.0.bias"].cpu(), state["final.0.bias"])


def test_phase_schedule_matches_oracle():
    """train_trgb_segnet_conf.py:361-367,577-592: critic warm-up, then alternating seg / critic phases."""
    from heatnet_amd.train_step import PhaseSchedule
    from oracle import heatnet_ref as R
    a, b = PhaseSchedule(initial_critic=5, seg=3, critic=4), R.PhaseSchedule(initial_critic=5, seg=3, critic=4)
    for _ in range(40):
        assert a.state == b.state
        assert a.tick() == b.tick()
    c = PhaseSchedule(no_conf=True)
    assert c.state == "train_seg" and not c.tick() and c.state == "train_seg"


def test_flat_arena_layout_on_cpu():
    """ParamArena: 256-B slots, parameter views keep their strides, gradients are views of one flat buffer."""
    from heatnet_amd import optim
    from heatnet_amd.critics import FCDiscriminator
    net = FCDiscriminator(13)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    arena = optim.ParamArena(list(net.parameters()))
    assert all(off % 64 == 0 for off in arena.offsets) and arena.numel % 64 == 0
    assert all(torch.equal(v, before[k]) for k, v in net.state_dict().items())
    assert net.conv1.weight.is_contiguous(memory_format=torch.channels_last)
    for p in net.parameters():
        assert p.grad is not None and p.grad.stride() == p.stride()
    arena.check()
    net.conv1.weight.grad.fill_(1.0)
    assert float(arena.grad.sum()) == net.conv1.weight.numel()
    groups = optim.arenas_for(net)
    assert len(groups) == 1


def test_flat_optimizer_state_interchange(tmp_path):
    """optim.FlatRMSprop / FlatAdam state_dict() <-> torch.optim.RMSprop / Adam (the reference's checkpoints keep
    `optimizer.state_dict()`, cm/train_trgb_segnet_conf.py:648-654), and checkpoint.resume() end to end."""
    import copy
    import torch.nn as nn
    from heatnet_amd import checkpoint, optim

    def make():
        torch.manual_seed(0)
        net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.PReLU(), nn.Conv2d(8, 4, 1))
        net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)  # as HnConv2d keeps them
        return net

    for flat_cls, ref_cls, bufs, kw in ((optim.FlatRMSprop, torch.optim.RMSprop, ("square_avg",), dict(lr=1e-4)),
                                        (optim.FlatAdam, torch.optim.Adam, ("exp_avg", "exp_avg_sq"), dict(lr=1e-3))):
        ref_net = make()
        ref_opt = ref_cls(ref_net.parameters(), **kw)
        for _ in range(2):  # two real torch steps -> a reference-format optimizer state
            ref_opt.zero_grad()
            ref_net(torch.randn(2, 3, 8, 8)).sum().backward()
            ref_opt.step()
        net = copy.deepcopy(ref_net)
        flat = flat_cls(optim.arenas_for(net), **kw)
        flat.load_state_dict(ref_opt.state_dict())                       # torch -> flat
        assert flat.steps == [2]
        arena = flat.arenas[0]
        for p_ref, p, off in zip(ref_net.parameters(), arena.params, arena.offsets):
            for name in bufs:
                got = getattr(flat, name)[0][off:off + p.numel()].as_strided(p.size(), p.stride())
                assert torch.equal(got, ref_opt.state[p_ref][name]), name
        back = ref_cls(copy.deepcopy(ref_net).parameters(), **kw)
        back.load_state_dict(flat.state_dict())                           # flat -> torch
        for pa, pb in zip(back.param_groups[0]["params"], ref_opt.param_groups[0]["params"]):
            for name in bufs:
                assert torch.equal(back.state[pa][name], ref_opt.state[pb][name]), name
            assert float(back.state[pa]["step"]) == 2.0
        # a checkpoint file written the reference's way (DataParallel prefix) resumes into the flat optimizer
        sched = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=30, gamma=0.5)
        path = str(tmp_path / ("ckpt_%s.pth.tar" % flat_cls.__name__))
        checkpoint.save_checkpoint({"epoch": 7, "state_dict": {"module." + k: v for k, v in ref_net.state_dict().items()},
                                    "best_iou": 0.42, "optimizer": ref_opt.state_dict(),
                                    "lr_scheduler": sched.state_dict()}, False, filename=path)
        net2 = make()
        flat2 = flat_cls(optim.arenas_for(net2), **kw)
        sched2 = torch.optim.lr_scheduler.StepLR(flat2, step_size=30, gamma=0.5)
        epoch, best = checkpoint.resume(path, net2, flat2, sched2)
        assert (epoch, best) == (7, 0.42) and flat2.steps == [2]
        assert all(torch.equal(a, b) for a, b in zip(net2.state_dict().values(), ref_net.state_dict().values()))
        assert torch.equal(getattr(flat2, bufs[0])[0], getattr(flat, bufs[0])[0])


def test_feedback_seg_layout_matches_reference(golden_dir):
    """conv_segnet(feedback_seg=True): the DownNets and the widened critics carry the reference's state_dict keys and
    shapes (cm/models/conf_segnet.py:61-70; fixture written from the reference by make_golden.py --part feedback)."""
    import numpy as np
    from heatnet_amd import conf_segnet
    g = np.load(os.path.join(golden_dir, "feedback_seg.npz"), allow_pickle=False)
    model = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="pspnet", late_fusion=True,
                                    feedback_seg=True)
    sd = model.state_dict()
    mine = [k for k in sd.keys() if not k.startswith("trgb_segnet.")]
    assert mine == [str(k) for k in g["keys"]]
    for k, shape in zip(mine, g["shapes"]):
        assert list(sd[k].shape) == [int(v) for v in shape[:sd[k].dim()]], k
    assert "12 channels" in str(g["forward_error"])  # the reference cannot run this configuration on 13 classes either


def test_data_parallel_replication_is_refused():
    """nn.DataParallel over several devices replicates modules through `_replicate_for_data_parallel`; the HIP-backed layers
    refuse it with a message that names the supported wrappers (the INTEGRATION.md claim was untested)."""
    from heatnet_amd.layers import HnBatchNorm2d, HnConv2d
    for m in (HnConv2d(4, 4, 1), HnBatchNorm2d(4)):
        with pytest.raises(RuntimeError, match="DistributedDataParallel"):
            m._replicate_for_data_parallel()
    assert isinstance(HnConv2d(4, 4, 1), torch.nn.Conv2d) and isinstance(HnBatchNorm2d(4), torch.nn.BatchNorm2d)


def test_plan_names_the_kernel_instantiation_of_a_launch():
    """hn_conv2d_plan is host logic only: the name rocprofv3 will print for a plane-tensor launch -- tile code 7 is the 224-row
    tile (LIVE = 14), and has_planes & 4 names the instantiation whose epilogue carries a BatchNorm-backward reduction (the
    kernel's last template argument)."""
    import ctypes
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    d = _lib.HnConvDesc(8, 40, 88, 512, 512, 40, 88, 512, 512, 3, 3, 1, 4, 4)

    def name(kind, has_planes, tile):
        buf = ctypes.create_string_buffer(128)
        assert lib.hn_conv2d_plan(kind, ctypes.byref(d), has_planes, ops._variant(tile & 3, tile >> 2, 0, 4), buf, 128) >= 0
        return buf.value.decode()
    assert name(0, 3, 7) == "conv_igemm4_kernel<2, 4, 4, 2, 2, 2, 14, 0>"
    assert name(1, 3, 7) == "conv_igemm4_kernel<2, 4, 4, 2, 2, 2, 14, 0>"
    assert name(1, 7, 7) == "conv_igemm4_kernel<2, 4, 4, 2, 2, 2, 14, 1>"
    assert name(1, 7, 2) == "conv_igemm4_kernel<2, 4, 2, 2, 3, 2, 0, 1>"
    assert name(0, 3, 0) == "conv_igemm4_kernel<2, 2, 2, 2, 2, 2, 0, 0>"

"""Segmentation network + domain critics wrapper (the object the HeatNet trainer drives).

models/confusion_maximization/models/conf_segnet.py:22-140 for arch='pspnet' and arch='custom' (ResNeXt-50 + ASPP,
trgb_segnet.py), disc_arch='cyclegan', with the `feedback_seg` branch (DownNets appended to the critic inputs,
:61-70,121-125).  The remaining optional branches of the reference (ResNet critics, input adapter) are not working
configurations of its training loop and raise NotImplementedError; `cert_branch` builds the certainty head
of the custom architecture (early fusion; trgb_segnet.py:344-346, 383-387) and train_step adds the trainer's certainty loss.
"""
import torch
import torch.nn as nn

from . import build_net, ops
from .critics import FCDiscriminator
from .downscale_network import DownNet


def weights_init_normal(m):
    """models/confusion_maximization/utils.py:126-132."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm2d") != -1:
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


def create_critic(disc_arch, input_num):
    if disc_arch == "cyclegan":
        return FCDiscriminator(input_num)
    # cm/models/critic_resnet.py: a ResNet with a one-logit fc head.  Its [N, 1] output cannot pass the trainer's own
    # train_seg phase (`c_a.size(2)` at cm/train_trgb_segnet_conf.py:534 raises IndexError -- verified by running the
    # reference), so only the fully-convolutional critic is a working configuration of the path.
    raise NotImplementedError("critic architecture %r: only 'cyclegan' (FCDiscriminator) is built; the reference's ResNet "
                              "critics cannot complete its own train_seg phase (2-D critic output)" % disc_arch)


class conv_segnet(nn.Module):
    def __init__(self, pretrained=True, disc_arch="resnet", num_critics=6, feedback_seg=False, no_conf=False,
                 modalities="ir_rgb", input_adapter=False, cert_branch=False, arch="custom", late_fusion=False):
        super().__init__()
        if arch not in ("pspnet", "custom"):
            raise NotImplementedError("arch=%r: the reference itself only knows 'pspnet' and 'custom'" % arch)
        if input_adapter:
            # cm/models/conf_segnet.py:108-112 hands the [rgb, ir] LIST to the UNet adapter: with arch='pspnet' the
            # reference raises TypeError in its first conv (verified by running it); there is nothing to be compatible with
            raise NotImplementedError("input_adapter: the reference's adapter takes a single tensor (arch='custom'); with the "
                                      "PSPNet path's [rgb, ir] inputs its own forward raises TypeError")
        self.cert_branch = bool(cert_branch)  # (arch='custom' only builds the head; PSPNet returns cert = None, as in the reference)
        num_input_channels = (3 if "rgb" in modalities else 0) + (1 if "ir" in modalities else 0)
        if arch == "custom":  # :38-43 -- ResNeXt-50 (64 groups) + ASPP, the reference's default architecture
            from .trgb_segnet import ResNeXt
            self.trgb_segnet = ResNeXt(structure=[3, 4, 6, 3], input_channels=num_input_channels, cert_branch=cert_branch,
                                       late_fusion=late_fusion)
            critic_num = [13, 768, 1024, 512, 256 * 2, 64 * 2] if late_fusion else [13, 512, 1024, 512, 256, 64]
        else:
            self.trgb_segnet = build_net.build_network(None, "resnet50", in_channels=num_input_channels,
                                                       late_fusion=late_fusion)
            critic_num = [13, 2048, 1024, 512 * 2, 256 * 2, 64 * 2] if late_fusion else [13, 2048, 1024, 512, 256, 64]
        self.trgb_segnet.apply(weights_init_normal)
        self.feedback_seg = feedback_seg
        self.input_adapter = False
        if not no_conf:
            if feedback_seg:  # :61-70 -- every critic but the first also sees the (downscaled) segmentation output
                num_downscale = [3, 3, 3, 2, 2]
                critic_num = [critic_num[0]] + [c + 12 for c in critic_num[1:]]
                self.downscale_nets = nn.ModuleList([DownNet(k) for k in num_downscale])
            self.critics = nn.ModuleList([create_critic(disc_arch, c) for c in critic_num[0:num_critics]])
        if pretrained:
            # the reference loads 'models_finished/training_nc_irrgb_best.pth' relative to the cwd here
            from .checkpoint import init_model_renamed
            init_model_renamed(self.trgb_segnet, "models_finished/training_nc_irrgb_best.pth", "module.", "")
        self.phase = "train_seg"
        self.no_conf = no_conf

    def setLearningModel(self, module, val):
        for p in module.parameters():
            p.requires_grad = val

    def setPhase(self, phase):
        self.phase = phase
        print("Switching to phase: %s" % self.phase)
        if phase not in ("train_seg", "train_critic"):
            return
        seg = phase == "train_seg"
        if not self.no_conf:
            for c in self.critics:
                self.setLearningModel(c, not seg)
        self.setLearningModel(self.trgb_segnet, seg)

    def forward(self, input_a, input_b):
        output = {}
        pred_label_day, inter_f_a, cert_a = self.trgb_segnet(*input_a)
        pred_label_night, inter_f_b, cert_b = self.trgb_segnet(*input_b)
        if not self.no_conf:
            output["critics_a"], output["critics_b"] = [], []
            for i, c in enumerate(self.critics):
                if self.feedback_seg and i > 0:  # :121-125 (raises for the 13-class PSPNet head, as the reference does)
                    inter_f_a[i] = ops.concat([inter_f_a[i], self.downscale_nets[i - 1](pred_label_day)])
                    inter_f_b[i] = ops.concat([inter_f_b[i], self.downscale_nets[i - 1](pred_label_night)])
                output["critics_a"].append(c(inter_f_a[i]))
                output["critics_b"].append(c(inter_f_b[i]))
        output["pred_label_a"] = pred_label_day
        output["pred_label_b"] = pred_label_night
        output["cert_a"] = cert_a
        output["cert_b"] = cert_b
        output["inter_f_b"] = inter_f_b
        return output

"""heatnet-pub_amd: MI355X-native implementation of HeatNet's data-parallel training hot path.

Public surface (mirrors the reference's model-construction boundary):
    heatnet_amd.build_net.build_network            -- both reference signatures
    heatnet_amd.pspnet.PSPNet / extractors.ResNet  -- state_dict-compatible networks
    heatnet_amd.conf_segnet.conv_segnet            -- segnet + critics wrapper
    heatnet_amd.train_step                         -- the confusion-maximisation iteration
    heatnet_amd.ddp.GradBuckets                    -- RCCL gradient all-reduce
All arithmetic runs in libheatnet_hip.so (hand-written gfx950 kernels, C ABI in include/heatnet_hip.h).
"""
__version__ = "0.1.0"

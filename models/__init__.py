"""Drop-in for the reference's `models` package on the PSPNet path (MI355X build).

Same module paths and names as /models/{build_net,pspnet,extractors}.py and
/models/confusion_maximization/... of jzuern/heatnet-pub, re-exporting the HIP-backed implementations in
`heatnet-pub_amd/` so that `scripts/main.py` / `train_trgb_segnet_conf.py` import them unchanged.
"""

"""torch.hub entry point (drop-in for /root/reference/hubconf.py:19-93).

    model = torch.hub.load('<this repo>', 'MPMAE', model_name='convnextv2_atto',
                           ckpt_name='pt-all_mod_atto_1M_64_uncertainty_56-8', pretrained=True, linear_probe=True)

Returns the dense ConvNeXt V2 with the encoder of an MP-MAE pretraining checkpoint loaded through
`remap_checkpoint_keys`. `ckpt_name` is one of the reference's released names (downloaded with
torch.hub.load_state_dict_from_url) **or a local path / file:// URL of a checkpoint written by
main_pretrain.py of this repository** - the produced checkpoints are consumable without the reference."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mmearth_train_amd import convnextv2  # noqa: E402
from mmearth_train_amd.helpers import load_state_dict, remap_checkpoint_keys  # noqa: E402

dependencies = ["torch"]

CKPT_URLS = {
    name: f"https://sid.erda.dk/share_redirect/g23YOnaaTp/{name}/checkpoint-199.pth"
    for name in ("pt-all_mod_atto_1M_64_uncertainty_56-8", "pt-all_mod_atto_1M_64_unweighted_56-8",
                 "pt-all_mod_atto_1M_128_uncertainty_112-16", "pt-S2_atto_1M_64_uncertainty_56-8")
}


def load_custom_checkpoint(model, checkpoint, linear_probe):
    """hubconf.py:21-75: drop a mismatching classifier, drop decoder / mask_token / proj / pred entries, remap the
    sparse-encoder keys, non-strict load; fine-tuning re-initialises the head (std 2e-5)."""
    ck = checkpoint["model"] if "model" in checkpoint else checkpoint
    ck = dict(ck)
    own = model.state_dict()
    for k in ("head.weight", "head.bias"):
        if k in ck and ck[k].shape != own[k].shape:
            print(f"Removing key {k} from pretrained checkpoint")
            del ck[k]
    for k in list(ck):
        if "decoder" in k or "mask_token" in k or "proj" in k or "pred" in k:
            del ck[k]
    load_state_dict(model, remap_checkpoint_keys(ck))
    if not linear_probe:
        torch.nn.init.trunc_normal_(model.head.weight, std=2e-5, a=-2.0, b=2.0)
        torch.nn.init.constant_(model.head.bias, 0.0)
    return model


def MPMAE(model_name="convnextv2_atto", ckpt_name="pt-all_mod_atto_1M_64_uncertainty_56-8", pretrained=True,
          linear_probe=True, **kwargs):
    model = convnextv2.__dict__[model_name](**kwargs)
    if pretrained:
        if ckpt_name in CKPT_URLS:
            checkpoint = torch.hub.load_state_dict_from_url(CKPT_URLS[ckpt_name], map_location="cpu")
        else:
            path = ckpt_name[7:] if ckpt_name.startswith("file://") else ckpt_name
            if not os.path.isfile(path):
                raise ValueError(f"unknown checkpoint {ckpt_name!r}: not a released name {sorted(CKPT_URLS)} nor a file")
            checkpoint = torch.load(path, map_location="cpu", weights_only=False)
        model = load_custom_checkpoint(model, checkpoint, linear_probe)
    return model

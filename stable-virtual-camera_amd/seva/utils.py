"""`seva.utils` pass-throughs needed by callers of the hot path (reference seva/utils.py).

`load_model` keeps the reference's weight-format contract (safetensors with the 1146 keys of
`Seva`, bf16) but only loads from a local directory: there is no network in this environment.
"""

from __future__ import annotations

import os

import torch

from .model import Seva, SevaParams


def seed_everything(seed: int = 0):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def load_model(
    pretrained_model_name_or_path: str = "stabilityai/stable-virtual-camera",
    weight_name: str = "model.safetensors",
    device: str | torch.device = "cuda",
    verbose: bool = False,
) -> Seva:
    import safetensors.torch

    if not os.path.isdir(pretrained_model_name_or_path):
        raise FileNotFoundError(
            f"{pretrained_model_name_or_path!r} is not a local directory; this build does not "
            "download checkpoints (reference seva/utils.py:38-43 fetches from the HF hub)."
        )
    weight_path = os.path.join(pretrained_model_name_or_path, weight_name)
    state_dict = safetensors.torch.load_file(weight_path, device=str(device))
    with torch.device("meta"):
        model = Seva(SevaParams()).to(torch.bfloat16)
    missing, unexpected = model.load_state_dict(state_dict, strict=False, assign=True)
    if verbose and (missing or unexpected):
        print(f"missing keys: {missing}\nunexpected keys: {unexpected}")
    return model

"""Device helper with the reference's semantics (riffusion/util/torch_util.py:7-18)."""
import warnings

import torch


def check_device(device: str, backup: str = "cpu") -> str:
    """Return `device` when it is usable, else warn and return `backup`."""
    name = device.lower()
    missing = (name.startswith("cuda") and not torch.cuda.is_available()) or (
        name.startswith("mps") and not torch.backends.mps.is_available()
    )
    if missing:
        warnings.warn(f"WARNING: {device} is not available, using {backup} instead.", stacklevel=3)
        return backup
    return device

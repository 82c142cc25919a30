"""Uncertainty weighting of the per-modality losses (drop-in for the reference's
custom_loss.UncertaintyWeightingStrategy, /root/reference/custom_loss.py:10-30).

In the HIP path the weighting itself runs inside mpmae_loss_finalize; this module only carries
the learnable `log_vars` (so that `model.parameters()` / the state dict contain
`loss_fn.log_vars`, as in the reference) and offers the same callable for host-side use."""
from typing import List

import torch
import torch.nn as nn
from torch import Tensor


class UncertaintyWeightingStrategy(nn.Module):
    def __init__(self, tasks: int):
        super().__init__()
        self.tasks = tasks
        self.log_vars = nn.Parameter(torch.zeros(tasks))

    def forward(self, task_losses: List[Tensor]):
        losses = torch.stack(list(task_losses))
        nz = losses != 0.0
        weighted = (torch.exp(-self.log_vars) * losses + self.log_vars) * nz
        return weighted, self.log_vars.tolist()

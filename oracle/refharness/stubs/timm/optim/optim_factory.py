"""Stub of timm.optim.optim_factory.param_groups_weight_decay (timm 0.9 semantics):
parameters with ndim <= 1, names ending in '.bias', or in no_weight_decay_list get wd 0."""


def param_groups_weight_decay(model, weight_decay=1e-5, no_weight_decay_list=()):
    no_weight_decay_list = set(no_weight_decay_list)
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if param.ndim <= 1 or name.endswith(".bias") or name in no_weight_decay_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [
        {"params": no_decay, "weight_decay": 0.0},
        {"params": decay, "weight_decay": weight_decay},
    ]

"""Dense-backed emulator of the MinkowskiEngine symbols the reference touches.

TEST INFRASTRUCTURE (container-only; see ../../README.md). MinkowskiEngine is an
absent third-party CUDA dependency of the reference (fork shwoo93/MinkowskiEngine,
no pinned SHA, /root/reference/.gitmodules:1-3). Semantics restated here from its
published behaviour; PARITY AT THIS BOUNDARY IS UNPINNED.

Restated semantics (call sites: /root/reference/models/convnextv2_sparse.py:11-22,
37-45,101-129,143-150,199,218; models/sparse_norm_layers.py:14,30-33,74-77):
  * a SparseTensor is a list of active coordinates (n, h, w) at a tensor stride and
    a feature row per coordinate;
  * stride-1 odd-kernel convolution: output coords == input coords,
    y[u] = sum_{k : u+k active} K[k]^T x[u+k] + b (inactive neighbours contribute 0);
  * kernel_size == stride (>1) convolution: output coord = floor-aligned block,
    active iff any child active, sums the children present, + b at active outputs;
  * depthwise = per-channel version of the above;
  * kernel index runs first spatial coordinate fastest:
    dense W[o,i,kh,kw] = K[kw*ks + kh, i, o] (/root/reference/helpers.py:676-687);
  * dense() zero-fills.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Manager:
    def __init__(self, B, H, W):
        self.B, self.H, self.W = B, H, W


class _Key:
    """coords: int64 [M,3] = (n, h, w) in units of `stride` pixels."""

    def __init__(self, coords, stride):
        self.coords = coords
        self.stride = stride


class SparseTensor:
    def __init__(self, features, coordinate_map_key=None, coordinate_manager=None):
        self.F = features
        self.coordinate_map_key = coordinate_map_key
        self.coordinate_manager = coordinate_manager

    # -- helpers -----------------------------------------------------------
    def _dense_map(self):
        cm, key = self.coordinate_manager, self.coordinate_map_key
        Hs, Ws = cm.H // key.stride, cm.W // key.stride
        C = self.F.shape[1]
        dense = self.F.new_zeros(cm.B, Hs, Ws, C)
        c = key.coords
        dense = dense.index_put((c[:, 0], c[:, 1], c[:, 2]), self.F)
        return dense.permute(0, 3, 1, 2)  # [B,C,Hs,Ws]

    def dense(self):
        key = self.coordinate_map_key
        d = self._dense_map()
        c = key.coords
        bmax = int(c[:, 0].max()) + 1
        hmax = int(c[:, 1].max()) + 1
        wmax = int(c[:, 2].max()) + 1
        return d[:bmax, :, :hmax, :wmax], torch.zeros(2, dtype=torch.int32), key.stride

    @property
    def decomposed_coordinates(self):
        c = self.coordinate_map_key.coords
        return [c[c[:, 0] == b, 1:] for b in range(self.coordinate_manager.B)]

    def __add__(self, other):
        assert self.coordinate_map_key is other.coordinate_map_key
        return SparseTensor(
            self.F + other.F,
            coordinate_map_key=self.coordinate_map_key,
            coordinate_manager=self.coordinate_manager,
        )


def _gather(dense_bchw, coords):
    return dense_bchw.permute(0, 2, 3, 1)[coords[:, 0], coords[:, 1], coords[:, 2]]


def _occupancy(x: SparseTensor):
    cm, key = x.coordinate_manager, x.coordinate_map_key
    Hs, Ws = cm.H // key.stride, cm.W // key.stride
    occ = torch.zeros(cm.B, 1, Hs, Ws)
    c = key.coords
    occ[c[:, 0], 0, c[:, 1], c[:, 2]] = 1.0
    return occ


def _down_key(x: SparseTensor, s: int):
    occ = F.max_pool2d(_occupancy(x), s)
    b, h, w = torch.where(occ[:, 0] != 0)
    return _Key(torch.stack([b, h, w], 1), x.coordinate_map_key.stride * s)


class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, bias=False, dimension=None):
        super().__init__()
        assert dimension == 2
        self.ks, self.stride = kernel_size, stride
        kv = kernel_size**dimension
        if kv == 1:
            self.kernel = nn.Parameter(torch.zeros(in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.zeros(kv, in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None
        self.in_channels, self.out_channels = in_channels, out_channels

    def _dense_weight(self):
        ks = self.ks
        k = self.kernel.reshape(ks * ks, self.in_channels, self.out_channels)
        # W[o,i,kh,kw] = K[kw*ks + kh, i, o]
        return k.permute(2, 1, 0).reshape(self.out_channels, self.in_channels, ks, ks).transpose(3, 2)

    def forward(self, x: SparseTensor):
        d = x._dense_map()
        w = self._dense_weight()
        if self.stride == 1:
            assert self.ks % 2 == 1
            y = F.conv2d(d, w, padding=self.ks // 2)
            key = x.coordinate_map_key
        else:
            assert self.ks == self.stride
            y = F.conv2d(d, w, stride=self.stride)
            key = _down_key(x, self.stride)
        f = _gather(y, key.coords)
        if self.bias is not None:
            f = f + self.bias
        return SparseTensor(f, coordinate_map_key=key, coordinate_manager=x.coordinate_manager)


class MinkowskiDepthwiseConvolution(nn.Module):
    def __init__(self, in_channels, kernel_size=-1, stride=1, bias=False, dimension=None):
        super().__init__()
        assert dimension == 2
        self.ks, self.stride = kernel_size, stride
        self.kernel = nn.Parameter(torch.zeros(kernel_size**dimension, in_channels))
        self.bias = nn.Parameter(torch.zeros(1, in_channels)) if bias else None
        self.in_channels = in_channels

    def _dense_weight(self):
        ks = self.ks
        # W[c,0,kh,kw] = K[kw*ks + kh, c]
        return self.kernel.permute(1, 0).reshape(self.in_channels, 1, ks, ks).transpose(3, 2)

    def forward(self, x: SparseTensor):
        d = x._dense_map()
        w = self._dense_weight()
        if self.stride == 1:
            assert self.ks % 2 == 1
            y = F.conv2d(d, w, padding=self.ks // 2, groups=self.in_channels)
            key = x.coordinate_map_key
        else:
            assert self.ks == self.stride
            y = F.conv2d(d, w, stride=self.stride, groups=self.in_channels)
            key = _down_key(x, self.stride)
        f = _gather(y, key.coords)
        if self.bias is not None:
            f = f + self.bias
        return SparseTensor(f, coordinate_map_key=key, coordinate_manager=x.coordinate_manager)


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x: SparseTensor):
        return SparseTensor(
            self.linear(x.F),
            coordinate_map_key=x.coordinate_map_key,
            coordinate_manager=x.coordinate_manager,
        )


class MinkowskiGELU(nn.Module):
    def forward(self, x: SparseTensor):
        return SparseTensor(
            F.gelu(x.F),
            coordinate_map_key=x.coordinate_map_key,
            coordinate_manager=x.coordinate_manager,
        )


def _to_sparse(x: torch.Tensor) -> SparseTensor:
    B, C, H, W = x.shape
    b, h, w = torch.where(x.abs().sum(1) != 0)
    coords = torch.stack([b, h, w], 1)
    feats = x.permute(0, 2, 3, 1)[b, h, w]
    return SparseTensor(feats, coordinate_map_key=_Key(coords, 1), coordinate_manager=_Manager(B, H, W))

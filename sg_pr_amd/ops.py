"""`torch.library` custom ops over the C-ABI (SURVEY §8b "Custom ops"): the hot path as four dispatcher-visible ops.

    torch.ops.sgpr.embed(centers [G,N,3] f32, labels [G,N] i32, weights_blob [48689] f32, k) -> (pooled, att)
    torch.ops.sgpr.score_pairs(pooled1 [B,32], pooled2 [B,32], weights_blob)                -> score [B]
    torch.ops.sgpr.score_all_pairs(pooled_rows [R,32], pooled_cols [M,32], weights_blob)    -> score [R,M]
    torch.ops.sgpr.forward_dense(features_1 [B,15,N], features_2 [B,15,N], weights_blob, k) -> (score, att1, att2)

`weights_blob` is the flat fp32 tensor of `engine.blob_from_state_dict` (order in include/sgpr.h); one engine handle is
kept per (blob tensor, version, device) and released with the tensor.  GPU tensors only: there is no CPU implementation - a CPU call raises.  Fake (meta)
kernels are registered so that the ops trace under torch.compile / FakeTensor.
"""
import collections
import weakref

import torch

from . import engine as _engine

_ENGINES = collections.OrderedDict()      # key -> Engine, least recently used first
_MAX_ENGINES = 8


def _drop(key):
    eng = _ENGINES.pop(key, None)
    if eng is not None:
        eng.close()


def _engine_for(blob, device):
    """The engine handle for this weights tensor.  The entry is tied to the tensor OBJECT (its id is part of the key
    and a weakref finaliser removes the entry - and frees the handle and its device copy - when the tensor dies), so
    a later blob that happens to be allocated at the same address can never meet a stale handle; `_version` catches
    in-place updates; the cache is bounded (least recently used handles are closed)."""
    if device.type != "cuda":
        raise RuntimeError("sgpr ops run on the MI355X only (got a %s tensor); there is no CPU fallback" % device.type)
    key = (id(blob), blob.data_ptr(), blob.numel(), blob._version, device.index)
    eng = _ENGINES.get(key)
    if eng is None:
        for stale in [k for k in _ENGINES if k[0] == key[0] and k[4] == key[4]]:   # same tensor, older version
            _drop(stale)
        eng = _engine.Engine(blob, device=device.index or 0)
        _ENGINES[key] = eng
        weakref.finalize(blob, _drop, key)
        while len(_ENGINES) > _MAX_ENGINES:
            _drop(next(iter(_ENGINES)))
    else:
        _ENGINES.move_to_end(key)
    return eng


@torch.library.custom_op("sgpr::embed", mutates_args=())
def embed(centers: torch.Tensor, labels: torch.Tensor, weights_blob: torch.Tensor, k: int) -> tuple[torch.Tensor, torch.Tensor]:
    eng = _engine_for(weights_blob, centers.device)
    pooled, att, _ = eng.embed(centers, labels, k, want_att=True)
    return pooled, att


@embed.register_fake
def _(centers, labels, weights_blob, k):
    g, n = labels.shape
    return centers.new_empty((g, _engine.F3)), centers.new_empty((g, n))


@torch.library.custom_op("sgpr::score_pairs", mutates_args=())
def score_pairs(pooled1: torch.Tensor, pooled2: torch.Tensor, weights_blob: torch.Tensor) -> torch.Tensor:
    return _engine_for(weights_blob, pooled1.device).score_pairs(pooled1, pooled2)


@score_pairs.register_fake
def _(pooled1, pooled2, weights_blob):
    return pooled1.new_empty((pooled1.shape[0],))


@torch.library.custom_op("sgpr::score_all_pairs", mutates_args=())
def score_all_pairs(pooled_rows: torch.Tensor, pooled_cols: torch.Tensor, weights_blob: torch.Tensor) -> torch.Tensor:
    return _engine_for(weights_blob, pooled_rows.device).score_all_pairs(pooled_rows, pooled_cols)


@score_all_pairs.register_fake
def _(pooled_rows, pooled_cols, weights_blob):
    return pooled_rows.new_empty((pooled_rows.shape[0], pooled_cols.shape[0]))


@torch.library.custom_op("sgpr::forward_dense", mutates_args=())
def forward_dense(features_1: torch.Tensor, features_2: torch.Tensor, weights_blob: torch.Tensor,
                  k: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    eng = _engine_for(weights_blob, features_1.device)
    score, a1, a2 = eng.forward_dense(features_1, features_2, k)
    return score, a1.clone(), a2.clone()     # the engine returns two views of one buffer: ops may not alias


@forward_dense.register_fake
def _(features_1, features_2, weights_blob, k):
    b, _, n = features_1.shape
    return features_1.new_empty((b,)), features_1.new_empty((b, n)), features_1.new_empty((b, n))

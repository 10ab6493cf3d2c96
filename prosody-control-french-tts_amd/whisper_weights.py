"""Whisper audio-encoder weights: names, shapes, the flat float32 blob ``pce_whisper_load``
expects, and a fixed-seed synthetic initialisation (no checkpoint is available offline; a real
``model.encoder.state_dict()`` converted to numpy packs the same way)."""
from __future__ import annotations

import numpy as np

DIMS = {
    "tiny": dict(n_mels=80, n_ctx=1500, n_state=384, n_head=6, n_layer=4),
    "base": dict(n_mels=80, n_ctx=1500, n_state=512, n_head=8, n_layer=6),
    "small": dict(n_mels=80, n_ctx=1500, n_state=768, n_head=12, n_layer=12),
    "medium": dict(n_mels=80, n_ctx=1500, n_state=1024, n_head=16, n_layer=24),
}


def tensor_order(dims):
    d, m = dims["n_state"], dims["n_mels"]
    order = [("conv1.weight", (d, m, 3)), ("conv1.bias", (d,)), ("conv2.weight", (d, d, 3)), ("conv2.bias", (d,))]
    for l in range(dims["n_layer"]):
        p = f"blocks.{l}."
        order += [(p + "attn_ln.weight", (d,)), (p + "attn_ln.bias", (d,)),
                  (p + "attn.query.weight", (d, d)), (p + "attn.query.bias", (d,)), (p + "attn.key.weight", (d, d)),
                  (p + "attn.value.weight", (d, d)), (p + "attn.value.bias", (d,)),
                  (p + "attn.out.weight", (d, d)), (p + "attn.out.bias", (d,)),
                  (p + "mlp_ln.weight", (d,)), (p + "mlp_ln.bias", (d,)),
                  (p + "mlp.0.weight", (4 * d, d)), (p + "mlp.0.bias", (4 * d,)),
                  (p + "mlp.2.weight", (d, 4 * d)), (p + "mlp.2.bias", (d,))]
    order += [("ln_post.weight", (d,)), ("ln_post.bias", (d,))]
    return order


def synthetic_weights(dims, seed=20240930):
    """PyTorch-default-like initialisation (uniform +-1/sqrt(fan_in)), LayerNorm near identity."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, shape in tensor_order(dims):
        if name.endswith("ln.weight") or name == "ln_post.weight":
            W[name] = (1.0 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif "ln" in name and name.endswith("bias"):
            W[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            b = 1.0 / np.sqrt(fan_in)
            W[name] = rng.uniform(-b, b, size=shape).astype(np.float32)
    return W


def pack(W, dims) -> np.ndarray:
    return np.concatenate([np.asarray(W[name], dtype=np.float32).reshape(-1) for name, _ in tensor_order(dims)])

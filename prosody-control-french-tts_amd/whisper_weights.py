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


# ------------------------------------------------------------------ text decoder
TEXT_DIMS = {
    "tiny": dict(n_vocab=51865, n_text_ctx=448, n_state=384, n_head=6, n_layer=4),
    "base": dict(n_vocab=51865, n_text_ctx=448, n_state=512, n_head=8, n_layer=6),
    "small": dict(n_vocab=51865, n_text_ctx=448, n_state=768, n_head=12, n_layer=12),
    "medium": dict(n_vocab=51865, n_text_ctx=448, n_state=1024, n_head=16, n_layer=24),
}


def decoder_tensor_order(dims):
    d = dims["n_state"]
    order = [("token_embedding.weight", (dims["n_vocab"], d)), ("positional_embedding", (dims["n_text_ctx"], d))]
    for l in range(dims["n_layer"]):
        p = f"blocks.{l}."
        for att in ("attn", "cross_attn"):
            order += [(p + att + "_ln.weight", (d,)), (p + att + "_ln.bias", (d,)),
                      (p + att + ".query.weight", (d, d)), (p + att + ".query.bias", (d,)), (p + att + ".key.weight", (d, d)),
                      (p + att + ".value.weight", (d, d)), (p + att + ".value.bias", (d,)),
                      (p + att + ".out.weight", (d, d)), (p + att + ".out.bias", (d,))]
        order += [(p + "mlp_ln.weight", (d,)), (p + "mlp_ln.bias", (d,)), (p + "mlp.0.weight", (4 * d, d)), (p + "mlp.0.bias", (4 * d,)),
                  (p + "mlp.2.weight", (d, 4 * d)), (p + "mlp.2.bias", (d,))]
    order += [("ln.weight", (d,)), ("ln.bias", (d,))]
    return order


def synthetic_decoder_weights(dims, seed=448):
    rng = np.random.default_rng(seed)
    W = {}
    for name, shape in decoder_tensor_order(dims):
        if name.endswith("ln.weight"):
            W[name] = (1.0 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("ln.bias"):
            W[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        elif name in ("token_embedding.weight", "positional_embedding"):   # (checked before the generic Linear branch)
            W[name] = (0.5 * rng.standard_normal(shape)).astype(np.float32)
        else:
            b = 1.0 / np.sqrt(shape[1] if len(shape) > 1 else shape[0])
            W[name] = rng.uniform(-b, b, size=shape).astype(np.float32)
    # make cross-attention peaky enough to resemble a trained aligner: larger query/key gains
    for name in W:
        if "cross_attn.query.weight" in name or "cross_attn.key.weight" in name:
            W[name] *= 3.0
    return W


def pack_decoder(W, dims) -> np.ndarray:
    return np.concatenate([np.asarray(W[name], dtype=np.float32).reshape(-1) for name, _ in decoder_tensor_order(dims)])


def greedy_test_decoder_weights(dims, seed=79):
    """Synthetic decoder weights for free-running decoding tests: with the usual initialisation the tied output
    projection makes the decoder repeat its input token for ever; small token embeddings and stronger cross-attention
    values let position and audio decide, so that the decoding rules (timestamps, suppression, end of text) get exercised."""
    W = synthetic_decoder_weights(dims, seed)
    rng = np.random.default_rng(seed + 1000)
    W["token_embedding.weight"] = (0.08 * rng.standard_normal(W["token_embedding.weight"].shape)).astype(np.float32)
    W["positional_embedding"] = (0.6 * rng.standard_normal(W["positional_embedding"].shape)).astype(np.float32)
    for name in W:
        if "cross_attn.value.weight" in name or "cross_attn.out.weight" in name:
            W[name] = (W[name] * 2.5).astype(np.float32)
    return W


"""Weights of the break-prediction token classifier (``BertForTokenClassification``,
Code/baseline_models/pause_bert.py:14-21,127-132: bert-base-multilingual-uncased, two labels): tensor names and
shapes in the order of the flat float32 blob ``pce_bert_load`` expects, a packer for a Hugging Face ``state_dict``
and a fixed-seed synthetic initialisation (no checkpoint is reachable offline)."""
from __future__ import annotations

import numpy as np

DIMS = {
    # bert-base-multilingual-uncased, num_labels = 2 (NO_BREAK / BREAK)
    "mbert-base-uncased": dict(n_vocab=105879, n_pos=512, n_type=2, n_state=768, n_head=12, n_layer=12, n_labels=2),
    "tiny": dict(n_vocab=300, n_pos=64, n_type=2, n_state=128, n_head=2, n_layer=2, n_labels=2),
}
MAX_LENGTH = 128        # pause_bert.py:16


def tensor_order(dims):
    d = dims["n_state"]
    order = [("bert.embeddings.word_embeddings.weight", (dims["n_vocab"], d)),
             ("bert.embeddings.position_embeddings.weight", (dims["n_pos"], d)),
             ("bert.embeddings.token_type_embeddings.weight", (dims["n_type"], d)),
             ("bert.embeddings.LayerNorm.weight", (d,)), ("bert.embeddings.LayerNorm.bias", (d,))]
    for l in range(dims["n_layer"]):
        p = f"bert.encoder.layer.{l}."
        order += [(p + "attention.self.query.weight", (d, d)), (p + "attention.self.query.bias", (d,)),
                  (p + "attention.self.key.weight", (d, d)), (p + "attention.self.key.bias", (d,)),
                  (p + "attention.self.value.weight", (d, d)), (p + "attention.self.value.bias", (d,)),
                  (p + "attention.output.dense.weight", (d, d)), (p + "attention.output.dense.bias", (d,)),
                  (p + "attention.output.LayerNorm.weight", (d,)), (p + "attention.output.LayerNorm.bias", (d,)),
                  (p + "intermediate.dense.weight", (4 * d, d)), (p + "intermediate.dense.bias", (4 * d,)),
                  (p + "output.dense.weight", (d, 4 * d)), (p + "output.dense.bias", (d,)),
                  (p + "output.LayerNorm.weight", (d,)), (p + "output.LayerNorm.bias", (d,))]
    order += [("classifier.weight", (dims["n_labels"], d)), ("classifier.bias", (dims["n_labels"],))]
    return order


def dims_of_config(cfg) -> dict:
    """``transformers.BertConfig`` -> dims (intermediate size must be 4 * hidden, GELU, absolute positions)."""
    assert cfg.intermediate_size == 4 * cfg.hidden_size and cfg.hidden_act == "gelu"
    return dict(n_vocab=cfg.vocab_size, n_pos=cfg.max_position_embeddings, n_type=cfg.type_vocab_size, n_state=cfg.hidden_size,
                n_head=cfg.num_attention_heads, n_layer=cfg.num_hidden_layers, n_labels=cfg.num_labels)


def synthetic_weights(dims, seed=1):
    """BERT-style initialisation (normal, std 0.02 scaled up so that the logits are not degenerate; LayerNorm near identity)."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, shape in tensor_order(dims):
        if name.endswith("LayerNorm.weight"):
            W[name] = (1.0 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("LayerNorm.bias"):
            W[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        elif "embeddings" in name:
            W[name] = (0.5 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith(".bias"):
            W[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:
            W[name] = (rng.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
    return W


def pack(W, dims) -> np.ndarray:
    """dict of arrays (numpy, or torch tensors of a ``state_dict``) -> the flat float32 blob."""
    out = []
    for name, shape in tensor_order(dims):
        a = W[name]
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        out.append(np.asarray(a, dtype=np.float32).reshape(-1))
    return np.concatenate(out)

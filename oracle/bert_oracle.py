"""oracle/bert_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp32 restatement (torch, CPU) of ``transformers.BertForTokenClassification.forward`` in eval mode, the model
Code/baseline_models/pause_bert.py:127-132 trains for break prediction (the reference has no inference path of its own).
Pinned: tests/golden/make_goldens_bert.py runs the installed ``transformers`` implementation itself on a random-init
two-layer model and stores ids, weights seed and logits (tests/golden/bert_tiny.npz); tests/test_bert.py holds this
restatement to those logits.  Only tests/ and __graft_entry__.smoke() may import this module."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def forward(token_lists, W, dims):
    """-> list of float32 [len][n_labels] logits, one per sequence (token_type 0, no padding needed: per sequence)."""
    d, H, L = dims["n_state"], dims["n_head"], dims["n_layer"]
    t = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in W.items()}
    out = []
    with torch.no_grad():
        for ids in token_lists:
            ids = torch.as_tensor(np.asarray(ids, dtype=np.int64))
            n = len(ids)
            x = t["bert.embeddings.word_embeddings.weight"][ids] + t["bert.embeddings.token_type_embeddings.weight"][0] \
                + t["bert.embeddings.position_embeddings.weight"][:n]
            x = F.layer_norm(x, (d,), t["bert.embeddings.LayerNorm.weight"], t["bert.embeddings.LayerNorm.bias"], 1e-12)
            for l in range(L):
                p = f"bert.encoder.layer.{l}."
                q = F.linear(x, t[p + "attention.self.query.weight"], t[p + "attention.self.query.bias"]).view(n, H, d // H).transpose(0, 1)
                k = F.linear(x, t[p + "attention.self.key.weight"], t[p + "attention.self.key.bias"]).view(n, H, d // H).transpose(0, 1)
                v = F.linear(x, t[p + "attention.self.value.weight"], t[p + "attention.self.value.bias"]).view(n, H, d // H).transpose(0, 1)
                a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(d // H), dim=-1) @ v
                a = a.transpose(0, 1).reshape(n, d)
                x = F.layer_norm(x + F.linear(a, t[p + "attention.output.dense.weight"], t[p + "attention.output.dense.bias"]), (d,),
                                 t[p + "attention.output.LayerNorm.weight"], t[p + "attention.output.LayerNorm.bias"], 1e-12)
                h = F.gelu(F.linear(x, t[p + "intermediate.dense.weight"], t[p + "intermediate.dense.bias"]))
                x = F.layer_norm(x + F.linear(h, t[p + "output.dense.weight"], t[p + "output.dense.bias"]), (d,),
                                 t[p + "output.LayerNorm.weight"], t[p + "output.LayerNorm.bias"], 1e-12)
            out.append(F.linear(x, t["classifier.weight"], t["classifier.bias"]).numpy())
    return out

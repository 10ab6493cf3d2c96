"""Generates tests/golden/bert_tiny.npz with the INSTALLED transformers implementation of the model the reference
trains (Code/baseline_models/pause_bert.py:127-132: BertForTokenClassification, num_labels=2), random-init, eval mode,
right-padded batch with an attention mask -- run in the build container:  python tests/golden/make_goldens_bert.py"""
import os
import sys

import numpy as np
import torch
from transformers import BertConfig, BertForTokenClassification

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from prosody_control_french_tts_amd import bert_weights as BW  # noqa: E402

cfg = BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                 max_position_embeddings=64, type_vocab_size=2, num_labels=2)
model = BertForTokenClassification(cfg).eval()
dims = BW.dims_of_config(cfg)
# weights: the package's fixed-seed numpy initialisation, loaded INTO the transformers model (the fixture then only has
# to hold ids and logits; transformers' own default init, std 0.02 and zero biases, gives near-constant logits)
W = BW.synthetic_weights(dims, seed=20250301)
missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
assert not missing.unexpected_keys and all("position_ids" in k for k in missing.missing_keys), missing
rng = np.random.default_rng(7)
lens = [24, 7, 1, 40, 64]
S = max(lens)
ids = np.zeros((len(lens), S), dtype=np.int64); mask = np.zeros((len(lens), S), dtype=np.int64)
for i, n in enumerate(lens):
    ids[i, :n] = rng.integers(0, cfg.vocab_size, size=n); mask[i, :n] = 1
with torch.no_grad():
    logits = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits.numpy()
np.savez_compressed(os.path.join(HERE, "bert_tiny.npz"), seed=np.array([20250301]), ids=ids.astype(np.int32), lens=np.array(lens, dtype=np.int32),
                    logits=logits.astype(np.float32), dims=np.array([dims[k] for k in ("n_vocab", "n_pos", "n_type", "n_state", "n_head", "n_layer", "n_labels")], dtype=np.int32))
print("wrote bert_tiny.npz", logits.shape, float(np.abs(logits).max()))

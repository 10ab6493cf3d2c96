"""Break-prediction token classifier (Code/baseline_models/pause_bert.py:127-132): the torch fp32
restatement against logits produced by the installed transformers implementation itself (tests/golden/bert_tiny.npz,
made by tests/golden/make_goldens_bert.py), and the weight packer."""
import os

import numpy as np

from oracle import bert_oracle as BO
from prosody_control_french_tts_amd import bert_weights as BW

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bert_tiny.npz")
KEYS = ("n_vocab", "n_pos", "n_type", "n_state", "n_head", "n_layer", "n_labels")


def load_gold():
    g = np.load(GOLD)
    dims = dict(zip(KEYS, (int(v) for v in g["dims"])))
    W = BW.synthetic_weights(dims, seed=int(g["seed"][0]))
    toks = [g["ids"][i, :n].tolist() for i, n in enumerate(g["lens"])]
    return dims, W, toks, g["logits"], g["lens"]


def test_restatement_matches_transformers_logits():
    dims, W, toks, logits, lens = load_gold()
    got = BO.forward(toks, W, dims)
    for i, n in enumerate(lens):
        assert got[i].shape == (n, dims["n_labels"])
        assert np.max(np.abs(got[i] - logits[i, :n])) <= 2e-5, i          # fp32 both sides; padded batch + mask vs per sequence


def test_packer_layout():
    dims = BW.DIMS["tiny"]
    W = BW.synthetic_weights(dims, seed=3)
    blob = BW.pack(W, dims)
    d, L = dims["n_state"], dims["n_layer"]
    per_layer = 4 * (d * d + d) + 2 * d + (4 * d * d + 4 * d) + (4 * d * d + d) + 2 * d
    assert blob.dtype == np.float32 and blob.size == (dims["n_vocab"] + dims["n_pos"] + dims["n_type"]) * d + 2 * d + L * per_layer + dims["n_labels"] * (d + 1)
    assert np.array_equal(blob[:d], W["bert.embeddings.word_embeddings.weight"][0])
    assert np.array_equal(blob[-dims["n_labels"]:], W["classifier.bias"])
    full = BW.DIMS["mbert-base-uncased"]
    assert full["n_state"] // full["n_head"] == 64 and BW.MAX_LENGTH == 128


def test_word_piece_bookkeeping():
    from prosody_control_french_tts_amd.Preprocessing import break_bert as BB
    words = [[5], [6, 7, 8], [9], [10, 11]]
    ids, wids = BB.encode_words(words, 101, 102, max_length=128)
    assert ids == [101, 5, 6, 7, 8, 9, 10, 11, 102] and wids == [None, 0, 1, 1, 1, 2, 3, 3, None]
    assert BB.first_subtoken_positions(wids, 4) == [1, 2, 5, 6]
    ids, wids = BB.encode_words(words, 101, 102, max_length=6)          # truncation keeps max_length - 2 pieces, may cut a word
    assert ids == [101, 5, 6, 7, 8, 102] and BB.first_subtoken_positions(wids, 4) == [1, 2, None, None]
    assert BB.encode_words([], 1, 2) == ([1, 2], [None, None])

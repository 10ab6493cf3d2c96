"""Device time of the break classifier forward at the reference's shape: bert-base width / depth, 256 x 128 tokens."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import bert_weights as BW

dims = dict(BW.DIMS["mbert-base-uncased"], n_vocab=8000)
eng = pkg.ProsodyEngine(0)
eng.bert_load(dims, BW.pack(BW.synthetic_weights(dims, seed=1), dims))
rng = np.random.default_rng(0)
for n in (64, 256, 1024):
    toks = [rng.integers(0, dims["n_vocab"], size=128).tolist() for _ in range(n)]
    for _ in range(3):
        eng.bert_run(toks)
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(10):
        eng.bert_run(toks)
    eng.sync()
    p = eng.profile()["bert_forward"]
    ms = p["total_ms"] / p["launches"]
    d, L, S = dims["n_state"], dims["n_layer"], 128
    flops = n * S * L * (24.0 * d * d) + n * L * 4.0 * S * S * d
    print(f"{n} sequences x 128 tokens: {ms:.2f} ms per forward, {flops / ms / 1e9:.0f} TFLOP/s, {n * S / ms * 1e3:.3g} tokens/s")
    eng.profile_enable(False)
eng.close()

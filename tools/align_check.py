import sys; sys.path.insert(0, '.')
import numpy as np
import prosody_control_french_tts_amd as P
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from oracle import whisper_oracle as WO
edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=4)
We, Wd = WW.synthetic_weights(edims), WW.synthetic_decoder_weights(tdims)
clips = [synth.synth_clip(0, seconds=10.0), synth.synth_clip(1, seconds=3.3)]
eng = P.ProsodyEngine(0); eng.upload(clips, 16000); eng.logmel_run(80)
eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_encode_run(); eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
rng = np.random.default_rng(21)
toks = [rng.integers(0, 300, size=n).tolist() for n in (37, 70)]
nf = [len(c) // 160 for c in clips]
res = eng.whisper_align(toks, nf, 3, want_cost=True)
for i in range(2):
    enc = eng.whisper_encode_fetch(i)
    cost, ti, tj = WO.find_alignment(toks[i], enc, Wd, tdims, nf[i], 3)
    g = res[i]
    d = np.abs(g["cost"] - cost)
    jg = g["time_indices"][np.r_[True, np.diff(g["text_indices"]) > 0]]; jw = tj[np.r_[True, np.diff(ti) > 0]]
    print(i, cost.shape, "std", cost.std(), "max err", d.max(), "mean err", d.mean(), "rel L2", np.linalg.norm(g["cost"] - cost) / np.linalg.norm(cost),
          "jump diffs", np.abs(jg - jw).max(), (jg == jw).mean())

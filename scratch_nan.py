import sys; sys.path.insert(0, '.')
import numpy as np
import prosody_control_french_tts_amd as P
from prosody_control_french_tts_amd import synth, whisper_weights as WW
dims = WW.DIMS["small"]; W = WW.synthetic_weights(dims)
for nclips in (64,):
    clips = [synth.synth_clip(i) for i in range(nclips)]
    eng = P.ProsodyEngine(0); eng.upload(clips, 16000)
    eng.whisper_load(dims, WW.pack(W, dims))
    for it in range(2):
        eng.logmel_run(dims["n_mels"]); eng.whisper_encode_run(); eng.sync()
        bad = []
        for c in range(nclips):
            g = eng.whisper_encode_fetch(c)
            n = int(np.isnan(g).sum())
            if n: bad.append((c, n, np.where(np.isnan(g).any(axis=1))[0][:5].tolist()))
        print("nclips", nclips, "iter", it, "bad", bad[:6], len(bad))
    eng.close()

"""Debug aid: is a clip's encoder output the same in a batch of N (twice) and in a batch of 4?  argv: N layers"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dims = dict(WW.DIMS["small"], n_layer=layers)
W = WW.pack(WW.synthetic_weights(dims), dims)
clips = synth.synth_batch(n, 10.0, 16000, first=0)
pick = [0, 97 % n, 200 % n, n - 1]
with pkg.ProsodyEngine(0) as eng:
    eng.whisper_load(dims, W)
    def run(idx, fetch):
        eng.upload([clips[i] for i in idx], 16000); eng.logmel_run(80); eng.whisper_encode_run()
        return [eng.whisper_encode_fetch(k) for k in fetch]
    a = run(list(range(n)), pick); b = run(list(range(n)), pick); b2 = run(list(range(n)), pick); s = run(pick, range(4))
    for rep in range(int(os.environ.get('REPS', '0'))):
        x = run(list(range(n)), pick); print('rep', rep, [int((u != v).sum()) for u, v in zip(x, s)])
    print('third run vs second:', [int((x != y).sum()) for x, y in zip(b, b2)], 'third vs first', [int((x != y).sum()) for x, y in zip(a, b2)])
    for k, i in enumerate(pick):
        d1 = np.argwhere(a[k] != b[k]); d2 = np.argwhere(a[k] != s[k])
        if len(d1): print("   rerun: rows", np.unique(d1[:, 0]), "max", float(np.abs(a[k] - b[k]).max()), "cols of the first row", d1[d1[:, 0] == d1[0, 0]][:, 1][:20])
        print("clip", i, "rerun diffs", len(d1), "vs batch of 4:", len(d2), "rows", np.unique(d2[:, 0])[:8] if len(d2) else "", "n rows", len(np.unique(d2[:, 0])) if len(d2) else 0,
              "max", float(np.abs(a[k] - s[k]).max()))

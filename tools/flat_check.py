"""Debug aid: encoder output of the persistent 256 x 256 GEMM path against the tiled kernels (PCE_GEMM_FLAT=0), same weights."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dims = dict(WW.DIMS["small"], n_layer=layers)
W = WW.pack(WW.synthetic_weights(dims), dims)
clips = [synth.synth_clip(i, seconds=10.0 if i % 2 == 0 else 3.3) for i in range(n_clips)]
outs = {}
for flat in (os.environ.get("ORDER", "0,1").split(",")):
    os.environ["PCE_GEMM_FLAT"] = flat
    with pkg.ProsodyEngine(0) as eng:
        eng.upload(clips, 16000); eng.logmel_run(80); eng.whisper_load(dims, W); eng.whisper_encode_run()
        outs[flat + str(len(outs))] = [eng.whisper_encode_fetch(i) for i in range(n_clips)]
for i in range(n_clips):
    keys = list(outs); a, b = outs[keys[0]][i], outs[keys[-1]][i]
    bad = ~np.isfinite(b)
    print("clip", i, "nan rows", np.unique(np.where(bad)[0])[:10], "count", int(bad.sum()), "rel", float(np.linalg.norm(np.where(bad, 0, b) - a) / np.linalg.norm(a)))

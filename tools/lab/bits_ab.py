"""Bits of the front-end stages with one library build (PCE_LIBRARY=...): sha1 of the log-mel, STFT-dB, F0, LUFS and encoder output of fixed clips.
Two builds that print the same lines compute the same bits.  usage: PCE_LIBRARY=path python3 tools/lab/bits_ab.py"""
import hashlib, os, sys
ROOT = os.environ.get("BITS_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
clips = [synth.synth_clip(i, seconds=3.0 + i) for i in range(4)] + [np.zeros(16000, np.int16)]
eng = pkg.ProsodyEngine(0)
eng.upload(clips, 16000)
eng.logmel_run(80)
print("logmel", [h(eng.logmel_fetch(i)) for i in range(len(clips))])
eng.stft_db_run(1024, 256)
print("stft  ", [h(eng.stft_db_fetch(i)) for i in range(len(clips))])
sl = eng.whole_clip_slices()
pi = eng.pitch(sl, pkg.PitchParams.praat(150.0, 600.0))
print("f0    ", h(pi["f0"]), h(pi["summary"]))
print("lufs  ", h(eng.lufs(sl)[0]))
edims = dict(n_mels=80, n_ctx=1500, n_state=256, n_head=4, n_layer=2)
eng.whisper_load(edims, WW.pack(WW.synthetic_weights(edims, seed=3), edims)); eng.whisper_encode_run()
print("enc   ", [h(eng.whisper_encode_fetch(i)) for i in range(2)])
eng.close()

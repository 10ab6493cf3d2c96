"""Dump a checksum and a sample of the STFT-dB output (to compare two builds bit for bit)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth

rng = np.random.default_rng(3)
clips = [synth.synth_clip(i, seconds=10.0) for i in range(6)] + [rng.integers(-32768, 32768, size=12345).astype(np.int16),
         np.zeros(3000, np.int16), np.full(777, -32768, np.int16)]
eng = pkg.ProsodyEngine(0)
eng.upload(clips, 16000)
eng.stft_db_run(1024, 256)
h = hashlib.sha256()
for i in range(len(clips)):
    h.update(eng.stft_db_fetch(i).tobytes())
print("stft sha256", h.hexdigest())
eng.close()

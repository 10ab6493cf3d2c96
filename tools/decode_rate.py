"""Device time of one free-running decoding step at Whisper-small size (256 x 10 s clips, prefixes of 8 / 40 tokens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from prosody_control_french_tts_amd.Aligners import decoding as DEC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edims, tdims = WW.DIMS["small"], WW.TEXT_DIMS["small"]
eng = pkg.ProsodyEngine(0)
eng.upload(synth.synth_batch(n, 10.0, 16000, first=0), 16000)
eng.logmel_run(80)
eng.whisper_load(edims, WW.pack(WW.synthetic_weights(edims), edims))
eng.whisper_decoder_load(tdims, WW.pack_decoder(WW.greedy_test_decoder_weights(tdims), tdims))
eng.whisper_encode_run()
V = tdims["n_vocab"]
rules = dict(eot=50257, no_timestamps=50363, timestamp_begin=50364, suppress_tokens=list(range(50258, 50363)), blank_tokens=[220, 50257], max_initial_timestamp_index=50)
mask = DEC.vocab_mask(V, rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
rng = np.random.default_rng(0)
for L in (8, 40):
    seqs = [[50258, 50265, 50359] + rng.integers(0, 50000, size=L - 3).tolist() for _ in range(n)]
    eng.whisper_decode_step(seqs, 3, rules["eot"], rules["timestamp_begin"], mask, 50)         # first call computes the cross K / V
    t0 = time.perf_counter()
    for _ in range(5):
        eng.whisper_decode_step(seqs, 3, rules["eot"], rules["timestamp_begin"], mask, 50)
    dt = (time.perf_counter() - t0) / 5
    print(f"{n} clips, prefix {L}: {dt * 1e3:.1f} ms per step (host wall, upload + sync included)")
t0 = time.perf_counter()
out = DEC.greedy_decode(eng, V, [50258, 50265, 50359], rules, sample_len=24)
print(f"24 free steps: {(time.perf_counter() - t0) * 1e3:.0f} ms; lengths {sorted(set(len(o) for o in out))[:5]}")
eng.close()

"""Race screen at the bench's batch size: the Whisper-small encoder over 256 clips and a 32-step device-resident decoding loop, repeated;
every repetition must reproduce the first bit for bit (encoder states of every clip, chosen tokens and their log-probabilities).  A kernel
that reads a staged buffer before its LDS-DMA has landed, or re-stages one too early, shows up here as a repetition that differs.
usage: stress_repeat.py [repetitions = 12] [operands = fp16]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from prosody_control_french_tts_amd.Aligners import decoding as DEC

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ops = sys.argv[2] if len(sys.argv) > 2 else "fp16"
n, steps = 256, 32
edims, tdims = WW.DIMS["small"], WW.TEXT_DIMS["small"]
eng = pkg.ProsodyEngine(0)
eng.whisper_set_operands(ops)
eng.upload(synth.synth_batch(n, 10.0, 16000, first=0), 16000)
eng.logmel_run(80)
eng.whisper_load(edims, WW.pack(WW.synthetic_weights(edims), edims))
eng.whisper_decoder_load(tdims, WW.pack_decoder(WW.greedy_test_decoder_weights(tdims), tdims))
mask = DEC.vocab_mask(tdims["n_vocab"], list(range(50258, 50363)) + [50257], [220, 50257], 50363)
init = [[50258, 50265, 50359] for _ in range(n)]
first = None
bad = 0
for r in range(reps):
    eng.logmel_run(80)
    eng.whisper_encode_run()
    h = hashlib.sha256()
    for i in range(0, n, 17):                                   # every 17th clip's full encoder state
        h.update(eng.whisper_encode_fetch(i).tobytes())
    toks, lps, _ = eng.whisper_decode_loop(init, 3, 50257, 50364, mask, steps, 50)
    h.update(np.ascontiguousarray(toks).tobytes()); h.update(np.ascontiguousarray(lps).tobytes())
    d = h.hexdigest()
    if first is None:
        first = d
    elif d != first:
        bad += 1
    print(f"repetition {r}: {d[:16]} {'' if d == first else 'DIFFERS'}", flush=True)
print(f"{reps} repetitions, operands {ops}: {bad} differ from the first")
eng.close()
sys.exit(1 if bad else 0)

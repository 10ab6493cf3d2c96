"""Device time of free-running decoding at Whisper-small size (256 x 10 s clips): the host-driven step (one round trip per step) and
the device-resident loop (pce_whisper_decode_loop), with the per-kernel split of a step.  usage: decode_rate.py [clips] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from prosody_control_french_tts_amd.Aligners import decoding as DEC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
    print(f"{n} clips, prefix {L}: {dt * 1e3:.1f} ms per PREFIX step (host wall, upload + sync included)")
init = [[50258, 50265, 50359] for _ in range(n)]
# suppress end-of-text so that every step is a full step (no early stop): the rate of a live step
m2 = mask.copy(); m2[rules["eot"]] |= 1
for name, fn in (("host-driven", lambda: DEC.decode_batch(eng, V, init, [3] * n, dict(rules, suppress_tokens=rules["suppress_tokens"] + [rules["eot"]]), steps, device_loop=False)),
                 ("device loop", lambda: eng.whisper_decode_loop(init, 3, rules["eot"], rules["timestamp_begin"], m2, steps, 50))):
    fn()
    eng.sync()
    t0 = time.perf_counter(); fn(); eng.sync(); dt = time.perf_counter() - t0
    print(f"{name}: {steps} steps in {dt * 1e3:.1f} ms = {dt / steps * 1e3:.2f} ms per step (wall, first step = prefix run included)")
eng.profile_enable(True); eng.profile_reset()
eng.whisper_decode_loop(init, 3, rules["eot"], rules["timestamp_begin"], m2, steps, 50)
eng.sync()
prof = eng.profile()
for k, p in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"  {k:22s} {p['total_ms']:9.2f} ms  {p['launches']:6d} launches  {p['total_ms'] / max(p['launches'], 1) * 1e3:9.1f} us each")
xkv = 12 * n * (1500 * 768 * 2 + 768 * 1536 * 2) / 1e9
loop = prof.get("whisper_decode_loop", {}).get("total_ms")
if loop:
    print(f"cross K / V per step: {xkv:.2f} GB -> {xkv / 8.0:.2f} ms at 8 TB/s; loop {loop / steps:.2f} ms per step = {xkv / (loop / steps):.2f} TB/s counting only those bytes")
eng.close()

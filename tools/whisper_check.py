import sys, time; sys.path.insert(0, '.')
import numpy as np
import prosody_control_french_tts_amd as P
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from oracle import whisper_oracle as WO
name = sys.argv[1] if len(sys.argv) > 1 else "small"
nclips = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dims = WW.DIMS[name]; W = WW.synthetic_weights(dims)
clips = [synth.synth_clip(i) for i in range(nclips)]
eng = P.ProsodyEngine(0); eng.upload(clips, 16000)
eng.whisper_load(dims, WW.pack(W, dims))
eng.profile_enable(True)
for it in range(3):
    t = time.time(); eng.logmel_run(dims["n_mels"]); eng.whisper_encode_run(); eng.sync(); dt = time.time() - t
    print("iter", it, "ms", dt * 1e3)
print(eng.profile())
got = eng.whisper_encode_fetch(0)
t = time.time(); want = WO.encoder_forward(WO.log_mel(clips[0], dims["n_mels"]), W, dims); print("cpu s", time.time() - t)
print("rel L2", np.linalg.norm(got - want) / np.linalg.norm(want), "max abs", np.abs(got - want).max(), "std", want.std())
d, L = dims["n_state"], dims["n_layer"]
flop = nclips * (2 * 3000 * d * 240 + 2 * 1500 * d * 3 * d + L * (2 * 1500 * d * 3 * d + 4 * 1500 * 1500 * d + 2 * 1500 * d * d + 16 * 1500 * d * d))
print("GFLOP", flop / 1e9, "TFLOP/s", flop / dt / 1e12)

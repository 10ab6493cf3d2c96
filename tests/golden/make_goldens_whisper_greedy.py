"""Generates tests/golden/whisper_greedy_tiny.npz: greedy decoding (temperature 0) of the two-layer random-init Whisper of
make_goldens_whisper_hf.py, driven by the INSTALLED transformers model and its own ports of openai-whisper's logit
filters (SuppressTokensLogitsProcessor, SuppressTokensAtBeginLogitsProcessor, WhisperTimeStampLogitsProcessor).
Run in the build container:  python tests/golden/make_goldens_whisper_greedy.py"""
import os
import sys
import types

import numpy as np
import torch
from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                    WhisperTimeStampLogitsProcessor)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_goldens_whisper_hf as M  # noqa: E402
from prosody_control_french_tts_amd import synth, whisper_weights as WW  # noqa: E402
from oracle import whisper_oracle as WO  # noqa: E402

# a miniature of Whisper's vocabulary layout: text ids, then eot, sot / language / task ids, no_timestamps, timestamps
RULES = dict(eot=250, no_timestamps=262, timestamp_begin=263, suppress_tokens=[1, 2, 7, 8, 9] + list(range(251, 262)),
             blank_tokens=[220, 250], max_initial_timestamp_index=20)
INITIAL = [251, 255, 260]                  # <|startoftranscript|><|fr|><|transcribe|>
# a decoder whose next token depends on position and audio rather than on repeating its input: small token embeddings
# under the tied output projection (WW.greedy_test_weights)
We, Wd = WW.synthetic_weights(M.edims, seed=77), WW.greedy_test_decoder_weights(M.tdims, seed=79)
model = M.build_model(We, Wd)
gen_cfg = types.SimpleNamespace(no_timestamps_token_id=RULES["no_timestamps"], eos_token_id=RULES["eot"], bos_token_id=RULES["eot"],
                                max_initial_timestamp_index=RULES["max_initial_timestamp_index"])
procs = [SuppressTokensLogitsProcessor(RULES["suppress_tokens"]), SuppressTokensAtBeginLogitsProcessor(RULES["blank_tokens"], len(INITIAL)),
         WhisperTimeStampLogitsProcessor(gen_cfg, begin_index=len(INITIAL))]
out = {}
for ci in (5, 6, 7):
    clip = synth.synth_clip(ci, seconds=4.0)
    mel = torch.from_numpy(WO.log_mel(clip, 80))[None]
    ids = torch.tensor([INITIAL])
    with torch.no_grad():
        enc = model.model.encoder(mel).last_hidden_state
        for _ in range(48):
            hid = model.model.decoder(input_ids=ids, encoder_hidden_states=enc).last_hidden_state
            scores = model.proj_out(hid[:, -1]).float()
            for p in procs:
                scores = p(ids, scores)
            nxt = int(scores.argmax(-1))
            ids = torch.cat([ids, torch.tensor([[nxt]])], dim=1)
            if nxt == RULES["eot"]:
                break
    out[f"tokens_{ci}"] = ids[0].numpy().astype(np.int32)
    print(ci, ids[0].tolist())
np.savez_compressed(os.path.join(HERE, "whisper_greedy_tiny.npz"), clips=np.array([5, 6, 7]), initial=np.array(INITIAL, dtype=np.int32),
                    suppress=np.array(RULES["suppress_tokens"], dtype=np.int32), blank=np.array(RULES["blank_tokens"], dtype=np.int32),
                    layout=np.array([RULES["eot"], RULES["no_timestamps"], RULES["timestamp_begin"], RULES["max_initial_timestamp_index"]], dtype=np.int32), **out)
print("wrote whisper_greedy_tiny.npz")

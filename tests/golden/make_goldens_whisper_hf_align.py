"""Generates tests/golden/whisper_hf_align.npz: the MIDDLE of ``find_alignment`` (openai-whisper timing.py: alignment-head selection ->
softmax -> ``std_mean`` normalisation over the token axis -> median filter -> mean over heads) and the token times that follow from it,
computed by the INSTALLED transformers implementation: ``WhisperGenerationMixin._extract_token_timestamps`` (the function behind
``generate(return_token_timestamps=True)``) is called on the cross-attentions of the transformers model itself (random-init two-layer
model carrying the package's fixed-seed weights, teacher-forced over the token list), and the matrix it hands to its DTW port is
captured.

Why the function is called directly and not through ``generate``: transformers' recipe differs from openai-whisper's in WHICH ROWS take
part in the normalisation -- ``generate`` drops the rows of the decoder prompt BEFORE the std / mean (``num_input_ids``) and has no row for
the last token, openai-whisper normalises over all rows and crops ``[len(sot_sequence):-1]`` afterwards (timing.py) -- so through
``generate`` the two can never agree.  Called with ``num_input_ids=0`` on a teacher-forced pass the function normalises over all rows,
exactly openai-whisper's arithmetic; the row crop is then applied to ITS matrix and ITS DTW port (``_dynamic_time_warping``) gives the
path.  Every arithmetic stage of the stored vectors is transformers' code; the one slice ``[sot_len:-1]`` is openai-whisper's.
(Window of 3000 frames: openai-whisper crops the attention LOGITS to ``num_frames // 2`` before the softmax, transformers crops the
probabilities after it; the two coincide only without a crop.  The cropped form stays a restatement.)

Run in the build container:  python tests/golden/make_goldens_whisper_hf_align.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)
import transformers.models.whisper.generation_whisper as G  # noqa: E402
from transformers.generation.utils import GenerateEncoderDecoderOutput  # noqa: E402
from make_goldens_whisper_hf import build_model, edims, tdims  # noqa: E402
from oracle import whisper_oracle as WO  # noqa: E402
from prosody_control_french_tts_amd import synth, whisper_weights as WW  # noqa: E402

SOT_LEN = 3
CASES = {"upper_half": [(1, 0), (1, 1)],          # openai-whisper's default when a model has no alignment heads: the upper half of the layers
         "picked": [(0, 1), (1, 0)]}              # a built-in style head list (layer, head)


def main():
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.synthetic_decoder_weights(tdims, seed=78)
    model = build_model(We, Wd)
    out = {}
    for ci, seconds, n_text, seed in ((5, 4.0, 17, 6), (8, 9.0, 40, 9)):
        clip = synth.synth_clip(ci, seconds=seconds)
        mel = WO.log_mel(clip, 80)
        tokens = np.random.default_rng(seed).integers(3, 300, size=SOT_LEN + 1 + n_text + 1).tolist()
        with torch.no_grad():
            fwd = model(input_features=torch.from_numpy(mel)[None], decoder_input_ids=torch.tensor([tokens]), output_attentions=True)
        gen = GenerateEncoderDecoderOutput(sequences=torch.tensor([tokens]), cross_attentions=(tuple(fwd.cross_attentions),))
        for name, heads in CASES.items():
            captured = []
            real = G._dynamic_time_warping
            G._dynamic_time_warping = lambda m: (captured.append(np.array(m)), real(m))[1]
            try:
                ts = model._extract_token_timestamps(gen, heads, time_precision=0.02, num_frames=None, num_input_ids=0)
            finally:
                G._dynamic_time_warping = real
            neg = captured[0]                                                  # -matrix over ALL rows, float64 [T][1500]
            assert neg.shape == (len(tokens), 1500) and ts.shape == (1, len(tokens) + 1)
            ti, tj = real(neg[SOT_LEN:-1])                                     # openai-whisper's row crop, transformers' DTW
            jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
            key = f"c{ci}_{name}"
            out[key + "_tokens"] = np.array(tokens, dtype=np.int32)
            out[key + "_heads"] = np.array(heads, dtype=np.int32)
            out[key + "_matrix"] = (-neg).astype(np.float32)                  # what the restatement's normalise / filter / mean must reproduce
            out[key + "_all_rows_times"] = ts[0].numpy().astype(np.float32)    # the function's own return value (DTW over all rows)
            out[key + "_text_idx"] = ti.astype(np.int32); out[key + "_time_idx"] = tj.astype(np.int32)
            out[key + "_jump_times"] = (tj[jumps] * 0.02).astype(np.float64)
            print(key, len(tokens), "tokens; first jump times", np.round(tj[jumps][:6] * 0.02, 2))
    out["clips"] = np.array([[5, 4.0], [8, 9.0]]); out["sot_len"] = np.array([SOT_LEN])
    np.savez_compressed(os.path.join(HERE, "whisper_hf_align.npz"), **out)
    print("wrote whisper_hf_align.npz", os.path.getsize(os.path.join(HERE, "whisper_hf_align.npz")))


if __name__ == "__main__":
    main()

"""Whisper checkpoints -> the flat float32 blobs ``pce_whisper_load`` / ``pce_whisper_decoder_load`` take.

The reference calls ``whisper.load_model(model_size, device=...)`` (Code/Aligners/use_whisper_timestamped.py:97), which
downloads ``<size>.pt``: ``{"dims": {...}, "model_state_dict": {...}}`` with parameter names ``encoder.conv1.weight``,
``encoder.blocks.N.attn.query.weight``, ``decoder.token_embedding.weight`` ...  Nothing can be downloaded here, so the
model directory is named by the caller (``PCE_WHISPER_DIR`` or the additive config key ``whisper_dir``):

    <dir>/<size>.pt                    openai-whisper checkpoint (torch.load), or
    <dir>/<size>.safetensors|.npz      the same state dict, or a Hugging Face ``WhisperForConditionalGeneration`` one
    <dir>/<size>.json                  optional: {"dims": {...}, "alignment_heads": [[layer, head], ...]}
    <dir>/multilingual.tiktoken        vocabulary (openai-whisper assets format), see Aligners/tokenizer.py

Hugging Face parameter names are mapped onto openai-whisper's; both carry the same tensors.  The tests write a
random-init miniature model in these formats (no trained weights exist offline).
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Optional, Tuple

import numpy as np

from .. import whisper_weights as WW

_HF_LAYER = {
    "self_attn.q_proj": "attn.query", "self_attn.k_proj": "attn.key", "self_attn.v_proj": "attn.value", "self_attn.out_proj": "attn.out",
    "self_attn_layer_norm": "attn_ln", "encoder_attn.q_proj": "cross_attn.query", "encoder_attn.k_proj": "cross_attn.key",
    "encoder_attn.v_proj": "cross_attn.value", "encoder_attn.out_proj": "cross_attn.out", "encoder_attn_layer_norm": "cross_attn_ln",
    "fc1": "mlp.0", "fc2": "mlp.2", "final_layer_norm": "mlp_ln",
}


def _hf_to_openai(name: str) -> Optional[str]:
    """``model.encoder.layers.3.self_attn.q_proj.weight`` -> ``encoder.blocks.3.attn.query.weight`` (None: not a tensor we load)."""
    name = name[6:] if name.startswith("model.") else name
    if name.startswith("proj_out."):
        return None                                              # tied to the token embedding
    m = re.match(r"(encoder|decoder)\.layers\.(\d+)\.(.+)\.(weight|bias)$", name)
    if m:
        side, l, mid, leaf = m.groups()
        return f"{side}.blocks.{l}.{_HF_LAYER[mid]}.{leaf}" if mid in _HF_LAYER else None
    flat = {"encoder.conv1": "encoder.conv1", "encoder.conv2": "encoder.conv2", "encoder.layer_norm": "encoder.ln_post",
            "decoder.layer_norm": "decoder.ln", "decoder.embed_tokens": "decoder.token_embedding"}
    for hf, oa in flat.items():
        if name.startswith(hf + "."):
            return oa + name[len(hf):]
    if name == "decoder.embed_positions.weight":
        return "decoder.positional_embedding"
    if name == "encoder.embed_positions.weight":
        return None                                              # the fixed sinusoids: the engine builds them itself
    return None


def _to_numpy(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().to("cpu").float().numpy()
    return np.asarray(v, dtype=np.float32)


def split_state_dict(sd: Dict[str, object]) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Any supported state dict -> (encoder tensors, decoder tensors) under the names of ``whisper_weights``."""
    if any(k.startswith("model.") or ".layers." in k for k in sd):
        sd = {n: v for n, v in ((_hf_to_openai(k), v) for k, v in sd.items()) if n}
    enc = {k[len("encoder."):]: _to_numpy(v) for k, v in sd.items() if k.startswith("encoder.")}
    dec = {k[len("decoder."):]: _to_numpy(v) for k, v in sd.items() if k.startswith("decoder.")}
    enc.pop("positional_embedding", None)
    return enc, dec


def dims_from_tensors(enc: Dict[str, np.ndarray], dec: Dict[str, np.ndarray]) -> Tuple[dict, dict]:
    d, n_mels, _ = enc["conv1.weight"].shape
    n_layer = 1 + max(int(k.split(".")[1]) for k in enc if k.startswith("blocks."))
    n_vocab, dd = dec["token_embedding.weight"].shape
    n_text_layer = 1 + max(int(k.split(".")[1]) for k in dec if k.startswith("blocks."))
    return (dict(n_mels=int(n_mels), n_ctx=1500, n_state=int(d), n_head=int(d) // 64, n_layer=n_layer),
            dict(n_vocab=int(n_vocab), n_text_ctx=int(dec["positional_embedding"].shape[0]), n_state=int(dd), n_head=int(dd) // 64,
                 n_layer=n_text_layer))


def pad_vocab(dec: Dict[str, np.ndarray], tdims: dict, n_vocab: int):
    """Checkpoints whose embedding has fewer rows than the tokenizer has ids cannot decode them; more rows are fine."""
    if tdims["n_vocab"] < n_vocab:
        raise ValueError(f"the checkpoint embeds {tdims['n_vocab']} tokens, the vocabulary has {n_vocab}")


# Cross-attention heads whisper.load_model installs as ``model.alignment_heads`` (openai-whisper ``__init__.py``
# ``_ALIGNMENT_HEADS``; the ``.pt`` file does not carry them) and whisper_timestamped reads for its DTW.  openai-whisper ships
# them as base85 bit masks; these are the same sets as [layer, head] pairs, the form the models' published
# ``generation_config.json`` uses.  Third party, absent here: restated from those published lists, unverifiable offline --
# a sidecar ``<size>.json`` always wins.
ALIGNMENT_HEADS = {
    "tiny.en": [[1, 0], [2, 0], [2, 5], [3, 0], [3, 1], [3, 2], [3, 3], [3, 4]],
    "tiny": [[2, 2], [3, 0], [3, 2], [3, 3], [3, 4], [3, 5]],
    "base.en": [[3, 3], [4, 7], [5, 1], [5, 5], [5, 7]],
    "base": [[3, 1], [4, 2], [4, 3], [4, 7], [5, 1], [5, 2], [5, 4], [5, 6]],
    "small.en": [[6, 6], [7, 0], [7, 3], [7, 8], [8, 2], [8, 5], [8, 7], [9, 0], [9, 4], [9, 8], [9, 10], [10, 0], [10, 1], [10, 2],
                 [10, 3], [10, 6], [10, 11], [11, 2], [11, 4]],
    "small": [[5, 3], [5, 9], [8, 0], [8, 4], [8, 7], [8, 8], [9, 0], [9, 7], [9, 9], [10, 5]],
    "medium.en": [[11, 4], [14, 1], [14, 12], [14, 14], [15, 4], [16, 0], [16, 4], [16, 9], [17, 12], [17, 14], [18, 7], [18, 10],
                  [18, 15], [20, 0], [20, 3], [20, 9], [20, 14], [21, 12]],
    "medium": [[13, 15], [15, 4], [15, 15], [16, 1], [20, 0], [23, 4]],
    "large-v1": [[9, 19], [11, 2], [11, 4], [11, 17], [22, 7], [22, 11], [22, 17], [23, 2], [23, 15]],
    "large-v2": [[10, 12], [13, 17], [16, 11], [16, 12], [16, 13], [17, 15], [17, 16], [18, 4], [18, 11], [18, 19], [19, 11],
                 [21, 2], [21, 3], [22, 3], [22, 9], [22, 12], [23, 5], [23, 7], [23, 13], [25, 5], [26, 1], [26, 12], [27, 15]],
    "large-v3": [[7, 0], [10, 17], [12, 18], [13, 12], [16, 1], [17, 14], [19, 11], [21, 4], [24, 1], [25, 6]],
    "large-v3-turbo": [[2, 4], [2, 11], [3, 3], [3, 6], [3, 11], [3, 14]],
}
ALIGNMENT_HEADS["large"] = ALIGNMENT_HEADS["large-v3"]          # openai-whisper 20240930: "large" -> large-v3, "turbo" -> large-v3-turbo
ALIGNMENT_HEADS["turbo"] = ALIGNMENT_HEADS["large-v3-turbo"]


def builtin_alignment_heads(model_size: str, n_layer: int, n_head: int):
    """The table entry for ``model_size`` if it fits a decoder of ``n_layer`` x ``n_head`` (a miniature test model saved
    under an official name does not), else None."""
    heads = ALIGNMENT_HEADS.get(str(model_size).lower())
    if heads is None or any(l >= n_layer or h >= n_head for l, h in heads):
        return None
    return heads


class WhisperModel:
    """What ``whisper.load_model`` returns, reduced to what the engine needs: dims, packed weights, alignment heads."""

    def __init__(self, enc: Dict[str, np.ndarray], dec: Dict[str, np.ndarray], alignment_heads=None, name: str = ""):
        self.name = name
        self.dims, self.text_dims = dims_from_tensors(enc, dec)
        missing = [n for n, _ in WW.tensor_order(self.dims) if n not in enc] + [n for n, _ in WW.decoder_tensor_order(self.text_dims) if n not in dec]
        if missing:
            raise KeyError(f"checkpoint lacks {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        self.encoder_blob = WW.pack(enc, self.dims)
        self.decoder_blob = WW.pack_decoder(dec, self.text_dims)
        L, H = self.text_dims["n_layer"], self.text_dims["n_head"]
        mask = np.zeros((L, H), dtype=np.uint8)
        if alignment_heads is None:
            mask[L // 2:] = 1                                    # model.py: "use the last half among the decoder layers by default"
        else:
            for l, h in alignment_heads:
                mask[int(l), int(h)] = 1
        self.alignment_heads = mask

    def load_into(self, engine):
        engine.whisper_load(self.dims, self.encoder_blob)
        engine.whisper_decoder_load(self.text_dims, self.decoder_blob)
        return self


def _read_state_dict(path: str) -> Tuple[Dict[str, object], Optional[dict]]:
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}, None
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return load_file(path), None
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(ck, dict) and "model_state_dict" in ck:
        return ck["model_state_dict"], ck.get("dims")
    return ck, None


def find_checkpoint(model_size: str, model_dir: Optional[str] = None) -> str:
    model_dir = model_dir or os.environ.get("PCE_WHISPER_DIR")
    if not model_dir:
        raise FileNotFoundError(f'no Whisper checkpoint directory: set PCE_WHISPER_DIR (or the "whisper_dir" config key) to a directory '
                                f"holding {model_size}.pt / .safetensors / .npz and multilingual.tiktoken (this build cannot download models)")
    for ext in (".pt", ".safetensors", ".npz"):
        p = os.path.join(model_dir, model_size + ext)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"{model_dir} holds no {model_size}.pt / .safetensors / .npz")


def load_model(model_size: str, model_dir: Optional[str] = None) -> WhisperModel:
    path = find_checkpoint(model_size, model_dir)
    sd, _ = _read_state_dict(path)
    heads = None
    side = os.path.splitext(path)[0] + ".json"
    if os.path.exists(side):
        with open(side, encoding="utf-8") as f:
            heads = json.load(f).get("alignment_heads")
    enc, dec = split_state_dict(sd)
    if heads is None:
        _, tdims = dims_from_tensors(enc, dec)
        heads = builtin_alignment_heads(model_size, tdims["n_layer"], tdims["n_head"])
        if heads is None:
            import logging
            logging.getLogger(__name__).warning(
                "no alignment heads known for Whisper model %r (no %s, not an official size): falling back to every head of the last "
                "half of the decoder layers; word timestamps will differ from whisper.load_model's", model_size, os.path.basename(side))
    return WhisperModel(enc, dec, heads, name=model_size)


def load_tokenizer(model_dir: Optional[str] = None, language: str = "fr", n_vocab: Optional[int] = None):
    from .tokenizer import WhisperTokenizer
    model_dir = model_dir or os.environ.get("PCE_WHISPER_DIR")
    path = os.path.join(model_dir or "", "multilingual.tiktoken")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing (openai-whisper assets/multilingual.tiktoken)")
    tok = WhisperTokenizer.from_tiktoken_file(path, language=language)
    if n_vocab is not None and n_vocab >= tok.n_vocab + 1:       # large-v3 family: one more language token
        tok = WhisperTokenizer.from_tiktoken_file(path, language=language, num_languages=100)
    return tok

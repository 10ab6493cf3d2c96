"""oracle/whisper_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

torch (CPU, float32) restatement of openai-whisper==20240930's ``log_mel_spectrogram`` and
``AudioEncoder.forward`` (whisper/audio.py, whisper/model.py), the device work behind
``whisper_timestamped.transcribe`` (Code/Aligners/use_whisper_timestamped.py:139,150-163).
openai-whisper is a third-party dependency absent from /root/reference and not installed here, and
no checkpoint is available offline: the architecture is restated from its published definition and
exercised with fixed-seed synthetic weights.  Pinned against an independent implementation of the
same network: tests/golden/whisper_hf_tiny.npz holds log-mel columns, encoder rows, teacher-forced
decoder logits and a cross-attention map produced by the installed transformers port
(tests/golden/make_goldens_whisper_hf.py; tests/test_whisper_hf_crosscheck.py).  The stage of
``find_alignment`` after the attention logits (head selection, softmax, std / mean normalisation,
median filter, head mean, DTW, token jump times) is pinned to transformers'
``_extract_token_timestamps`` on the full 30 s window (tests/golden/whisper_hf_align.npz, made by
tests/golden/make_goldens_whisper_hf_align.py, which says how the two recipes differ in their row
crop and why the function is driven directly); with a crop to ``num_frames // 2`` columns
openai-whisper cuts the logits before the softmax and transformers the probabilities after it:
that form stays a restatement.
"""
import numpy as np
import torch
import torch.nn.functional as F

N_FFT, HOP, N_SAMPLES, N_FRAMES = 400, 160, 480000, 3000


def mel_filters(n_mels=80, sr=16000, n_fft=N_FFT):
    """librosa.filters.mel(sr, n_fft, n_mels) (Slaney scale + normalisation) -- what whisper ships as mel_filters.npz."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=float)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=float)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def log_mel(pcm_i16: np.ndarray, n_mels=80) -> np.ndarray:
    """whisper.log_mel_spectrogram(audio, n_mels, padding=N_SAMPLES)[:, :3000] for int16 16 kHz audio."""
    audio = torch.from_numpy(pcm_i16.astype(np.float32) / 32768.0)
    audio = F.pad(audio, (0, N_SAMPLES))
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP, window=window, return_complex=True)
    mag = stft[..., :-1].abs() ** 2
    mel = torch.from_numpy(mel_filters(n_mels)) @ mag
    log_spec = torch.clamp(mel, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec[:, :N_FRAMES].numpy()


def log_mel_window(pcm_i16: np.ndarray, start_frame: int, n_mels=80) -> np.ndarray:
    """The mel segment ``whisper.transcribe`` hands to the model at seek position ``start_frame``:
    ``log_mel_spectrogram(audio, padding=N_SAMPLES)`` of the WHOLE recording (reflect-padded STFT, drop the last frame,
    clamp at the global maximum - 8, (x + 4) / 4), then ``mel[:, seek : seek + N_FRAMES]`` (padded to 3000 frames)."""
    audio = torch.from_numpy(np.asarray(pcm_i16, dtype=np.float32) / 32768.0)
    audio = F.pad(audio, (0, N_SAMPLES))
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP, window=window, return_complex=True)
    mag = stft[..., :-1].abs() ** 2
    mel = torch.from_numpy(mel_filters(n_mels)) @ mag
    log_spec = torch.clamp(mel, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    seg = log_spec[:, start_frame:start_frame + N_FRAMES]
    if seg.shape[1] < N_FRAMES:
        seg = F.pad(seg, (0, N_FRAMES - seg.shape[1]))
    return seg.numpy().astype(np.float32)


def sinusoids(length, channels, max_timescale=10000):
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    st = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def encoder_forward(mel: np.ndarray, W: dict, dims: dict) -> np.ndarray:
    """AudioEncoder.forward on one [n_mels, 3000] log-mel window -> [1500, n_state] (float32)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    d, H = dims["n_state"], dims["n_head"]
    x = t(mel)[None]
    x = F.gelu(F.conv1d(x, t(W["conv1.weight"]), t(W["conv1.bias"]), padding=1))
    x = F.gelu(F.conv1d(x, t(W["conv2.weight"]), t(W["conv2.bias"]), stride=2, padding=1))
    x = x.permute(0, 2, 1)
    x = x + sinusoids(dims["n_ctx"], d)
    for l in range(dims["n_layer"]):
        p = f"blocks.{l}."
        h = F.layer_norm(x, (d,), t(W[p + "attn_ln.weight"]), t(W[p + "attn_ln.bias"]))
        q = F.linear(h, t(W[p + "attn.query.weight"]), t(W[p + "attn.query.bias"]))
        k = F.linear(h, t(W[p + "attn.key.weight"]))
        v = F.linear(h, t(W[p + "attn.value.weight"]), t(W[p + "attn.value.bias"]))
        n_b, n_t, _ = q.shape
        scale = (d // H) ** -0.25
        q = q.view(n_b, n_t, H, -1).permute(0, 2, 1, 3) * scale
        k = k.view(n_b, n_t, H, -1).permute(0, 2, 3, 1) * scale
        v = v.view(n_b, n_t, H, -1).permute(0, 2, 1, 3)
        w = F.softmax((q @ k).float(), dim=-1)
        a = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
        x = x + F.linear(a, t(W[p + "attn.out.weight"]), t(W[p + "attn.out.bias"]))
        h = F.layer_norm(x, (d,), t(W[p + "mlp_ln.weight"]), t(W[p + "mlp_ln.bias"]))
        h = F.linear(F.gelu(F.linear(h, t(W[p + "mlp.0.weight"]), t(W[p + "mlp.0.bias"]))), t(W[p + "mlp.2.weight"]), t(W[p + "mlp.2.bias"]))
        x = x + h
    x = F.layer_norm(x, (d,), t(W["ln_post.weight"]), t(W["ln_post.bias"]))
    return x[0].numpy()


def dtw_path(x: np.ndarray):
    """openai-whisper timing.py ``dtw_cpu`` + ``backtrace`` (numba there, plain numpy loops here)."""
    x = np.asarray(x, dtype=np.float64)
    N, M = x.shape
    cost = np.ones((N + 1, M + 1), dtype=np.float32) * np.inf            # float32 there too: x + c is rounded at every cell
    trace = -np.ones((N + 1, M + 1), dtype=np.int64)
    cost[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = x[i - 1, j - 1] + c
            trace[i, j] = t
    i, j = N, M
    trace[0, :] = 2
    trace[:, 0] = 1
    out = []
    while i > 0 or j > 0:
        out.append((i - 1, j - 1))
        if trace[i, j] == 0:
            i -= 1; j -= 1
        elif trace[i, j] == 1:
            i -= 1
        else:
            j -= 1
    out = np.array(out)[::-1]
    return out[:, 0], out[:, 1]


def median_filter(x: torch.Tensor, width: int) -> torch.Tensor:
    """openai-whisper timing.py median_filter (reflect padding along the last axis)."""
    pad = width // 2
    if x.shape[-1] <= pad:
        return x
    x = F.pad(x[None], (pad, pad, 0, 0), mode="reflect")[0] if x.ndim == 3 else F.pad(x, (pad, pad, 0, 0), mode="reflect")
    return x.unfold(-1, width, 1).sort()[0][..., width // 2]


def find_alignment(tokens, enc_out: np.ndarray, W: dict, dims: dict, num_frames: int, sot_len: int, head_mask=None,
                   medfilt_width: int = 7, qk_scale: float = 1.0, want_internal: bool = False, want_matrix: bool = False):
    """TextDecoder.forward (teacher forced) + the cross-attention / DTW part of timing.py find_alignment.
    Returns (cost matrix fed to the DTW, text_indices, time_indices)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    d, H, L = dims["n_state"], dims["n_head"], dims["n_layer"]
    tok = torch.tensor(tokens, dtype=torch.long)
    T = len(tokens)
    xa = t(enc_out)[None]
    x = t(W["token_embedding.weight"])[tok][None] + t(W["positional_embedding"])[:T]
    mask = torch.empty(T, T).fill_(-np.inf).triu_(1)
    if head_mask is None:
        head_mask = np.zeros((L, H), dtype=bool); head_mask[L // 2:] = True
    head_mask = np.asarray(head_mask, dtype=bool).reshape(L, H)
    qks = []

    def mha(xq, xkv, p, m):
        q = F.linear(xq, t(W[p + "query.weight"]), t(W[p + "query.bias"]))
        k = F.linear(xkv, t(W[p + "key.weight"]))
        v = F.linear(xkv, t(W[p + "value.weight"]), t(W[p + "value.bias"]))
        n_b, n_q, _ = q.shape
        scale = (d // H) ** -0.25
        qh = q.view(n_b, n_q, H, -1).permute(0, 2, 1, 3) * scale
        kh = k.view(n_b, k.shape[1], H, -1).permute(0, 2, 3, 1) * scale
        vh = v.view(n_b, v.shape[1], H, -1).permute(0, 2, 1, 3)
        qk = qh @ kh
        if m is not None:
            qk = qk + m
        w = F.softmax(qk.float(), dim=-1)
        out = (w @ vh).permute(0, 2, 1, 3).flatten(start_dim=2)
        return F.linear(out, t(W[p + "out.weight"]), t(W[p + "out.bias"])), qk[0]

    for l in range(L):
        p = f"blocks.{l}."
        a, _ = mha(F.layer_norm(x, (d,), t(W[p + "attn_ln.weight"]), t(W[p + "attn_ln.bias"])), None if False else
                   F.layer_norm(x, (d,), t(W[p + "attn_ln.weight"]), t(W[p + "attn_ln.bias"])), p + "attn.", mask)
        x = x + a
        a, qk = mha(F.layer_norm(x, (d,), t(W[p + "cross_attn_ln.weight"]), t(W[p + "cross_attn_ln.bias"])), xa, p + "cross_attn.", None)
        x = x + a
        for hh in range(H):
            if head_mask[l, hh]:
                qks.append(qk[hh])
        h = F.layer_norm(x, (d,), t(W[p + "mlp_ln.weight"]), t(W[p + "mlp_ln.bias"]))
        x = x + F.linear(F.gelu(F.linear(h, t(W[p + "mlp.0.weight"]), t(W[p + "mlp.0.bias"]))), t(W[p + "mlp.2.weight"]), t(W[p + "mlp.2.bias"]))
    if want_internal:                     # the teacher-forced decoder's own outputs (cross-checks against other implementations)
        hidden = F.layer_norm(x, (d,), t(W["ln.weight"]), t(W["ln.bias"]))[0]
        logits = (hidden @ t(W["token_embedding.weight"]).T).numpy()
        return dict(hidden=hidden.numpy(), logits=logits, cross_qk=[q.numpy() for q in qks])
    weights = torch.stack(qks)[:, :, : num_frames // 2]
    weights = (weights * qk_scale).softmax(dim=-1)
    std, mean = torch.std_mean(weights, dim=-2, keepdim=True, unbiased=False)
    weights = (weights - mean) / std
    weights = median_filter(weights, medfilt_width)
    matrix = weights.mean(axis=0)
    if want_matrix:                       # the normalised, filtered, head-averaged matrix over ALL rows (pinned: tests/golden/whisper_hf_align.npz)
        full = matrix.numpy().copy()
    matrix = matrix[sot_len:-1]
    cost = (-matrix).double().numpy()
    ti, tj = dtw_path(cost)
    if want_matrix:
        return cost, ti, tj, full
    return cost, ti, tj


def apply_decoding_rules(logits: np.ndarray, tokens, sample_begin: int, rules: dict) -> np.ndarray:
    """openai-whisper decoding.py logit filters of a default ``DecodingTask`` at temperature 0, in its order:
    SuppressBlank, SuppressTokens, ApplyTimestampRules.  ``logits``: float32 [n_vocab] of the last position;
    ``tokens``: the sequence so far (prompt included).  rules: eot, no_timestamps, timestamp_begin, suppress_tokens,
    blank_tokens (the ids of " " and eot), max_initial_timestamp_index."""
    x = torch.from_numpy(np.array(logits, dtype=np.float32))
    eot, tsb = rules["eot"], rules["timestamp_begin"]
    if len(tokens) == sample_begin:
        x[list(rules["blank_tokens"])] = -np.inf
    x[list(rules["suppress_tokens"])] = -np.inf
    x[rules["no_timestamps"]] = -np.inf
    seq = list(tokens[sample_begin:])
    last_ts = len(seq) >= 1 and seq[-1] >= tsb
    pen_ts = len(seq) < 2 or seq[-2] >= tsb
    if last_ts:
        if pen_ts:
            x[tsb:] = -np.inf
        else:
            x[:eot] = -np.inf
    stamps = [t for t in seq if t >= tsb]
    if stamps:
        last = stamps[-1] if (last_ts and not pen_ts) else stamps[-1] + 1
        x[tsb:last] = -np.inf
    if len(tokens) == sample_begin:
        x[:tsb] = -np.inf
        if rules.get("max_initial_timestamp_index") is not None:
            x[tsb + rules["max_initial_timestamp_index"] + 1:] = -np.inf
    logprobs = F.log_softmax(x.float(), dim=-1)
    if logprobs[tsb:].logsumexp(dim=-1) > logprobs[:tsb].max():
        x[:tsb] = -np.inf
    return x.numpy()


def greedy_decode(enc_out: np.ndarray, W: dict, dims: dict, initial_tokens, rules: dict, sample_len: int):
    """``DecodingTask._main_loop`` with a ``GreedyDecoder`` at temperature 0 for one utterance: arg-max of the filtered
    logits until end-of-text or ``sample_len`` new tokens.  -> the full token list (prompt included)."""
    tokens = list(initial_tokens)
    sample_begin = len(tokens)
    for _ in range(sample_len):
        logits = find_alignment(tokens, enc_out, W, dims, 2, 0, want_internal=True)["logits"][-1]
        nxt = int(np.argmax(apply_decoding_rules(logits, tokens, sample_begin, rules)))
        tokens.append(nxt)
        if nxt == rules["eot"]:
            break
    return tokens


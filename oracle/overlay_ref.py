"""CPU restatement of the int16 mix in add_noise (core/process_data.py:283-297):
``noise_track.overlay(speech_track, gain_during_overlay=dB)``.

TEST INFRASTRUCTURE ONLY (imported by tests/; the product path is vad_overlay_i16 on the GPU).

PINNED for its arithmetic: the standard library's audioop (the C module pydub calls for both steps)
is importable here, and tests/test_oracle_overlay.py holds this restatement against audioop.mul +
audioop.add (width 2) on the reference's three gains, the saturation edges and ragged lengths.
pydub itself is not importable and the reference holds no vectors for this step, so pydub's control
flow around the two calls is restated from its published behaviour:
  * pydub.AudioSegment.overlay: over the part of the base track (the noise) that the other
    segment (the speech) covers, the base is first passed through audioop.mul with factor
    db_to_float(gain) = 10 ** (gain / 20) when gain_during_overlay is truthy (None and 0
    skip it), then audioop.add mixes the two; the rest of the base track is copied.
  * audioop.mul (width 2): val = sample * factor in double; values above 32767 become 32767,
    values below -32767 become -32768, then floor.
  * audioop.add (width 2): integer sum clamped to [-32768, 32767].
"""
import numpy as np


def overlay(noise_frames, speech_frames, gain_during_overlay=None):
    noise = np.asarray(noise_frames, dtype=np.int16).reshape(-1)
    speech = np.asarray(speech_frames, dtype=np.int16).reshape(-1)
    n = min(noise.size, speech.size)
    base = noise[:n].astype(np.float64)
    if gain_during_overlay:
        v = base * (10.0 ** (float(gain_during_overlay) / 20.0))
        v = np.where(v > 32767.0, 32767.0, np.where(v < -32767.0, -32768.0, v))
        base = np.floor(v)
    out = noise.astype(np.int64)
    out[:n] = np.clip(base.astype(np.int64) + speech[:n].astype(np.int64), -32768, 32767)
    return out.astype(np.int16).reshape(-1, 480) if out.size % 480 == 0 else out.astype(np.int16)

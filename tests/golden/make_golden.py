"""Regenerates tests/golden/*.npy from the reference's own asset (run in the authoring container,
where /root/reference exists; the GPU box only sees the committed .npy files).

BASELINE config 5: decode /root/reference/assets/music.wav (PCM16 LE, 2 ch, 44.1 kHz) directly,
keep a 32 768-sample excerpt that contains both quiet and loud passages, and write
  music_excerpt_i16.npy      the raw i16 samples
  music_excerpt_f32.npy      dasp_sample 0.11.0 `i16 -> f32`:  s as f32 / 32768.0   (sample.rs:42-44)
  music_excerpt_6to2.npy     the f32 stream re-framed as 6 channels -> ChannelCountConverter(6 -> 2):
                             channels >= 2 of every frame are dropped (channels.rs:57-85)
  music.wav                  the asset itself (1.8 MB, cc-by-sa: /root/reference/assets/README.md), byte for byte: BASELINE config 5
                             at full size feeds the real RIFF image through rh_wav_probe_host / rh_wav_decode
The expected arrays are computed here with numpy from the cited formulas, independently of oracle/.

tests/golden/wav/ (since round 6): the six files the reference's tests/wav_test.rs decodes (assets/audacity16bit.wav, lmms16bit.wav, lmms24bit.wav,
audacity32bit.wav, lmms32bit.wav, audacity32bit_int.wav), copied byte for byte -- `cp /root/reference/assets/<name> tests/golden/wav/` -- as INPUT data of
tests/test_wav_assets.py; nothing is derived from them here (the expected samples come from scipy.io.wavfile at test time).
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = "/root/reference/assets/music.wav"


def read_pcm16(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        body = b[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    assert fmt[0] == 1 and fmt[1] == 2 and fmt[2] == 44100 and fmt[5] == 16, fmt
    return np.frombuffer(data, dtype="<i2")


if __name__ == "__main__":
    pcm = read_pcm16(WAV)
    assert len(pcm) == 894654, len(pcm)  # 447 327 stereo frames
    start = 2 * 200000
    ex = pcm[start:start + 32768].copy()
    f32 = (ex.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    frames6 = f32[: (len(f32) // 6) * 6].reshape(-1, 6)
    np.save(os.path.join(HERE, "music_excerpt_i16.npy"), ex)
    np.save(os.path.join(HERE, "music_excerpt_f32.npy"), f32)
    np.save(os.path.join(HERE, "music_excerpt_6to2.npy"), np.ascontiguousarray(frames6[:, :2]).reshape(-1))
    import shutil

    shutil.copyfile(WAV, os.path.join(HERE, "music.wav"))
    os.chmod(os.path.join(HERE, "music.wav"), 0o644)
    print("excerpt", ex.shape, "min/max", ex.min(), ex.max(), "nonzero", int(np.count_nonzero(ex)))

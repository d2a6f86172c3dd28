import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rodio_amd as rh
from oracle import rodio_oracle as O
rh.init(0)
S, N = 16, 262144
host = np.stack([(np.random.default_rng(1234 + s).uniform(-1, 1, 2 * N) / 32).astype(np.float32) for s in range(S)])
ref = O.pipeline_resample_lowpass_mix(host.reshape(S, N, 2), 44100, 48000, O.SPAN_NONE, 200, 0.5, want_output=True)
data = torch.from_numpy(host).cuda()
for mf in (False, True):
    for R in range(2, 21):
        for NS in (2, 3):
            try:
                p = rh.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=N, frames_per_lane=R, ring_stages=NS)
            except Exception as e:
                print(mf, R, NS, "create:", e); continue
            p.set_exclusive(False)
            p.set_mix_first(mf)
            p.set_sources([data[s] for s in range(S)])
            out = p.run().cpu().numpy()
            p.check_status()
            print(mf, R, NS, p.geometry()["mix_first"], float(np.abs(out - ref).max()), flush=True)
            p.close()

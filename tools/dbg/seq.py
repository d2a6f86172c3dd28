import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rodio_amd as rh
from oracle import rodio_oracle as O
rh.init(0)
S, N = 16, 262144
host = np.stack([(np.random.default_rng(1234 + s).uniform(-1, 1, 2 * N) / 32).astype(np.float32) for s in range(S)])
ref = O.pipeline_resample_lowpass_mix(host.reshape(S, N, 2), 44100, 48000, O.SPAN_NONE, 200, 0.5, want_output=True)
data = torch.from_numpy(host).cuda()
def err(t): return float(np.abs(t.cpu().numpy() - ref).max())
for excl in (True, False):
    for tune in (False, True):
        p = rh.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=N)
        p.set_exclusive(excl)
        p.set_sources([data[s] for s in range(S)])
        if tune: p.autotune()
        outs = [torch.empty(p.out_frames * 2, device="cuda") for _ in range(2)]
        for k in range(5): p.run(outs[k & 1])
        p.check_status()
        print("excl", excl, "tune", tune, "mix-first", p.geometry()["mix_first"], err(outs[0]), flush=True)
        p.set_mix_first(False)
        p.set_sources([data[s] for s in range(S)])
        o = torch.empty(p.out_frames * 2, device="cuda")
        p.run(o); p.check_status()
        print("   per-source before tune", p.geometry(), err(o), flush=True)
        if tune: p.autotune()
        for k in range(3): p.run(o)
        p.check_status()
        print("   per-source", p.geometry()["frames_per_lane"], p.geometry()["ring_stages"], err(o), flush=True)
        p.set_mix_first(True)
        p.set_sources([data[s] for s in range(S)])
        p.run(o); p.check_status()
        print("   mix-first again", err(o), flush=True)
        p.close()

"""Where does the time-parallel filter stay within 1e-5 of the REFERENCE's own f32 recurrence?  (VERDICT r03 weak #1.)
Full-scale stereo noise (|x| <= 1, what one rodio source may carry) through rh_biquad mode 1 (the scan), mode 0 (the reference's
order: bit-exact) and an f64 evaluation, for the cutoffs rodio users give low_pass()/high_pass(); error relative to the input peak.
    python tools/filter_contract.py > profiles/r04_filter_contract.txt"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodio_amd as rh
from scipy.signal import lfilter
rh.init(0)
n = 400_000
x = (np.random.default_rng(99).uniform(-1, 1, 2 * n)).astype(np.float32)
xd = torch.from_numpy(x).cuda().reshape(1, -1)
rows = []
for fs in (44100, 48000, 96000):
    for kind in ("low_pass", "high_pass"):
        for f in (10, 20, 30, 50, 100, 200, 300, 500, 1000, 5000):
            co = rh.biquad_coeffs(kind, f, 0.5, fs)
            seq = rh.biquad_batch(xd, co, mode=0).cpu().numpy()[0]
            par = rh.biquad_batch(xd, co, mode=1).cpu().numpy()[0]
            b = np.array(co[:3], dtype=np.float64); a = np.array([1.0, co[3], co[4]], dtype=np.float64)
            truth = np.empty(2 * n)
            for c in range(2):
                truth[c::2] = lfilter(b, a, x[c::2].astype(np.float64))
            pole = float(np.max(np.abs(np.roots(a))))
            rows.append({"fs": fs, "kind": kind, "freq": f, "pole_radius": round(pole, 6),
                         "scan_vs_reference": float(np.max(np.abs(par - seq))), "reference_vs_f64": float(np.max(np.abs(seq - truth))), "scan_vs_f64": float(np.max(np.abs(par - truth)))})
            print(json.dumps(rows[-1]), flush=True)

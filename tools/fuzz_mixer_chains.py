"""Random chains of the host mirror's whole adapter vocabulary (without dither) handed to a GpuMixer, against the oracle's Mixer -- on the CPU
stand-in of the C ABI (tests/cpp/host_mirror_test_fake).  Not part of the suite: it documents what is still open at the mixer boundary.

    python tools/fuzz_mixer_chains.py [first_seed last_seed]

Known categories it reports ("what is open"): a chain that ENDS INSIDE A FRAME (reverb with a delay that is no whole number of
frames) handed to a mixer -- pulled through the host the samples of its last, open frame are dropped; on the device the mix's last frame
differs by what the zero padding of that frame makes of it.  Loud refusals (span arithmetic that is not mirrored) are counted, not failures.
`channel_volume` is left out here only because its list of gains collides with the spec file's separator."""
import os
import pathlib
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import numpy as np
import test_host_logic_fuzz_cpu as F
import test_host_mirror as M
from oracle import rodio_oracle as O
bad=0; refused=0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
for seed in range(lo,hi):
    rng=np.random.default_rng(33000+seed)
    S=int(rng.integers(1,5)); mixer_ch=int(rng.choice([1,2,2,6])); to_rate=int(rng.choice([22050,44100,48000])); block=int(rng.choice([777,4096,20000])); on_device=bool(rng.integers(0,2))
    kind=str(rng.choice(["test","buffer","mixed","spans:2304","spans:1000"]))
    tmp=pathlib.Path(tempfile.mkdtemp())
    try:
        lines=[];adds=[]
        for i in range(S):
            gain=float(np.float32(rng.choice([0.5,0.8,1.0])))
            ch0,rate0=int(rng.choice([1,2,2,6])),int(rng.choice(F.RATES))
            x=M.rnd(33000+1000*seed+10*i,int(rng.integers(1,9000))*ch0,0.2); x.tofile(tmp/f"src_{i}.f32")
            ops=[o for o in F._full_ops(rng,ch0,int(rng.integers(1,3)),False) if not o.startswith('channel_volume')] if rng.random()<0.6 else []
            lines.append(f"{ch0} {rate0} {gain} -1 0 {','.join(ops) if ops else '-'}\n")
            adds.append((x,ch0,rate0,i,ops,gain))
        (tmp/"spec.txt").write_text("".join(lines))
        r=subprocess.run([F.FAKE,"chainmix",str(tmp),str(S),str(mixer_ch),str(to_rate),str(block),"1" if on_device else "0"],capture_output=True,text=True,timeout=300,env=dict(os.environ,RH_TEST_SOURCE=kind))
        what=(seed,S,mixer_ch,to_rate,block,on_device,kind,lines)
        if r.returncode!=0:
            if r.returncode==1 and "unsupported" in r.stderr.lower(): refused+=1; continue
            bad+=1; print('ERR',what,r.stderr[:300]); continue
        got=np.fromfile(tmp/"out.f32",dtype=np.float32)
        m=O.Mixer(mixer_ch,to_rate)
        for x,ch0,rate0,i,ops,gain in adds:
            m.add(O.UniformSourceIterator(F._oracle_full(O,M._span_source(O,kind,x,ch0,rate0,i),ops).amplify(gain),mixer_ch,to_rate))
        ref=m.collect()
        if len(got)!=len(ref): bad+=1; print('LEN',what,len(got),len(ref)); continue
        if len(ref):
            tol=2e-5*max(1.0,float(np.max(np.abs(ref))))*(8 if any('agc' in l or 'distortion' in l for l in lines) else 1)
            e=float(np.max(np.abs(got-ref)))
            if e>tol: bad+=1; print('TOL',what,e,int(np.argmax(np.abs(got-ref))))
    finally:
        shutil.rmtree(tmp,ignore_errors=True)
print('bad',bad,'refused',refused)

import importlib, sys, json
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
m = importlib.import_module("gpu-homomorphic-encryption_amd")
from oracle import pyoracle as orc
out = {}
for n in (8, 16, 64):
    q = 12289
    e = m.RnsNttEngine(n, [q])
    for name, vec in (("e0", [1] + [0]*(n-1)), ("e1", [0,1] + [0]*(n-2)), ("ramp", list(range(1, n+1)))):
        x = orc.to_limbs(vec)
        d = m.DeviceBuffer.from_numpy(x)
        e.forward(d, 1)
        got = d.download()
        want = orc.Plan(n, q).forward(x)
        out[f"{n}_{name}"] = dict(got=[int(v) for v in got[:, 0]], want=[int(v) for v in want[:, 0]], upper=int(got[:,1:].sum()))
        print(n, name, "OK" if np.array_equal(got, want) else "MISMATCH")
json.dump(out, open("gpurun_out/dbg256.json", "w"))

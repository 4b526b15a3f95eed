import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import chord_detection_amd as cd
from oracle import esacf as oe, thirdparty as tp, dsp
warnings.simplefilter("ignore")
d = np.load("tests/golden/clips.npz")
eng = cd.get_engine(0)
FS = 22050
for name in ("piano_like_Cmaj", "poly_seed2"):
    x = d[name]
    total, per = eng.esacf(x, FS, 1023, return_frames=True)
    want = oe.esacf_frames(x, FS)
    bad = np.argwhere(~np.isclose(per, want, rtol=1e-5, atol=1e-12))
    print(name, "bad entries", bad.tolist())
    e = eng.esacf_stage("esacf", x, FS, 1023)
    for f in sorted(set(b[0] for b in bad)):
        print(" frame", f, "gpu", per[f], "\n   want", want[f])
        ch, pk, it = oe.frame_chroma(e[f], FS, detail=True)
        print("   oracle-on-gpu-esacf chroma", ch)
        print("   peaks", pk, "\n   interp", it)
        for i, tau in enumerate(it):
            midi = 12 * (np.log2(FS / tau) - np.log2(440.0)) + 69
            print("     tau %.6f midi %.6f pc %d weight %.6f" % (tau, midi, int(np.round(midi)) % 12, e[f][pk[i]]))

import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chord_detection_amd as cd
from chord_detection_amd import corpus
from oracle import esacf as o_es
warnings.simplefilter("ignore")
eng = cd.get_engine(0)
fs = 22050
for mode in ("unicode", "ascii"):
  for cid in range(8):
    x = corpus.synth_chunk([cid], fs, 2.0, "cuda:0").cpu().numpy()[0]
    tot, per = eng.esacf(x, fs, 1023, return_frames=True, note_names=mode)
    want = o_es.esacf_frames(x, fs, note_names=mode)
    rows = eng.esacf_stage("esacf", x, fs, 1023)
    bad = [f for f in range(per.shape[0]) if not np.allclose(per[f], want[f], rtol=1e-5, atol=1e-12)]
    if not bad:
        continue
    for f in bad:
        frag = o_es.frame_fragility(rows[f], fs)
        frag10 = o_es.frame_fragility(rows[f], fs, trials=10)
        same = o_es.frame_chroma(rows[f], fs, note_names=mode)
        sh, bins = o_es.runaway_fit_bins(rows[f], fs)
        _, peaks, interp = o_es.frame_chroma(rows[f], fs, detail=True, note_names="ascii")
        print(mode, "clip", cid, "frame", f, "fragile(2)", frag, "fragile(10)", frag10, "shifted", sh, "bins", bins,
              "gpu==oracle(same input)", bool(np.allclose(per[f], same, rtol=1e-5, atol=1e-12)))
        print("   gpu   ", np.round(per[f], 4)); print("   oracle", np.round(want[f], 4)); print("   same  ", np.round(same, 4))
        print("   peaks", list(peaks), "interp", [round(float(t), 3) for t in interp])

"""Loader for the committed golden fixtures (tests/golden/*.npz, written by make_golden.py)."""
import glob
import os

import numpy as np

from vectordb_amd.flat_index import FlatIndex

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = lambda k: z["index_" + k] if ("index_" + k) in z.files else None  # noqa: E731
    fi = FlatIndex(dim=int(z["dim"]), metric=int(z["metric"]), code_kind=int(z["code_kind"]), centroids=g("centroids"),
                   child_offsets=g("child_offsets"), children=g("children"), roots=g("roots"),
                   vec_offsets=g("vec_offsets"), ids=g("ids"), vectors=g("vectors"), threshold=float(z["threshold"]),
                   codes=g("codes"), rabitq_rot_t=g("rabitq_rot_t"), rabitq_codes=g("rabitq_codes"),
                   rabitq_norm=g("rabitq_norm"), rabitq_ip_oo=g("rabitq_ip_oo")).normalised()

    def ragged(key):
        lens = z["exp_" + key + "_len"]
        flat = z["exp_" + key]
        out, o = [], 0
        for n in lens:
            out.append(flat[o:o + n])
            o += n
        return out

    exp = {k: ragged(k) for k in ("leaves", "cand_pos", "cand_key", "sel_pos", "sel_key", "sel_dist")}
    exp.update(ids=z["exp_ids"], dist=z["exp_dist"], count=z["exp_count"])
    par = dict(k=int(z["k"]), probes=int(z["probes"]), rf=int(z["rf"]), flags=int(z["flags"]))
    return fi, z["queries"], par, exp

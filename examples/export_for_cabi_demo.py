#!/usr/bin/env python3
"""Writes the two input files of examples/cabi_demo (weights.bin: a state_dict with the reference's key names; inputs.bin: source and
character windows plus cnt_mean / cnt_std), from the package's synthetic generators or from your own arrays through write_*()."""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_weights(path, state_dict, layout="mocha"):
    with open(path, "wb") as f:
        f.write(b"MOCHAW01"); f.write(struct.pack("<ii", 0 if layout == "mocha" else 1, len(state_dict)))
        for name, arr in state_dict.items():
            a = np.ascontiguousarray(np.asarray(arr), dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb))); f.write(nb)
            f.write(struct.pack("<i", a.ndim)); f.write(struct.pack(f"<{a.ndim}q", *a.shape)); f.write(a.tobytes())


def write_inputs(path, src, cha, cnt_mean, cnt_std):
    B, T, V, C = src.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", B, V))
        for a in (src, cha, cnt_mean, cnt_std):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())


if __name__ == "__main__":
    from mocha_sigasia2023_amd import synthetic, weights
    out = sys.argv[1] if len(sys.argv) > 1 else "."
    layout = sys.argv[2] if len(sys.argv) > 2 else "mocha"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    V = 24 if layout == "mocha" else 22
    write_weights(os.path.join(out, "weights.bin"), weights.synthetic_state_dict(1777, 1.0, layout), layout)
    m, s = synthetic.cnt_norm(7)
    write_inputs(os.path.join(out, "inputs.bin"), synthetic.pose_windows(1, B, V), synthetic.pose_windows(2, B, V), m, s)
    print(f"wrote {out}/weights.bin and {out}/inputs.bin ({B} windows, {layout}); run: examples/cabi_demo {out}/weights.bin {out}/inputs.bin {out}/out.bin")

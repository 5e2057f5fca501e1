"""Image-regression harness with the reference's metric (raycheck.py:22-32,103-186):
render every ``*.ray`` under --scenes with --exec (always ``-r 5``, raycheck.py:113),
compare with the image of the same name under --refs, PASS if
RMS = ||a - b||_2 / sqrt(#bytes) over uint8 RGB is below --maxrms (default 10.0,
raycheck.py:229-232).  The subtraction is done in float64 — the original subtracts
uint8 arrays, which wraps modulo 256 (raycheck.py:30).

    python -m pathtracer_amd.raycheck --scenes DIR --refs DIR [--exec pathtracer_amd/bin/ray]
"""
import argparse
import os
import subprocess
import sys

import numpy as np


def rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        raise ValueError(f"image shapes differ: {a.shape} vs {b.shape}")
    return float(np.linalg.norm((a - b).ravel()) / np.sqrt(a.size))


def load_rgb(path):
    from PIL import Image
    return np.array(Image.open(path).convert("RGB"))


def main(argv=None):
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--scenes", required=True)
    ap.add_argument("--refs", required=True)
    ap.add_argument("--exec", dest="exe", default=os.path.join(here, "bin", "ray"))
    ap.add_argument("--out", default="raycheck_out")
    ap.add_argument("--maxrms", type=float, default=10.0)
    ap.add_argument("--timelimit", type=float, default=180.0)   # raycheck.py:152-161
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--samples", type=int, default=2)
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    failed = 0
    for base, _, files in sorted(os.walk(args.scenes)):
        for f in sorted(files):
            if not f.endswith(".ray"):
                continue
            name = os.path.splitext(f)[0]
            out = os.path.join(args.out, name + ".png")
            ref = os.path.join(args.refs, name + ".png")
            try:
                subprocess.run([args.exe, "-r", "5", "-w", str(args.width), "-s", str(args.samples), os.path.join(base, f), out],
                               check=True, timeout=args.timelimit, capture_output=True)
                r = rms(load_rgb(out), load_rgb(ref))
                ok = r < args.maxrms
                print(f"{'PASS' if ok else 'FAIL'} {name}: rms {r:.3f}")
            except (subprocess.SubprocessError, OSError, ValueError) as e:
                ok = False
                print(f"FAIL {name}: {e}")
            failed += 0 if ok else 1
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())

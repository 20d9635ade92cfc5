"""Digests of the synthetic capture generator (slamhip/synth.py): python tests/golden/make_synth_digest.py
Pins the byte stream of two small frames; the device generator must produce the same bytes (GPU test).  No reference
involved: the generator is this build's own."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "slam-process_amd"))
from slamhip import synth  # noqa: E402

doc = {"4x128_f3": hashlib.sha256(synth.make_frame_text(4, 128, 3).tobytes()).hexdigest(),
       "8x256_f0": hashlib.sha256(synth.make_frame_text(8, 256, 0).tobytes()).hexdigest()}
json.dump(doc, open(os.path.join(HERE, "synth_digest.json"), "w"), indent=1)
print(doc)

#!/usr/bin/env python3
"""tests/golden/buildlib.json: digests of what the COMPILED REFERENCE's `megagta buildlib` writes for tests.helpers.write_buildlib_inputs
(build container only: needs oracle/_ref).   python tests/golden/make_golden_buildlib.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402

d = tempfile.mkdtemp(prefix="mgta_bl_")
lib = H.write_buildlib_inputs(d)
subprocess.run([os.path.join(ROOT, "oracle", "_ref", "megagta"), "buildlib", lib, f"{d}/out"], check=True, capture_output=True)
fx = {"bin_md5": hashlib.md5(open(f"{d}/out.bin", "rb").read()).hexdigest(), "bin_bytes": os.path.getsize(f"{d}/out.bin"),
      "lib_info": open(f"{d}/out.lib_info").read()}
json.dump(fx, open(os.path.join(ROOT, "tests", "golden", "buildlib.json"), "w"), indent=1)
print(fx)

"""Developer aid: `import _variant` first in a tools/ script makes CLAVA_LIB_VARIANT=<name> load
clavaddpm_amd/csrc/variants/libclava_hip_<name>.so (built by tools/build_variant.py) instead of the shipped library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from clavaddpm_amd import _lib  # noqa: E402

_name = os.environ.get("CLAVA_LIB_VARIANT")
if _name:
    _lib.LIB_PATH = os.path.join(ROOT, "clavaddpm_amd", "csrc", "variants", f"libclava_hip_{_name}.so")
    print(f"[variant] using {_lib.LIB_PATH}", flush=True)

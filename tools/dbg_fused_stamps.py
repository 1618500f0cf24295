"""Developer aid: where one wavefront of cls_tail spends its cycles (s_memtime stamps of workgroup 0, wavefront 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: F401
sys.argv = [sys.argv[0], "3"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "run_guided_steps.py")).read())
from clavaddpm_amd._lib import lib
buf = (ctypes.c_longlong * 16)()
torch.cuda.synchronize()
assert lib().clava_debug_fused_stamps_(buf) == 0
v = list(buf)
if os.environ.get("CLAVA_LIB_VARIANT") == "bstamps":
    names = ["start", "loads + transposition", "dL2 mfma", "dL2 epilogue", "dL1 mfma", "dL1 epilogue", "dL0 mfma", "gx + Gaussian step"]
    for i in range(1, 8):
        print("%-22s %8d cycles" % (names[i], v[i] - v[i - 1]))
    print("total", v[7] - v[0])
    sys.exit(0)
names = ["start", "a6 loads issued", "L7 mfma", "L7 epilogue", "L8 + softmax", "dL8 mfma", "dL8 epilogue", "dL7a mfma", "dL7a epi+store", "dL7b mfma", "dL7b epi+store"]
for i in range(1, 11):
    print("%-18s %8d cycles" % (names[i], v[i] - v[i - 1]))
print("total", v[10] - v[0])

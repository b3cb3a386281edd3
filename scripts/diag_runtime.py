"""Diagnose HIP-runtime coexistence with torch: which libamdhip64 copies are mapped, and does the path run."""
import faulthandler, os, re, sys, time
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    m = open('/proc/self/maps').read()
    return sorted(set(os.path.basename(os.path.dirname(p)) + '/' + os.path.basename(p) for p in re.findall(r'(/\S*(?:amdhip64|hsa-runtime)\S*)', m)))
t0 = time.time()
def log(*a): print(f"[{time.time()-t0:6.1f}s]", *a, flush=True)
if order == "torch_first":
    import torch
    log("torch imported", torch.cuda.is_available(), maps())
    from superslam_amd import _lib
    _lib.lib(); log("lib loaded", maps())
    _lib.init(0); log("sship_init ok")
else:
    from superslam_amd import _lib
    _lib.lib(); log("lib loaded", maps())
    _lib.init(0); log("sship_init ok")
    import torch
    log("torch imported", torch.cuda.is_available(), maps())
import numpy as np, tempfile
from superslam_amd import SuperPoint
from superslam_amd.weights import make_superpoint_weights, save_safetensors
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors")
sp = SuperPoint(d + "/sp.safetensors", 600, 0.005, 4, max_batch=4)
log("init", sp.initialize(), sp.last_error)
img = (np.random.default_rng(0).random((376, 1376)) * 255).astype(np.uint8)
f = sp.extract(img); log("extract host", len(f.keypoints), sp.last_error)
x = torch.from_numpy(np.stack([img] * 4)).cuda(); log("torch tensor on gpu")
desc, kp, n = sp.extract_batch_device(x); torch.cuda.synchronize(); log("batch device", n.tolist())
t = time.time()
for _ in range(5): sp.extract_batch_device(x, desc, kp, n)
torch.cuda.synchronize(); log("5 batch-4 extracts", (time.time() - t) / 5 * 1e3, "ms each")

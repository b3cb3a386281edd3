"""Time SuperPoint layers in isolation (sship_sp_bench_layer; also the target for rocprofv3 --pmc passes).
usage: prof_layer.py <layer[,layer...]> [iters] [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '.')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import SuperPoint, _lib
from superslam_amd.weights import make_superpoint_weights, save_safetensors
layers = [int(x) for x in sys.argv[1].split(",")]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5; B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
_lib.init(0)
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prof_w"); os.makedirs(d, exist_ok=True)
save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors")
sp = SuperPoint(d + "/sp.safetensors", 600, 0.005, 4, max_batch=B); assert sp.initialize(), sp.last_error
img = torch.from_numpy((np.random.default_rng(0).random((B, 376, 1376)) * 255).astype(np.uint8)).cuda()
sp.extract_batch_device(img); torch.cuda.synchronize()
ms = C.c_float(0); macs = C.c_double(0)
out = []
for layer in layers:
    _lib.check(_lib.lib().sship_sp_bench_layer(sp._h, layer, B, 376, 1376, iters, C.byref(ms), C.byref(macs)))
    out.append(f"L{layer} {ms.value*1e3:.1f}us {2*macs.value/ms.value/1e9:.0f}TF")
print(os.environ.get("TAG", ""), " | ".join(out))

"""Throughput of vad_threshold_sweep_f32 (FAR-at-fixed-FRR search over all 218 candidate thresholds in one pass)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voice_activity_detection_amd as pkg
from voice_activity_detection_amd import _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000_000
L = pkg.lib()
g = torch.Generator(device='cuda').manual_seed(5)
lab = (torch.rand(n, device='cuda', generator=g) < 0.4)
s = torch.sigmoid(torch.randn(n, device='cuda', generator=g) * 3 + torch.where(lab, 4.0, -6.0))
lab8 = lab.to(torch.uint8)
thr = np.asarray(pkg.metrics.frr_thresholds('-3'))
T = thr.size
thr_dev = torch.from_numpy(thr).cuda()
hist = torch.empty(2 * (T + 1), dtype=torch.int64, device='cuda')
ge = torch.empty((T, 2), dtype=torch.int64, device='cuda')
total = torch.empty(2, dtype=torch.int64, device='cuda')
def run():
    _lib.check(L.vad_threshold_sweep_f32(ctypes.c_void_p(s.data_ptr()), 1, ctypes.c_void_p(lab8.data_ptr()), n,
                                         ctypes.c_void_p(thr_dev.data_ptr()), T, ctypes.c_void_p(hist.data_ptr()),
                                         ctypes.c_void_p(ge.data_ptr()), ctypes.c_void_p(total.data_ptr()),
                                         _lib.current_stream()))
for _ in range(2):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
gbs = 5 * n / (ms * 1e-3) / 1e9
print(json.dumps({'kernel': 'sweep_hist_kernel', 'scores': n, 'thresholds': T, 'ms_per_launch': ms,
                  'scores_per_s': n / (ms * 1e-3), 'bound': 'hbm', 'bytes_per_score': 5, 'achieved': gbs, 'peak': 8000.0,
                  'unit': 'GB/s', 'frac': gbs / 8000.0, 'audio_hours_per_s': n / 120000 / (ms * 1e-3)}))

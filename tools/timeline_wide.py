"""Per-phase timeline of the stride-1 wide-GRU split kernel from a diagnostic build (-DVAD_TIMELINE=<block>):

    tools/build_variant.sh tl voice_activity_detection_amd/csrc -DVAD_TIMELINE=300
    VAD_AMD_LIB=$PWD/variants/libvad_tl.so python tools/timeline_wide.py [mode] [hours]

Waves 0 and 4 of the chosen block share SIMD 0.  Stamps per step: 0 step start; per unit tile mt: 1+4mt matrix work
issued, 2+4mt its results arrived, 3+4mt gate math done, 4+4mt rows moved / next row requested; 13 state split;
14 head done; 15 last tile's rows moved.  Prints the mean cycles of every phase over steps 3..28 for both waves and how
the two waves' matrix phases lie against each other."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voice_activity_detection_amd as pkg
from bench import make_net

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x8'
hours = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
n = int(hours * 120000)
net = make_net('gru_h90')
pkg.set_matrix_mode(mode)
feat = torch.randn(n, 24, device='cuda') * 8
for _ in range(3):
    p, d = net.score(feat, window_stride=24)
torch.cuda.synchronize()
L = pkg.lib()
buf = np.zeros(2 * 30 * 16, np.uint64)
L.vad_debug_timeline.restype = ctypes.c_int
assert L.vad_debug_timeline(buf.ctypes.data_as(ctypes.c_void_p)) == buf.size
tl = buf.reshape(2, 30, 16).astype(np.int64)
names = ['start->mm0 issued', 'mm0 done', 'gates0', 'rows0', 'mm1 issued', 'mm1 done', 'gates1', 'rows1', 'mm2 issued',
         'mm2 done', 'gates2', '(rows2 -)', 'split', 'head', 'rows2', 'loop']
print('s_memtime ticks (100 MHz constant clock on this part unless it counts shader cycles); steps 3..28')
for w in range(2):
    t = tl[w]
    d = np.diff(t, axis=1)                                  # (30, 15) phase lengths inside a step
    nxt = t[1:, 0] - t[:-1, 15]                             # loop back edge
    step = t[1:, 0] - t[:-1, 0]
    print('wave %d: mean step %.0f ticks' % (4 * w, step[3:28].mean()))
    for k in range(15):
        print('   %-20s %8.1f' % (names[k], d[3:28, k].mean()))
    print('   %-20s %8.1f' % (names[15], nxt[3:27].mean()))
# relative phase of the two waves: where in wave 0's step does wave 4's step start
off = (tl[1, 3:28, 0] - tl[0, 3:28, 0])
step0 = (tl[0, 4:29, 0] - tl[0, 3:28, 0])
print('wave 4 step start - wave 0 step start (ticks):', off.tolist())
print('wave 0 step length:', step0.tolist())

"""Per-stage timeline of the MFCC kernel from a diagnostic build (-DVAD_FE_TIMELINE=<block>):

    tools/build_variant.sh fetl voice_activity_detection_amd/csrc -DVAD_FE_TIMELINE=2000
    VAD_AMD_LIB=$PWD/variants/libvad_fetl.so python tools/timeline_fe.py [hours]

Stamps of wave 0 of the chosen block per frame: 0 frame start (samples requested next), 1 samples arrived and
pre-emphasised, 2 pass A + W1024 twiddles + exchange stores, 3 exchange loads, 4 pass B + W64 twiddles, 5 pass C
(quad DPP), 6 the sixteen cross-lane reads issued, 7 untangle + power-spectrum stores, 8 mel runs, 9 frame done;
every fourth frame also 10 filter sums, 11 log + DCT tail, 12 row emit.  A stamp drains the wave's LDS queue, so
stages that overlap in the product are serialised here: the sum is an upper bound of a frame's time."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voice_activity_detection_amd as pkg

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
n = int(hours * 120000)
stream = pkg.synth_stream(0, n, seed=1337)
L = pkg.lib()
L.vad_debug_fe_timeline.restype = ctypes.c_int
buf = np.zeros(96 * 16, np.uint64)
feat = pkg.mfcc_delta_stream(stream, n, 166)
torch.cuda.synchronize()
L.vad_debug_fe_timeline(buf.ctypes.data_as(ctypes.c_void_p))           # discard the first launch, reset the frame counter
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
feat = pkg.mfcc_delta_stream(stream, n, 166)
e1.record()
torch.cuda.synchronize()
assert L.vad_debug_fe_timeline(buf.ctypes.data_as(ctypes.c_void_p)) == buf.size
tl = buf.reshape(96, 16).astype(np.int64)
print('instrumented kernel: %.2f ms per 100 h' % (e0.elapsed_time(e1) * 100 / hours))
names = ['load + pre-emphasis', 'pass A + tw1024 + ex stores', 'exchange loads', 'pass B + tw64', 'pass C (DPP)',
         'cross-lane reads issued', 'untangle + power stores', 'mel runs', 'frame end']
frames = tl[8:88]
d = np.diff(frames[:, :10], axis=1)
full = frames[1:, 0] - frames[:-1, 0]
print('ticks per frame (frame start to next frame start, median / mean): %.0f / %.0f' % (np.median(full), full.mean()))
for k, name in enumerate(names):
    print('   %-30s %8.1f   %5.1f %%' % (name, np.median(d[:, k]), 100 * np.median(d[:, k]) / np.median(full)))
tail = frames[frames[:, 12] > 0]
if len(tail):
    t = tail[:, 10:13] - tail[:, 9:12]
    for k, name in enumerate(['filter sums (last frame)', 'log + DCT tail', 'row emit']):
        print('   %-30s %8.1f   per 4 frames = %5.1f %% of a frame' % (name, np.median(t[:, k]), 100 * np.median(t[:, k]) / 4 / np.median(full)))

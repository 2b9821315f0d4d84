"""MI355X-native VAD inference hot path (MFCC front end + GRU / LSTM / DenseNet window models).

The compute path is libvad_amd.so (HIP, gfx950) behind the C ABI in
include/vad_amd.h; this package is the Python host side that mirrors the
reference's entry points.  There is no CPU fallback: without the library and a
GPU the calls raise ``VadError``.
"""
from ._lib import VadError, build, init, lib
from .common import (ALIGN_FRAMES, BATCH_SIZE, FEATURES, FRAME_SIZE, FRAMES, SAMPLE_RATE, SLICE_MAX, STEP_SIZE)
from .models import MODEL_STACK, DenseNet, Net, NickNet, set_matrix_mode
from .process_data import (add_noise, mfcc_delta_ragged, mfcc_delta_slices, mfcc_delta_stream, overlay, overlay_device,
                           process_test_data, process_test_frames, process_training_data, recording_slices,
                           training_slices)
from . import metrics
from . import train
from .generator import DataGenerator
from .metrics import fix_frr, roc_curve, threshold_sweep
from .stream import StreamWorkspace, vad_stream
from .sharding import HALO_FRAMES, all_gather_posteriors, gather_posteriors, shard_plan, shard_range, window_owner_ranges
from .synth import synth_audio, synth_stream

__all__ = ['VadError', 'build', 'init', 'lib', 'Net', 'DenseNet', 'NickNet', 'MODEL_STACK', 'set_matrix_mode', 'process_test_frames', 'process_test_data', 'recording_slices', 'add_noise',
           'overlay', 'overlay_device', 'mfcc_delta_slices', 'mfcc_delta_stream', 'mfcc_delta_ragged', 'process_training_data', 'training_slices', 'vad_stream', 'StreamWorkspace',
           'synth_audio', 'metrics', 'train', 'DataGenerator', 'fix_frr', 'roc_curve', 'threshold_sweep',
           'synth_stream', 'SAMPLE_RATE', 'FRAME_SIZE', 'FRAMES', 'FEATURES', 'BATCH_SIZE', 'SLICE_MAX',
           'ALIGN_FRAMES', 'STEP_SIZE']

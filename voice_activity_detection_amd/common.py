"""Constants of the hot path, mirroring the reference's module constants."""

# core/common.py:17-25
SAMPLE_RATE = 16000
SAMPLE_CHANNELS = 1
SAMPLE_WIDTH = 2
BATCH_SIZE = 2048
FRAMES = 30
FEATURES = 24
NOISE_LEVELS_DB = {'None': None, '-15': -15, '-3': -3}

# core/process_data.py:41-56
SEED = 1337
SLICE_MIN_MS = 1000
SLICE_MAX_MS = 5000
FRAME_SIZE_MS = 30
MFCC_WINDOW_FRAME_SIZE = 4
SLICE_MIN = int(SLICE_MIN_MS / FRAME_SIZE_MS)
SLICE_MAX = int(SLICE_MAX_MS / FRAME_SIZE_MS)
FRAME_SIZE = int(SAMPLE_RATE * (FRAME_SIZE_MS / 1000.0))
ALIGN_FRAMES = MFCC_WINDOW_FRAME_SIZE - 1

# core/train.py:54  (window step used when batching for training / test_predict)
STEP_SIZE = 6

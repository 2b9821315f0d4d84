"""``DataGenerator`` with the reference's interface (core/generator.py:13-165): the batch provider that
sits between the stored features and the models.

``get_batch`` reproduces the reference's sampling (the same numpy global-RNG calls in the same order, so
the same seed draws the same windows) and returns the materialised ``(x, y)`` batch as the reference
does.  ``get_batch_rows`` returns only the absolute window-start rows and labels; with
``Net.score_rows`` / ``DenseNet.score_rows`` a model scores those windows straight from a device-resident
feature matrix, which is what ``test_predict`` (core/train.py:394-445) needs without building batches.

``data`` is any mapping of array-likes that slice like the reference's HDF5 datasets: ``labels``,
``frames-<lvl>``, ``mfcc-<lvl>``, ``delta-<lvl>`` for the noise levels of core/common.py:25.
"""
import numpy as np

from .common import NOISE_LEVELS_DB

SEED = 1337


class DataGenerator:

    def __init__(self, data, size_limit=0):
        self.data = data
        self.size = size_limit if size_limit > 0 else len(data['labels'])
        self.data_mode = 0                                   # training data
        self.indices = np.arange(self.size)                  # drawn but unused, as in the reference; keeps
        np.random.seed(SEED)                                 # the global RNG stream identical
        np.random.shuffle(self.indices)

    def set_noise_level_db(self, level, reset_data_mode=True):
        if level not in NOISE_LEVELS_DB:
            raise Exception(f'Noise level "{level}" not supported! Options are: {list(NOISE_LEVELS_DB.keys())}')
        self.noise_level = level
        if reset_data_mode:
            (self.use_train_data, self.use_validate_data, self.use_test_data)[self.data_mode]()

    def setup_generation(self, frame_count, step_size, batch_size, val_part=0.1, test_part=0.1):
        self.frame_count = frame_count
        self.step_size = step_size
        self.batch_size = batch_size
        self.train_index = 0
        self.val_index = int(self.size * (1 - val_part - test_part))
        self.test_index = int(self.size * (1 - test_part))
        self.train_size = self.val_index
        self.val_size = self.test_index - self.val_index
        self.test_size = self.size - self.test_index
        assert self.size == self.train_size + self.val_size + self.test_size

    def _use(self, size, first, mode):
        n = int((size - self.frame_count) / self.step_size) + 1
        self.batch_count = int(n / self.batch_size)
        self.initial_pos = first
        self.data_mode = mode

    def use_train_data(self):
        self._use(self.train_size, self.train_index, 0)

    def use_validate_data(self):
        self._use(self.val_size, self.val_index, 1)

    def use_test_data(self):
        self._use(self.test_size, self.test_index, 2)

    def get_data(self, index_from, index_to):
        assert index_from >= 0 and index_to <= self.size
        try:
            return (self.data['frames-' + self.noise_level][index_from: index_to],
                    self.data['mfcc-' + self.noise_level][index_from: index_to],
                    self.data['delta-' + self.noise_level][index_from: index_to],
                    self.data['labels'][index_from: index_to])
        except KeyError:
            return self.data['frames'][index_from: index_to], None, None, self.data['labels'][index_from: index_to]

    def _sample(self, index, skip_single_class):
        """The draw of get_batch: position of the batch's data span, its labels, the sampled offsets."""
        pos = self.initial_pos + (self.batch_size * index) * self.step_size
        span = self.frame_count + self.step_size * self.batch_size
        assert pos >= 0 and pos + span <= self.size
        labels = np.asarray(self.data['labels'][pos: pos + span])
        class_0 = np.where(labels == 0)[0]
        class_1 = np.where(labels == 1)[0]
        if skip_single_class and (len(class_0) == 0 or len(class_1) == 0):
            print(f"Batch {index} - Skipping due to missing class")
            return pos, span, labels, None
        per_class = self.batch_size // 2
        sampled_0 = np.random.choice(class_0, per_class, replace=True)
        sampled_1 = np.random.choice(class_1, per_class, replace=True)
        balanced = np.hstack((sampled_0, sampled_1))
        np.random.shuffle(balanced)
        # offsets whose window would run past the span are dropped (the reference's shape check)
        return pos, span, labels, balanced[balanced + self.frame_count <= span]

    def get_batch(self, index, skip_single_class=False):
        pos, span, labels, offsets = self._sample(index, skip_single_class)
        if offsets is None:
            return [], []
        _, mfcc, delta, _ = self.get_data(pos, pos + span)
        x = [np.hstack((mfcc[i: i + self.frame_count], delta[i: i + self.frame_count])) for i in offsets]
        return np.array(x), np.array([labels[i] for i in offsets])

    def get_batch_rows(self, index, skip_single_class=False):
        """``(rows, y)``: absolute start rows (into the stored feature arrays) of the windows get_batch
        would return for this call, and their labels."""
        pos, _, labels, offsets = self._sample(index, skip_single_class)
        if offsets is None:
            return np.zeros(0, np.int64), np.zeros(0, labels.dtype)
        return (pos + offsets).astype(np.int64), labels[offsets]

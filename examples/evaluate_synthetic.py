"""End to end on synthetic audio, with the reference's own call sequence: speech + noise frames ->
``process_training_data`` (overlay at the three noise levels + MFCC/delta, the dataset layout the reference stores
in HDF5) -> the evaluation loops of core/train.py (``far``, ``roc_auc``, ``netvad``) on a randomly initialised model.

    python examples/evaluate_synthetic.py [model] [speech frames]        (needs the GPU; model: a MODEL_STACK key,
                                                                         gru_large or nicknet; default gru_large)

Labels are "the frame is not silent" in the synthetic stream, so the numbers say nothing about detection quality (the
model is untrained); the point is that every stage runs on the device kernels behind the reference's names.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_activity_detection_amd as vad                                            # noqa: E402


def make_model(name):
    if name in vad.MODEL_STACK:
        return vad.MODEL_STACK[name]['model']
    if name == 'nicknet':
        return vad.NickNet(large=True)
    return vad.Net(large=True, lstm=False)                                           # the GRU-30k of the benchmark


def synthetic_sets(n_speech, n_noise=20000):
    """(speech frames, labels, noise frames): int16 (n,480) arrays from the library's deterministic generator."""
    speech = vad.synth_audio(0, n_speech * vad.FRAME_SIZE, seed=1337).cpu().numpy().reshape(-1, vad.FRAME_SIZE)
    noise = vad.synth_audio(0, n_noise * vad.FRAME_SIZE, seed=7).cpu().numpy().reshape(-1, vad.FRAME_SIZE)
    noise = (noise // 8).astype(np.int16)
    labels = (np.abs(speech).max(axis=1) > 0).astype(np.int8)
    return speech, labels, noise


def main(model='gru_large', n_speech=70000):
    net = make_model(model)
    speech, labels, noise = synthetic_sets(n_speech)
    data = vad.process_training_data(speech, labels, noise)
    print('dataset rows: %d (written %d), noise levels: %s' % (len(data['labels']), data['count'], vad.train.NOISE_LEVELS))
    results = vad.train.far(net, data, frr=1, model_name=model)
    curves = vad.train.roc_auc({model: net}, data, '-3')
    print('AUC at -3 dB: %.3f' % curves[model][3])
    raw, lab, decisions = vad.train.netvad(net, data, noise_level='-3', init_pos=50, length=700)
    print('netvad: %d decisions, %.1f %% speech, agreement with the labels %.1f %%'
          % (len(decisions), 100 * decisions.mean(), 100 * (decisions == lab).mean()))
    return results, curves, (raw, lab, decisions)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gru_large', int(sys.argv[2]) if len(sys.argv) > 2 else 70000)

"""The evaluation-side callers (voice_activity_detection_amd/train.py: test_predict, far, roc_auc, netvad) on the
GPU against their CPU restatement (oracle/train_ref.py) run with the CPU model oracle on the same data and seed.

Tolerance: labels and window draws identical; scores within 1e-5 absolute; argmax decisions identical except where
the two class posteriors are within 2e-5 of each other; FAR / FRR exact against the reference's loop applied to
the same scores."""
import numpy as np
import pytest
import sklearn.metrics as skm

from oracle import metrics_ref, models_ref, train_ref

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    labels = (np.sin(np.arange(n) / 53.0) + 0.3 * rng.normal(size=n) > 0).astype(np.int64)
    data = {'labels': labels}
    for k, lvl in enumerate(('None', '-15', '-3')):
        data['frames-' + lvl] = np.arange(n, dtype=np.int64).reshape(n, 1) + 1000 * k
        sep = (labels[:, None] * 2.0 - 1.0) * (1.5 - 0.5 * k)          # the classes are separable, less so with noise
        data['mfcc-' + lvl] = (rng.normal(0, 8, size=(n, 12)) + sep).astype(np.float32)
        data['delta-' + lvl] = rng.normal(0, 2, size=(n, 12)).astype(np.float32)
    return data


def _nets(vad, kind):
    ref = models_ref.make_model(kind)
    if kind.startswith('densenet'):
        net = vad.DenseNet(large=kind.endswith('large'))
    else:
        net = vad.Net(large='large' in kind, lstm=kind.startswith('lstm'))
    net.load_state_dict(ref.state_dict())
    return net, ref


@pytest.mark.parametrize('kind', ['gru_large', 'densenet'])
def test_test_predict_draws_and_scores_like_the_reference_loop(vad, kind, capsys):
    data = _data(260000)
    net, ref = _nets(vad, kind)
    y_true, y_score = vad.train.test_predict(net, data, 0, '-15')
    r_true, r_score = train_ref.test_predict(ref, data, 0, '-15')
    assert 4000 < len(y_true) == len(r_true) <= 2 * 2048     # offsets whose window would pass the span's end are dropped
    np.testing.assert_array_equal(np.asarray(y_true), np.asarray(r_true))
    assert np.abs(np.asarray(y_score) - np.asarray(r_score)).max() <= ATOL
    assert 'y_true distribution' in capsys.readouterr().out
    # the reference's early exits, with its messages
    assert vad.train.test_predict(net, data, 0, '-7') is None
    assert 'invalid noise level' in capsys.readouterr().out
    assert vad.train.test_predict(net, {'labels': np.zeros(0)}, 0, 'None') is None
    assert 'no test data' in capsys.readouterr().out


def test_far_and_roc_auc(vad, capsys):
    data = _data(260000, seed=2)
    net, ref = _nets(vad, 'gru_large')
    got = vad.train.far(net, data, frr=1, model_name='gru')
    text = capsys.readouterr().out
    assert 'Network metrics: gru' in text and text.count('FAR: ') == 3
    want = train_ref.far(ref, data, frr=1)
    for lvl in ('None', '-15', '-3'):
        y_true, y_score = vad.train.test_predict(net, data, 0, lvl)
        # exact against the reference's threshold loop on the same scores
        assert got[lvl] == pytest.approx(metrics_ref.fix_frr(y_true, y_score, 1, lvl), abs=1e-12)
        # and the CPU model's scores (within 1e-5 of these) land on the same or a neighbouring threshold
        assert abs(got[lvl][0] - want[lvl][0]) <= 1.0 and abs(got[lvl][1] - want[lvl][1]) <= 0.2
    curves = vad.train.roc_auc({'a': net}, data, '-3')
    fpr, tpr, thr, area = curves['a']
    y_true, y_score = vad.train.test_predict(net, data, 0, '-3')
    sf, st, sthr = skm.roc_curve(y_true, y_score)
    np.testing.assert_allclose(fpr, sf, atol=1e-12)
    np.testing.assert_allclose(tpr, st, atol=1e-12)
    assert abs(area - skm.auc(sf, st)) <= 1e-12


@pytest.mark.parametrize('kind,init_pos,length', [('gru_large', 50, 700), ('lstm_small', 0, 2500), ('densenet', 7, 31)])
def test_netvad_decisions_per_frame(vad, kind, init_pos, length, capsys):
    data = _data(30000, seed=5)
    net, ref = _nets(vad, kind)
    raw, labels, out = vad.train.netvad(net, data, noise_level='-3', init_pos=init_pos, length=length)
    r_raw, r_labels, r_out, post = train_ref.netvad(ref, data, noise_level='-3', init_pos=init_pos, length=length)
    assert 'Network processed' in capsys.readouterr().out
    np.testing.assert_array_equal(raw, r_raw)
    np.testing.assert_array_equal(labels, r_labels)
    assert out.shape == r_out.shape == (length - 30,)
    clear = np.abs(post[:, 1] - post[:, 0]) > 2 * ATOL
    np.testing.assert_array_equal(out[clear], r_out[clear])
    assert clear.mean() > 0.99
    assert vad.train.netvad(net, data, noise_level='0') is None


@pytest.mark.parametrize('model', ['gru_large', 'densenet', 'nicknet'])
def test_end_to_end_example_runs(vad, model, capsys):
    """examples/evaluate_synthetic.py: synthetic audio -> process_training_data -> far / roc_auc / netvad, every
    stage on the device kernels; outputs are well formed (the stages themselves are checked against the oracle above
    and in the front-end / overlay / model tests)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'evaluate_synthetic.py')
    spec = importlib.util.spec_from_file_location('evaluate_synthetic', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    results, curves, (raw, lab, decisions) = mod.main(model, 70000)
    assert set(results) == {'None', '-15', '-3'}
    for far, frr in results.values():
        assert 0.0 <= far <= 100.0 and 0.0 <= frr <= 100.0
    fpr, tpr, thr, area = curves[model]
    assert fpr[0] == 0.0 and tpr[-1] == 1.0 and 0.0 <= area <= 1.0
    assert raw.shape == (670, 480) and lab.shape == decisions.shape == (670,)
    assert set(np.unique(decisions)) <= {0.0, 1.0}
    assert 'FAR: ' in capsys.readouterr().out

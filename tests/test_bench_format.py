"""The line bench.py prints for the driver: short enough to survive the driver's output tail, and carrying
every key of the benchmark contract.  Runs bench.py's formatter on a canned full record (a real run's)."""
import json
import os

import bench


def _sample(golden_dir):
    with open(os.path.join(golden_dir, 'bench_record_sample.json')) as f:
        out = json.load(f)
    out.setdefault('dtype_note', 'f32 operands split exactly into 3 bf16, 8 of 9 partial products on the bf16 MFMA, '
                                 'f32 accumulation')
    out.setdefault('dist_backend', None)
    out['detail'] = 'gpurun_out/bench_detail.json'
    return out


def _strings(x):
    if isinstance(x, str):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _strings(v)
    elif isinstance(x, list):
        for v in x:
            yield from _strings(v)


def test_compact_line_is_short_and_complete(golden_dir):
    out = _sample(golden_dir)
    assert len(json.dumps(out)) > 10000                       # the full record is what the driver could not keep
    line = bench.compact_line(out)
    assert len(line) < bench.COMPACT_LIMIT == 3000 and '\n' not in line
    c = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'matrix_mode', 'value_f32_mfma'):
        assert key in c, key
    assert set(c['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(c['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    assert 'workload' in c['config'] and 'model' not in c['config']
    assert abs(c['value'] - out['value']) <= 1e-4 * out['value']
    assert abs(c['roofline']['frac'] - out['roofline']['frac']) <= 1e-4
    assert c['value_f32_mfma'] == bench._r(out['matrix_modes']['f32']['value'])
    assert set(c['configs']) == set(out['configs'])
    assert all(len(s) < 120 for s in _strings({k: v for k, v in c.items() if k != 'cpu_baseline'}))


def test_compact_line_sheds_optional_parts_before_growing(golden_dir):
    out = _sample(golden_dir)
    for i in range(40):                                        # far more configurations than any run times
        out['configs']['extra_%d' % i] = out['configs']['gru_h90']
    line = bench.compact_line(out)
    assert len(line) < bench.COMPACT_LIMIT
    c = json.loads(line)
    assert 'configs' not in c and 'roofline' in c and 'cpu_baseline' in c and 'value' in c

"""CPU oracle for the VAD inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this
directory; it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / the timed CPU baseline.

* ``mfcc_ref``     float64 numpy/scipy restatement of the third-party
                   ``python_speech_features`` calls made at
                   core/process_data.py:272-277 (PARITY UNPINNED: the package
                   is not vendored in the reference and not installed here).
* ``models_ref``   torch.nn restatement of core/models.py ``Net``/``DenseNet``,
                   pinned to fixtures produced by importing the reference.
* ``pipeline_ref`` slice loop / window assembly / decisions glue
                   (core/process_data.py:171-255, core/train.py:539-626).
* ``synth_ref``    numpy twin of the counter-based synthetic audio generator.
"""

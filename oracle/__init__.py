"""CPU oracle for the audio hot path (log-mel -> encoder -> projector).

TEST INFRASTRUCTURE ONLY.  Nothing under ``asr_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the reported CPU baseline.

It is a CPU restatement (numpy / torch-CPU fp32) of the arithmetic the reference
runs on this path.  The arithmetic owner is the third-party ``transformers``
package (reference pins 4.56.1, uv.lock:3383-3384; this container has 5.15.0),
called from src/train.py:54-59,172-180 and src/modeling.py:33-40,77-84.

Pinning: the reference ships no tests or golden vectors.
The restatement is pinned against fixtures under ``tests/golden/`` that
``tests/golden/make_golden.py`` generated in the build container by running the
installed ``transformers`` 5.15.0 code (WhisperFeatureExtractor, WhisperEncoder)
and the reference's own ``AudioProjector`` class on deterministic inputs.
"""

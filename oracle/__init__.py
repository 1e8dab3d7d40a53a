"""CPU oracle for the image-in-speech watermark train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the
checker / timed CPU baseline, never as the thing measured or shipped.

Parity status: PINNED for the model path (UformerAudio / ModelA / HiDDeN
Decoder forward, losses, gradients, STFT conventions) by golden vectors
generated from the reference itself in the build container
(``tests/golden/make_golden.py``).  ``low_pass`` is pinned against
``scipy.signal.butter/filtfilt`` (the reference's own call,
uformerWM/audio_attack.py:27-29).  ``resampling`` (librosa, unpinned version,
not installed) and the AWGN random stream are **parity unpinned** — see
``oracle/attacks.py``.
"""

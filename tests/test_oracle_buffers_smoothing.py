"""Host-logic oracles: ring buffers (index arithmetic) and the EMA recurrence."""
import numpy as np

from koemorph_amd import synth
from oracle import buffers, smoothing


def test_ringbuffer_fifo_wrap_and_overflow():
    rb = buffers.RingBufferOracle(10)
    assert rb.read(1) is None
    rb.write(np.arange(7))
    assert np.array_equal(rb.read(4), [0, 1, 2, 3])
    rb.write(np.arange(7, 20))               # only 7 fit (3 left + 4 freed): overflow dropped
    assert rb.available == 10
    assert np.array_equal(rb.read(10), [4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
    assert rb.read(1) is None


def test_mel_audio_buffer_quirks():
    b = buffers.MelAudioBufferOracle()
    assert b.buffer_size == 136000 and b.hop_length == 532      # int(16000/(1/0.0333))
    assert not b.add_audio_frame(np.zeros(530, np.float32))     # more than +/-1 off: rejected
    x = synth.uniform(1, (300, 533))
    for i in range(255):
        assert b.add_audio_frame(x[i]) and b.get_current_audio() is None
    assert b.add_audio_frame(x[255]) and b.is_full              # 256*532 >= 136000
    w = b.get_current_audio()
    stream = np.concatenate([r[:532] for r in x[:256]])          # 533-sample frames truncated to 532
    assert np.array_equal(w, stream[-136000:])
    b.add_audio_frame(x[256][:531])                              # short frame zero-padded
    w2 = b.get_current_audio()
    assert np.array_equal(w2[:-532], stream[-136000 + 532:])
    assert np.array_equal(w2[-532:-1], x[256][:531]) and w2[-1] == 0.0


def test_ema_first_call_and_batch_change():
    sm = smoothing.TemporalSmootherOracle(0.8)
    a = synth.uniform(2, (5, 4, 52), 0, 1)
    y0 = sm(a[0])
    assert np.array_equal(y0, a[0])                              # first call: passthrough
    y1 = sm(a[1])
    alpha = 1.0 / (1.0 + np.exp(-0.8))
    np.testing.assert_allclose(y1, alpha * a[1] + (1 - alpha) * a[0], atol=1e-7)
    # output lies between previous state and current input (reference tests/model/test_decoder.py:203-205)
    assert np.all(y1 <= np.maximum(a[0], a[1]) + 1e-7) and np.all(y1 >= np.minimum(a[0], a[1]) - 1e-7)
    y2 = sm(a[2][:3])                                            # batch-size change resets
    assert np.array_equal(y2, a[2][:3])
    sm.reset()
    assert np.array_equal(sm(a[3]), a[3])

"""Host logic of the cross-mixture candidate batcher (acousticswarms_speech_amd/batching.py) with a CPU stand-in for
the spot model: requests of concurrent searches are merged into multi-mixture launches, every search gets exactly its
own slice, a full internal batch is launched without waiting for stragglers, and a search that dies does not leave
the others waiting.  (The real model runs it in tests/test_gpu_configs.py.)"""
import threading
import time

import numpy as np
import pytest
import torch

from acousticswarms_speech_amd.batching import CandidateBatcher


class _FakeModel(object):
    """energies / waveforms that depend on (mixture, offsets) only, so any mis-routing shows"""
    batch_size = 8
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def shift_and_sep_device_multi(self, mixes, offs, idx, strict, want_wave=True, want_energy=True, window=12000,
                                   circular=True):
        self.calls.append((len(offs), sorted(set(idx.tolist())), int(strict)))
        base = mixes[idx.long(), 0, 0].double() * 1000 + offs.double().abs().sum(1) + 10 * strict
        en = torch.stack([base, base + 0.5], dim=1)
        wave = (base[:, None] + torch.arange(mixes.shape[2], dtype=torch.float64)[None, :]).float() if want_wave else None
        return wave, en


def _expected(k, offs, strict):
    b = (k + 1) * 1000 + np.abs(offs).sum(1) + 10 * strict
    return np.stack([b, b + 0.5], axis=1)


def test_requests_are_merged_and_routed_back():
    K, T = 5, 16
    mixes = torch.zeros(K, 3, T)
    mixes[:, 0, 0] = torch.arange(1, K + 1)
    model = _FakeModel()
    batcher = CandidateBatcher(model, mixes, n_workers=K)
    out, errs = {}, []

    def work(k):
        try:
            sc = batcher.proxy(k)
            rng = np.random.default_rng(k)
            for rnd in range(4):
                offs = rng.integers(-9, 10, size=(2 + (k + rnd) % 3, 2)).astype(np.int32)
                strict = rnd % 2
                if rnd < 2:
                    class P:                                  # the search hands patches over
                        def __init__(self, o):
                            self.sample_offset = o
                    en = sc.shift_and_score(None, [P(o.astype(float)) for o in offs], Strict=strict)
                    np.testing.assert_array_equal(en, _expected(k, offs, strict))
                else:
                    wave, en = batcher.request(k, offs, strict, 12000, True)
                    np.testing.assert_array_equal(en.numpy(), _expected(k, offs, strict))
                    assert wave.shape == (len(offs), T) and abs(float(wave.mean())) < 1e-3      # mean-removed rows
                time.sleep(0.001 * (k % 3))
            out[k] = True
        except BaseException as e:
            errs.append(e)
        finally:
            batcher.worker_done()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    [t.start() for t in ts]
    [t.join(timeout=60) for t in ts]
    assert not errs, errs
    assert len(out) == K and not any(t.is_alive() for t in ts)
    n_requests = K * 4
    assert batcher.launches < n_requests                      # requests really were merged
    assert any(len(ks) > 1 for _n, ks, _s in model.calls)     # several mixtures in one launch
    assert batcher.candidates == sum(n for n, _k, _s in model.calls)


def test_full_batch_goes_out_without_waiting_for_a_straggler():
    mixes = torch.zeros(3, 2, 4)
    model = _FakeModel()
    batcher = CandidateBatcher(model, mixes, n_workers=3, target=6)
    got = {}

    def requester(k, n):
        _w, en = batcher.request(k, np.zeros((n, 1), dtype=np.int32), 1, 12000, False)
        got[k] = len(en)
        batcher.worker_done()

    a = threading.Thread(target=requester, args=(0, 3))
    b = threading.Thread(target=requester, args=(1, 3))
    a.start(); b.start()
    a.join(timeout=10); b.join(timeout=10)                    # 6 candidates = target: launched although worker 2 never asked
    assert got == {0: 3, 1: 3} and batcher.launches == 1
    batcher.worker_done()                                     # the straggler leaves


def test_a_dying_search_does_not_block_the_others():
    mixes = torch.zeros(2, 2, 4)
    batcher = CandidateBatcher(_FakeModel(), mixes, n_workers=2, target=100)
    res = {}

    def good():
        _w, en = batcher.request(0, np.zeros((2, 1), dtype=np.int32), 0, 12000, False)
        res["good"] = len(en)
        batcher.worker_done()

    def bad():
        try:
            raise ValueError("search failed")
        except ValueError:
            pass
        finally:
            batcher.worker_done()

    t1, t2 = threading.Thread(target=good), threading.Thread(target=bad)
    t1.start(); time.sleep(0.05); t2.start()
    t1.join(timeout=10); t2.join(timeout=10)
    assert res == {"good": 2}


def test_launch_error_reaches_every_waiting_search():
    class Broken(_FakeModel):
        def shift_and_sep_device_multi(self, *a, **k):
            raise RuntimeError("device fault")
    batcher = CandidateBatcher(Broken(), torch.zeros(2, 2, 4), n_workers=2, target=100)
    errs = []

    def work(k):
        try:
            batcher.request(k, np.zeros((1, 1), dtype=np.int32), 0, 12000, False)
        except RuntimeError as e:
            errs.append(str(e))
        finally:
            batcher.worker_done()
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join(timeout=10) for t in ts]
    assert len(errs) == 2 and all("device fault" in e for e in errs)


def test_queue_ahead_rule_launches_until_two_are_in_flight_then_merges():
    """the launch rule's device side, driven through the inflight hook: with fewer than max_inflight launches queued a
    lone, part-filled request goes out at once; with two queued the requests wait and are merged when one drains."""
    mixes = torch.zeros(3, 2, 4)
    mixes[:, 0, 0] = torch.arange(1, 4)
    model = _FakeModel()
    queued = {"n": 0}
    batcher = CandidateBatcher(model, mixes, n_workers=3, target=100, inflight=lambda: queued["n"], poll_s=1e-3)
    offs = np.ones((2, 1), dtype=np.int32)
    _w, en = batcher.request(0, offs, 0, 12000, False)          # workers 1, 2 never asked, the batch is far from full
    np.testing.assert_array_equal(en.numpy(), _expected(0, offs, 0))
    assert batcher.launches == 1 and batcher.awaiting == 0 and batcher.reasons == {"queue_ahead": 1}
    queued["n"] = 2
    got = {}

    def requester(k):
        _w, e = batcher.request(k, offs, 0, 12000, False)
        got[k] = e.numpy()

    ts = [threading.Thread(target=requester, args=(k,)) for k in (1, 2)]
    [t.start() for t in ts]
    time.sleep(0.05)
    assert all(t.is_alive() for t in ts) and batcher.launches == 1     # two launches queued on the device: they wait
    queued["n"] = 1                                             # one drained
    [t.join(timeout=10) for t in ts]
    assert not any(t.is_alive() for t in ts) and batcher.launches == 2  # ... and went out together
    assert model.calls[-1][0] == 4 and model.calls[-1][1] == [1, 2]
    for k in (1, 2):
        np.testing.assert_array_equal(got[k], _expected(k, offs, 0))

"""Cross-mixture candidate batching (BASELINE configs[3]: a batch of mixtures per GPU).

The reference fills its 128-wide network batches from ONE mixture at a time
(sep/training/JointModel/network.py:75-96); a search issues a 30-candidate coarse call and then
fine-stage chunks whose sizes follow the subdivision, so most internal batches run part empty and
the GPU idles during the SRP-PHAT and clustering host work of the mixture.  Here several searches
run concurrently -- one host thread per mixture, each with its own view of the (shared, read-only)
geometry tables -- and every scoring request they make goes through a ``CandidateBatcher``: the
requests of different mixtures are concatenated into one candidate stream with a per-candidate
mixture index and evaluated by ONE ``asw_spot_shift_and_sep_multi`` launch sequence, as soon as a
full internal batch is waiting or fewer than two launches are queued on the device
(``CandidateBatcher``'s launch rule).  Each search sees
its own slice of the result, so its decisions are those of the plain per-mixture loop.

Only the spot network's workspace is shared between the searches, and only the batcher touches
it (under its lock, on the batcher's own stream); everything else a search launches (SRP map,
SI-SDR matrices) allocates its own buffers and runs on the search's streams.
"""
import copy
import threading
import time

import numpy as np

from .spot import offsets_from_patches


class _Request(object):
    __slots__ = ("k", "offs", "key", "want_wave", "wave", "energy", "done", "error", "event")

    def __init__(self, k, offs, key, want_wave):
        self.k, self.offs, self.key, self.want_wave = k, offs, key, want_wave
        self.wave = self.energy = self.error = self.event = None
        self.done = False


class CandidateBatcher(object):
    """Merges the scoring requests of ``n_workers`` concurrent searches over ``mix_stack [K,M,T]``
    (cuda float32) into multi-mixture launches of ``model`` (a SpotModel on that device).

    Launch rule (``_pump``): the waiting requests of one kind go out together as soon as
      * they fill an internal batch (``target`` candidates), or
      * fewer than ``max_inflight`` (2) launches of this batcher are queued on the device: the device always has its
        next launch behind the running one, so it never waits for a host round trip (poll, enqueue: 1-2 ms per
        launch), and whatever arrives while two are queued accumulates and is merged.
    A search leaves ``request`` when its launch has been ENQUEUED and waits for the launch's end event outside the
    lock, so the host work it does next (subdivision, clustering, SRP-PHAT of its next mixture) overlaps the launches
    the other searches queued meanwhile.  (Two earlier rules, both measured on the 64-mixture run: "launch when a
    batch is full or every search is blocked" runs the searches in lock-step -- the device idles during every host
    phase; "launch when the device is idle, wait otherwise" never merges and leaves a bubble at every launch
    boundary, because half of the searches are always outside the batcher in their own small kernels and read-backs.)
    ``inflight`` replaces the event query (host tests); a CPU stand-in model computes inside ``_launch``, nothing is
    ever in flight there and the rule falls back to: merge until every live search is waiting."""

    def __init__(self, model, mix_stack, n_workers, target=None, inflight=None, max_inflight=2, poll_s=1e-3):
        self.model, self.mixes = model, mix_stack
        self.live = int(n_workers)
        self.target = int(target or model.batch_size)
        self.cv = threading.Condition()
        self.pending = []                                    # requests not launched yet (one per stopped search at most)
        self.awaiting = 0                                    # searches waiting for a launch in flight
        self.inflight = []                                   # end events of launches not yet seen complete
        self.max_inflight, self.poll_s = int(max_inflight), float(poll_s)
        self._inflight_hook = inflight
        # The launches run on a stream of their own.  On the searches' (default) stream every event a search records
        # and every small read-back it waits for would queue behind whole launches of the OTHER searches (40-100 ms
        # each): the searches then spend most of their time blocked outside the batcher, are never "all stopped",
        # and nothing gets merged or queued ahead (measured: 97 ms per request outside the batcher against 4 ms in
        # the plain loop).
        self.stream = None
        if mix_stack.device.type == "cuda":
            import torch
            self.stream = torch.cuda.Stream(device=mix_stack.device)
        self.requests = 0
        self.reasons = {}                                    # launches by the rule that sent them (stats)
        self.launches = 0
        self.candidates = 0
        self.sizes = []                                      # candidates per launch
        self.host_s = 0.0                                    # host time spent enqueueing the launches
        self.events = []                                     # (start, end) device events per launch (stats)
        self.wait_pending_s = 0.0                            # summed over searches: waiting for a launch to go out,
        self.wait_device_s = 0.0                             # ... and for its end event

    def proxy(self, k):
        return MixtureScorer(self, int(k))

    # ---- called by the searches -------------------------------------------------------------
    def request(self, k, offs, strict, window, want_wave):
        req = _Request(k, offs, (int(strict), int(window)), bool(want_wave))
        t0 = time.perf_counter()
        with self.cv:
            self.pending.append(req)
            self.requests += 1
            self._pump()
            while not req.done:
                self.cv.wait(self.poll_s)                     # a launch draining on the device changes the rule's answer
                if not req.done:
                    self._pump()
        t1 = time.perf_counter()
        try:
            if req.event is not None:
                req.event.synchronize()                      # outside the lock: the others keep queueing
                if self.stream is not None:
                    import torch
                    cur = torch.cuda.current_stream(self.mixes.device)
                    for t in (req.wave, req.energy):         # allocated on the launch stream, consumed on the search's
                        if t is not None:
                            t.record_stream(cur)
        finally:
            with self.cv:
                self.awaiting -= 1
                self.wait_pending_s += t1 - t0
                self.wait_device_s += time.perf_counter() - t1
                self._pump()
        if req.error is not None:
            raise RuntimeError(f"batched scoring failed: {req.error}")
        return req.wave, req.energy

    def worker_done(self):
        """A search has finished (or died): the others must not wait for it."""
        with self.cv:
            self.live -= 1
            self._pump()

    # ---- internals (lock held) -----------------------------------------------------------
    def _inflight(self):
        """launches of this batcher the device has not finished yet; None when that cannot be asked (CPU stand-in)"""
        if self._inflight_hook is not None:
            return int(self._inflight_hook())
        if self.mixes.device.type != "cuda":
            return None
        self.inflight = [e for e in self.inflight if not e.query()]
        return len(self.inflight)

    def _pump(self):
        while self.pending:
            key = self.pending[0].key                       # oldest request decides which kind goes first
            group = [r for r in self.pending if r.key == key]
            total = sum(len(r.offs) for r in group)
            why = "full"
            if total < self.target:
                n = self._inflight()
                if n is None:
                    if len(self.pending) < self.live:
                        return                              # somebody is still on the host: wait for more candidates
                    why = "all_waiting"
                elif n >= self.max_inflight:
                    return                                  # enough queued ahead; the poll in request() comes back
                else:
                    why = "queue_ahead"
            self.reasons[why] = self.reasons.get(why, 0) + 1
            self.pending = [r for r in self.pending if r.key != key]
            self._launch(group, key)
            self.cv.notify_all()

    def _launch(self, group, key):
        import contextlib
        import torch
        self.awaiting += len(group)
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()
        try:
            with ctx:
                self._launch_on_stream(group, key)
        except Exception as exc:                             # every waiting search gets the error
            for r in group:
                r.error = f"{type(exc).__name__}: {exc}"
        for r in group:
            r.done = True

    def _launch_on_stream(self, group, key):
        import torch
        from . import native
        dev = self.mixes.device
        offs = np.ascontiguousarray(np.concatenate([r.offs for r in group], axis=0))
        idx = np.concatenate([np.full(len(r.offs), r.k, dtype=np.int32) for r in group])
        assert idx.min() >= 0 and idx.max() < self.mixes.shape[0]
        on_gpu = dev.type == "cuda"                          # (a CPU stand-in model drives this class in the host tests)
        off_d, idx_d = torch.from_numpy(offs), torch.from_numpy(idx)
        if on_gpu:
            off_d = off_d.pin_memory().to(dev, non_blocking=True)
            idx_d = idx_d.pin_memory().to(dev, non_blocking=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        want_wave = any(r.want_wave for r in group)
        t_host = time.perf_counter()
        wave, en = self.model.shift_and_sep_device_multi(self.mixes, off_d, idx_d, key[0], want_wave=want_wave,
                                                         want_energy=True, window=key[1])
        self.host_s += time.perf_counter() - t_host
        self.sizes.append(len(offs))
        if want_wave and on_gpu:
            native.torch_ops().center_rows_(wave)           # the stage loops compare mean-removed outputs (Mic_Array.py:291)
        elif want_wave:
            wave -= wave.mean(dim=1, keepdim=True)
        if on_gpu:
            e1.record()
            self.events.append((e0, e1))
            self.inflight.append(e1)
        pos = 0
        for r in group:
            n = len(r.offs)
            r.wave = wave[pos:pos + n] if r.want_wave else None
            r.energy = en[pos:pos + n]
            r.event = e1 if on_gpu else None
            pos += n
        self.launches += 1
        self.candidates += pos


class MixtureScorer(object):
    """The spot-model surface one search uses (coarse ``shift_and_score``, fine ``shift_and_sep_resident``,
    the device SI-SDR helpers), bound to mixture ``k`` of a batcher.  The mixture argument of the calls is
    ignored: the candidates are scored against ``batcher.mixes[k]``."""

    def __init__(self, batcher, k):
        self.batcher, self.k = batcher, k
        self.inner_model = batcher.model
        self.device = batcher.model.device
        self.batch_size = batcher.model.batch_size

    def _offsets(self, patch_list):
        return offsets_from_patches(patch_list, self.batcher.mixes.shape[1] - 1)

    def shift_and_score(self, input_channels, patch_list, Strict=0, window=12000, keep_waveforms=False):
        if len(patch_list) == 0:
            return np.empty((0, 2), dtype=np.float64)
        _w, en = self.batcher.request(self.k, self._offsets(patch_list), Strict, window, False)
        return en.cpu().numpy()

    def shift_and_sep_resident(self, input_channels, patch_list, Strict=0, window=12000, device_energies=False):
        wave, en = self.batcher.request(self.k, self._offsets(patch_list), Strict, window, True)
        return wave, (en if device_energies else en.cpu().numpy())

    def shift_and_sep(self, input_channels, patch_list, Strict=0, save_input=False):
        raise RuntimeError("a batched search scores through shift_and_score / shift_and_sep_resident only")

    def pair_sisdr(self, waves):
        return self.inner_model.pair_sisdr(waves)

    def segment_sisdr(self, waves, segments):
        return self.inner_model.segment_sisdr(waves, segments)


def mixture_view(mic_array):
    """A per-search view of a MicArray: the geometry tables (tens of MB, read-only) are shared, everything a
    search writes -- the SRP map and its voxel image, counters, caches, the decision trace, the side stream --
    is the view's own."""
    v = copy.copy(mic_array)
    v.SRP_node = copy.copy(mic_array.SRP_node)
    v.SRP_node.POWER_MAP = mic_array.SRP_node.POWER_MAP.copy()
    v._side_stream = None
    v._seg_cache, v._dev_cache = {}, {}
    v.trace = {"coarse_kept": [], "fine_clusters": {}, "final_clusters": []}
    return v


def search_batched(joint_model, mixes, concurrent=2):
    """The complete localization search (SRP-PHAT -> coarse -> fine -> clustering) of every mixture in
    ``mixes`` on this rank's GPU: ``concurrent`` worker threads pull mixtures from a queue -- a finished
    search is replaced at once, so the GPU never waits for a group to drain -- and score through one
    CandidateBatcher.  Returns the per-mixture result dicts of shard.localize_batch, in order, and a stats dict."""
    import torch
    spot = joint_model.spot_model
    mp = joint_model.Mic_processor
    assert mp is not None, "call JointModel.setup() first"
    dev = spot.device
    n = len(mixes)
    results = [None] * n
    hosts = [torch.as_tensor(m).to(dtype=torch.float32).cpu() for m in mixes]
    stack = torch.stack(hosts).to(dev).contiguous()          # [K,M,T]: 1.3 MB per 7-mic, 48 000-sample mixture
    torch.cuda.synchronize(dev)                              # the launch stream reads it
    n_workers = max(1, min(int(concurrent), n))
    batcher = CandidateBatcher(spot, stack, n_workers)
    errors, todo, todo_lock = [], list(range(n)), threading.Lock()

    def search(k):
        view, scorer = mixture_view(mp), batcher.proxy(k)
        times = [0.0] * 5
        t0 = time.time()
        patch_list, _ = view.Apply_SRP_PHAT(hosts[k])
        times[0] = time.time() - t0
        patches, spot_times = [], 0
        if len(patch_list) > 0:
            t0 = time.time()
            kept = view.Spotform_Big_Patch(stack[k], patch_list, scorer)
            times[1] = time.time() - t0
            if len(kept) > 0:
                t0 = time.time()
                pairs = view.Spotform_Small_Patch_Parallel(stack[k], kept, scorer)
                times[2] = time.time() - t0
                if len(pairs) > 0:
                    t0 = time.time()
                    _audio, patches, spot_times, _ = view.Clustering_new(pairs)
                    times[3] = time.time() - t0
        if len(patches) == 0:
            patches, spot_times = [], 0                    # the reference's empty-result early returns (:151-199)
        results[k] = {"centres": np.array([p[0].center_pos() for p in patches]).reshape(-1, 3),
                      "powers": np.array([p[2] for p in patches]), "names": [p[3] for p in patches],
                      "spot_times": spot_times, "times": times}

    worker_s = []

    def work():
        t_begin = time.perf_counter()
        try:
            while not errors:
                with todo_lock:
                    if not todo:
                        break
                    k = todo.pop(0)
                search(k)
        except BaseException as exc:                           # reported after the join; the others finish their mixture
            errors.append(exc)
        finally:
            batcher.worker_done()
            worker_s.append(time.perf_counter() - t_begin)

    # A search alternates short bursts of numpy / Python with blocking device calls (event waits, read-backs); after
    # each of those it needs the interpreter lock back, and with the default 5 ms switch interval it queues behind
    # every other search that is in a burst: about 90 ms per request went there (10 blocking calls x 2-3 others x 5 ms),
    # against 4 ms of host time per request in the single-threaded loop.
    import sys
    switch = sys.getswitchinterval()
    sys.setswitchinterval(min(switch, 2e-4))
    threads = [threading.Thread(target=work, name=f"asw-search-{w}") for w in range(n_workers)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        sys.setswitchinterval(switch)
    if errors:
        raise errors[0]
    torch.cuda.synchronize(dev)
    stats = {"launches": batcher.launches, "requests": batcher.requests, "launch_rule": dict(batcher.reasons), "candidates": batcher.candidates, "launch_sizes": list(batcher.sizes),
             "spot_gpu_s": sum(a.elapsed_time(b) for a, b in batcher.events) * 1e-3, "enqueue_host_s": batcher.host_s,
             "wait_pending_s": batcher.wait_pending_s, "wait_device_s": batcher.wait_device_s,
             "worker_s": sum(worker_s), "workers": n_workers}
    return results, stats

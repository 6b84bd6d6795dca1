"""Pipeline driver with the call surface of the reference ``JointModel``
(sep/training/JointModel/network.py:106-215): ``setup`` / ``forward`` /
``localize_by_separation`` / ``separate_by_localization`` and the five stage timers
``times[0..4]`` = [SRP, coarse, fine, clustering, joint separation] seconds.

Unlike the reference's timers (host ``time.time()`` with no device synchronisation,
:143-148) each stage boundary here synchronises the device, so the numbers are true
stage latencies.  ``sep_model`` is the joint separation network (``sep.SepModel``, the HIP
implementation of sep/training/SpeakerSeparation/network.py) or any object with
``infer(mix, patches)``; with ``sep_model=None`` the separation stage is skipped, ``audio`` is
None and ``times[4]`` stays 0 (callers must then report a localize-only latency).
"""
import time

import numpy as np

from .mic_array import MicArray


def _sync():
    """Stage boundary: every stream of the device drained, then one event round trip on the current stream.
    The round trip is not decoration.  After hipDeviceSynchronize / hipStreamSynchronize alone the ROCm 7.2
    runtime intermittently (about every second forward) starts the FIRST command of the next stage 20-25 ms
    late: the whole stage is enqueued within a few ms but nothing executes until that long into the stage's
    final blocking copy (rocprofv3 --hip-trace --kernel-trace; the separation stage read 23 or 45-60 ms,
    tests/micro/e2e_stats.py).  With a hipEventRecord + hipEventSynchronize after the drain the next
    submission is dispatched at once: 12 of 12 forwards at 23 ms.  (Event record without the wait, stream
    synchronize, or no synchronisation at all: the late start stays.)"""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
    except ImportError:          # pragma: no cover
        pass


class JointModel(object):
    def __init__(self, spot_model, sep_model=None, device=None):
        self.spot_model = spot_model
        self.sep_model = sep_model
        self.device = device
        self.times = [0, 0, 0, 0, 0]
        self.previous_config = None
        self.Mic_processor = None
        self._mix_dev = self._mix_src = None

    def setup(self, mic_positions, speaker_range, cached=False, cached_folder=None):
        """(Re)build the geometry tables unless the configuration is unchanged (:125-137).
        One-off per geometry and excluded from latency, as the reference's README notes."""
        key = '~'.join([f"{x:.05f}" for x in np.asarray(mic_positions).flatten()]) \
            + '|' + '~'.join([f"{x:.05f}" for x in speaker_range])
        if key == self.previous_config:
            print("reuse the previous recycle!")
            return
        import gc
        gc.unfreeze()                       # a previous geometry may go now
        self.Mic_processor = MicArray(mic_positions, Spk_Range=speaker_range, device=self.device)
        self.previous_config = key
        # The geometry tables are tens of thousands of small arrays and lists that live as long as
        # this configuration.  Left in the collector's oldest generation they make every full
        # collection of the interpreter a 50 ms pause that lands at random inside a forward()
        # (measured: 22-66 ms on the first statement after the search).  Move them to the permanent
        # generation: later collections only look at what a forward() itself allocates.
        gc.collect()
        gc.freeze()

    def forward(self, mix_data):
        """-> (patches, audio_loc, audio, SRP_drop, stage1_drop, spot_times) (:142-149)."""
        self.times = [0, 0, 0, 0, 0]
        patches, audio_loc, SRP_drop, stage1_drop, spot_times = self.localize_by_separation(mix_data)
        _sync()
        t0 = time.time()
        audio = self.separate_by_localization(mix_data, patches)
        _sync()
        self.times[4] = time.time() - t0
        self._mix_dev = self._mix_src = None              # the resident copy does not outlive the forward
        return patches, audio_loc, audio, SRP_drop, stage1_drop, spot_times

    __call__ = forward

    def _timed(self, slot, fn, *args):
        _sync()
        t0 = time.time()
        out = fn(*args)
        _sync()
        self.times[slot] = time.time() - t0
        return out

    def localize_by_separation(self, mix_data):
        """The four search stages with the reference's empty-result early returns (:151-199)."""
        assert self.previous_config is not None, \
            "Microphone positions and spk range were not provided, did you forget to call .setup()?"
        mp = self.Mic_processor
        # One upload of the mixture per forward, inside the first timed stage: the HIP models take
        # the resident copy (a host tensor would be copied again by every stage -- and a pageable
        # H2D right after the fine stage was measured at 29 ms for these 1.3 MB).  Any other
        # duck-typed model keeps receiving the caller's tensor.
        self._mix_dev = self._mix_src = None

        def srp_stage(m):
            self._mix_dev = self._resident(m)
            self._mix_src = m                             # the resident copy stands for THIS object only
            return mp.Apply_SRP_PHAT(m)
        patch_list, _ = self._timed(0, srp_stage, mix_data)
        if self._takes_resident(self.spot_model) and self._mix_dev is not None:
            mix_data = self._mix_dev
        if len(patch_list) <= 0:
            print("No spk picked in SRP-PHAT")
            return [], [], 0, 0, 0
        patch_list = self._timed(1, mp.Spotform_Big_Patch, mix_data, patch_list, self.spot_model)
        if len(patch_list) <= 0:
            print("No spk picked in Spotform_Big_Patch")
            return [], [], 0, 0, 0
        output_pair = self._timed(2, mp.Spotform_Small_Patch_Parallel, mix_data, patch_list, self.spot_model)
        if len(output_pair) <= 0:
            print("No spk picked in Spotform_Small_Patch")
            return [], [], 0, 0, 0
        audio_final, patch_final, spot_times, _ = self._timed(3, mp.Clustering_new, output_pair)
        if len(patch_final) <= 0:
            print("No spk picked in Clustering")
            return [], [], 0, 0, 0
        return patch_final, np.array(audio_final), 0, 0, spot_times

    @staticmethod
    def _takes_resident(model):
        """HIP models of this package (SpotModel / SepModel) accept a device-resident mixture."""
        dev = getattr(model, "device", None)
        return getattr(dev, "type", None) == "cuda" and (hasattr(model, "shift_and_sep_device") or hasattr(model, "infer_device"))

    def _resident(self, mix_data):
        for m in (self.spot_model, self.sep_model):
            if m is not None and self._takes_resident(m):
                import torch
                return torch.as_tensor(mix_data).to(m.device, dtype=torch.float32).contiguous()
        return None

    def separate_by_localization(self, mix_data, target_patches):
        if len(target_patches) == 0 or self.sep_model is None:
            return None
        # the resident copy uploaded by localize_by_separation() is reused only for the very object it was made
        # from (or for itself): a different mixture of the same shape must not be swapped for the cached one
        cached = getattr(self, "_mix_dev", None)
        if self._takes_resident(self.sep_model) and cached is not None \
                and (mix_data is getattr(self, "_mix_src", None) or mix_data is cached):
            mix_data = cached
        return self.sep_model.infer(mix_data, [p[0] for p in target_patches])

    def to(self, device=None):
        if device is not None:
            self.device = device
            if hasattr(self.spot_model, "to"):
                self.spot_model.to(device)
            if self.sep_model is not None and hasattr(self.sep_model, "to"):
                self.sep_model.to(device)
        return self

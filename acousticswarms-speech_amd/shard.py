"""Candidate sharding across the GPUs of one node.

Candidates of one stage share only the read-only mixture and the weights, so they are
partitioned with no data-path collective; the single exchange step of a stage is an
all-gather of the per-candidate energies ([N,2] float64, a few KB) before the host-side
thresholding / clustering.  One process per GPU, ``torch.distributed`` (backend "nccl"
is RCCL over xGMI; "gloo" on CPU for tests).  This replaces the per-call replicate /
scatter / gather of ``nn.DataParallel`` (sep/training/JointModel/network.py:30,93):
weights are resident per rank, only offsets and energies move.
"""
from typing import Callable, List, Sequence

import numpy as np


def shard_bounds(n_items: int, world: int) -> List[int]:
    """Balanced contiguous partition: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(n_items, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


def shard_groups(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Partition whole groups (all fine candidates of one surviving coarse patch stay on
    one rank so its SI-SDR clustering is local, sep/Mic_Array.py:339-383) with a
    longest-processing-time greedy balance.  Returns group indices per rank."""
    order = sorted(range(len(sizes)), key=lambda i: -sizes[i])
    load = [0] * world
    owner = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[r].append(i)
        load[r] += sizes[i]
    return [sorted(o) for o in owner]


def offsets_fingerprint(patches_or_offsets) -> int:
    """CRC32 of the rounded int32 offsets of a candidate list (ndarray [N,P] or patches)."""
    import zlib
    if len(patches_or_offsets) == 0:
        return 0
    offs = np.stack([np.asarray(getattr(p, "sample_offset", p), dtype=np.float64) for p in patches_or_offsets])
    return zlib.crc32(np.ascontiguousarray(np.rint(offs.astype(np.float32)).astype(np.int32)).tobytes())


def assert_same_on_all_ranks(dist, group, device, what: str, *values: int):
    """Every rank derives shard widths and group ownership locally from its own (deterministic)
    host search.  If the ranks ever disagreed -- another candidate count, other group sizes,
    other offsets -- the fixed-shape exchange that follows would hang or silently mis-assemble
    the energies.  So each exchange is preceded by an all-gather of this small fingerprint and
    a mismatch raises on every rank."""
    import torch
    world = dist.get_world_size(group)
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    box = torch.empty((world * len(values),), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(box, mine, group=group)
    box = box.view(world, len(values)).cpu()
    if not bool((box == box[0:1]).all()):
        raise RuntimeError(f"ranks disagree before the {what} exchange (per-rank fingerprints {box.tolist()}): "
                           "the candidate lists are not identical on every rank")


class ShardedScorer:
    """score(mix, offsets[N,P]) -> energies [N,2] on every rank.

    ``local_score(mix, offsets_local) -> ndarray|tensor [n_local,2] float64`` is the
    single-GPU scorer (SpotModel.shift_and_score bound to a window)."""

    def __init__(self, local_score: Callable, group=None):
        import torch.distributed as dist
        self.local_score = local_score
        self.group = group
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1

    def score(self, mix, offsets: np.ndarray, device=None) -> np.ndarray:
        import torch
        N = len(offsets)                                     # ndarray [N,P] or a list of patches
        b = shard_bounds(N, self.world)
        lo, hi = b[self.rank], b[self.rank + 1]
        local = self.local_score(mix, offsets[lo:hi])
        local = torch.as_tensor(local, dtype=torch.float64)
        if self.world == 1:
            return local.cpu().numpy().reshape(N, 2)
        width = max(b[r + 1] - b[r] for r in range(self.world))     # pad to equal shards
        dev = device if device is not None else local.device
        assert_same_on_all_ranks(self.dist, self.group, dev, "energy", N, offsets_fingerprint(offsets))
        buf = torch.zeros((width, 2), dtype=torch.float64, device=dev)
        buf[:hi - lo] = local.to(dev)
        out = torch.empty((self.world * width, 2), dtype=torch.float64, device=dev)
        self.dist.all_gather_into_tensor(out, buf, group=self.group)   # the stage's one exchange
        out = out.view(self.world, width, 2).cpu().numpy()
        return np.concatenate([out[r, :b[r + 1] - b[r]] for r in range(self.world)], axis=0)


class ShardedSpotModel:
    """Duck-typed spot model for ``MicArray`` / ``JointModel`` with one rank per GPU.

    Every rank runs the same (deterministic) host search on the same mixture; the candidate
    evaluations are what is sharded:
      * coarse stage: ``shift_and_score`` -> contiguous shard + all-gather of [N,2] energies;
      * fine stage:   ``MicArray.Spotform_Small_Patch_Parallel`` asks ``my_groups`` for the
        coarse patches this rank owns (LPT-balanced whole groups), evaluates only those through
        the wrapped model, all-gathers the energies (``all_gather_groups``) and the finished
        output tuples (``gather_pairs``, object gather of a few cluster heads).
    Replaces nn.DataParallel's per-call replicate/scatter/gather
    (sep/training/JointModel/network.py:30,93).  ``inner`` is a SpotModel (HIP) or any object
    with the reference's ``shift_and_sep`` surface (used by the gloo CPU tests)."""

    def __init__(self, inner, group=None, device=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("ShardedSpotModel needs an initialised torch.distributed process group")
        self.inner, self.group, self.dist = inner, group, dist
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
        self._scorer = ShardedScorer(self._local_score, group)
        self._strict = 0
        # the device-resident fine stage is offered only when the wrapped model has it
        if hasattr(inner, "shift_and_sep_resident"):
            self.shift_and_sep_resident = inner.shift_and_sep_resident
            self.pair_sisdr = inner.pair_sisdr

    def to(self, device):
        if hasattr(self.inner, "to"):
            self.inner.to(device)
        return self

    # ---- candidate-sharded scoring -------------------------------------------------------
    def _local_score(self, mix, patches):
        if len(patches) == 0:
            return np.zeros((0, 2))
        if hasattr(self.inner, "shift_and_score"):
            return self.inner.shift_and_score(mix, patches, Strict=self._strict, keep_waveforms=False)
        from .hostdsp import max_avg_power
        sep = self.inner.shift_and_sep(mix, patches, Strict=self._strict)
        out = np.zeros((sep.shape[0], 2))
        for i in range(sep.shape[0]):
            x = sep[i, :] - np.mean(sep[i, :])
            out[i] = (np.sum(x ** 2), max_avg_power(x))
        return out

    def shift_and_score(self, mix, patch_list, Strict=0, keep_waveforms=False):
        if keep_waveforms:
            raise RuntimeError("sharded scoring returns energies only")
        self._strict = Strict

        return self._scorer.score(mix, list(patch_list), device=self.device)

    def shift_and_sep(self, mix, patch_list, Strict=0, save_input=False):
        """Reference surface on the LOCAL candidates handed in (the fine stage passes only
        this rank's groups)."""
        return self.inner.shift_and_sep(mix, patch_list, Strict=Strict)

    # ---- fine-stage group sharding -------------------------------------------------------
    def my_groups(self, sizes):
        return shard_groups(sizes, self.world)[self.rank]

    def deal_groups(self, weights):
        """Owners of whole coarse patches from a weight every rank can compute WITHOUT subdividing them
        (the number of 1 cm grid points of a patch, which is what drives the size of its subdivision):
        the same longest-processing-time deal as ``my_groups``.  Returns the group indices per rank."""
        return shard_groups([int(w) for w in weights], self.world)

    def gather_sizes(self, local_sizes: dict, n_groups: int):
        """{coarse patch: number of fine candidates} of the patches each rank subdivided -> the full list."""
        box = [None] * self.world
        self.dist.all_gather_object(box, dict(local_sizes), group=self.group)
        sizes = [None] * n_groups
        for part in box:
            for g, n in part.items():
                sizes[int(g)] = int(n)
        if any(v is None for v in sizes):
            raise RuntimeError("fine-stage sizes: some coarse patch was subdivided by no rank")
        return sizes

    def all_gather_groups(self, local_energies, mine, bounds, owners=None):
        """All-gather of the fine-stage energies: local rows are this rank's groups in
        ``mine`` order; returns the full [N,2] table in the global candidate order.  ``owners``:
        the deal that was used (default: the size-balanced one of ``my_groups``)."""
        import torch
        sizes = [bounds[i + 1] - bounds[i] for i in range(len(bounds) - 1)]
        if owners is None:
            owners = shard_groups(sizes, self.world)
        width = max(1, max(sum(sizes[g] for g in o) for o in owners))
        import zlib
        assert_same_on_all_ranks(self.dist, self.group, self.device, "fine-stage energy", len(sizes), int(bounds[-1]),
                                 zlib.crc32(np.asarray(sizes, dtype=np.int64).tobytes()))
        buf = torch.zeros((width, 2), dtype=torch.float64, device=self.device)
        loc = torch.as_tensor(np.asarray(local_energies, dtype=np.float64).reshape(-1, 2))
        buf[:loc.shape[0]] = loc.to(self.device)
        out = torch.empty((self.world * width, 2), dtype=torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(out, buf, group=self.group)
        out = out.view(self.world, width, 2).cpu().numpy()
        full = np.zeros((bounds[-1], 2))
        for r, o in enumerate(owners):
            pos = 0
            for g in o:
                full[bounds[g]:bounds[g + 1]] = out[r, pos:pos + sizes[g]]
                pos += sizes[g]
        return full

    def gather_pairs(self, local_pairs):
        """Object all-gather of the finished (centre, audio, power, "g_head", ...) tuples,
        merged back into coarse-patch order (the order the single-GPU loop emits)."""
        box = [None] * self.world
        self.dist.all_gather_object(box, local_pairs, group=self.group)
        merged = [p for part in box for p in part]
        key = lambda p: int(p[3].split("_")[0])              # stable: heads keep their per-group order
        return sorted(merged, key=key)


def localize_batch(joint_model, mixes, group=None, concurrent=2):
    """A batch of mixtures over the ranks of one node (BASELINE config "batch of 64 mixtures"):
    with at least as many mixtures as ranks the cheapest partition is by whole mixture --
    contiguous balanced blocks, each rank runs the complete search of its mixtures on its own
    GPU with no per-candidate traffic -- followed by ONE object all-gather of the per-mixture
    results.  (Fewer mixtures than ranks: wrap the spot model in ``ShardedSpotModel`` instead,
    which shards the candidates of a single mixture.)

    On a rank, the searches of its mixtures run ``concurrent`` at a time and share their network
    launches (batching.search_batched: one candidate stream with a per-candidate mixture index, so the
    internal batches stay full and one search's host stages overlap the others' GPU work) when the spot
    model is the HIP model; ``concurrent=1`` or any other duck-typed model gives the plain loop.

    ``joint_model.spot_model`` must be the plain per-rank model here.  Returns, on every rank and
    in mixture order, a list of dicts {centres [K,3], powers [K], names, spot_times, times[5]}."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    if getattr(joint_model.spot_model, "world", 1) > 1:
        raise RuntimeError("localize_batch shards by mixture: pass the un-sharded per-rank spot model")
    b = shard_bounds(len(mixes), world)
    mine = list(range(b[rank], b[rank + 1]))
    spot = joint_model.spot_model
    batched = (concurrent > 1 and len(mine) > 1 and hasattr(spot, "shift_and_sep_device_multi")
               and getattr(getattr(spot, "device", None), "type", None) == "cuda")
    local = []
    if batched:
        from .batching import search_batched
        res, stats = search_batched(joint_model, [mixes[k] for k in mine], concurrent=concurrent)
        localize_batch.last_stats = stats                    # diagnostic: launches, candidates, GPU seconds inside them
        local = list(zip(mine, res))
    else:
        for k in mine:
            patches, _audio_loc, _audio, _d0, _d1, spot_times = joint_model.forward(mixes[k])
            local.append((k, {"centres": np.array([p[0].center_pos() for p in patches]).reshape(-1, 3),
                              "powers": np.array([p[2] for p in patches]),
                              "names": [p[3] for p in patches],
                              "spot_times": spot_times, "times": list(joint_model.times)}))
    if world == 1:
        return [r for _k, r in local]
    box = [None] * world
    dist.all_gather_object(box, local, group=group)
    merged = sorted((kr for part in box for kr in part), key=lambda kr: kr[0])
    return [r for _k, r in merged]

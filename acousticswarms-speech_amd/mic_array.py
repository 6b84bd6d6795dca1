"""Search orchestration of localization-by-separation (SURVEY.md §8 a-M): the four stage
methods of the reference ``Mic_Array`` (sep/Mic_Array.py:95-500) with the same names,
arguments and return layouts.  The per-candidate arithmetic (shift, spot network,
energies) is behind ``spot_model`` on the GPU; SRP-PHAT behind ``SRPPhat``; what remains
here is list bookkeeping, thresholds and the greedy SI-SDR clustering, which follow the
reference including its order-dependent quirks (documented inline).
"""
import numpy as np

from .hostdsp import max_avg_power, si_sdr, split_wav, split_wise_sisdr
from .patch import FS, SPEED_OF_SOUND, Patch, pair_offsets
from .search import (INIT_WIDTH, SPOT_POWER_THRESHOLD2, USE_RELATIVE_SPOT_POWER, binary_search_baseline,
                     search_area)
from .srp import SRPPhat

FINE_CHUNK_EDGES = (0.0, 0.07, 0.40, 0.73, 0.93, 1.0)      # pipelined fine stage, see _fine_stage_pipelined

# sep/helpers/constants.py:24-27
BIN0, BIN1, N_FFT = 2, 200, 2048
FREQ_BINS = np.arange(BIN0, BIN1)


def check_sisnr_win(sisnr_list, SISNR_THRESHOLD=-2, SISNR_THRESHOLD2=-7):
    """Same talker if some segment is similar (> thr) and none is very different (< thr2)
    (sep/Mic_Array.py:18-28)."""
    v = np.asarray(list(sisnr_list), dtype=np.float64)
    return bool(np.any(v > SISNR_THRESHOLD) and not np.any(v < SISNR_THRESHOLD2))


def weight_mean_pos(patch_list, powers, id_lists):
    """Power-weighted mean position / offsets over the cluster members within 75 % of the
    head's power (sep/Mic_Array.py:32-47)."""
    head = powers[id_lists[0]]
    pos = np.zeros((3,))
    offs = np.zeros(patch_list[0].sample_offset.shape)
    tot = 0
    for i in id_lists:
        if powers[i] < head * 0.75:
            continue
        pos += powers[i] * patch_list[i].center_pos()
        offs += powers[i] * patch_list[i].sample_offset
        tot += powers[i]
    return pos / tot, offs / tot


def find_merge_center(merged_offests, init_area, mic_positions, Big_patch_center, init_samples=None):
    """Patch of width 3 around the merged offsets holding the coarse patch's points that fall
    inside; falls back to the coarse centre (sep/Mic_Array.py:50-81).  The reference's
    widening loop leaves after its first pass (factor 0), which repeats the width-3 test.
    ``init_samples`` may carry pair_offsets(init_area, mic_positions) when the caller tests many
    cluster heads against the same coarse patch (same values, computed once)."""
    P = mic_positions.shape[0] - 1
    patch = Patch(merged_offests, [3 for _ in range(P)], None)
    if init_samples is None:
        init_samples = pair_offsets(init_area, mic_positions, SPEED_OF_SOUND, FS)
    inside = patch.hyperbola_sample(init_samples) == 1
    if np.sum(inside) == 0:
        patch.width_list = [3 for _ in range(P)]
        inside = patch.hyperbola_sample(init_samples) == 1
        if np.sum(inside) > 0:
            patch.area_points = init_area[:, inside]
        else:
            patch.peak_pos = Big_patch_center
    else:
        patch.area_points = init_area[:, inside]
    return patch


class MicArray(object):
    def __init__(self, mic_positions, demo=False, Spk_Range=None, grid_size=0.05, Prone_method="SRP",
                 MIN_TRIGGER_POWER=0.5, SRP_fast=False, cached=False, cached_folder=None, device=None):
        if Prone_method != "SRP":
            raise RuntimeError("only the SRP-PHAT pruner is provided (MUSIC/TOPS are out of scope)")
        self.Prone_method = Prone_method
        self.MIN_TRIGGER_POWER = MIN_TRIGGER_POWER
        self.visual_save = False
        self.Range_spk = Spk_Range
        print("Init the Range_spk: ", Spk_Range)
        self.mic_positions = mic_positions
        self.num_mic = mic_positions.shape[0]
        # physical TDoA bound per pair, 8 cm slack (sep/Mic_Array.py:114-116)
        self.upper_bound_pairwise = (np.linalg.norm(mic_positions[1:] - mic_positions[0], axis=1) + 0.08) \
            / SPEED_OF_SOUND * FS
        self.SRP_node = SRPPhat(mic_pos=mic_positions, freq_bins=FREQ_BINS, Range_spk=Spk_Range, grid_size=grid_size,
                                FS=FS, n_fft=N_FFT, threshold=[0.15, 0.015, 0.05], WIDTH=INIT_WIDTH, device=device)
        self.original_times = 0
        self.spotforming_times = 0
        self.big_spotforming_times = 0
        self._device_scorer = None      # set by the fine stage when the spot model offers the GPU SI-SDR kernels
        self._seg_cache = {}
        self._dev_cache = {}
        # decision trace of the latest search (cheap bookkeeping, used by the precision flip-rate and
        # parity tests): coarse kept indices, per coarse patch {head: members} of the fine-stage
        # clustering, and the global clusters as lists of "g_head" names
        self.trace = {"coarse_kept": [], "fine_clusters": {}, "final_clusters": []}

    # ---- stage 1: SRP-PHAT pruning (sep/Mic_Array.py:152-194) ---------------------------
    def Apply_SRP_PHAT(self, mix_data):
        self.SRP_node.reset()
        self.spotforming_times = 0
        self.original_times = 0
        mix_np = mix_data.numpy() if hasattr(mix_data, "numpy") else np.asarray(mix_data)
        win = 36000 if mix_np.shape[1] >= 72000 else 24000
        self.SRP_node.SRP_Map_WINDOW_new(mix_np, window=win)
        patch_list = self.SRP_node.local_source_adaptive()
        return patch_list, np.zeros((3, 3))

    # ---- stage 2: coarse Spotforming, relaxed window (sep/Mic_Array.py:196-222) ---------
    def Spotform_Big_Patch(self, mix_data, patch_list, spot_model):
        self.big_spotforming_times = len(patch_list)
        kept, _powers_with_dis, rel_thr = binary_search_baseline(mix_data, spot_model, patch_list,
                                                                 self.mic_positions)
        self.Relative_Threshold = rel_thr
        pos = {id(p): i for i, p in enumerate(patch_list)}
        self.trace = {"coarse_kept": [pos[id(p)] for p in kept], "fine_clusters": {}, "final_clusters": []}
        return kept

    # ---- stage 3: fine Spotforming, strict window (sep/Mic_Array.py:225-395) ------------
    def _subdivide(self, big):
        """Fine candidates of one coarse patch: its hypercube subdivision plus the centre
        candidate, which goes last (sep/Mic_Array.py:245-262)."""
        P = self.num_mic - 1
        fine = search_area([big], self.mic_positions, self.upper_bound_pairwise)
        centre_patch = Patch(big.sample_offset, [2 for _ in range(P)], None, big.peak_pos)
        c = centre_patch.center_pos()
        if c is not None:
            fine.append(centre_patch)
        else:
            print("it is impossible to be here")
        return fine, c

    def _cluster_group(self, g, big, patches, powers, powers2, area, centre, T_len, thr_new, sample_gt,
                       sim_of, audio_of):
        """Thresholding + SI-SDR clustering of the candidates of ONE coarse patch and the output
        tuples of its cluster heads (sep/Mic_Array.py:283-383).  ``sim_of(k, h)`` gives
        si_sdr(candidate k, candidate h); ``audio_of(heads)`` the heads' waveforms."""
        out = []
        big_label = -1
        if sample_gt is not None:
            for k in range(sample_gt.shape[1]):
                if np.amax(np.abs(big.sample_offset - sample_gt[:, k])) < 3.5:
                    big_label = k
                    break
        c = big.center_pos()
        d = np.linalg.norm(c - self.mic_positions[0]) if c.shape[0] == 3 else 4
        if np.amax(powers2) < thr_new / (1 + d):
            return out
        order = np.argsort(-1 * np.array(powers))                           # sorted by total power (:339)
        clusters = {}
        # the reference scales the trigger by the length of the LAST candidate row (:343)
        min_trigger = self.MIN_TRIGGER_POWER / (3 * 48000) * T_len
        for k in order:
            d = np.linalg.norm(patches[k].center_pos() - self.mic_positions[0])
            if powers2[k] < thr_new / (1 + d) or powers[k] < min_trigger:
                continue
            home = None
            for head in clusters:
                if sim_of(k, clusters[head][0]) > -4:                       # SI_SDR_THRESHOLD (:340)
                    home = head
                    break
            if home is None:
                clusters[k] = [k]
            else:
                clusters[home].append(k)
        self.trace["fine_clusters"][int(g)] = {int(h): [int(k) for k in m] for h, m in clusters.items()}
        if len(clusters) == 0:
            return out
        heads = list(clusters.keys())
        audio = audio_of(heads)
        area_samples = None
        for n, head in enumerate(heads):
            _position, offs = weight_mean_pos(patches, powers, clusters[head])
            if area_samples is None:                                        # once per coarse patch
                area_samples = pair_offsets(area, self.mic_positions, SPEED_OF_SOUND, FS)
            merged = find_merge_center(offs, area, self.mic_positions, centre, area_samples)
            if merged.center_pos() is None:
                print("Warning some bug happen one source may be drop")
            out.append((merged, audio[n], powers[head], str(g) + '_' + str(head),
                        {"audio_offset": patches[head].sample_offset, "localization_offset": offs}, big_label))
        return out

    def _resident_group(self, g, big, patches, waves_g, energies_g, area, centre, T_len, thr_new, sample_gt,
                        spot_model):
        """One coarse patch with the waveforms on the GPU: similarities from one Gram launch (made
        only if a second candidate survives the thresholds), heads copied to the host."""
        sim = {}

        def sim_of(k, h):
            if "m" not in sim:
                sim["m"] = spot_model.pair_sisdr(waves_g)
            return sim["m"][k, h]

        kept = {}

        def audio_of(heads):
            kept["rows"] = waves_g[heads]
            return kept["rows"].cpu().numpy()
        out = self._cluster_group(g, big, patches, list(energies_g[:, 0]), list(energies_g[:, 1]), area, centre,
                                  T_len, thr_new, sample_gt, sim_of, audio_of)
        for n, pair in enumerate(out):     # the global clustering needs these; here they hide behind the GPU
            self._seg_cache[id(pair[1])] = (pair[1], split_wav(pair[1]))
            self._dev_cache[id(pair[1])] = (pair[1], kept["rows"][n])      # the same waveform, still on the GPU
        return out

    def Spotform_Small_Patch_Parallel(self, mix_data, candidate_finished, spot_model, sample_gt=None,
                                      run_demo_folder=None):
        thr_new = min([SPOT_POWER_THRESHOLD2, self.Relative_Threshold]) if USE_RELATIVE_SPOT_POWER \
            else SPOT_POWER_THRESHOLD2
        resident = hasattr(spot_model, "shift_and_sep_resident")
        inner = getattr(spot_model, "inner", spot_model)                 # ShardedSpotModel wraps the scorer
        self._device_scorer = inner if (resident and hasattr(inner, "segment_sisdr")) else None
        sharded = getattr(spot_model, "world", 1) > 1
        n_groups = len(candidate_finished)
        self.spotforming_times = 0
        self._seg_cache = {}               # id(waveform) -> (waveform, voiced segments), filled by the resident path
        self._dev_cache = {}               # id(waveform) -> (waveform, its device row)
        if resident and not sharded and n_groups >= 3 and getattr(spot_model, "device", None) is not None:
            return self._fine_stage_pipelined(mix_data, candidate_finished, spot_model, sample_gt, thr_new)
        if resident and sharded and getattr(inner, "device", None) is not None:
            return self._fine_stage_sharded(mix_data, candidate_finished, spot_model, sample_gt, thr_new)

        total_patch, bounds, areas, centers = [], [0], [], []
        for big in candidate_finished:
            fine, c = self._subdivide(big)
            areas.append(big.area_points)
            centers.append(c)
            self.spotforming_times += len(fine)
            total_patch.extend(fine)
            bounds.append(self.spotforming_times)

        # one rank per GPU (shard.ShardedSpotModel): whole coarse patches are dealt to ranks so
        # the per-patch clustering below stays local; energies are all-gathered (the stage's one
        # collective) and only the finished output tuples travel.
        mine = spot_model.my_groups([bounds[i + 1] - bounds[i] for i in range(n_groups)]) if sharded \
            else list(range(n_groups))
        if sharded:
            gbounds = bounds
            total_patch_all = total_patch
            total_patch, bounds = [], [0]
            for i in mine:
                total_patch.extend(total_patch_all[gbounds[i]:gbounds[i + 1]])
                bounds.append(len(total_patch))
        slot = {g: n for n, g in enumerate(mine)}              # coarse patch -> local group slot

        # the hot call.  With the HIP spot model the N x T outputs stay on the GPU: energies come
        # from the device reduction, SI-SDR similarities from the device Gram kernel, and only
        # the cluster heads' waveforms are copied to the host.  Any other
        # duck-typed model goes through the reference's host loops.
        if len(total_patch) == 0:                          # a rank that was dealt no coarse patch
            waves_dev, energies, sep_all = None, np.zeros((0, 2)), None
            T_len = int(mix_data.shape[1])
        elif resident:
            waves_dev, energies = spot_model.shift_and_sep_resident(mix_data, total_patch, Strict=1)
            T_len = int(waves_dev.shape[1])
        else:
            sep_all = spot_model.shift_and_sep(mix_data, total_patch, Strict=1)
            T_len = int(sep_all.shape[1])
        if sharded and resident:
            self.fine_energies = spot_model.all_gather_groups(energies, mine, gbounds)

        output_pair = []
        for g in mine:
            i = slot[g]                                                     # local slot of coarse patch g
            big = candidate_finished[g]
            patches = total_patch[bounds[i]:bounds[i + 1]]
            if resident:
                output_pair.extend(self._resident_group(g, big, patches, waves_dev[bounds[i]:bounds[i + 1]],
                                                        energies[bounds[i]:bounds[i + 1]], areas[g], centers[g],
                                                        T_len, thr_new, sample_gt, spot_model))
                continue
            sep = sep_all[bounds[i]:bounds[i + 1]]
            powers, powers2 = [], []
            for j in range(len(patches)):
                sep[j, :] = sep[j, :] - np.mean(sep[j, :])                 # in place, as :291
                powers.append(np.sum(sep[j, :] ** 2))
                powers2.append(max_avg_power(sep[j, :]))
            output_pair.extend(self._cluster_group(
                g, big, patches, powers, powers2, areas[g], centers[g], T_len, thr_new, sample_gt,
                lambda k, h, sep=sep: si_sdr(sep[k, :], sep[h]), lambda heads, sep=sep: [sep[h, :] for h in heads]))
        if sharded:
            # the voiced segments of every head travel with its tuple: the global clustering of every rank
            # then finds them cached for the remote heads too, as it does for its own
            tagged = [p + (self._seg_cache.get(id(p[1]), (None, None))[1],) for p in output_pair]
            merged = spot_model.gather_pairs(tagged)
            output_pair = [t[:-1] for t in merged]
            for t, p in zip(merged, output_pair):
                if t[-1] is not None:
                    self._seg_cache[id(p[1])] = (p[1], t[-1])
        return output_pair

    def _fine_stage_sharded(self, mix_data, candidate_finished, spot_model, sample_gt, thr_new):
        """One rank per GPU with the HIP model (shard.ShardedSpotModel).  Whole coarse patches are dealt to
        ranks by a weight every rank knows without subdividing anything -- the number of 1 cm grid points of
        the patch, which is what drives the size of its subdivision; each rank then subdivides and runs the
        pipelined fine stage on ITS patches only.  The stage's exchanges follow: the sizes (a few integers),
        the energy all-gather, and the object gather of the finished output tuples with their voiced
        segments.  Same output_pair list on every rank as on one GPU."""
        n_groups = len(candidate_finished)
        owners = spot_model.deal_groups([max(1, big.area_size()) for big in candidate_finished])
        mine = owners[spot_model.rank]
        sizes_mine = {}
        output_pair, energies = self._fine_stage_pipelined(mix_data, candidate_finished, spot_model, sample_gt, thr_new,
                                                           owned=mine, sizes_out=sizes_mine)
        sizes = spot_model.gather_sizes(sizes_mine, n_groups)
        self.spotforming_times = int(sum(sizes))
        gbounds = [0]
        for n in sizes:
            gbounds.append(gbounds[-1] + n)
        self.fine_energies = spot_model.all_gather_groups(energies, mine, gbounds, owners=owners)
        # the voiced segments of every head travel with its tuple: the global clustering of every rank
        # then finds them cached for the remote heads too, as it does for its own
        tagged = [p + (self._seg_cache.get(id(p[1]), (None, None))[1],) for p in output_pair]
        merged = spot_model.gather_pairs(tagged)
        output_pair = [t[:-1] for t in merged]
        for t, p in zip(merged, output_pair):
            if t[-1] is not None:
                self._seg_cache[id(p[1])] = (p[1], t[-1])
        return output_pair

    def _fine_stage_pipelined(self, mix_data, candidate_finished, spot_model, sample_gt, thr_new, owned=None, sizes_out=None):
        """Single-GPU fine stage with the host work hidden behind the GPU: the coarse patches are
        processed in contiguous chunks; while the GPU evaluates the candidates of chunk c
        the host subdivides the patches of chunk c+1, and the clustering of chunk c (energies,
        Gram launches, head copies -- issued on a side stream that only waits for chunk c) runs
        while the GPU is already on chunk c+1.  Same candidates, same order, same output.
        ``owned`` (sharded use): the coarse patches this rank owns; the call then returns (output_pair,
        energies of these groups in that order) and reports their subdivision sizes in ``sizes_out``."""
        import torch
        dev = getattr(spot_model, "inner", spot_model).device
        mix_dev = torch.as_tensor(mix_data).to(dev, dtype=torch.float32).contiguous()
        T_len = int(mix_dev.shape[1])
        order = list(range(len(candidate_finished))) if owned is None else [int(g) for g in owned]
        n_groups = len(order)
        local_energies = []
        if n_groups == 0:                                  # a rank that was dealt no coarse patch
            return ([], np.zeros((0, 2))) if owned is not None else []
        if getattr(self, "_side_stream", None) is None:
            # high priority: its SI-SDR kernels and read-backs are tiny and sit on the search's critical path, while the
            # main stream is running whole network launches that fill the device
            self._side_stream = torch.cuda.Stream(device=dev, priority=-1)
        side, main = self._side_stream, torch.cuda.current_stream(dev)
        # Chunk edges (fractions of the coarse patches).  Only two pieces of host work cannot hide behind
        # the GPU: the subdivision of the first chunk and the clustering of the last one -- so those two
        # chunks are small (about 1/15 of the patches each) and the middle ones large enough to keep the
        # internal batches full.  (Bench scene, 710 candidates, same box: three equal chunks 361 ms; these
        # edges 340; [0, .1, .5, .9, 1] 346; [0, .13, .5, .87, 1] 349; seven chunks 345.)
        edges = sorted(set(min(n_groups, max(0, round(n_groups * f))) for f in FINE_CHUNK_EDGES) | {0, n_groups})
        n_chunks = len(edges) - 1
        output_pair, inflight = [], None

        def finish(job):
            groups, fines, centres, waves, en_dev, ev = job
            with torch.cuda.stream(side):
                side.wait_event(ev)
                waves.record_stream(side)
                en_dev.record_stream(side)
                energies = en_dev.cpu().numpy()
                local_energies.append(energies)
                pos = 0
                for g, fine, centre in zip(groups, fines, centres):
                    n = len(fine)
                    output_pair.extend(self._resident_group(
                        g, candidate_finished[g], fine, waves[pos:pos + n], energies[pos:pos + n],
                        candidate_finished[g].area_points, centre, T_len, thr_new, sample_gt, spot_model))
                    pos += n

        for k in range(n_chunks):
            groups = order[edges[k]:edges[k + 1]]
            if not groups:
                continue
            fines, centres, flat = [], [], []
            for g in groups:                                   # host: subdivision of this chunk
                fine, c = self._subdivide(candidate_finished[g])
                self.spotforming_times += len(fine)
                if sizes_out is not None:
                    sizes_out[int(g)] = len(fine)
                fines.append(fine)
                centres.append(c)
                flat.extend(fine)
            waves, en_dev = spot_model.shift_and_sep_resident(mix_dev, flat, Strict=1, device_energies=True)
            ev = torch.cuda.Event()
            ev.record(main)
            if inflight is not None:
                finish(inflight)                               # host clustering of the previous chunk
            inflight = (groups, fines, centres, waves, en_dev, ev)
        if inflight is not None:
            finish(inflight)
        if owned is not None:
            return output_pair, (np.concatenate(local_energies, axis=0) if local_energies else np.zeros((0, 2)))
        return output_pair

    # ---- stage 4: global non-max suppression (sep/Mic_Array.py:399-500) -----------------
    def Clustering_new(self, output_pair, simple_pos=None, sample_gt=None):
        cands = sorted(output_pair, key=lambda x: -x[2])
        # With the HIP spot model the O(n^2) waveform comparisons of this stage run on the GPU:
        # one launch for the full-length SI-SDR matrix, one for the segment-wise tensor
        # (SURVEY.md §8f-2).  Any other model keeps the reference's host loops.
        scorer = getattr(self, "_device_scorer", None)
        cache = getattr(self, "_seg_cache", {})
        seg_all = []
        for c in cands:
            hit = cache.get(id(c[1]))
            seg_all.append(hit[1] if hit is not None and hit[0] is c[1] else split_wav(c[1]))
        full_dev = seg_dev = None
        if scorer is not None and len(cands) > 1:
            import torch
            dev_rows = [getattr(self, "_dev_cache", {}).get(id(c[1])) for c in cands]
            if all(r is not None and r[0] is c[1] for r, c in zip(dev_rows, cands)):
                waves = torch.stack([r[1] for r in dev_rows])         # the fine stage left every row on the GPU
            else:
                waves = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(c[1], dtype=np.float32) for c in cands])))
                waves = waves.to(scorer.device if getattr(scorer, "device", None) is not None else "cuda")
            full_dev = scorer.pair_sisdr(waves)
            seg_dev, _ = scorer.segment_sisdr(waves, seg_all)
        clusters = {}
        wrong = []
        centres = [c[0].center_pos() for c in cands]
        win_dev = None
        if seg_dev is not None:
            # check_sisnr_win of every ordered pair at once (entries beyond a candidate's own segment
            # count are NaN and compare false, like the slice [:len(segs)] of the per-pair form)
            with np.errstate(invalid="ignore"):
                win_dev = np.any(seg_dev > -2, axis=2) & ~np.any(seg_dev < -7, axis=2)
        near = None
        if win_dev is not None and all(c is not None for c in centres):
            # dis < 0.45 for every pair; entries within rounding of the threshold are settled by the very
            # call the per-pair form makes
            xy = np.array([c[:2] for c in centres], dtype=np.float64)
            d = xy[:, None, :] - xy[None, :, :]
            dist = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])
            near = dist < 0.45
            for a, b in zip(*np.nonzero(np.abs(dist - 0.45) < 1e-9)):
                near[a, b] = np.linalg.norm(centres[a][:2] - centres[b][:2]) < 0.45
            merge = (full_dev > -1) | win_dev | near             # (:401,458) for every ordered pair
        heads = []                                               # cluster heads in creation order
        for i, cand in enumerate(cands):
            centre1, audio1, power1, big_label = centres[i], cand[1], cand[2], cand[-1]
            segs = seg_all[i]
            if len(segs) == 0:
                print("discard because no invalid split!!!")
                continue
            unique, belong = True, -1
            seg_tab, seen = [], []
            if near is not None:
                hit = np.flatnonzero(merge[i, heads]) if heads else np.empty(0, dtype=np.int64)
                if hit.size:                                     # first head in creation order wins
                    belong = heads[int(hit[0])]
                    clusters[belong].append(i)
                    unique = False
                    seen = heads[:int(hit[0]) + 1]
                else:
                    seen = list(heads)
            else:
                for head in clusters:
                    h = clusters[head][0]
                    audio2, centre2 = cands[h][1], centres[h]
                    sim = si_sdr(audio1, audio2)
                    per_seg = split_wise_sisdr(audio1, audio2, segs)
                    seg_tab.append(per_seg)
                    dis = np.linalg.norm(centre1[:2] - centre2[:2])
                    if sim > -1 or check_sisnr_win(per_seg) or dis < 0.45:      # (:401,458)
                        clusters[h].append(i)
                        unique, belong = False, head
                        break
            if seen:
                seg_tab = seg_dev[i, seen, :len(segs)]
            if len(seg_tab) != 0:
                best = np.amax(np.array(seg_tab), axis=0)
                if check_sisnr_win(best, SISNR_THRESHOLD=-1, SISNR_THRESHOLD2=-5):
                    unique = False
            if unique:
                clusters[i] = [i]
                heads.append(i)
            elif big_label >= 0 and sample_gt is not None and belong >= 0:
                h = clusters[belong][0]
                if cands[h][-1] == -1:
                    delta = (cands[h][-2]["audio_offset"] - sample_gt[:, big_label]).astype(int)
                    wrong.append((big_label, cands[h][-1], delta, power1 / cands[h][2]))
        print("final speaker number is ", len(clusters.keys()))
        self.trace["final_clusters"] = [[cands[i][3] for i in clusters[h]] for h in clusters]
        patch_final = [cands[clusters[h][0]] for h in clusters]
        audio_final = [p[1] for p in patch_final]
        return audio_final, patch_final, self.big_spotforming_times + self.spotforming_times, wrong

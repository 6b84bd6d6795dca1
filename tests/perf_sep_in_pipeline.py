import io, sys, time, os
from contextlib import redirect_stdout
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from acousticswarms_speech_amd.config import FULL, SEP_FULL
from acousticswarms_speech_amd.joint import JointModel
from acousticswarms_speech_amd.scenes import make_scene
from acousticswarms_speech_amd.sep import SepModel
from acousticswarms_speech_amd.spot import SpotModel
from acousticswarms_speech_amd.weights import make_sep_state_dict, make_spot_state_dict
scene = make_scene(1010, 5, 7, 48000, reverb=True)
spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=256, precision="f16x3").to("cuda")
sep = SepModel(SEP_FULL, make_sep_state_dict(SEP_FULL, 9), precision="f16x3").to("cuda")
jm = JointModel(spot, sep, device="cuda")
mix = torch.from_numpy(scene.mix)
with redirect_stdout(io.StringIO()):
    jm.setup(scene.mic_positions, scene.speaker_range)
    out = jm.forward(mix)
    out = jm.forward(mix)
print("times", [round(t * 1e3, 1) for t in jm.times], "talkers", len(out[0]))
patches = [p[0] for p in out[0]]
def t(f, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("sep.infer x3 back to back: %.1f ms" % t(lambda: sep.infer(mix, patches)))
from acousticswarms_speech_amd.sep import rounded_offsets
offs = torch.from_numpy(rounded_offsets([p.sample_offset for p in patches], 6)).cuda()
mix_d = mix.cuda()
print("infer_device only: %.1f ms" % t(lambda: sep.infer_device(mix_d, offs)))
print("offsets abs max", int(offs.abs().max()), "S", offs.shape[0])
time.sleep(0.3)
print("after 0.3 s idle, one call: %.1f ms" % t(lambda: sep.infer_device(mix_d, offs), 1))
print("next call: %.1f ms" % t(lambda: sep.infer_device(mix_d, offs), 1))
ro = torch.randint(-140, 141, offs.shape, dtype=torch.int32).cuda()
print("random offsets: %.1f ms" % t(lambda: sep.infer_device(mix_d, ro)))
with redirect_stdout(io.StringIO()):
    res = jm.localize_by_separation(mix)
torch.cuda.synchronize()
pp = [p[0] for p in res[0]]
for k in range(3):
    t0 = time.perf_counter(); sep.infer(mix, pp); torch.cuda.synchronize()
    print("right after the search, call %d: %.1f ms" % (k, (time.perf_counter() - t0) * 1e3))
with redirect_stdout(io.StringIO()):
    res = jm.localize_by_separation(mix)
torch.cuda.synchronize()
t0 = time.perf_counter(); mix_d2 = mix.to("cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
y = sep.infer_device(mix_d2, offs); torch.cuda.synchronize(); t2 = time.perf_counter()
yh = y.cpu().numpy(); t3 = time.perf_counter()
print("after the search: H2D %.1f ms, infer_device %.1f ms, D2H %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print("---- what is slow right after the search?")
for trial in ("tiny H2D", "kernel launch", "D2H small"):
    with redirect_stdout(io.StringIO()):
        res = jm.localize_by_separation(mix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if trial == "tiny H2D":
        o = torch.from_numpy(np.zeros((27, 6), dtype=np.int32)).to("cuda")
    elif trial == "kernel launch":
        o = torch.zeros(1024, device="cuda") + 1
    else:
        o = offs.cpu()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    o2 = torch.from_numpy(np.zeros((27, 6), dtype=np.int32)).to("cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: first %.2f ms, then a tiny H2D %.2f ms" % (trial, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
import gc
with redirect_stdout(io.StringIO()):
    res = jm.localize_by_separation(mix)
torch.cuda.synchronize()
t0 = time.perf_counter(); gc.collect(); t1 = time.perf_counter()
print("gc.collect after the search: %.2f ms" % ((t1 - t0) * 1e3))
print("---- allocator counters around the first allocation after the search")
for trial in range(2):
    with redirect_stdout(io.StringIO()):
        res = jm.localize_by_separation(mix)
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    o = torch.empty(1024, device="cuda")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print("torch.empty: %.2f ms (+sync %.2f ms); device mallocs %d -> %d, frees %d -> %d, reserved %.1f -> %.1f GB" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, s0["num_device_alloc"], s1["num_device_alloc"], s0["num_device_free"], s1["num_device_free"],
        s0["reserved_bytes.all.current"] / 1e9, s1["reserved_bytes.all.current"] / 1e9))
print("---- pageable vs pinned H2D right after the search")
pin = torch.zeros((27, 6), dtype=torch.int32).pin_memory()
for trial in ("pinned", "pageable", "pinned", "pageable"):
    with redirect_stdout(io.StringIO()):
        res = jm.localize_by_separation(mix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if trial == "pinned":
        o = pin.to("cuda", non_blocking=True)
    else:
        o = torch.from_numpy(np.zeros((27, 6), dtype=np.int32)).to("cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    y = sep.infer_device(mix_d, offs)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s H2D: %.2f ms; then infer_device %.1f ms" % (trial, (t1 - t0) * 1e3, (t2 - t1) * 1e3))

"""cProfile of the reference's training step (bench.py --workload c1_train) on the Seoul-shaped stand-in: where the HOST time of a
launch-bound step goes.  Usage: python tools/c1_train_profile.py [reference|device] [steps]"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "device"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
model, step, coo, ids = bench.seoul_train_setup(pkg, dev, 1024, mode)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step (no profiler)")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])

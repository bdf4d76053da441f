"""r04 lab: what would fusing the SpMM into the dense half of a layer (`ngcf_layer_fused_f32` as ONE kernel) save
at most?  Today LE [N, d] leaves the SpMM through HBM and is read back by the dense kernel (0.56 GB each way per layer at C3).
The dense kernel's share of that round trip is bounded from above by running it with LE ALIASED to E: both operands are then the
same bytes, a row tile's second read hits the vector L1 / L2, and the kernel moves 0.56 GB less from HBM - the traffic of a dense
half whose LE rows "are already on the chip".  (The result is a different product, the instruction stream is the same.)
The SpMM's share - 0.56 GB of stores under 51 GB of gathers - is priced by arithmetic in DESIGN.md 4.2.

    python tools/le_roundtrip_lab.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402

eng = pkg.engine
dev = torch.device("cuda:0")
d = 128
g = torch.Generator().manual_seed(1)
W1, W2 = ((torch.rand((d, d), generator=g) - 0.5).to(dev) * 0.2 for _ in range(2))
b1, b2 = ((torch.rand((d,), generator=g) - 0.5).to(dev) * 0.1 for _ in range(2))
ws = eng.Workspace()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for N in (1_100_000, 131_072):
    E = (torch.rand((N, d), generator=g) - 0.5).to(dev) * 0.3          # the bench's operand scale (small embeddings)
    LE = (torch.rand((N, d), generator=g) - 0.5).to(dev) * 0.3
    carry, norm = torch.empty((N, d), device=dev), torch.empty((N, d), device=dev)
    t_today = timeit(lambda: eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, ws))
    t_alias = timeit(lambda: eng.layer_dense(E, E, W1, b1, W2, b2, carry, norm, ws))
    t_nocarry = timeit(lambda: eng.layer_dense(LE, E, W1, b1, W2, b2, None, norm, ws))
    gb = N * d * 4 / 1e9
    print(f"N={N}: dense half today (LE and E from memory, carry + norm written: {4 * gb:.2f} GB) {t_today * 1e3:.1f} us; LE aliased to E "
          f"({3 * gb:.2f} GB: LE 'already on the chip') {t_alias * 1e3:.1f} us -> at most {(t_today - t_alias) * 1e3:.1f} us per layer for the dense "
          f"kernel's half of the LE round trip; last layer (no carry) {t_nocarry * 1e3:.1f} us", flush=True)
    del E, LE, carry, norm
